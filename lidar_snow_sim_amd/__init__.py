"""lidar_snow_sim_amd -- MI355X-native snowfall / wet-ground augmentation engine.

The hot path of SysCV/LiDAR_snow_sim (tools/snowfall/simulation.py::augment and what it calls) as
hand-written HIP for gfx950 behind a C ABI (include/snowgpu.h), with Python host code that mirrors the
reference's importable surface:

    lidar_snow_sim_amd.tools.snowfall.simulation   augment, augment_batch
    lidar_snow_sim_amd.tools.snowfall.sampling     compute_occupancy, snowfall_rate_to_rainfall_rate, ...
    lidar_snow_sim_amd.tools.wet_ground.augmentation   ground_water_augmentation, estimate_laser_parameters
    lidar_snow_sim_amd.tools.wet_ground.planes     calculate_plane
"""
__version__ = "0.1.0"
