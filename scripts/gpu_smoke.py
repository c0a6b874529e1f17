import sys, time, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0,'/root/repo/tests')
t0=time.time()
from lidar_snow_sim_amd import engine
from oracle import snow_oracle as so
eng = engine.get_engine(0)
print('ctx ok', time.time()-t0, flush=True)
T = np.load('/root/repo/tests/golden/tables.npz'); tl=[T[f't{i%4}'] for i in range(64)]
d = np.load('/root/repo/tests/golden/L5_augment_portable.npz')
from lidar_snow_sim_amd.tools.snowfall.simulation import augment
for c in range(8):
    pc=d[f'c{c}_pc']; plane=(d[f'c{c}_plane_w'], float(d[f'c{c}_plane_h']))
    stats,aug,src = augment(pc,'x',float(d['bd']),only_camera_fov=False,plane=plane,order=list(d[f'c{c}_order']),particles=tl,return_src=True,device_prepass=False)
    print(c, pc.dtype, tuple(int(s) for s in stats), tuple(int(s) for s in d[f'c{c}_stats']), flush=True)
