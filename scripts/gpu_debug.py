import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0,'/root/repo/tests')
from lidar_snow_sim_amd import engine
from oracle import snow_oracle as so
from lidar_snow_sim_amd.tools.wet_ground.augmentation import noise_threshold_poly, ground_water_augmentation, estimate_laser_parameters
eng = engine.get_engine(0)
G='/root/repo/tests/golden/'
T = np.load(G+'tables.npz'); tl=[T[f't{i%4}'] for i in range(64)]
d = np.load(G+'L5_augment_portable.npz')
for c in range(8):
    pc=d[f'c{c}_pc']; plane=(d[f'c{c}_plane_w'], float(d[f'c{c}_plane_h'])); order=list(d[f'c{c}_order'])
    tids = eng.table_ids_from_arrays(tl, order)
    _,_,cnt,stats,thr = eng.ctx.augment_batch(pc,[0,pc.shape[0]],[tids],float(d['bd']),plane=[[*plane[0],plane[1]]],want_thr=True)
    srt = pc[np.argsort(pc[:,4],kind='stable')]
    host = noise_threshold_poly(srt, plane[0], plane[1], 0.7)
    dist=np.linspace(3,80,50)
    print(c, pc.dtype, 'dev', thr[0], 'host', host, 'maxrel', np.max(np.abs(np.polyval(thr[0],dist)-np.polyval(host,dist))/np.abs(np.polyval(host,dist))), 'stats', stats[0], d[f'c{c}_stats'])
d = np.load(G+'L6_wet_ground_portable.npz')
PLANE=(np.array([0.,0.,-1.]),-1.7)
for c in range(8):
    pc=d[f'c{c}_pc']
    out,src = ground_water_augmentation(pc, water_height=0.0008, pavement_depth=0.001, flat_earth=bool(d[f'c{c}_flat']), debug=False, replace=bool(d[f'c{c}_replace']), plane=PLANE, return_src=True)
    ref=d[f'c{c}_out']
    print('L6',c,pc.dtype,out.shape,ref.shape, end=' ')
    if out.shape==ref.shape:
        print('xyz/l eq', np.array_equal(out[:,[0,1,2,4]],ref[:,[0,1,2,4]]), 'int maxrel', np.max(np.abs(out[:,3]-ref[:,3])/np.maximum(np.abs(ref[:,3]),1e-300)))
    else: print()
