"""-m gpu: the torch-tensor boundary of augment() / augment_batch() (lidar_snow_sim_amd/tensors.py): CUDA tensors in, CUDA tensors
out, snowgpu_augment_batch_device on torch's current stream -- against the oracle, against the host entry's bytes, asynchronously,
with the wet-ground model chained behind it, and with the reference's exception types (tools/snowfall/simulation.py:53, :427-429;
SURVEY 8 b "Ownership")."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

PLANE = (np.array([0.0, 0.0, -1.0]), -1.7)
BD = float(np.degrees(3e-3))


@pytest.fixture(scope="module")
def so():
    from oracle import snow_oracle
    snow_oracle.build()
    return snow_oracle


def _tables64(tables):
    return [tables["t"][i % 4] for i in range(64)]


def _ragged_frames(dtype=np.float32):
    from lidar_snow_sim_amd.synthetic import synthetic_sweep
    full = [synthetic_sweep(64, 2048, seed=1200 + f, intensity="lambert").reshape(64, 2048, 5) for f in range(3)]
    return [np.ascontiguousarray(full[0][:, ::8, :].reshape(-1, 5)).astype(dtype), np.ascontiguousarray(full[1][:, 1::16, :].reshape(-1, 5)).astype(dtype),
            np.ascontiguousarray(full[2][:, 3::8, :].reshape(-1, 5)).astype(dtype)]


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["float32", "float64"])
def test_tensor_batch_matches_the_oracle_and_the_host_entry(so, tables, dtype):
    """A ragged batch of three frames as CUDA tensors: kept rows, labels, intensities and statistics are the oracle's, and the
    bytes are the ones the NumPy boundary returns for the same frames (same kernels, other plumbing)."""
    from lidar_snow_sim_amd.tools.snowfall.simulation import augment_batch
    frames = _ragged_frames(dtype)
    tl = _tables64(tables)
    orders = [list(np.random.default_rng(5 + f).permutation(64)) for f in range(3)]
    dev = torch.device("cuda:0")
    t_frames = [torch.from_numpy(f).to(dev) for f in frames]
    res = augment_batch(t_frames, "unused", BD, planes=[PLANE] * 3, orders=orders, particles=tl, return_src=True)
    host = augment_batch(frames, "unused", BD, planes=[PLANE] * 3, orders=orders, particles=tl, return_src=True)
    for f in range(3):
        st, aug, src = res[f]
        assert aug.is_cuda and src.is_cuda and aug.dtype == t_frames[f].dtype and aug.shape[1] == 5
        s0, a0, src0 = so.augment(frames[f], tl, BD, orders[f], plane=PLANE)
        got, gsrc = aug.cpu().numpy(), src.cpu().numpy()
        assert tuple(int(v) for v in st) == tuple(int(v) for v in s0)
        assert np.array_equal(gsrc, src0) and np.array_equal(got[:, 3:], a0[:, 3:])
        np.testing.assert_allclose(got[:, :3], a0[:, :3], rtol=1e-6 if dtype == np.float32 else 1e-12, atol=0)
        hs, ha, hsrc = host[f]
        assert tuple(int(v) for v in hs) == tuple(int(v) for v in st) and np.array_equal(hsrc, gsrc) and ha.tobytes() == got.tobytes()
    assert (np.concatenate([r[1].cpu().numpy()[:, 4] for r in res]) == 2).sum() > 20


def test_single_tensor_through_augment_with_the_reference_call_shape(so, tables):
    """augment(pc, prefix, beam_divergence, only_camera_fov=False) with pc a CUDA tensor: global `random` draws the channel permutation
    exactly as for an array (simulation.py:482-486), the plane is calculate_plane's flat-earth answer, the result is a tensor."""
    from lidar_snow_sim_amd.tools.snowfall.simulation import augment
    pc = _ragged_frames()[0]
    tl = _tables64(tables)
    random.seed(11)
    st, aug = augment(torch.from_numpy(pc).cuda(), "unused", BD, only_camera_fov=False, particles=tl)
    random.seed(11)
    st_h, aug_h = augment(pc, "unused", BD, only_camera_fov=False, particles=tl)
    assert aug.is_cuda and tuple(int(v) for v in st) == tuple(int(v) for v in st_h)
    assert aug.cpu().numpy().tobytes() == aug_h.tobytes()
    random.seed(11)
    order = list(range(64))
    random.shuffle(order)
    s0, a0, _ = so.augment(pc, tl, BD, order, plane=(np.array([0.0, 0.0, 1.0]), -1.55))
    assert tuple(int(v) for v in st) == tuple(int(v) for v in s0) and np.array_equal(aug.cpu().numpy()[:, 3:], a0[:, 3:])


def test_further_columns_ride_through_on_the_device(tables):
    """A sixth column (the source index the golden fixtures carry) comes back attached to the row it belonged to (simulation.py:447,
    :508-523 index whole rows)."""
    from lidar_snow_sim_amd.tools.snowfall.simulation import augment_batch
    pc = _ragged_frames()[1]
    pc6 = np.column_stack((pc, np.arange(len(pc)))).astype(np.float32)
    (st, aug, src), = augment_batch([torch.from_numpy(pc6).cuda()], "unused", BD, planes=[PLANE], orders=[list(range(64))],
                                    particles=_tables64(tables), return_src=True)
    assert aug.shape[1] == 6 and torch.equal(aug[:, 5].to(torch.int32), src)


def test_asynchronous_call_on_a_side_stream_and_reused_result_tensors(so, tables):
    """sync=False returns before anything has been waited for, on torch's CURRENT stream; a second call reuses the first one's result
    tensors (out=) and, queued behind a device-side change of the input on the same stream, sees the changed rows."""
    from lidar_snow_sim_amd.tensors import DeviceBatch, DeviceResult, augment_batch
    frames = [f for f in _ragged_frames() if True]
    n = min(f.shape[0] for f in frames)
    frames = [np.ascontiguousarray(f[:n]) for f in frames]          # (equal sizes: the stacked F x N x 5 form)
    tl = _tables64(tables)
    orders = np.asarray([np.random.default_rng(9 + f).permutation(64) for f in range(3)])
    stack = torch.from_numpy(np.stack(frames)).cuda()
    other = torch.from_numpy(np.stack(frames[::-1])).cuda()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        batch = DeviceBatch(stack.clone())
        r1 = augment_batch(batch, "unused", BD, planes=[PLANE] * 3, orders=orders, particles=tl, sync=False)
        assert isinstance(r1, DeviceResult) and r1.stream == s
        first = [(st, a.clone(), i.clone()) for st, a, i in r1.frames(return_src=True)]
        batch.rows.copy_(other.reshape(-1, 5))                       # device-side, same stream: ordered ahead of the next call
        r2 = augment_batch(batch, "unused", BD, planes=[PLANE] * 3, orders=orders, particles=tl, sync=False, out=r1)
        assert r2.rows.data_ptr() == r1.rows.data_ptr()
        second = r2.frames(return_src=True)
    for f in range(3):
        for got, pc in ((first[f], frames[f]), (second[f], frames[2 - f])):
            s0, a0, src0 = so.augment(pc, tl, BD, list(orders[f]), plane=PLANE)
            assert tuple(int(v) for v in got[0]) == tuple(int(v) for v in s0)
            assert np.array_equal(got[2].cpu().numpy(), src0) and np.array_equal(got[1].cpu().numpy()[:, 3:], a0[:, 3:])


@pytest.mark.parametrize("n_lanes", [2, 3])
def test_compute_lanes_in_flight_give_the_results_of_one(so, tables, n_lanes):
    """lane=k: the call runs on compute lane k (an engine context of its own, every kernel on ONE stream of that context -- lanes 0, 1, 2 on
    streams of three different priorities: snowgpu_lane_stream), the caller's stream does not wait for it, and the result is claimed through the
    DeviceResult (.join() on the consumer's stream, .wait() on the host).  Batches dealt round over the lanes, results reused per lane: every
    batch equals the oracle."""
    from lidar_snow_sim_amd import engine
    from lidar_snow_sim_amd.tensors import LANE_SLOT0, augment_batch
    frames = _ragged_frames()
    tl = _tables64(tables)
    nb = 2 * n_lanes
    orders = [[list(np.random.default_rng(40 + 3 * b + f).permutation(64)) for f in range(3)] for b in range(nb)]
    t_frames = [torch.from_numpy(f).cuda() for f in frames]
    res, got = [None] * n_lanes, []
    for b in range(nb):
        k = b % n_lanes
        if res[k] is not None:                                  # the lane's earlier batch: claim it before its tensors are reused
            got.append([(st, a.clone(), i.clone()) for st, a, i in res[k].frames(return_src=True)])
        res[k] = augment_batch(t_frames, "unused", BD, planes=[PLANE] * 3, orders=orders[b], particles=tl, sync=False, lane=k, out=res[k])
        assert res[k].stream != torch.cuda.current_stream()
    assert len({r.stream.cuda_stream for r in res}) == n_lanes   # a stream per lane ...
    ctxs = [engine.get_engine(0, LANE_SLOT0 + k).ctx for k in range(n_lanes)]
    assert all(r.stream.cuda_stream in {c.lane_stream(lv) for lv in range(3)} for r, c in zip(res, ctxs))     # ... and it is its context's own
    for r in res:
        r.join()                                                # (the consumer's stream waits; the host does not)
    torch.cuda.current_stream().synchronize()
    for r in res:
        got.append([(st, a.clone(), i.clone()) for st, a, i in r.frames(return_src=True)])
    assert len(got) == nb
    for b in range(nb):
        for f in range(3):
            s0, a0, src0 = so.augment(frames[f], tl, BD, orders[b][f], plane=PLANE)
            st, aug, src = got[b][f]
            assert tuple(int(v) for v in st) == tuple(int(v) for v in s0), (b, f)
            assert np.array_equal(src.cpu().numpy(), src0) and np.array_equal(aug.cpu().numpy()[:, 3:], a0[:, 3:]), (b, f)


def test_options_of_the_array_boundary_work_on_tensors_with_the_same_bytes(tables):
    """The keyword options the two boundaries share -- the camera crop inside the compaction (calib=, simulation.py:532-540), calculate_plane on
    the device by least squares (plane_method='lsq'), tables sampled on the device (particles='device'), caller polynomials (thr_polys=),
    shuffled permutations from Python's global generator -- give, on CUDA tensors, the bytes the NumPy boundary gives."""
    import random as pyrandom
    from lidar_snow_sim_amd.calibration import Calibration
    from lidar_snow_sim_amd.tools.snowfall import sampling as smp
    from lidar_snow_sim_amd.tools.snowfall.simulation import augment_batch
    cal = Calibration(P2=np.array([[700.0, 0, 960, 0], [0, 700.0, 512, 0], [0, 0, 1, 0]]), R0=np.eye(3),
                      V2C=np.array([[0, -1.0, 0, 0], [0, 0, -1.0, 0], [1.0, 0, 0, 0]]))
    frames = _ragged_frames()[:2]
    t_frames = [torch.from_numpy(f).cuda() for f in frames]
    tl = _tables64(tables)
    occ, rate = smp.compute_occupancy(2.5, 1.6), smp.snowfall_rate_to_rainfall_rate(2.5, 1.6)
    cases = [dict(particles=tl, planes=[PLANE] * 2, calib=cal),
             dict(particles=tl, plane_method="lsq"),
             dict(particles=tl, thr_polys=[[0.0, 0.01, 2.0]] * 2),
             dict(particles="device", planes=[PLANE] * 2)]
    for kw in cases:
        pyrandom.seed(77)
        want = augment_batch(frames, f"gunn_{rate}_{occ}", BD, return_src=True, **kw)
        pyrandom.seed(77)
        got = augment_batch(t_frames, f"gunn_{rate}_{occ}", BD, return_src=True, **kw)
        for (s0, a0, i0), (s1, a1, i1) in zip(want, got):
            assert tuple(int(v) for v in s0) == tuple(int(v) for v in s1), kw.keys()
            assert np.array_equal(i0, i1.cpu().numpy()) and a0.tobytes() == a1.cpu().numpy().tobytes(), kw.keys()
    assert want[0][1].shape[0] > 0


def test_wet_ground_chained_on_the_device(so, tables):
    """wet=...: snowfall -> ground_water_augmentation with the viewer's keyword arguments (pointcloud_viewer.py:2807-2821), one launch
    sequence, float64 rows out (wet_ground/augmentation.py:150); against the oracle chain."""
    from lidar_snow_sim_amd.synthetic import synthetic_sweep
    from lidar_snow_sim_amd.tools.snowfall.simulation import augment_batch
    frames = [synthetic_sweep(64, 512, seed=1300 + f, intensity="lambert") for f in range(2)]
    tl = _tables64(tables)
    order = list(range(64))
    wet = dict(water_height=0.0008, pavement_depth=0.001, power_factor=15, flat_earth=False, delta=0.5, replace=False)
    res = augment_batch([torch.from_numpy(f).cuda() for f in frames], "unused", BD, planes=[PLANE] * 2, orders=[order] * 2, particles=tl,
                        return_src=True, wet=dict(wet, noise_floor=0.7, plane=PLANE))
    for f in range(2):
        st, out, src = res[f]
        s0, a0, src0 = so.augment(frames[f], tl, BD, order, plane=PLANE)
        o0, wsrc0 = so.ground_water_augmentation(a0, noise_floor=0.7, plane=PLANE, return_src=True, **wet)
        got = out.cpu().numpy()
        assert out.dtype == torch.float64 and tuple(int(v) for v in st) == tuple(int(v) for v in s0)
        assert got.shape == o0.shape and np.array_equal(got[:, 4], o0[:, 4]) and np.array_equal(src.cpu().numpy(), src0[wsrc0])
        np.testing.assert_allclose(got[:, :4], o0[:, :4], rtol=1e-6, atol=0)


def test_reference_exception_types_from_the_status_words(tables):
    """A point at >= 120 m: the reference's IndexError (simulation.py:149); a frame without ground rows: its TypeError (:462).  The
    message is the library's (snowgpu_status_error), the same one the host entry gives."""
    from lidar_snow_sim_amd.tools.snowfall.simulation import augment_batch
    tl = _tables64(tables)
    pc = _ragged_frames()[0]
    far = np.array([[125.0, 1.0, 0.0, 30.0, 3.0], [10.0, 1.0, -1.0, 30.0, 3.0]], np.float32)
    with pytest.raises(IndexError, match="range grid"):
        augment_batch([torch.from_numpy(far).cuda()], "unused", float(np.degrees(3e-2)), thr_polys=[[0.0, 0.0, 0.0]], shuffle=False, particles=tl)
    sky = np.array([[10.0, 1.0, 5.0, 30.0, 3.0], [12.0, 1.0, 6.0, 30.0, 3.0]], np.float32)          # nothing near the plane
    with pytest.raises(TypeError, match="ground"):
        augment_batch([torch.from_numpy(sky).cuda()], "unused", BD, planes=[PLANE], shuffle=False, particles=tl)
    with pytest.raises(ValueError, match="q8"):
        augment_batch([torch.from_numpy(pc).cuda()], "unused", BD, planes=[PLANE], particles=tl, q8="numpy")
    (st, aug), = augment_batch([torch.empty((0, 5), dtype=torch.float32, device="cuda")], "unused", BD, planes=[PLANE], particles=tl)   # an empty frame
    assert tuple(int(v) for v in st) == (0, 0, 0) and tuple(aug.shape) == (0, 5) and aug.is_cuda
