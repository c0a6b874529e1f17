"""Device-resident boundary of augment_batch(): torch CUDA tensors in, torch CUDA tensors out, no host copy of a row.

The reference's training-time caller (the OpenPCDet DENSE dataset: `root_path`, tools/snowfall/simulation.py:53, :324-325; SURVEY 8 b
"Ownership": NumPy array or torch tensor data_ptr) holds its sweeps on the GPU.  `augment_batch(frames, ...)` /
`augment(pc, ...)` of lidar_snow_sim_amd.tools.snowfall.simulation hand anything that is a torch CUDA tensor to this module:

  * the rows are read where they lie (`tensor.data_ptr()`), through snowgpu_augment_batch_device (include/snowgpu.h) -- or
    snowgpu_augment_wet_batch_device with `wet=...` (pointcloud_viewer.py:2807-2821) -- on the CALLER's current torch stream;
  * results are torch tensors on the same device, allocated by torch's caching allocator (no hipMalloc after the first call of a size);
  * `sync=False` returns a DeviceResult at once -- nothing has been waited for, the counts are still a device tensor -- so that the
    call can be chained in front of the consumer's kernels; `sync=True` (default) waits, checks the status words and returns
    the reference-shaped list of (stats, aug_pc) with aug_pc a view of the result tensor.

What crosses the link per call: the frame offsets, the n_frames x n_lasers table ids and the planes (a few KB, cached by value: a
training loop that reshuffles `order` per frame uploads 64 int32 per frame).  PyTorch is plumbing here -- device memory and streams;
no arithmetic of the simulation happens in this file.
"""
from __future__ import annotations

import os
import random
from collections import OrderedDict

import numpy as np

from . import _native


LANE_SLOT0 = 1000           # compute lane k is engine slot LANE_SLOT0 + k: lanes never share a context with plain calls (slot=)


def is_device_input(frames) -> bool:
    """True for a torch CUDA tensor, a DeviceBatch, or a non-empty sequence whose first element is a torch CUDA tensor."""
    if isinstance(frames, DeviceBatch):
        return True
    t = frames
    if isinstance(frames, (list, tuple)):
        if not frames:
            return False
        t = frames[0]
    return type(t).__module__.startswith("torch") and bool(getattr(t, "is_cuda", False))


class DeviceBatch:
    """Frames that lie back to back in ONE N_total x 5 CUDA tensor, with their (host) offsets -- the device twin of FlatBatch.
    `frame_rows=n`: every frame has n rows (a stack of sweeps); else `offsets` (n_frames + 1, host integers)."""

    def __init__(self, rows, offsets=None, frame_rows=None):
        if rows.dim() == 3:                                    # F x N x 5
            frame_rows = int(rows.shape[1])
            rows = rows.reshape(-1, rows.shape[2])
        if rows.dim() != 2 or rows.shape[1] != 5 or not rows.is_cuda or not rows.is_contiguous():
            raise ValueError("a DeviceBatch is a contiguous N x 5 (or F x N x 5) CUDA tensor")
        self.rows = rows
        if offsets is None:
            if not frame_rows or rows.shape[0] % int(frame_rows):
                raise ValueError("give `offsets`, or `frame_rows` dividing the row count")
            offsets = np.arange(rows.shape[0] // int(frame_rows) + 1, dtype=np.int64) * int(frame_rows)
        self.offsets = np.ascontiguousarray(offsets, np.int64)
        if self.offsets[0] != 0 or self.offsets[-1] > rows.shape[0] or np.any(np.diff(self.offsets) < 0):
            raise ValueError("bad frame offsets")

    def __len__(self):
        return len(self.offsets) - 1


class DeviceResult:
    """What an asynchronous device call leaves behind: `rows` (N_total x 5; the first counts[f] rows of frame f's slot
    [offsets[f], offsets[f + 1]) are its output), `src` (input row of every output row, frame-local), `counts` (n_frames, int64),
    `stats` (n_frames x 3: num_attenuated, num_removed, avg_intensity_diff -- simulation.py:516-538), `status` (8 int32 words),
    `flags` (fused wet ground: 1 where a frame had fewer than 1000 ground rows and came back as the snowfall result) -- all device
    tensors, all still being written until the stream the call was made on has caught up."""

    def __init__(self, ctx, rows, src, counts, stats, status, offsets, stream, flags=None, keep=()):
        self.ctx, self.rows, self.src, self.counts, self.stats, self.status = ctx, rows, src, counts, stats, status
        self.offsets, self.stream, self.flags = offsets, stream, flags
        self._keep = keep                                      # inputs of the call: alive until the result has been waited for

    def wait(self):
        """Wait for the call, raise what the status words say (as the host entry does) and drop the references to the inputs."""
        self.stream.synchronize()
        st = self.status.cpu().numpy()
        self._keep = ()
        self.ctx.check_status(st)                              # SnowGPUError with the library's message
        return self

    def join(self, stream=None):
        """torch's current stream (or `stream`) waits for the call; the host does not.  For results of a lane call (lane=k)."""
        import torch
        (stream or torch.cuda.current_stream(self.rows.device)).wait_stream(self.stream)
        return self

    def frames(self, return_src=False):
        """The reference-shaped result: [(stats, aug_pc)] (or (stats, aug_pc, src)), aug_pc / src views of the result tensors."""
        self.wait()
        counts = self.counts.cpu().numpy()
        stats = self.stats.cpu().numpy()
        out = []
        for i in range(len(self.offsets) - 1):
            a, n = int(self.offsets[i]), int(counts[i])
            st = (np.int64(stats[i, 0]), np.int64(stats[i, 1]), int(stats[i, 2]))
            out.append((st, self.rows[a:a + n], self.src[a:a + n]) if return_src else (st, self.rows[a:a + n]))
        return out


class _SmallUploads:
    """Device copies of the small per-call arrays (offsets, table ids, planes, polynomials), by value: a few KB each, least
    recently used first out."""

    def __init__(self, cap=64):
        self.cap, self.d = cap, OrderedDict()

    def get(self, torch, dev, arr, user):
        """The device copy of `arr`, safe to read on stream `user`: a copy made by an earlier call on another stream is waited for by event."""
        key = (str(dev), arr.dtype.str, arr.shape, arr.tobytes())
        hit = self.d.get(key)
        if hit is None:
            t = torch.from_numpy(np.ascontiguousarray(arr)).to(dev)                # (on torch's current stream)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            hit = self.d[key] = (t, ev)
            while len(self.d) > self.cap:
                self.d.popitem(last=False)
        else:
            self.d.move_to_end(key)
        if not hit[1].query():
            user.wait_event(hit[1])
        return hit[0]


def _uploads(eng):
    u = eng.__dict__.get("_small_uploads")
    if u is None:
        u = eng.__dict__["_small_uploads"] = _SmallUploads()
    return u


def _as_batch(torch, frames):
    """(rows N_total x 5 contiguous, host offsets, extras or None) from a DeviceBatch, an F x N x C tensor or a list of N_i x C tensors."""
    if isinstance(frames, DeviceBatch):
        return frames.rows, frames.offsets, None
    if not isinstance(frames, (list, tuple)):
        if frames.dim() == 3:
            frames = list(frames.unbind(0))
        else:
            frames = [frames]
    for f in frames:
        if f.dim() != 2 or f.shape[1] < 5:
            raise ValueError("pc must be N x 5 (x, y, z, intensity, channel)")
        if f.dtype != frames[0].dtype or f.device != frames[0].device:
            raise TypeError("all frames of a batch must share one dtype and device")
    if frames[0].dtype not in (torch.float32, torch.float64):
        frames = [f.to(torch.float64) for f in frames]
    offsets = np.zeros(len(frames) + 1, np.int64)
    offsets[1:] = np.cumsum([int(f.shape[0]) for f in frames])
    extra = any(f.shape[1] > 5 for f in frames)
    if len(frames) == 1 and frames[0].shape[1] == 5 and frames[0].is_contiguous():
        rows = frames[0]                                                          # read in place
    else:
        rows = torch.cat([f[:, :5] for f in frames]).contiguous()                 # device-to-device; never through the host
    return rows, offsets, (frames if extra else None)


def table_ids_for(eng, n_frames, particle_file_prefix, root_path, particles, orders, shuffle):
    """n_frames x n_lasers int32 device table ids: channel c of frame f reads line orders[f][c] + 1 (simulation.py:78, :482-486)."""
    from .tools.snowfall import simulation as _sim
    nl = eng.n_lasers
    if orders is None:
        orders = np.empty((n_frames, nl), np.int64)
        for f in range(n_frames):
            order = list(range(nl))                                               # simulation.py:483
            if shuffle:
                random.shuffle(order)                                             # simulation.py:485-486: Python's global generator
            orders[f] = order
    else:
        orders = np.asarray([list(o)[:nl] for o in orders] if not isinstance(orders, np.ndarray) else orders[:, :nl], np.int64)
        if orders.shape != (n_frames, nl):
            raise ValueError("orders must be n_frames permutations of the laser lines")
    if isinstance(particles, str):
        if particles not in ('device', 'missing'):
            raise ValueError("particles must be a sequence of tables, 'device' or 'missing'")
        ids = _sim._LazyFileIds(eng, particle_file_prefix, root_path, sample='all' if particles == 'device' else 'missing')
    else:
        ids = _sim._ArrayIds(eng, particles) if particles is not None else _sim._LazyFileIds(eng, particle_file_prefix, root_path)
    return np.ascontiguousarray(ids[orders.reshape(-1)].reshape(n_frames, nl), np.int32)


def augment_batch(frames, particle_file_prefix, beam_divergence, shuffle=True, noise_floor=0.7, root_path=None, *, planes=None,
                  orders=None, particles=None, thr_polys=None, device=None, return_src=False, slot=0, calib=None, pre_crop=False,
                  q8='first', plane_method='reference', plane_seed=0, plane_trials=1000, sync=True, wet=None, out=None, lane=None, **_ignored):
    """augment_batch() of tools/snowfall/simulation.py for torch CUDA tensors (see that docstring for the shared arguments).

    frames   a list of N_i x 5 CUDA tensors (concatenated on the device), an F x N x 5 tensor, one N x 5 tensor, or a DeviceBatch
             (read in place).  float32 or float64.
    sync     True: wait, check, return [(stats, aug_pc)] with aug_pc device tensors.  False: return a DeviceResult immediately
             (asynchronous on torch's current stream of the device).
    wet      optional dict of ground_water_augmentation()'s keyword arguments (water_height, pavement_depth, noise_floor,
             power_factor, flat_earth, delta, replace, plane): the wet-ground model runs behind the snowfall on the same stream
             (snowgpu_augment_wet_batch_device); the result rows are float64 then (wet_ground/augmentation.py:150).
    out      optional DeviceResult of an earlier call with the same shapes whose tensors are reused (no allocation at all).
    lane     None (default): the call is part of torch's current stream -- it starts when that stream gets there and whatever the caller
             queues behind it waits for it.  An integer k: the call runs on compute lane k -- an engine context of its own (streams,
             scratch, tables) with a torch stream of its own -- which waits for what the caller's stream holds NOW (the inputs) and
             nothing waits for it: the caller's stream goes on, a call on another lane may run beside this one (the memory-bound sort and
             compaction of one batch beside the latency-bound per-beam kernels of the other), and the result is claimed through the
             DeviceResult: .wait() (host), .join() (torch's current stream waits, the host does not).  Keep as many results alive as
             lanes in flight; `sync=True` with a lane is a plain synchronous call on that lane.  A lane's context runs every kernel of a
             batch on ONE stream (snowgpu_set_serial) of the context's own (snowgpu_lane_stream); lanes k, k + 1, k + 2 get streams of
             different priorities -- the HIP runtime keeps a queue pool per priority, so they never share a hardware queue (streams of one
             priority may, and then run one after the other).  Measured per 256-sweep batch: two lanes 3.8 ms against 4.0 for one batch at
             a time; with GPU_MAX_HW_QUEUES=32 in the environment before the process first touches the GPU all lanes run at one priority
             on queues of their own: three lanes 3.66 ms (profiles/r06_lanes_ab.txt).  Lanes are engine contexts of their own (slot
             LANE_SLOT0 + k): a plain call never runs on a lane's context.
    """
    import torch
    from . import engine as _engine
    if q8 != 'first':
        raise ValueError("q8='numpy' selects the histogram minima with the HOST's NumPy: it needs host arrays, not CUDA tensors")
    if calib is not None and pre_crop:
        raise ValueError("pre_crop is a step of the host entry (precompute.py:96-99); crop the tensors before the call")
    if plane_method not in _native.PLANE_METHODS:
        raise ValueError("plane_method must be 'reference', 'lsq' or 'ransac'")
    rows, offsets, extras = _as_batch(torch, frames)
    dev = rows.device
    if device is not None and int(device) != dev.index:
        raise ValueError(f"the tensors live on {dev}, device={device} was asked for")
    eng = _engine.get_engine(dev.index, slot if lane is None else LANE_SLOT0 + int(lane))
    if lane is not None and not eng.__dict__.get("_lane_serial"):
        # a compute lane: its batches overlap OTHER lanes' batches, not their own side streams -- one stream per lane (snowgpu_set_serial; with
        # GPU_MAX_HW_QUEUES >= 16 in the environment every lane's stream has a hardware queue of its own: include/snowgpu.h)
        eng.ctx.set_serial(True)
        eng.__dict__["_lane_serial"] = True
    nf, n = len(offsets) - 1, int(offsets[-1])
    if nf == 0:
        return []
    code = 0 if rows.dtype == torch.float32 else 1
    max_rows = int(np.diff(offsets).max())
    up = _uploads(eng)
    with torch.cuda.device(dev):
        tids = table_ids_for(eng, nf, particle_file_prefix, root_path, particles, orders, shuffle)
        if n == 0:                                   # nothing to simulate (the C ABI wants non-null buffers): empty frames come back empty
            zero = (np.int64(0), np.int64(0), 0)
            empty = [(zero, rows[:0], torch.empty(0, dtype=torch.int32, device=dev)) if return_src else (zero, rows[:0]) for _ in range(nf)]
            if sync:
                return empty
            z = lambda *shape, dt=torch.int64: torch.zeros(shape, dtype=dt, device=dev)   # noqa: E731
            return DeviceResult(eng.ctx, rows[:0], z(0, dt=torch.int32), z(nf), z(nf, 3), z(8, dt=torch.int32), offsets, torch.cuda.current_stream(dev))
        stream = torch.cuda.current_stream(dev)
        # The C ABI reads stream = NULL as "the context's own stream", which is not ordered against anything of torch's.  torch's legacy
        # default stream HAS the handle 0, so a call made on it runs on a side stream of the engine, forked from and joined back into the
        # default stream (two event waits): uploads queued before the call are seen, and whoever reads the results on the caller's stream
        # afterwards -- or synchronises it -- waits for the call.
        run = stream
        if stream.cuda_stream == 0 or lane is not None:
            run = eng.__dict__.get("_torch_side_stream")
            if run is None or run.device != dev:
                if lane is not None:
                    # lanes k, k + 1, k + 2 on streams of three different priorities: three queue pools of the runtime, so three hardware
                    # queues whatever else the process has created (include/snowgpu.h: snowgpu_lane_stream)
                    many = int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) >= 16      # (then every stream has a queue anyway: one priority)
                    order = [int(v) for v in os.environ.get("SNOWGPU_LANE_LEVELS", "1" if many else "2,1,0").split(",")]
                    run = torch.cuda.ExternalStream(eng.ctx.lane_stream(order[int(lane) % len(order)]), device=dev)
                else:
                    run = torch.cuda.Stream(device=dev)
                eng.__dict__["_torch_side_stream"] = run
        d_off = up.get(torch, dev, offsets, run)
        d_tids = up.get(torch, dev, tids, run)
        d_poly = d_plane = None
        if thr_polys is not None:
            d_poly = up.get(torch, dev, np.ascontiguousarray(thr_polys, np.float64).reshape(nf, 3), run)
        elif planes is not None:
            if isinstance(planes, np.ndarray) and planes.shape == (nf, 4):          # (wx, wy, wz, h) rows, as the C ABI takes them
                pl = np.ascontiguousarray(planes, np.float64)
            else:
                pl = np.asarray([[float(w[0]), float(w[1]), float(w[2]), float(h)] for w, h in planes], np.float64).reshape(nf, 4)
            d_plane = up.get(torch, dev, pl, run)
        d_wet_plane = None
        if wet is not None:
            wet = dict(wet)
            wp = wet.pop("plane", None)
            if wp is not None:
                wp = [wp] * nf if len(wp) == 2 and np.ndim(wp[1]) == 0 else wp
                d_wet_plane = up.get(torch, dev, np.asarray([[float(w[0]), float(w[1]), float(w[2]), float(h)] for w, h in wp], np.float64), run)
        out_dt = torch.float64 if wet is not None else rows.dtype
        if out is not None and out.rows.shape[0] >= n and out.rows.dtype == out_dt and out.rows.device == dev and out.counts.shape[0] == nf:
            o_rows, o_src, o_cnt, o_st, o_status, o_flags = out.rows, out.src, out.counts, out.stats, out.status, out.flags
        else:
            o_rows = torch.empty((n, 5), dtype=out_dt, device=dev)
            o_src = torch.empty(n, dtype=torch.int32, device=dev)
            o_cnt = torch.empty(nf, dtype=torch.int64, device=dev)
            o_st = torch.empty((nf, 3), dtype=torch.int64, device=dev)
            o_status = torch.empty(8, dtype=torch.int32, device=dev)
            o_flags = None
        if wet is not None and o_flags is None:
            o_flags = torch.empty(nf, dtype=torch.int32, device=dev)
        if run is not stream:
            run.wait_stream(stream)
            if lane is not None:                    # (tensors of the caller's stream used on the lane's: the allocator must know)
                for t in (rows, o_rows, o_src, o_cnt, o_st, o_status, o_flags):
                    if t is not None:
                        t.record_stream(run)
        ptr = lambda t: 0 if t is None else t.data_ptr()   # noqa: E731
        with eng.batch_lock:
            if calib is not None:
                eng.ctx.set_fov(calib, (1024, 1920))                                # simulation.py:536
            device_plane = d_poly is None and d_plane is None                       # calculate_plane (simulation.py:449) on the device
            if device_plane or (wet is not None and d_wet_plane is None):
                eng.ctx.set_plane_method(plane_method, seed=plane_seed, trials=plane_trials, min_rows=5)
            try:
                if wet is None:
                    eng.ctx.augment_batch_device(nf, n, max_rows, d_off.data_ptr(), rows.data_ptr(), code, d_tids.data_ptr(),
                                                 float(beam_divergence), ptr(d_poly), ptr(d_plane), float(noise_floor), 0, o_rows.data_ptr(),
                                                 o_src.data_ptr(), o_cnt.data_ptr(), o_st.data_ptr(), 0, o_status.data_ptr(), run.cuda_stream)
                else:
                    w = dict(water_height=0.001, pavement_depth=0.0012, noise_floor=0.7, power_factor=15, flat_earth=False, delta=0.5,
                             replace=True)
                    unknown = set(wet) - set(w) - {"estimation_method", "debug"}
                    if unknown:
                        raise TypeError(f"unknown wet-ground arguments: {sorted(unknown)}")
                    if wet.get("estimation_method", "linear") != "linear":
                        raise ValueError("the fused snowfall + wet-ground call fits estimation_method='linear'")
                    w.update({k: v for k, v in wet.items() if k in w})
                    eng.ctx.augment_wet_batch_device(nf, n, max_rows, d_off.data_ptr(), rows.data_ptr(), code, d_tids.data_ptr(),
                                                     float(beam_divergence), ptr(d_poly), ptr(d_plane), float(noise_floor), 0, ptr(d_wet_plane),
                                                     w["water_height"], w["pavement_depth"], w["noise_floor"], w["power_factor"],
                                                     w["flat_earth"], w["delta"], w["replace"], o_rows.data_ptr(), o_src.data_ptr(),
                                                     o_cnt.data_ptr(), o_st.data_ptr(), o_flags.data_ptr(), o_status.data_ptr(), run.cuda_stream)
            finally:
                if run is not stream and lane is None:
                    stream.wait_stream(run)
                if calib is not None:
                    eng.ctx.set_fov(None)
                if plane_method != 'reference':
                    eng.ctx.set_plane_method('reference')
    res = DeviceResult(eng.ctx, o_rows, o_src, o_cnt, o_st, o_status, offsets, stream if lane is None else run, flags=o_flags, keep=(rows, d_off, d_tids, d_poly, d_plane, d_wet_plane))
    if not sync:
        return res
    from .tools.snowfall.simulation import _raise_like_reference
    try:
        per_frame = res.frames(return_src=True)
    except _native.SnowGPUError as err:
        _raise_like_reference(err)
    results = []
    for i, (st, aug, src) in enumerate(per_frame):
        if extras is not None and extras[i].shape[1] > 5:        # further columns ride through (simulation.py:447, :508-523)
            aug = torch.cat((aug, extras[i][src.long(), 5:].to(aug.dtype)), dim=1)
        results.append((st, aug, src) if return_src else (st, aug))
    return results
