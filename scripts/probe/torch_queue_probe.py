"""Are PyTorch's pool streams concurrent under the runtime's default of four hardware queues?  torch.cuda._sleep spins one block for a number
of cycles; k streams that run it at once take one kernel time if each has a queue of its own, k times that if they share one.
usage: python scripts/probe/torch_queue_probe.py [engines]   (engines: create that many engine contexts first, as bench.py --lanes does)"""
import sys, time
import torch
sys.path.insert(0, ".")
n_eng = int(sys.argv[1]) if len(sys.argv) > 1 else 0
if n_eng:
    from lidar_snow_sim_amd import engine
    for k in range(n_eng):
        engine.get_engine(0, 1000 + k)
dev = torch.device("cuda:0")
streams = [torch.cuda.Stream(device=dev) for _ in range(8)]
cyc = 5_000_000
def run(ids, reps=3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for i in ids:
            with torch.cuda.stream(streams[i]):
                torch.cuda._sleep(cyc)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
run([0])
one = run([0])
print(f"engines={n_eng}; one stream: {one:.2f} ms")
for ids in ([0, 1], [0, 1, 2], [0, 1, 2, 3], [1, 2, 3], [2, 3, 4], [0, 2, 4], [4, 5, 6], [0, 1, 2, 3, 4, 5, 6, 7]):
    print(ids, f"{run(ids) / one:.2f} x")
