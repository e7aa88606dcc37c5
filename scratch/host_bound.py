"""Is the graph-replayed step bound by the host (hipGraphLaunch) or the GPU?  Host time per step to ENQUEUE vs time to complete."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py"]
import bench
job = bench.Job(1, 20, "bf16", torch.device("cuda:0"), 1, 0, use_graphs=True)
for _ in range(10):
    job.step()
torch.cuda.synchronize()
for trial in range(3):
    t0 = time.perf_counter()
    for _ in range(50):
        job.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("enqueue %.3f ms/step   complete %.3f ms/step   (GPU still busy %.1f ms after the last enqueue)" % ((t1 - t0) * 20, (t2 - t0) * 20, (t2 - t1) * 1e3))
# per-graph replay cost on the host
st = job.mw._graph_state
for name in ("gd", "gf", "gg"):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st[name].replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s: host %.3f ms, total %.3f ms" % (name, (t1 - t0) * 1e3, (t2 - t0) * 1e3))
