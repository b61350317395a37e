"""Host-side cost of enqueueing one 16-image step (plan.run() without a synchronize): how much of a CPU core one rank needs to keep the GPU fed."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_object_detection_amd.model.od import HalfInvertedStageFCOS
dev = "cuda:0"
torch.manual_seed(0)
model = HalfInvertedStageFCOS([512, 1024, 2048], 80, 256).eval().to(dev)
x = torch.randn(16, 3, 640, 640, device=dev)
for _ in range(3):
    model(x)
torch.cuda.synchronize()
built = next(iter(model._plans.values()))[1]
plan = [o for o in (built if isinstance(built, tuple) else (built,)) if hasattr(o, "steps")][0]
ts = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    plan.run()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ts.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
ts.sort()
print(f"{len(plan.steps)} launches per step: host enqueue {ts[len(ts)//2][0]:.2f} ms, step wall {ts[len(ts)//2][1]:.2f} ms (one batch in flight)")
