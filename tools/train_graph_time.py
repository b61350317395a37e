"""One HISFCOS-R50 training step (B = 16, 512 x 512: forward, loss, backward, fused SGD; FD_AMP=1: autocast + GradScaler) captured as ONE HIP graph and replayed, against the
same step enqueued eagerly -- the AMP step's ~1 050 launches are 16 ms of kernels behind 15 - 20 ms of Python / autograd / ctypes work (tools/train_host_time.py)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_object_detection_amd.model.loss import FCOSLoss
from pytorch_object_detection_amd.model.modules.head import FCOSGenTargets
from pytorch_object_detection_amd.model.od import HalfInvertedStageFCOS
dev = "cuda:0"
torch.manual_seed(0)
B = 16
AMP = os.environ.get("FD_AMP") == "1"
model = HalfInvertedStageFCOS([512, 1024, 2048], 20, 256).to(dev).train()
opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=1e-3, momentum=0.9, weight_decay=1e-4, fused=True)
x = torch.randn(B, 3, 512, 512, device=dev)
c = torch.rand(B, 8, 2, device=dev) * 400 + 50
s = torch.rand(B, 8, 2, device=dev) * 150 + 20
gt = torch.cat([c - s / 2, c + s / 2], -1).clamp(0, 511)
labels = torch.randint(1, 21, (B, 8), device=dev)
gen = FCOSGenTargets([8, 16, 32, 64, 128], [[-1, 32], [32, 96], [96, 192], [192, 384], [384, 9999999]])
crit = FCOSLoss("giou")
scaler = torch.amp.GradScaler("cuda", enabled=AMP)


def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.float16, enabled=AMP, cache_enabled=False):
        out = model(x)
        losses = crit([out, gen([out, gt, labels])])
    scaler.scale(losses[-1]).backward()
    scaler.step(opt)
    scaler.update()
    return losses[-1].detach()


def timeit(f, n=20):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(4):
        l_eager = step()
torch.cuda.current_stream().wait_stream(side)
print(f"eager: {timeit(step):.2f} ms / step   loss {float(l_eager):.4f}")
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    l_graph = step()
g.replay()
torch.cuda.synchronize()
print(f"graph: {timeit(g.replay):.2f} ms / step   loss {float(l_graph):.4f}   scale {scaler.get_scale() if AMP else 1.0}")
