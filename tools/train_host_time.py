"""Host-side cost of one HISFCOS-R50 training step (B = 16, 512 x 512): wall time of step() WITHOUT a synchronize against the synchronized step time -- whether the
step is bound by the GPU or by the Python / autograd / ctypes work that enqueues its ~1 100 launches.  FD_AMP=1: under autocast + GradScaler."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_object_detection_amd import ops
from pytorch_object_detection_amd.model.loss import FCOSLoss
from pytorch_object_detection_amd.model.modules.head import FCOSGenTargets
from pytorch_object_detection_amd.model.od import HalfInvertedStageFCOS
dev = "cuda:0"
torch.manual_seed(0)
B = 16
model = HalfInvertedStageFCOS([512, 1024, 2048], 20, 256).to(dev).train()
opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=1e-3, momentum=0.9, weight_decay=1e-4)
x = torch.randn(B, 3, 512, 512, device=dev)
c = torch.rand(B, 8, 2, device=dev) * 400 + 50
s = torch.rand(B, 8, 2, device=dev) * 150 + 20
gt = torch.cat([c - s / 2, c + s / 2], -1).clamp(0, 511)
labels = torch.randint(1, 21, (B, 8), device=dev)
gen = FCOSGenTargets([8, 16, 32, 64, 128], [[-1, 32], [32, 96], [96, 192], [192, 384], [384, 9999999]])
crit = FCOSLoss("giou")
AMP = os.environ.get("FD_AMP") == "1"
scaler = torch.amp.GradScaler("cuda", enabled=AMP)
def step(phase=None):
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.float16, enabled=AMP):
        out = model(x)
        losses = crit([out, gen([out, gt, labels])])
    t1 = time.perf_counter()
    scaler.scale(losses[-1]).backward()
    t2 = time.perf_counter()
    scaler.step(opt)
    scaler.update()
    t3 = time.perf_counter()
    if phase is not None:
        phase.append((t1 - t0, t2 - t1, t3 - t2))
for _ in range(3):
    step()
torch.cuda.synchronize()
ph, walls = [], []
for _ in range(6):
    torch.cuda.synchronize()
    t = time.perf_counter()
    l0 = ops.LAUNCHES[0]
    step(ph)
    th = time.perf_counter() - t
    torch.cuda.synchronize()
    walls.append((th, time.perf_counter() - t, ops.LAUNCHES[0] - l0))
walls.sort()
th, tw, nl = walls[len(walls) // 2]
f, b, o = [sorted(p[i] for p in ph)[len(ph) // 2] * 1e3 for i in range(3)]
print(f"{'AMP ' if AMP else 'fp32'} step: host enqueue {th * 1e3:.1f} ms (forward+loss {f:.1f}, backward {b:.1f}, optimizer {o:.1f}), synchronized {tw * 1e3:.1f} ms, {nl} library launches")
