"""Diagnostic: wall time of one HISFCOS-R50 training step (Cfg4 shape: batch 16, 512x512, 20 classes, M = 8 GT boxes):
forward (HIP convs) -> FCOSGenTargets -> FCOSLoss('giou') -> backward (HIP dgrad / wgrad) -> SGD.
FD_TRAIN_STOCK_CONV=1 routes the convolutions to the stock PyTorch-ROCm op for comparison."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_object_detection_amd.model.loss import FCOSLoss
from pytorch_object_detection_amd.model.modules.head import FCOSGenTargets
from pytorch_object_detection_amd.model.od import HalfInvertedStageFCOS
dev = "cuda:0"
torch.manual_seed(0)
B = int(os.environ.get("B", "16"))
model = HalfInvertedStageFCOS([512, 1024, 2048], 20, 256).to(dev).train()
opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=1e-3, momentum=0.9, weight_decay=1e-4)
x = torch.randn(B, 3, 512, 512, device=dev)
c = torch.rand(B, 8, 2, device=dev) * 400 + 50
s = torch.rand(B, 8, 2, device=dev) * 150 + 20
gt = torch.cat([c - s / 2, c + s / 2], -1).clamp(0, 511)
labels = torch.randint(1, 21, (B, 8), device=dev)
gen = FCOSGenTargets([8, 16, 32, 64, 128], [[-1, 32], [32, 96], [96, 192], [192, 384], [384, 9999999]])
crit = FCOSLoss("giou")
AMP = os.environ.get("FD_AMP") == "1"        # FD_AMP=1: the step under torch.autocast(float16) + GradScaler (train.py:175-181)
scaler = torch.amp.GradScaler("cuda", enabled=AMP)
def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.float16, enabled=AMP):
        out = model(x)
        losses = crit([out, gen([out, gt, labels])])
    scaler.scale(losses[-1]).backward()
    scaler.step(opt)
    scaler.update()
    return losses
for _ in range(2):
    l = step()
torch.cuda.synchronize()
t = time.perf_counter()
n = 5
for _ in range(n):
    l = step()
torch.cuda.synchronize()
el = (time.perf_counter() - t) / n
print(f"train step B={B} 512x512 {'AMP f16 ' if AMP else ''}({'stock' if os.environ.get('FD_TRAIN_STOCK_CONV') == '1' else 'HIP'} convs): {el * 1e3:.1f} ms "
      f"-> {B / el:.1f} img/s; losses {[round(float(v), 4) for v in l]}")
