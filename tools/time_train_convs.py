"""Every conv launch of one HISFCOS-R50 training step (B = 16, 512 x 512; FD_AMP=1: under autocast, f16 activation maps), replayed one by one with HIP events:
time, algorithmic bytes (input map + output map + residual + weights at their stored widths) and FLOPs per launch, grouped by shape -- which of the step's
launches sit far from BOTH of their bounds (HBM 8 TB/s; MFMA 157 TFLOP/s fp32 / 2 500 f16).  usage (GPU box): [FD_AMP=1] python tools/time_train_convs.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_object_detection_amd import ops
from pytorch_object_detection_amd.model.loss import FCOSLoss
from pytorch_object_detection_amd.model.modules.head import FCOSGenTargets
from pytorch_object_detection_amd.model.od import HalfInvertedStageFCOS
dev = "cuda:0"
torch.manual_seed(0)
B = 16
AMP = os.environ.get("FD_AMP") == "1"
model = HalfInvertedStageFCOS([512, 1024, 2048], 20, 256).to(dev).train()
x = torch.randn(B, 3, 512, 512, device=dev)
c = torch.rand(B, 8, 2, device=dev) * 400 + 50
s = torch.rand(B, 8, 2, device=dev) * 150 + 20
gt = torch.cat([c - s / 2, c + s / 2], -1).clamp(0, 511)
labels = torch.randint(1, 21, (B, 8), device=dev)
gen = FCOSGenTargets([8, 16, 32, 64, 128], [[-1, 32], [32, 96], [96, 192], [192, 384], [384, 9999999]])
crit = FCOSLoss("giou")


def step():
    model.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.float16, enabled=AMP):
        out = model(x)
        loss = crit([out, gen([out, gt, labels])])[-1]
    (loss * (1024.0 if AMP else 1.0)).backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
ops.CONV_LOG = []
step()
torch.cuda.synchronize()
log, ops.CONV_LOG = ops.CONV_LOG, None
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
groups = {}
for run, d in log:
    for _ in range(2):
        run()
    e0.record()
    for _ in range(5):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    key = (d.get("kind", "conv"), d["Cin"], d["Cout"], d["k"], d["stride"], d["dil"], d["rows_in"], d["rows_out"], d["io"], d["precision"], d["res"], getattr(run.params, "tile", -1))
    g = groups.setdefault(key, [0, 0.0, d])
    g[0] += 1
    g[1] += ms
peak_tf = 2500.0 if AMP else 157.3
print(f"# {'AMP f16' if AMP else 'fp32'} training step: {len(log)} conv + weight-gradient launches, conv {sum(g[1] for k, g in groups.items() if k[0] == 'conv'):.2f} ms, weight gradients {sum(g[1] for k, g in groups.items() if k[0] == 'wgrad'):.2f} ms replayed one by one")
print("# kind Cin Cout k s d rows_in rows_out io prec res tile | n  ms_total  us_each  GB/s  frac_hbm  TFLOP/s  frac_mfma  bound_us(max of the two)  us_each/bound")
for key, (n, ms, d) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
    us = ms / n * 1e3
    gbs = d["bytes"] / us / 1e3
    tf = d["flops"] / us / 1e6
    bound = max(d["bytes"] / 8e6, d["flops"] / (peak_tf * 1e6))
    print(key[0], " ".join(str(int(v)) for v in key[1:]), f"| {n} {ms:.3f} {us:.1f} {gbs:.0f} {gbs / 8000:.2f} {tf:.1f} {tf / peak_tf:.3f} {bound:.1f} {us / bound:.2f}")
