"""Per-parameter deviation of the HIP train step's gradients from the CPU oracle's autograd (the comparison of
tests/test_train_gpu.py::test_train_step_matches_oracle_autograd), printed for the direct and the Winograd conv path."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import torch_ref as R
from pytorch_object_detection_amd import engine
from pytorch_object_detection_amd.model.loss import FCOSLoss
from pytorch_object_detection_amd.model.modules.head import FCOSGenTargets
from pytorch_object_detection_amd.model.od import HalfInvertedStageFCOS
DEV = "cuda:0"
strides = [8, 16, 32, 64, 128]
ranges = [[-1, 32], [32, 96], [96, 192], [192, 384], [384, 9999999]]
gt = torch.tensor([[[10., 12., 60., 70.], [30., 30., 120., 110.], [-1, -1, -1, -1]], [[5., 5., 25., 30.], [0., 0., 127., 127.], [64., 20., 100., 90.]]])
labels = torch.tensor([[3, 7, -1], [1, 20, 12]])
for wino in [bool(int(c)) for c in os.environ.get("ORDER", "1001")]:
    engine.WINOGRAD = wino
    torch.manual_seed(0)
    model = HalfInvertedStageFCOS([512, 1024, 2048], 20, 256)
    gen = torch.Generator().manual_seed(1)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=gen) * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=gen) * 0.5 + 0.75)
    x = torch.randn(2, 3, 128, 128)
    sd = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in model.state_dict().items()}
    outs = R.hisfcos_forward(sd, x)
    tg = R.gen_targets([tuple(o.shape[2:]) for o in outs[0]], strides, ranges, gt, labels)
    R.fcos_loss(outs, tg, "giou")[3].backward()
    model.freeze_all_bn = True
    model.to(DEV).train()
    out = model(x.to(DEV))
    fe = max(float((a.cpu() - b).abs().max()) for o, r in zip(out, outs) for a, b in zip(o, r))
    target = FCOSGenTargets(strides, ranges)([out, gt.to(DEV), labels.to(DEV)])
    FCOSLoss("giou")([out, target])[-1].backward()
    rows = []
    for name, p in model.named_parameters():
        if p.grad is None or sd[name].grad is None:
            continue
        a, b = p.grad.cpu().double().flatten(), sd[name].grad.double().flatten()
        scale = float(b.abs().max()) + 1e-30
        if scale < 1e-12:
            continue
        d = (a - b).abs() / scale
        rows.append((float(d.median()), float(d.max()), name, a.numel()))
    rows.sort(reverse=True)
    print(f"== winograd={wino}: forward max |out - oracle| = {fe:.2e}; {len(rows)} gradients; medians > 5e-5: {sum(r[0] > 5e-5 for r in rows)}, > 2e-4: {sum(r[0] > 2e-4 for r in rows)}")
    for med, mx, name, n in rows[:4]:
        print(f"   median {med:.2e}  max {mx:.2e}  {name} ({n})")
