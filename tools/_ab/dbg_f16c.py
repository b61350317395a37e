import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorch_object_detection_amd import ops, _lib, train_ops as T
from pytorch_object_detection_amd.model.backbone.resnet50 import _Bottleneck
DEV = "cuda:0"
Cin, P, stride, ds = 512, 256, 2, True
torch.manual_seed(Cin + P + stride)
blk = _Bottleneck(Cin, P, stride, ds).to(DEV)
for m in blk.modules():
    if isinstance(m, torch.nn.BatchNorm2d):
        m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.1); m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
        m.weight.requires_grad_(False); m.bias.requires_grad_(False)
blk.eval()
B, H, W = 2, 18, 14
x0 = torch.randn(B, Cin, H, W, device=DEV).to(memory_format=torch.channels_last)
gy = torch.randn(B, 4 * P, (H - 1) // stride + 1, (W - 1) // stride + 1, device=DEV).to(memory_format=torch.channels_last) * 64.0
# capture intermediates by wrapping _conv_launch / _strided_dgrad
log = {}
orig_launch, orig_sd = T._conv_launch, T._strided_dgrad
def mk(store):
    cnt = {"n": 0}
    def launch(x, segs, wp, y, **kw):
        orig_launch(x, segs, wp, y, **kw)
        log[(store, "conv", cnt["n"])] = (y.detach().float().clone(), kw.get("k"), tuple(y.shape), x.dtype, y.dtype); cnt["n"] += 1
    def sd(g, *a, **kw):
        r = orig_sd(g, *a, **kw)
        log[(store, "sd", cnt["n"])] = (r.detach().float().clone(), a[3] if len(a) > 3 else None, tuple(r.shape), g.dtype, r.dtype); cnt["n"] += 1
        return r
    return launch, sd
for xdt in (torch.float32, torch.float16):
    log.clear()
    for store in (False, True):
        T.AMP_F16_STORE = store
        T._conv_launch, T._strided_dgrad = mk(store)
        blk.zero_grad(set_to_none=True)
        x = x0.to(xdt).detach().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.float16):
            y = T.bottleneck(blk, x)
        y.backward(gy.to(y.dtype))
    T._conv_launch, T._strided_dgrad = orig_launch, orig_sd
    keys = sorted(k[1:] for k in log if k[0] is True)
    for k in keys:
        a, b = log[(True,) + k], log[(False,) + k]
        d = (a[0] - b[0]).abs()
        print(xdt, k, "k", a[1], a[2], "dtypes", a[3], a[4], "| max", float(d.max()), "of", float(b[0].abs().max()), "n_bad", int((d > 0.02 * b[0].abs().max()).sum()), flush=True)
    a, b = log[(True, "conv", 4)][0], log[(False, "conv", 4)][0]
    y2a, y2b = log[(True, "conv", 1)][0], log[(False, "conv", 1)][0]
    d = (a - b).abs()
    idx = (d > 0.02 * b.abs().max()).nonzero()[:12]
    for i, j in idx.tolist():
        print("  bad", i, j, "f16-store", float(a[i, j]), "f32-store", float(b[i, j]), "| y2", float(y2a[i, j]), float(y2b[i, j]))
    print("  zero-vs-nonzero flips:", int(((a == 0) != (b == 0)).sum()))
