"""The F(4x4) layers of the batch-16 640 x 640 plan: plain launch against the persistent stream-K form (fd_conv_params.sk_wgs) at several grid sizes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_object_detection_amd import ops, _lib
from pytorch_object_detection_amd._lib import Segs
dev = "cuda:0"


def timeit(call, reps=20):
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps


def bench(name, B, hw, Cin, Cout, dil=1, grids=(256,)):
    segs = Segs.make(B, hw)
    x = ops.Rows(torch.randn(segs.rows, Cin, device=dev))
    y0, y1 = ops.new_rows(segs.rows, Cout, dev), ops.new_rows(segs.rows, Cout, dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) / (Cin * 9) ** 0.5
    wp = ops.pack_conv_weight_wino4(w)
    kw = dict(Cin=Cin, Cout=Cout, k=3, pad=dil, dil=dil, tile=_lib.WINO4_TILE)
    c0 = ops.conv_call(x, segs, wp, y0, **kw)
    m0 = timeit(c0)
    out = [f"{name}: plain {m0:.4f} ms ({ops.conv_workgroups(c0)[1]} items)"]
    for wgs in grids:
        ws = ops.sk_workspace(wgs, dev)
        try:
            c1 = ops.conv_call(x, segs, wp, y1, sk_wgs=wgs, workspace=ws, **kw)
            m1 = timeit(c1)
        except Exception as e:
            out.append(f"sk{wgs} n/a"); continue
        err = (y0.tensor() - y1.tensor()).abs().max().item()
        out.append(f"sk{wgs} {m1:.4f} ({m0 / m1:.2f}x, max|diff| {err:.1e})")
    print(" | ".join(out), flush=True)


pyr = [(80, 80), (40, 40), (20, 20), (10, 10), (5, 5)]
G = (256, 248, 240)
bench("tower 256>512 pyramid B16", 16, pyr, 256, 512, grids=G)
bench("HisBlock3.conv3 256>128 80x80", 16, [(80, 80)], 256, 128, grids=G)
bench("layer3.conv2 256>256 40x40", 16, [(40, 40)], 256, 256, grids=G)
bench("layer2.conv2 128>128 80x80", 16, [(80, 80)], 128, 128, grids=G)
bench("layer1.conv2 64>64 160x160", 16, [(160, 160)], 64, 64, grids=G)
bench("layer4.conv2 512>512 20x20", 16, [(20, 20)], 512, 512, grids=G)
bench("cls_logits 256>80 pyramid", 16, pyr, 256, 80, grids=G)
bench("HisBlock3.conv4 256>256 80x80 dil2", 16, [(80, 80)], 256, 256, 2, grids=G)
bench("HisBlock2.conv4 256>256 40x40 dil2", 16, [(40, 40)], 256, 256, 2, grids=G)
