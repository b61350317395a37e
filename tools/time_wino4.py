"""3x3 stride-1 dilation-1 convs of the bench model: Winograd F(2x2, 3x3) (FD_TILE_WINOGRAD) against F(4x4, 3x3) (FD_TILE_WINOGRAD4)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_object_detection_amd import ops, _lib
from pytorch_object_detection_amd._lib import Segs
dev = "cuda:0"


def timeit(call, reps=10):
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps


def bench(name, B, hw, Cin, Cout, dil=1):
    segs = Segs.make(B, hw)
    x = ops.Rows(torch.randn(segs.rows, Cin, device=dev))
    y2, y4 = ops.new_rows(segs.rows, Cout, dev), ops.new_rows(segs.rows, Cout, dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) / (Cin * 9) ** 0.5
    fl = 2 * segs.rows * Cout * Cin * 9
    c2 = ops.conv_call(x, segs, ops.pack_conv_weight_wino(w), y2, Cin=Cin, Cout=Cout, k=3, pad=dil, dil=dil, tile=_lib.WINO_TILE)
    c4 = ops.conv_call(x, segs, ops.pack_conv_weight_wino4(w), y4, Cin=Cin, Cout=Cout, k=3, pad=dil, dil=dil, tile=_lib.WINO4_TILE)
    m2, m4 = timeit(c2), timeit(c4)
    err = (y2.tensor() - y4.tensor()).abs().max().item()
    print(f"{name}: F(2x2) {m2:.4f} ms {fl / m2 / 1e9:.1f} TF/s-eq | F(4x4) {m4:.4f} ms {fl / m4 / 1e9:.1f} TF/s-eq ({m2 / m4:.2f}x) max|diff| {err:.2e}", flush=True)


pyr = [(80, 80), (40, 40), (20, 20), (10, 10), (5, 5)]
bench("tower 256>512 pyramid B16", 16, pyr, 256, 512)
bench("HisBlock3.conv3 256>128 80x80", 16, [(80, 80)], 256, 128)
bench("layer3.conv2 256>256 40x40", 16, [(40, 40)], 256, 256)
bench("layer2.conv2 128>128 80x80", 16, [(80, 80)], 128, 128)
bench("layer1.conv2 64>64 160x160", 16, [(160, 160)], 64, 64)
bench("layer4.conv2 512>512 20x20", 16, [(20, 20)], 512, 512)
bench("cls_logits 256>80 pyramid", 16, pyr, 256, 80)
bench("cnt_reg 256>8 pyramid", 16, pyr, 256, 8)
bench("HisBlock3.conv4 256>256 80x80 dil2", 16, [(80, 80)], 256, 256, 2)
bench("HisBlock2.conv4 256>256 40x40 dil2", 16, [(40, 40)], 256, 256, 2)
