"""3x3 stride-1 convs of the bench model: the direct implicit-GEMM kernel's best tile against FD_TILE_WINOGRAD."""
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


def bench(name, B, hw, Cin, Cout, dil, tiles):
    segs = Segs.make(B, hw)
    x = ops.Rows(torch.randn(segs.rows, Cin, device=dev))
    y = ops.new_rows(segs.rows, Cout, dev)
    y2 = ops.new_rows(segs.rows, Cout, dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) / (Cin * 9) ** 0.5
    fl = 2 * segs.rows * Cout * Cin * 9
    best = None
    for t in tiles:
        call = ops.conv_call(x, segs, ops.pack_conv_weight(w), y, Cin=Cin, Cout=Cout, k=3, pad=dil, dil=dil, tile=t)
        ms = timeit(call)
        if best is None or ms < best[0]:
            best = (ms, t)
    call = ops.conv_call(x, segs, ops.pack_conv_weight_wino(w), y2, Cin=Cin, Cout=Cout, k=3, pad=dil, dil=dil, tile=_lib.WINO_TILE)
    ms = timeit(call)
    err = (y.tensor() - y2.tensor()).abs().max().item()
    print(f"{name}: direct tile {best[1]} {best[0]:.4f} ms {fl / best[0] / 1e9:.1f} TF/s | winograd {ms:.4f} ms {fl / ms / 1e9:.1f} TF/s-equivalent "
          f"({best[0] / ms:.2f}x) max|diff| {err:.2e}", flush=True)


pyr = [(80, 80), (40, 40), (20, 20), (10, 10), (5, 5)]
if os.environ.get("WINO_SHORT"):
    bench("tower 256>512 pyramid B16", 16, pyr, 256, 512, 1, (7,))
    bench("HisBlock3.conv4 256>256 d2 80x80", 16, [(80, 80)], 256, 256, 2, (9,))
    bench("layer3.conv2 256>256 40x40", 16, [(40, 40)], 256, 256, 1, (4,))
    bench("layer1.conv2 64>64 160x160", 16, [(160, 160)], 64, 64, 1, (8,))
    bench("cls_logits 256>80 pyramid", 16, pyr, 256, 80, 1, (12,))
    sys.exit(0)
bench("tower 256>512 pyramid B16", 16, pyr, 256, 512, 1, (7, 1))
bench("HisBlock3.conv4 256>256 d2 80x80", 16, [(80, 80)], 256, 256, 2, (9, 7))
bench("HisBlock3.conv3 256>128 80x80", 16, [(80, 80)], 256, 128, 1, (9, 4))
bench("HisBlock2.conv4 256>256 d2 40x40", 16, [(40, 40)], 256, 256, 2, (4, 9))
bench("HisBlock2.conv3 256>128 40x40", 16, [(40, 40)], 256, 128, 1, (4, 9))
bench("HisBlock1.conv4 256>256 d2 20x20", 16, [(20, 20)], 256, 256, 2, (4, 9))
bench("layer4.conv2 512>512 20x20", 16, [(20, 20)], 512, 512, 1, (4, 9))
bench("layer3.conv2 256>256 40x40", 16, [(40, 40)], 256, 256, 1, (4, 9))
bench("layer2.conv2 128>128 80x80", 16, [(80, 80)], 128, 128, 1, (9, 8))
bench("layer1.conv2 64>64 160x160", 16, [(160, 160)], 64, 64, 1, (8, 4))
bench("cls_logits 256>80 pyramid", 16, pyr, 256, 80, 1, (12,))
bench("FCOS-B3 tower 256>256 104x168+..", 4, [(104, 168), (52, 84), (26, 42), (13, 21), (7, 11)], 256, 256, 1, (7, 9))
