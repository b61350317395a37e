"""Head-tower-shaped 3x3 convs: the generic kernel's best tile against FD_TILE_128x128_PATCH (patch staged in LDS)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_object_detection_amd import ops, _lib
from pytorch_object_detection_amd._lib import Segs
dev = "cuda:0"
def bench(name, B, hw, Cin, Cout, dil, tiles, prec):
    segs = Segs.make(B, hw)
    x = ops.Rows(torch.randn(segs.rows, Cin, device=dev))
    y = ops.new_rows(segs.rows, Cout, dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) / 48
    wp = (ops.pack_conv_weight_f16x3 if prec else ops.pack_conv_weight)(w)
    fl = 2 * segs.rows * Cout * Cin * 9
    for t in tiles:
        try:
            call = ops.conv_call(x, segs, wp, y, Cin=Cin, Cout=Cout, k=3, pad=dil, dil=dil, tile=t, precision=prec)
            for _ in range(3): call()
        except Exception as e:
            print(name, "tile", t, "n/a:", str(e)[:60]); continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): call()
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{name} prec={'f16x3' if prec else 'f32'} tile {t}: {ms:.4f} ms  {fl / ms / 1e9:.1f} TFLOP/s")
pyr = [(80, 80), (40, 40), (20, 20), (10, 10), (5, 5)]
for prec in (0, 1):
    bench("tower 256>512 pyramid B16", 16, pyr, 256, 512, 1, (7, 1, 13), prec)
    bench("HisBlock3.conv3 256>128 80x80", 16, [(80, 80)], 256, 128, 1, (9, 4, 13), prec)
    bench("layer3.conv2 256>256 40x40", 16, [(40, 40)], 256, 256, 1, (4, 9, 13), prec)
    bench("layer2.conv2 128>128 80x80", 16, [(80, 80)], 128, 128, 1, (9, 8, 13), prec)
    bench("HisBlock2.conv4 256>256 d2 40x40", 16, [(40, 40)], 256, 256, 2, (4, 9, 13), prec)
    bench("cls_logits 256>80 pyramid", 16, pyr, 256, 80, 1, (12, 13), prec)
