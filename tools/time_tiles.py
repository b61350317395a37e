"""Time one conv shape under every tile id (diagnostic).  usage: python tools/time_tiles.py [f32|f16x3]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_object_detection_amd import ops, _lib
from pytorch_object_detection_amd._lib import Segs
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
dev = "cuda:0"
B = 16
hw = [(80, 80), (40, 40), (20, 20), (10, 10), (5, 5)]
segs = Segs.make(B, hw)
Cin, Cout = 256, 512
x = ops.Rows(torch.randn(segs.rows, Cin, device=dev))
w = torch.randn(Cout, Cin, 3, 3, device=dev) / 48
wp = ops.pack_conv_weight_f16x3(w) if prec == "f16x3" else ops.pack_conv_weight(w)
y = ops.new_rows(segs.rows, Cout, dev)
flops = 2 * segs.rows * Cout * Cin * 9
only = [int(v) for v in os.environ.get('FD_TILES', '').split(',') if v]
for tile in (only or sorted(_lib.TILES)):
    if tile in (5, 6):
        continue
    run = ops.conv_call(x, segs, wp, y, Cin=Cin, Cout=Cout, k=3, pad=1, tile=tile, precision=1 if prec == "f16x3" else 0)
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{prec} tile {tile:2d} {_lib.TILES[tile]}: {ms:.3f} ms  {flops / ms / 1e9:.1f} TFLOP/s-equivalent")
