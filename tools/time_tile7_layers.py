"""The two 1x1 layers the tile table gives FD_TILE_128x128_SB (which spills after the epilogue diet): every tile on both shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_object_detection_amd import ops, _lib
from pytorch_object_detection_amd._lib import Segs
dev = "cuda:0"
pyr = [(80, 80), (40, 40), (20, 20), (10, 10), (5, 5)]
for name, hw, Cin, Cout, act in [("layer1.0.downsample", [(160, 160)], 64, 256, 0), ("head.pw1", pyr, 256, 512, 0)]:
    segs = Segs.make(16, hw)
    x = ops.Rows(torch.randn(segs.rows, Cin, device=dev))
    w = torch.randn(Cout, Cin, 1, 1, device=dev) / Cin ** 0.5
    wp = ops.pack_conv_weight(w)
    y = ops.new_rows(segs.rows, Cout, dev)
    sc, sf = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev)
    res = {}
    for rnd in range(2):
        for tile in (1, 2, 3, 4, 7, 8, 9):
            run = ops.conv_call(x, segs, wp, y, Cin=Cin, Cout=Cout, k=1, scale=sc, shift=sf, act=act, tile=tile)
            for _ in range(3): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): run()
            e1.record(); e1.synchronize()
            res.setdefault(tile, []).append(e0.elapsed_time(e1) / 30 * 1e3)
    print(name, {t: [round(v, 1) for v in vs] for t, vs in res.items()})
