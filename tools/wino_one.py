"""One Winograd launch shape, a few launches (for rocprofv3 --pmc passes): python3 tools/wino_one.py [tower|l3|l1|hb3] [2|4]   (F(2x2) / F(4x4)); ACT=1: with the
folded-BatchNorm scale / shift and ReLU of the trunk's conv2 layers (the epilogue path of DESIGN 4.1n)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_object_detection_amd import ops, _lib
from pytorch_object_detection_amd._lib import Segs
dev = "cuda:0"
which = sys.argv[1] if len(sys.argv) > 1 else "tower"
pyr = [(80, 80), (40, 40), (20, 20), (10, 10), (5, 5)]
B, hw, Cin, Cout, dil = {"tower": (16, pyr, 256, 512, 1), "l3": (16, [(40, 40)], 256, 256, 1), "l1": (16, [(160, 160)], 64, 64, 1),
                         "hb3": (16, [(80, 80)], 256, 256, 2)}[which]
segs = Segs.make(B, hw)
x = ops.Rows(torch.randn(segs.rows, Cin, device=dev))
y = ops.new_rows(segs.rows, Cout, dev)
w = torch.randn(Cout, Cin, 3, 3, device=dev) / (Cin * 9) ** 0.5
f4 = len(sys.argv) > 2 and sys.argv[2] == "4"
call = ops.conv_call(x, segs, ops.pack_conv_weight_wino4(w) if f4 else ops.pack_conv_weight_wino(w), y, Cin=Cin, Cout=Cout, k=3, pad=dil, dil=dil,
                     tile=_lib.WINO4_TILE if f4 else _lib.WINO_TILE,
                     **(dict(scale=torch.rand(Cout, device=dev) + 0.5, shift=torch.randn(Cout, device=dev), act=1) if os.environ.get("ACT") == "1" else {}))
for _ in range(6):
    call()
torch.cuda.synchronize()
print("done", which, 2 * segs.rows * Cout * Cin * 9 / 1e9, "GFLOP per launch")
