import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorch_object_detection_amd import ops, _lib
from pytorch_object_detection_amd._lib import Segs
DEV = "cuda:0"
Cin = Cout = 64
segs = Segs.make(1, [(8, 16)])
x = (torch.arange(segs.rows).view(-1, 1) * 1.0 + torch.arange(Cin).view(1, -1) / 128.0).to(DEV)     # row + channel/128: exactly representable in f16
w = torch.eye(Cout, Cin).view(Cout, Cin, 1, 1)
wp = ops.pack_conv_weight_hip(w.to(DEV), f16=True)
for x16 in (0, 1):
    xb = x.half() if x16 else x
    y = ops.new_rows(segs.rows, Cout, DEV)
    ops.conv_call(ops.Rows(xb.contiguous()), segs, wp, y, Cin=Cin, Cout=Cout, k=1, precision=_lib.PREC_F16, tile=4)()
    yt = y.tensor().cpu()
    print("x16", x16, "row0", (yt[0] * 128).round().int().tolist()[:40])
    print("x16", x16, "row5", ((yt[5] - 5) * 128).round().int().tolist()[:40], "rows:", yt[:8, 0].tolist())
