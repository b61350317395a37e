import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorch_object_detection_amd import ops, _lib
from pytorch_object_detection_amd._lib import Segs
DEV = "cuda:0"
def h(t): return t.half().float()
gen = torch.Generator().manual_seed(1)
for (Cin, Cout, k, stride, pad) in [(64, 256, 1, 1, 0), (128, 128, 3, 1, 1)]:
    B = 2; segs = Segs.make(B, [(20, 24)]); so = ops.conv_out_segs(segs, k, stride, pad, 1)
    x = h(torch.randn(segs.rows, Cin, generator=gen)).to(DEV); r = h(torch.randn(so.rows, Cout, generator=gen)).to(DEV)
    w = torch.randn(Cout, Cin, k, k, generator=gen) / np.sqrt(Cin * k * k)
    wp = ops.pack_conv_weight_hip(w.to(DEV), f16=True)
    for tile in (2, 0, 4):
        for use_res in (False, True):
            kw = dict(Cin=Cin, Cout=Cout, k=k, stride=stride, pad=pad, dil=1, precision=_lib.PREC_F16, tile=tile)
            y0 = ops.new_rows(so.rows, Cout, DEV)
            ops.conv_call(ops.Rows(x), segs, wp, y0, res=ops.Rows(r) if use_res else None, **kw)()
            for (x16, y16, r16) in [(1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 1)]:
                if r16 and not use_res: continue
                xb = x.half() if x16 else x; rb = r.half() if r16 else r
                yb = torch.zeros(so.rows, Cout, dtype=torch.float16 if y16 else torch.float32, device=DEV)
                ops.conv_call(ops.Rows(xb.contiguous()), segs, wp, ops.Rows(yb), res=ops.Rows(rb.contiguous()) if use_res else None, **kw)()
                want = y0.tensor().half().float() if y16 else y0.tensor()
                print(f"Cin {Cin} k {k} tile {tile} res {use_res} x16 {x16} y16 {y16} r16 {r16}: max err {float((yb.float() - want).abs().max()):.4g}", flush=True)
