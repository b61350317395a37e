"""fp32 against f16-operand weight gradient on the bench shapes (B = 16, 512 x 512 training maps).  usage: python tools/time_wgrad16.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_object_detection_amd import ops, _lib
from pytorch_object_detection_amd._lib import Segs
dev = "cuda:0"
B = 16
for Cin, Cout, k, stride, hw in ((256, 256, 3, 1, [(64, 64), (32, 32), (16, 16), (8, 8), (4, 4)]), (64, 256, 1, 1, [(128, 128)]), (256, 64, 1, 1, [(128, 128)]),
                                 (128, 512, 1, 1, [(64, 64)]), (512, 128, 1, 1, [(64, 64)]), (128, 128, 3, 1, [(64, 64)]), (256, 256, 3, 1, [(32, 32)]),
                                 (1024, 256, 1, 1, [(32, 32)]), (512, 512, 3, 1, [(16, 16)]), (256, 512, 3, 1, [(64, 64), (32, 32), (16, 16), (8, 8), (4, 4)])):
    segs = Segs.make(B, hw)
    pad = k // 2
    so = ops.conv_out_segs(segs, k, stride, pad, 1)
    x = ops.Rows(torch.randn(segs.rows, Cin, device=dev)); dy = ops.Rows(torch.randn(so.rows, Cout, device=dev))
    fl = 2 * so.rows * Cout * Cin * k * k
    res = []
    for prec in (0, 2):
        f = lambda: ops.conv_wgrad(x, dy, segs, Cin=Cin, Cout=Cout, k=k, stride=stride, pad=pad, oihw=True, precision=prec)
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); e1.synchronize()
        res.append(e0.elapsed_time(e1) / 10)
    print(f"{Cin:5d}>{Cout:<5d} k{k} rows {so.rows:7d}: f32 {res[0]*1e3:7.1f} us {fl/res[0]/1e9:6.1f} TF | f16 {res[1]*1e3:7.1f} us {fl/res[1]/1e9:6.1f} TF | x{res[0]/res[1]:.2f}")
