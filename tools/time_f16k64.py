"""The AMP conv kernel (FD_TILE_F16K64) on a few shapes of the training step, f16 maps in and out.  usage: python tools/time_f16k64.py   (FD_LIB selects a build)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_object_detection_amd import ops, _lib
from pytorch_object_detection_amd._lib import Segs
dev = "cuda:0"
B = 16
PYR = [(64, 64), (32, 32), (16, 16), (8, 8), (4, 4)]
SHAPES = [(256, 256, 3, PYR, False), (256, 256, 3, [(32, 32)], False), (128, 128, 3, [(64, 64)], False), (64, 256, 1, [(128, 128)], True), (256, 64, 1, [(128, 128)], False),
          (128, 512, 1, [(64, 64)], True), (512, 128, 1, [(64, 64)], False), (256, 1024, 1, [(32, 32)], True), (1024, 256, 1, [(32, 32)], False)]
for Cin, Cout, k, hw, use_res in SHAPES:
    segs = Segs.make(B, hw)
    pad = k // 2
    x = ops.Rows(torch.randn(segs.rows, Cin, device=dev).half())
    w = torch.randn(Cout, Cin, k, k, device=dev) / (Cin * k * k) ** 0.5
    wp = ops.pack_conv_weight_f16k64(w)
    y = ops.Rows(torch.empty(segs.rows, Cout, device=dev, dtype=torch.float16))
    res = ops.Rows(torch.randn(segs.rows, Cout, device=dev).half()) if use_res else None
    run = ops.conv_call(x, segs, wp, y, Cin=Cin, Cout=Cout, k=k, stride=1, pad=pad, res=res, act=1, tile=_lib.F16K64_TILE, precision=_lib.PREC_F16)
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    flops = 2 * segs.rows * Cout * Cin * k * k
    byts = 2 * segs.rows * (Cin + Cout * (2 if use_res else 1))
    print(f"{Cin:4d}>{Cout:4d} k{k} rows {segs.rows:6d} res {int(use_res)}: {us:7.1f} us  {flops / us / 1e6:6.1f} TFLOP/s  {byts / us / 1e3:6.0f} GB/s")
