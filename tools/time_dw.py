"""Time the depthwise 3x3 kernel on the head / HisBlock shapes of the bench (diagnostic)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_object_detection_amd import ops
from pytorch_object_detection_amd._lib import Segs
dev = "cuda:0"
B = 16
for name, hw, C in (("head.dw1 pyramid", [(80, 80), (40, 40), (20, 20), (10, 10), (5, 5)], 512), ("HisBlock3.conv1_1", [(80, 80)], 128),
                    ("HisBlock4.conv1_1", [(40, 40)], 128)):
    segs = Segs.make(B, hw)
    x, y = ops.Rows(torch.randn(segs.rows, C, device=dev)), ops.new_rows(segs.rows, C, dev)
    w = torch.randn(9, C, device=dev)
    sc, sf = torch.rand(C, device=dev), torch.randn(C, device=dev)
    f = lambda: ops.dwconv3x3(x, w, y, segs, sc, sf, 1)
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{name}: {ms * 1e3:.1f} us, {2 * segs.rows * C * 4 / ms / 1e9:.2f} TB/s (read + write once)")
