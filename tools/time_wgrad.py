"""Time the weight-gradient kernel on the train step's layer shapes (diagnostic).
usage: python tools/time_wgrad.py [nsplit,nsplit,...]   (0 = library choice); F16=1: the AMP form (f16 maps, FD_PREC_F16)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_object_detection_amd import ops
from pytorch_object_detection_amd._lib import Segs
dev = "cuda:0"
B = 16
splits = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0]
# Cin, Cout, k, stride, dil, H, W (forward input size)
SHAPES = [
    (256, 256, 3, 1, 2, 64, 64), (256, 256, 3, 1, 2, 32, 32), (256, 256, 3, 1, 2, 16, 16),
    (256, 128, 3, 1, 1, 64, 64), (256, 256, 3, 1, 1, 64, 64), (256, 256, 3, 1, 1, 32, 32),
    (128, 128, 3, 2, 1, 128, 128), (256, 128, 1, 1, 1, 128, 128), (128, 512, 1, 1, 1, 64, 64), (512, 128, 1, 1, 1, 64, 64),
    (128, 128, 3, 1, 1, 64, 64),
    (256, 256, 3, 1, 1, 32, 32), (1024, 256, 1, 1, 1, 32, 32), (256, 1024, 1, 1, 1, 32, 32),
    (512, 512, 3, 1, 1, 16, 16), (2048, 512, 1, 1, 1, 16, 16), (512, 2048, 1, 1, 1, 16, 16),
    (256, 512, 1, 1, 1, 64, 64), (512, 256, 1, 1, 1, 64, 64),
]
if os.environ.get('CUSTOM'):
    SHAPES = [tuple(int(v) for v in t.split(',')) for t in os.environ['CUSTOM'].split(';')]
elif os.environ.get('SHAPES'):
    SHAPES = [SHAPES[int(i)] for i in os.environ['SHAPES'].split(',')]
tot = {s: 0.0 for s in splits}
for Cin, Cout, k, stride, dil, H, W in SHAPES:
    pad = dil * (k - 1) // 2
    segs = Segs.make(B, [(H, W)])
    so = ops.conv_out_segs(segs, k, stride, pad, dil)
    F16 = os.environ.get('F16') == '1'
    x = ops.Rows(torch.randn(segs.rows, Cin, device=dev).to(torch.float16 if F16 else torch.float32))
    dy = ops.Rows(torch.randn(so.rows, Cout, device=dev).to(torch.float16 if F16 else torch.float32))
    flops = 2 * so.rows * Cout * Cin * k * k
    line = f"{Cin:4d}->{Cout:4d} k{k} s{stride} d{dil} {H:3d}x{W:<3d} M={so.rows:6d}:"
    for ns in splits:
        f = lambda: ops.conv_wgrad(x, dy, segs, Cin=Cin, Cout=Cout, k=k, stride=stride, pad=pad, dil=dil, nsplit=ns, precision=2 if F16 else 0)
        for _ in range(2):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1) / 10
        tot[ns] += ms
        line += f"  ns{ns}: {ms * 1e3:6.0f} us {flops / ms / 1e9:6.1f} TF"
    print(line)
print("total ms:", {k: round(v, 3) for k, v in tot.items()})
