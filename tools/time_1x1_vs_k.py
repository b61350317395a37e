"""Fixed cost of the 1x1 (GEMM-addressed) kernels: time a layer of fixed M, N at several K and fit t = a + b * K (a = prologue + epilogue, b = K loop)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_object_detection_amd import ops, _lib
from pytorch_object_detection_amd._lib import Segs
dev = "cuda:0"


def timeit(call, reps=20):
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, B, hw, N, res, tile in (("layer1-like 160x160 N=64", 16, [(160, 160)], 64, False, 4), ("layer1-like 160x160 N=256 +res", 16, [(160, 160)], 256, True, 9),
                                  ("layer3-like 40x40 N=1024 +res", 16, [(40, 40)], 1024, True, 8), ("layer3-like 40x40 N=256", 16, [(40, 40)], 256, False, 9)):
    segs = Segs.make(B, hw)
    out = []
    for K in (64, 128, 256, 512, 1024):
        x = ops.Rows(torch.randn(segs.rows, K, device=dev))
        y = ops.new_rows(segs.rows, N, dev)
        r = ops.Rows(torch.randn(segs.rows, N, device=dev)) if res else None
        w = torch.randn(N, K, 1, 1, device=dev) / K ** 0.5
        sc, sf = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)
        call = ops.conv_call(x, segs, ops.pack_conv_weight(w), y, Cin=K, Cout=N, k=1, scale=sc, shift=sf, res=r, act=1, tile=tile)
        out.append((K, timeit(call)))
    (k0, t0), (k1, t1) = out[1], out[-1]
    b = (t1 - t0) / (k1 - k0)
    a = t0 - b * k0
    print(f"{name} tile {tile}: " + "  ".join(f"K={k}: {t:.1f}us" for k, t in out) + f"   fit: {a:.1f} us + {b*32:.2f} us per K-tile of 32; MFMA-only would be {2*segs.rows*N*32/141e12*1e6:.2f} us per K-tile", flush=True)
