"""The F(4x4) kernel's fixed cost per workgroup: launches of exactly r x 256 workgroups (r whole rounds on 256 CUs, Cout = 64: one cout tile,
32 images of 64 x 64 = 256 M tiles per round) at Cin in {64, 128, 256, 512} = 8 / 16 / 32 / 64 chunks; least-squares fit t = a + b * chunks per round count.
With a -DFD_W4_TIMING build selected through FD_LIB: FD_W4_DBG switches parts of the kernel off, FD_W4_TS=<file> appends the in-kernel phase times per launch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_object_detection_amd import ops, _lib
from pytorch_object_detection_amd._lib import Segs
dev = "cuda:0"
ts = bool(os.environ.get("FD_W4_TS"))


def timeit(call, reps=20):
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3     # us


def make(rounds, Cin, Cout=64):
    segs = Segs.make(32 * rounds, [(64, 64)])
    x = ops.Rows(torch.randn(segs.rows, Cin, device=dev))
    y = ops.new_rows(segs.rows, Cout, dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) / (Cin * 9) ** 0.5
    return ops.conv_call(x, segs, ops.pack_conv_weight_wino4(w), y, Cin=Cin, Cout=Cout, k=3, pad=1, dil=1, tile=_lib.WINO4_TILE)


print("FD_W4_DBG", os.environ.get("FD_W4_DBG", "0"), "lib", os.path.basename(_lib.LIB_PATH), flush=True)
for rounds in (1, 2, 4):
    pts = []
    for Cin in (64, 128, 256, 512):
        call = make(rounds, Cin)
        if ts:
            call(); call(); torch.cuda.synchronize()
            continue
        t = timeit(call)
        pts.append((Cin // 8, t))
        print(f"rounds {rounds} Cin {Cin:4d} chunks {Cin // 8:3d}: {t:8.2f} us/launch  {t / rounds:8.2f} us/round", flush=True)
    if pts:
        n = len(pts); sx = sum(p[0] for p in pts); sy = sum(p[1] for p in pts); sxx = sum(p[0] ** 2 for p in pts); sxy = sum(p[0] * p[1] for p in pts)
        b = (n * sxy - sx * sy) / (n * sxx - sx * sx); a = (sy - b * sx) / n
        print(f"rounds {rounds}: t = {a:.2f} us + {b:.3f} us * chunks  (per round: a {a / rounds:.2f}, b {b / rounds:.3f})", flush=True)
