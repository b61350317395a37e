"""Time one conv shape under every tile id (diagnostic).
usage: python tools/time_conv.py Cin Cout k stride H W [res] [act] ; env FD_TILES=1,2 restricts, B=16"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_object_detection_amd import ops, _lib
from pytorch_object_detection_amd._lib import Segs
dev = "cuda:0"
Cin, Cout, k, stride, H, W = [int(v) for v in sys.argv[1:7]]
use_res = len(sys.argv) > 7 and sys.argv[7] == "1"
act = int(sys.argv[8]) if len(sys.argv) > 8 else 1
B = int(os.environ.get("B", "16"))
reps = int(os.environ.get("REPS", "20"))
segs = Segs.make(B, [(H, W)])
pad = (k - 1) // 2
so = ops.conv_out_segs(segs, k, stride, pad, 1)
x = ops.Rows(torch.randn(segs.rows, Cin, device=dev))
w = torch.randn(Cout, Cin, k, k, device=dev) / (Cin * k * k) ** 0.5
wp = ops.pack_conv_weight(w)
wf = ops.pack_conv_weight_wave(w) if ops.wave_ok(Cin, Cout, k, stride, pad) else None
y = ops.new_rows(so.rows, Cout, dev)
res = ops.Rows(torch.randn(so.rows, Cout, device=dev)) if use_res else None
sc, sf = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev)
flops = 2 * so.rows * Cout * Cin * k * k
byts = 4 * (segs.rows * Cin + so.rows * Cout * (2 if use_res else 1))
only = [int(v) for v in os.environ.get('FD_TILES', '').split(',') if v]
print(f"M={so.rows} Cin={Cin} Cout={Cout} k={k} s={stride} res={use_res}: {flops/1e9:.1f} GF, {byts/1e6:.0f} MB min traffic")
for tile in (only or sorted(_lib.TILES)):
    if (tile == 5 and Cout > 32) or (tile == 6 and Cout > 96) or (tile == 13 and k != 3) or (tile == 15 and wf is None):
        continue
    run = ops.conv_call(x, segs, wp, y, Cin=Cin, Cout=Cout, k=k, stride=stride, pad=pad, scale=sc, shift=sf, res=res, act=act, tile=tile, w_frag=wf)
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"tile {tile:2d} {_lib.TILES.get(tile, 'auto')}: {ms*1e3:7.1f} us  {flops / ms / 1e9:6.1f} TFLOP/s  {byts / ms / 1e9:6.2f} TB/s")
