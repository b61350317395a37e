"""Sample socket power / clocks (rocm-smi) while one conv shape runs in a loop (diagnostic).
usage: python tools/power_probe.py Cin Cout H W res tile seconds"""
import os, sys, subprocess, threading, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_object_detection_amd import ops, _lib
from pytorch_object_detection_amd._lib import Segs
Cin, Cout, H, W, use_res, tile = [int(v) for v in sys.argv[1:7]]
secs = float(sys.argv[7]) if len(sys.argv) > 7 else 3.0
dev = "cuda:0"
segs = Segs.make(16, [(H, W)])
x = ops.Rows(torch.randn(segs.rows, Cin, device=dev))
w = torch.randn(Cout, Cin, 1, 1, device=dev) / Cin ** 0.5
wp = ops.pack_conv_weight(w)
y = ops.new_rows(segs.rows, Cout, dev)
res = ops.Rows(torch.randn(segs.rows, Cout, device=dev)) if use_res else None
run = ops.conv_call(x, segs, wp, y, Cin=Cin, Cout=Cout, k=1, res=res, act=1, tile=tile)
samples = []
stop = False
def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=5).stdout
            samples.append(out.strip().splitlines()[-1])
        except Exception as e:
            samples.append("err %r" % e)
        time.sleep(0.2)
for _ in range(5): run()
torch.cuda.synchronize()
th = threading.Thread(target=sampler); th.start()
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    for _ in range(200): run()
    torch.cuda.synchronize(); n += 200
dt = time.time() - t0
stop = True; th.join()
print(f"Cin={Cin} Cout={Cout} {H}x{W} res={use_res} tile={tile}: {dt / n * 1e6:.1f} us/launch over {n} launches")
for s in samples[2:8]: print("  ", s)
