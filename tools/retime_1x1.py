"""Re-time the 1x1 stride-1 fp32 entries of tuned/gfx950_tiles.json on every plain tile and write a table with the consistent winners
(>= 3 % in both of two interleaved rounds) to the path given as argv[1].  Diagnostic / maintenance tool; run on the GPU box."""
import json, os, re, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_object_detection_amd import ops, _lib
from pytorch_object_detection_amd._lib import Segs
dev = "cuda:0"
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pytorch_object_detection_amd", "tuned", "gfx950_tiles.json")
table = json.load(open(path))
pat = re.compile(r"^B(\d+)\|([0-9x+]+)\|(\d+)>(\d+)\|k1s1p0d1\|res(\d)\|xcs(\d+)\|ycs(\d+)$")
changed = 0
for key, cur in sorted(table.items()):
    m = pat.match(key)
    if not m or cur >> 8:
        continue
    B, hws, Cin, Cout, res, xcs, ycs = int(m[1]), m[2], int(m[3]), int(m[4]), int(m[5]), int(m[6]), int(m[7])
    hw = [tuple(int(v) for v in p.split("x")) for p in hws.split("+")]
    segs = Segs.make(B, hw)
    if segs.rows * max(xcs, ycs) * 4 > 3e9:
        continue
    x = ops.Rows(torch.randn(segs.rows, xcs, device=dev), 0, Cin)
    y = ops.Rows(torch.empty(segs.rows, ycs, device=dev), 0, Cout)
    r = ops.Rows(torch.randn(segs.rows, Cout, device=dev)) if res else None
    w = torch.randn(Cout, Cin, 1, 1, device=dev) / Cin ** 0.5
    wp = ops.pack_conv_weight(w)
    wf = ops.pack_conv_weight_wave(w) if (ops.wave_ok(Cin, Cout, 1, 1, 0) and ycs % 4 == 0) else None
    sc, sf = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev)
    tiles = [t for t in (1, 2, 3, 4, 7, 8, 9, 5, 6, 12, 15) if not ((t == 5 and Cout > 32) or (t in (6, 12) and Cout > 96) or (t == 15 and wf is None))]
    times = {t: [] for t in tiles}
    for rnd in range(2):
        for t in tiles:
            run = ops.conv_call(x, segs, wp, y, Cin=Cin, Cout=Cout, k=1, scale=sc, shift=sf, res=r, act=1, tile=t, w_frag=wf)
            for _ in range(2): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(12): run()
            e1.record(); e1.synchronize()
            times[t].append(e0.elapsed_time(e1) / 12)
    c = cur & 0xFF
    if c not in times:
        continue
    best = min(tiles, key=lambda t: max(times[t]))
    if best != c and all(times[best][i] < 0.97 * times[c][i] for i in range(2)):
        print(f"{key}: tile {c} {times[c][0]*1e3:.1f}/{times[c][1]*1e3:.1f} us -> tile {best} {times[best][0]*1e3:.1f}/{times[best][1]*1e3:.1f} us", flush=True)
        table[key] = best
        changed += 1
print("changed", changed, "entries")
json.dump(dict(sorted(table.items())), open(sys.argv[1], "w"), indent=0)
