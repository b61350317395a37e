#!/usr/bin/env python3
"""Summarise gpurun_out/pmc_<tag>/ (tools_pmc.sh) into profiles/<tag>_pmc_summary.json + profiles/pmc_traffic.json.
HBM bytes follow MI355X_MICROARCH.md §HBM: FETCH_SIZE (KiB) is doubled on gfx950 (verified here on gn_apply /
se_scale / nchw3_to_nhwc4, whose read bytes are known: ratio 0.50), WRITE_SIZE is exact."""
import collections
import csv
import json
import sys

tag = sys.argv[1]
root = f"gpurun_out/pmc_{tag}"


def load(d):
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(f"{root}/{d}/pmc_counter_collection.csv")):
        e = disp.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"], "grid": int(r["Grid_Size"]), "c": {},
                                               "dur_ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        e["c"][r["Counter_Name"]] = float(r["Counter_Value"])
    return list(disp.values())


def longest_conv(rows):
    # the roofline kernel = the head-tower launch, which runs under its own symbol (TAG = 1): the Winograd kernel
    # conv3x3_wino_kernel<1, ..> (default) or, with FD_WINOGRAD=0, conv_igemm_kernel<..., 1, false>
    conv = ([v for v in rows if "conv3x3_wino4_kernel<1" in v["name"]] or [v for v in rows if "conv3x3_wino_kernel<1" in v["name"]]
            or [v for v in rows if "conv_igemm_kernel" in v["name"]])
    top = max(conv, key=lambda v: v["dur_ns"])
    same = [v for v in conv if v["name"] == top["name"] and v["grid"] == top["grid"] and v["dur_ns"] > 0.7 * top["dur_ns"]]
    return same


fe, wr, sq = load("fetch"), load("write"), load("sq")
f, w, q = longest_conv(fe), longest_conv(wr), longest_conv(sq)
fetch_kib = sum(v["c"]["FETCH_SIZE"] for v in f) / len(f)
write_kib = sum(v["c"]["WRITE_SIZE"] for v in w) / len(w)
cyc = [v["c"]["GRBM_GUI_ACTIVE"] / 8 for v in q]
util = [v["c"]["SQ_VALU_MFMA_BUSY_CYCLES"] / (c * 1024) for v, c in zip(q, cyc)]
clk = [c / v["dur_ns"] for v, c in zip(q, cyc)]
# calibration of the FETCH_SIZE factor on a normalise pass that reads exactly what it writes: gn_apply_kernel, or -- since the tower's GroupNorm is a slice
# pass (round 4) -- coef_apply_kernel; the largest launch of it in the run
cal = None
gf = [v for v in fe if "gn_apply_kernel" in v["name"] or "coef_apply_kernel" in v["name"]]
gw = [v for v in wr if "gn_apply_kernel" in v["name"] or "coef_apply_kernel" in v["name"]]
if gf and gw:
    gf, gw = max(gf, key=lambda v: v["c"]["FETCH_SIZE"]), max(gw, key=lambda v: v["c"]["WRITE_SIZE"])
    cal = gf["c"]["FETCH_SIZE"] / gw["c"]["WRITE_SIZE"]
def biggest(rows, key):
    ks = [v for v in rows if key in v["name"]]
    return max(ks, key=lambda v: v["grid"]) if ks else None


dec_f, dec_w = biggest(fe, "decode_coalesced"), biggest(wr, "decode_coalesced")
decode = None
if dec_f and dec_w:
    decode = {"kernel": dec_f["name"], "grid": dec_f["grid"], "FETCH_SIZE_KiB": dec_f["c"]["FETCH_SIZE"], "WRITE_SIZE_KiB": dec_w["c"]["WRITE_SIZE"],
              "hbm_bytes_per_launch": int((2 * dec_f["c"]["FETCH_SIZE"] + dec_w["c"]["WRITE_SIZE"]) * 1024),
              "algorithmic_bytes_batch16_640": 16 * 8525 * (85 * 4 + 24)}
import subprocess
import os
try:
    commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True, stderr=subprocess.DEVNULL).strip()
except Exception:
    commit = os.environ.get("FD_COMMIT")          # (the GPU box has no .git: tools/gpu_*.sh pass the hash along)
out = {"tag": tag, "commit_of_the_working_tree_summarised": commit, "decode_kernel_largest_launch": decode, "head_tower_conv": {"kernel": f[0]["name"], "launches": len(f), "FETCH_SIZE_KiB": fetch_kib, "WRITE_SIZE_KiB": write_kib,
                           "hbm_bytes_per_launch": int((2 * fetch_kib + write_kib) * 1024),
                           "mfma_busy_frac": sum(util) / len(util), "clock_ghz": sum(clk) / len(clk),
                           "avg_dur_ms_under_pmc": sum(v["dur_ns"] for v in q) / len(q) / 1e6},
       "fetch_size_calibration_normalise_pass(read/written)": cal,
       "source": f"rocprofv3 --kernel-trace --pmc <one counter set per pass> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline ({tag})"}
json.dump(out, open(f"profiles/{tag}_pmc_summary.json", "w"), indent=1)
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
