# Where the F(4x4) tower launch's time goes: a TIMING build of the library (-DFD_W4_TIMING: the shipped build has no such switches) run with parts of the
# kernel off (FD_W4_DBG: 1 = no loader stages, 4 = no epilogue, 5 = both; wrong results by design), stand-alone launch times + clock / MFMA-busy from PMC.
#   usage (GPU box, repo root): bash tools/gpu_w4_breakdown.sh   -> gpurun_out/w4_breakdown.txt     (rebuilds the shipped library at the end)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
C=pytorch_object_detection_amd/csrc
touch $C/fd_conv_wino4.hip; make -s -j8 -C $C FLAGS_fd_conv_wino4=-DFD_W4_TIMING > /dev/null 2>&1 || exit 1
O=$PWD/gpurun_out/w4_breakdown.txt; : > $O
for d in 0 1 5 4; do
  echo "== FD_W4_DBG=$d" >> $O
  FD_W4_DBG=$d timeout -k 10 200 python tools/time_wino4.py 2>&1 | grep -E "tower" >> $O
  out=$PWD/gpurun_out/pmc_w4dbg$d; mkdir -p $out
  FD_W4_DBG=$d rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --output-format csv -d $out/sq -o pmc -- python3 tools/wino_one.py tower 4 > $out/sq.log 2>&1
  python3 - >> $O <<PY
import csv, collections
rows = collections.OrderedDict()
for r in csv.DictReader(open("$out/sq/pmc_counter_collection.csv")):
    if "wino4_kernel" not in r["Kernel_Name"]: continue
    e = rows.setdefault(r["Dispatch_Id"], {"dur": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
    e[r["Counter_Name"]] = float(r["Counter_Value"])
last = list(rows.values())[-1]
cyc = last["GRBM_GUI_ACTIVE"] / 8
print("dbg $d dur_us", last["dur"] / 1e3, "clock GHz", round(cyc / last["dur"], 3), "mfma busy", round(last["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), 3))
PY
  rm -rf $out
done
touch $C/fd_conv_wino4.hip; make -s -j8 -C $C > /dev/null 2>&1
cat $O
