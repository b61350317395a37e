# clock / MFMA-busy of the F(4x4) tower launch with parts of the kernel switched off (FD_W4_DBG): does the loader cost time or clock?
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
for d in 0 1 5 4; do
  out=$PWD/gpurun_out/pmc_w4dbg$d; mkdir -p $out
  FD_W4_DBG=$d rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --output-format csv -d $out/sq -o pmc -- python3 tools/wino_one.py tower 4 > $out/sq.log 2>&1
  python3 - <<PY
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
