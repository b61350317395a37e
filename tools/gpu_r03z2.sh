# rocprof / PMC passes of the HEADLINE path only (--no-fast-mode: the f16x3 / mixed legs launch the tagged tower kernel too and pull its average)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03z; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
bash tools/prof.sh r03z 2>&1 | tail -3
cp gpurun_out/prof_r03z/trace_kernel_stats.csv $O/rocprofv3_kernel_stats.csv; rm -f gpurun_out/prof_r03z/trace_kernel_trace.csv
grep -o '"avg_launch_ms": [0-9.]*' gpurun_out/prof_r03z/bench.log | head -1; grep "wino_kernel<1" $O/rocprofv3_kernel_stats.csv
bash tools/pmc.sh r03z 2>&1 | tail -3
python tools/pmc_summary.py r03z > $O/pmc_summary.log 2>&1; tail -3 $O/pmc_summary.log
cp profiles/r03z_pmc_summary.json $O/pmc_summary.json; cp profiles/pmc_traffic.json $O/pmc_traffic.json
rm -rf gpurun_out/pmc_r03z/*/pmc_kernel_trace.csv gpurun_out/pmc_r03z/*/pmc_counter_collection.csv
