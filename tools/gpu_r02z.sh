# Final measurement set of round 2 (one gpurun call): rocprof kernel stats, PMC passes, layer times, bench lines.  -> gpurun_out/r02z/
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02z; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
bash tools/prof.sh r02z 2>&1 | tail -3
cp gpurun_out/prof_r02z/trace_kernel_stats.csv $O/rocprofv3_kernel_stats.csv; rm -f gpurun_out/prof_r02z/trace_kernel_trace.csv
bash tools/pmc.sh r02z 2>&1 | tail -3
python tools/pmc_summary.py r02z > $O/pmc_summary.log 2>&1; tail -3 $O/pmc_summary.log
cp profiles/r02z_pmc_summary.json $O/pmc_summary.json; cp profiles/pmc_traffic.json $O/pmc_traffic.json
rm -rf gpurun_out/pmc_r02z/*/pmc_kernel_trace.csv gpurun_out/pmc_r02z/*/pmc_counter_collection.csv
bash tools/pmc_wino.sh r02z_wino tower > $O/pmc_wino_tower.txt 2>&1; tail -5 $O/pmc_wino_tower.txt
rm -rf gpurun_out/pmc_r02z_wino/*/pmc_kernel_trace.csv gpurun_out/pmc_r02z_wino/*/pmc_counter_collection.csv
bash tools/prof_train.sh r02z_train 2>&1 | tail -3
cp gpurun_out/prof_r02z_train/last_step_kernel_stats.csv $O/train_step_kernel_stats.csv; rm -f gpurun_out/prof_r02z_train/trace_kernel_trace.csv
python bench.py --inflight 1 --layer-times $O/layer_times.tsv > /dev/null 2>&1; tail -1 $O/layer_times.tsv
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
timeout -k 10 300 python bench.py --model FCOS-B3 --size 832x1344 --steps 10 --warmup 3 --no-fast-mode --no-train-step > $O/bench_fcos_b3_832x1344.json 2> $O/bench_b3.err
timeout -k 10 300 python bench.py --model FCOS-B3 --size 832x1344 --layer-times $O/layer_times_fcos_b3.tsv > /dev/null 2>&1
timeout -k 10 300 python bench.py --model FCOS --no-fast-mode --no-train-step > $O/bench_fcos_r50.json 2> $O/bench_fcos.err
timeout -k 10 300 python bench.py --model MNFCOS --no-fast-mode --no-train-step --no-cpu-baseline > $O/bench_mnfcos.json 2> $O/bench_mn.err
timeout -k 10 300 python bench.py --mode train > $O/bench_train.json 2> $O/bench_train.err
for b in 1 2; do timeout -k 10 200 python bench.py --batch $b --size 512 --inflight 1 --steps 50 --warmup 10 --no-fast-mode --no-train-step --no-cpu-baseline > $O/bench_latency_b${b}_512.json 2>/dev/null; done
FD_WINOGRAD=0 timeout -k 10 300 python bench.py --no-fast-mode --no-train-step --no-cpu-baseline > $O/bench_direct_kernels_only.json 2> $O/bench_direct.err
timeout -k 10 200 python tools/time_wino.py 2>&1 | grep -v amdgpu.ids > $O/time_wino.txt
du -sh gpurun_out
# the 1x1 (GEMM-addressed) layers: time against K (compute + memory add up), persistent kernel on / off, socket power in three regimes
{ for a in "128 512 1 1 80 80 1" "256 512 1 1 80 80 1" "512 512 1 1 80 80 1" "128 512 1 1 80 80 0" "64 256 1 1 160 160 1" "256 64 1 1 160 160 0" "1024 256 1 1 40 40 0"; do
    for pw in 0 1; do echo "FD_CONV_PERSIST=$pw"; FD_CONV_PERSIST=$pw FD_TILES=4,8,9 timeout -k 10 100 python tools/time_conv.py $a; done; done
  timeout -k 10 60 python tools/power_probe.py 2048 512 80 80 0 9 3; timeout -k 10 60 python tools/power_probe.py 128 512 80 80 1 9 3
  timeout -k 10 60 python tools/power_probe.py 32 512 80 80 1 9 3; } 2>&1 | grep -v amdgpu.ids > $O/pw_study.txt
du -sh gpurun_out
