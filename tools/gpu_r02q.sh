set -x
cd $GRAFT_REPO_ROOT
export FD_COMMIT=519164b
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build_q.log 2>&1 || { tail -20 gpurun_out/build_q.log; exit 1; }
bash tools/prof.sh r02q 2>&1 | tail -3
rm -f gpurun_out/prof_r02q/trace_kernel_trace.csv
bash tools/pmc.sh r02q 2>&1 | tail -3
mkdir -p profiles_out && python tools/pmc_summary.py r02q > gpurun_out/pmc_r02q_summary.log 2>&1; tail -3 gpurun_out/pmc_r02q_summary.log
cp profiles/r02q_pmc_summary.json gpurun_out/r02q_pmc_summary.json
rm -rf gpurun_out/pmc_r02q/*/pmc_kernel_trace.csv gpurun_out/pmc_r02q/*/pmc_counter_collection.csv
bash tools/prof_train.sh r02q_train 2>&1 | tail -3
rm -f gpurun_out/prof_r02q_train/trace_kernel_trace.csv
python bench.py --inflight 1 --layer-times gpurun_out/layers_r02q.tsv > /dev/null 2>&1; tail -1 gpurun_out/layers_r02q.tsv
timeout -k 10 400 python bench.py > gpurun_out/bench_r02q.json 2> gpurun_out/bench_r02q.err; tail -2 gpurun_out/bench_r02q.err
timeout -k 10 300 python bench.py --model FCOS-B3 --size 832x1344 --steps 10 --warmup 3 --no-fast-mode --no-train-step > gpurun_out/bench_r02q_b3.json 2> gpurun_out/bench_r02q_b3.err
timeout -k 10 300 python bench.py --model FCOS-B3 --size 832x1344 --layer-times gpurun_out/layers_r02q_b3.tsv > /dev/null 2>&1
timeout -k 10 300 python bench.py --model FCOS --no-fast-mode --no-train-step > gpurun_out/bench_r02q_fcos.json 2> gpurun_out/bench_r02q_fcos.err
timeout -k 10 300 python bench.py --mode train > gpurun_out/bench_r02q_train.json 2> gpurun_out/bench_r02q_train.err
du -sh gpurun_out
