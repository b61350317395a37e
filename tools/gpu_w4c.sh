# per-layer in-plan times with the F(4x4) rule as shipped, with a smaller fixed-cost term (more layers on F(4x4)) and off
cd $GRAFT_REPO_ROOT
O=gpurun_out/w4c; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 300 python bench.py --inflight 1 --layer-times $O/layer_times_w4.tsv > /dev/null 2>&1
FD_WINOGRAD4_FIXED_US=8 timeout -k 10 300 python bench.py --inflight 1 --layer-times $O/layer_times_w4_fix8.tsv > /dev/null 2>&1
FD_WINOGRAD4=0 timeout -k 10 300 python bench.py --inflight 1 --layer-times $O/layer_times_w2.tsv > /dev/null 2>&1
tail -1 $O/layer_times_w4.tsv; tail -1 $O/layer_times_w4_fix8.tsv; tail -1 $O/layer_times_w2.tsv
timeout -k 10 400 python bench.py --no-train-step --no-cpu-baseline --no-fast-mode > $O/bench_w4.json 2> $O/bench_w4.err; cut -c1-200 $O/bench_w4.json
FD_WINOGRAD4_FIXED_US=8 timeout -k 10 400 python bench.py --no-train-step --no-cpu-baseline --no-fast-mode > $O/bench_w4_fix8.json 2> $O/bench_w4_fix8.err; cut -c1-200 $O/bench_w4_fix8.json
