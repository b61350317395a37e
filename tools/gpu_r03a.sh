# Round 3, first GPU call: the whole -m gpu suite (new: strict mode, SyncBatchNorm DDP, K > 1024, trunk dump, autotune force), the bench
# line with the new roofline fields, the self-launching --gpus 2 rehearsal over gloo.  -> gpurun_out/r03a/
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03a; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err; cut -c1-1500 $O/bench.json
FD_BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-fast-mode --no-train-step > $O/bench_gloo2.json 2> $O/bench_gloo2.err; echo "gloo2 rc=$?"; cut -c1-300 $O/bench_gloo2.json
FD_BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --mode train --steps 3 --warmup 1 > $O/bench_gloo2_train.json 2> $O/bench_gloo2_train.err; echo "gloo2 train rc=$?"; cut -c1-400 $O/bench_gloo2_train.json; tail -3 $O/bench_gloo2_train.err
python bench.py --inflight 1 --layer-times $O/layer_times.tsv > /dev/null 2>&1; tail -1 $O/layer_times.tsv
