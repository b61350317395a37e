set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02b; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/t_all.log 2>&1; echo "all tests rc=$?"
tail -15 $O/t_all.log
timeout -k 10 120 python tools/grad_strides.py > $O/strides.log 2>&1; tail -12 $O/strides.log
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cat $O/bench.json
timeout -k 10 200 python bench.py --model FCOS-B3 --size 832x1344 --steps 10 --warmup 3 --no-fast-mode --no-train-step > $O/bench_b3.json 2> $O/bench_b3.err; echo "b3 rc=$?"; cat $O/bench_b3.json; tail -3 $O/bench_b3.err
timeout -k 10 200 python bench.py --model FCOS-B3 --size 832x1344 --layer-times $O/layers_b3.tsv > /dev/null 2>&1; tail -5 $O/layers_b3.tsv
