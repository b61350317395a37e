cd $GRAFT_REPO_ROOT
O=gpurun_out/r02w; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 1000 python -m pytest tests -q -m gpu > $O/t.log 2>&1; tail -8 $O/t.log
timeout -k 10 400 python bench.py --no-cpu-baseline --no-fast-mode > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; cat $O/bench.json
timeout -k 10 300 python bench.py --inflight 1 --layer-times $O/layers.tsv > $O/lt.log 2>&1; tail -2 $O/lt.log
