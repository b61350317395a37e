cd $GRAFT_REPO_ROOT
O=gpurun_out/r02w; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $O/t.log 2>&1; tail -15 $O/t.log
timeout -k 10 400 python bench.py --no-cpu-baseline --no-fast-mode > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; cat $O/bench.json
