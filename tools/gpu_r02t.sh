cd $GRAFT_REPO_ROOT
O=gpurun_out/r02t; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 1000 python -m pytest tests -q -m gpu > $O/t.log 2>&1; tail -8 $O/t.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
