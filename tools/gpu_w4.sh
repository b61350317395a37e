cd $GRAFT_REPO_ROOT
O=gpurun_out/w4; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 300 python -m pytest tests/test_layers_gpu.py -q -m gpu -k "winograd_f4x4" 2>&1 | tail -6
for d in ${W4_DBGS:-0}; do echo "== FD_W4_DBG=$d"; FD_W4_DBG=$d timeout -k 10 200 python tools/time_wino4.py 2>&1 | grep -E "x" ; done | tee $O/time_wino4_dbg.txt
