cd $GRAFT_REPO_ROOT
O=gpurun_out/r02x; mkdir -p $O; rm -f $O/var.log
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
FD_WINO_GRP=16 timeout -k 10 300 python -m pytest tests/test_layers_gpu.py -q -m gpu -x -k "winograd" 2>&1 | tail -2
for v in 0 8 16 32 64; do
  echo "== FD_WINO_GRP=$v" | tee -a $O/var.log
  FD_WINO_GRP=$v WINO_SHORT=1 timeout -k 10 200 python tools/time_wino.py 2>&1 | grep -v amdgpu.ids | tee -a $O/var.log || exit 1
done
for v in 16 32; do
FD_WINO_GRP=$v bash tools/pmc_wino.sh r02x_grp$v tower 2>&1 | grep "fetch\|^sq"
done
