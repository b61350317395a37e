# round-4 experiment: F(4x4) with a transform pre-pass (FD_TILE_WINOGRAD4_PRE) against the fused kernel, stand-alone launches, same box, two rounds
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-w4pre}; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_layers_gpu.py -q -m gpu -x -k "winograd_f4x4" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for r in 1 2; do W4_PRE=1 timeout -k 10 300 python tools/time_wino4.py 2>&1 | grep -E "ms" | sed 's/F(2x2).*| F(4x4)/F(4x4)/' | sed 's/TF\/s-eq ([0-9.]*x) max|diff| [0-9.e-]*//'; done | tee $O/time_wino4_pre.txt
