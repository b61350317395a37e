# Round 3, third GPU call: the fused-GroupNorm consumers after their instruction diet (A/B/C), the tests that failed in r03b.  -> gpurun_out/r03c/
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests/test_layers_gpu.py tests/test_train_gpu.py tests/test_model_gpu.py -m gpu -q --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
B="python bench.py --no-fast-mode --no-train-step --no-cpu-baseline"
for r in 1 2; do
for cfg in "1 0" "0 0" "1 1"; do set -- $cfg; echo "== FD_GN_FUSED=$1 FD_GN_FUSED_TOWER=$2"; FD_GN_FUSED=$1 FD_GN_FUSED_TOWER=$2 timeout -k 10 300 $B 2>/dev/null | tail -1 | cut -c1-110; done
done
FD_GN_FUSED=1 python bench.py --inflight 1 --layer-times $O/layer_times_fused.tsv > /dev/null 2>&1; grep -E "head\.|total" $O/layer_times_fused.tsv
FD_GN_FUSED=1 FD_GN_FUSED_TOWER=1 python bench.py --inflight 1 --layer-times $O/layer_times_fused_tower.tsv > /dev/null 2>&1; grep -E "head\.tower|total" $O/layer_times_fused_tower.tsv
