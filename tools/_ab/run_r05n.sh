cd $GRAFT_REPO_ROOT
O=gpurun_out/r05n; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_effnet_gpu.py -x -q -m gpu 2>&1 | tail -3
for f in 1 0 1 0; do
  FD_MBCONV_FUSED=$f timeout -k 10 300 python bench.py --model FCOS-B3 --size 832x1344 --steps 10 --warmup 3 --no-fast-mode --no-train-step --no-cpu-baseline 2>/dev/null | tail -1 > $O/b3_fused$f.json
  echo "fused=$f $(cut -c50-150 $O/b3_fused$f.json)"
done
FD_MBCONV_FUSED=1 timeout -k 10 300 python bench.py --model FCOS-B3 --size 832x1344 --inflight 1 --layer-times $O/layers_fused.tsv > /dev/null 2>&1; head -40 $O/layers_fused.tsv | cut -c1-100
