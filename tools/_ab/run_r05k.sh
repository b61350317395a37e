cd $GRAFT_REPO_ROOT
O=gpurun_out/r05k; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_amp_gpu.py tests/test_train_gpu.py -x -q -m gpu 2>&1 | tail -5
for k in 1 0 1 0; do
  FD_AMP_K64=$k timeout -k 10 300 python bench.py --mode train --amp --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/train_amp_k64_$k.json
  echo "k64=$k $(cut -c1-175 $O/train_amp_k64_$k.json | cut -c75-)"
done
