cd $GRAFT_REPO_ROOT
O=gpurun_out/r05o; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_amp_gpu.py tests/test_train_gpu.py tests/test_train_nodes_gpu.py tests/test_ddp_gpu.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2 3; do
  timeout -k 10 300 python bench.py --mode train --amp --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/train_amp_$i.json
  echo "$(cut -c75-175 $O/train_amp_$i.json)"
done
FD_AMP=1 timeout -k 10 200 python tools/train_host_time.py 2>&1 | tail -1
timeout -k 10 300 python bench.py --mode train --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c75-175
