# same-box A/B of two builds of the library: csrc/_build/libold.so (previous commit) against the current one
cd $GRAFT_REPO_ROOT
L=pytorch_object_detection_amd/csrc
cp $L/libfcosdet_hip.so /tmp/libnew.so
for round in 1 2; do
  for which in new old; do
    if [ $which = old ]; then cp $L/_build/libold.so $L/libfcosdet_hip.so; else cp /tmp/libnew.so $L/libfcosdet_hip.so; fi
    echo "== $which (round $round)"
    timeout -k 10 300 python bench.py --no-fast-mode --no-train-step --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-110
    if [ $round = 1 ]; then
      timeout -k 10 300 python bench.py --model FCOS-B3 --size 832x1344 --steps 10 --warmup 3 --no-fast-mode --no-train-step --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-110
      timeout -k 10 300 python bench.py --inflight 1 --layer-times gpurun_out/layers_$which.tsv > /dev/null 2>&1; tail -1 gpurun_out/layers_$which.tsv
    fi
  done
done
cp /tmp/libnew.so $L/libfcosdet_hip.so
