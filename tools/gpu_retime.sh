cd $GRAFT_REPO_ROOT
T=pytorch_object_detection_amd/tuned/gfx950_tiles.json
cp $T /tmp/old_tiles.json
timeout -k 10 700 python tools/retime_1x1.py gpurun_out/new_tiles.json 2>&1 | grep -v amdgpu
for round in 1 2; do
  for which in new old; do
    if [ $which = old ]; then cp /tmp/old_tiles.json $T; else cp gpurun_out/new_tiles.json $T; fi
    echo "== $which table (round $round)"
    timeout -k 10 300 python bench.py --no-fast-mode --no-train-step --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-100
    if [ $round = 1 ]; then timeout -k 10 300 python bench.py --model FCOS-B3 --size 832x1344 --steps 10 --warmup 3 --no-fast-mode --no-train-step --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-100; fi
  done
done
cp /tmp/old_tiles.json $T
