# Round 3, second GPU call: full -m gpu suite, fused-GroupNorm A/B, wave-tile retime + table A/B.  -> gpurun_out/r03b/
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03b; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest.log
B="python bench.py --no-fast-mode --no-train-step --no-cpu-baseline"
for f in 1 0; do echo "== FD_GN_FUSED=$f"; FD_GN_FUSED=$f timeout -k 10 300 $B 2>/dev/null | tail -1 | cut -c1-110; done
FD_GN_FUSED=1 python bench.py --inflight 1 --layer-times $O/layer_times_fused.tsv > /dev/null 2>&1; grep -E "head\.|total" $O/layer_times_fused.tsv
T=pytorch_object_detection_amd/tuned/gfx950_tiles.json
cp $T /tmp/old_tiles.json
timeout -k 10 600 python tools/retime_1x1.py $O/new_tiles.json 2>&1 | grep -v amdgpu | tee $O/retime.log | tail -40
for round in 1 2; do
  for which in new old; do
    if [ $which = old ]; then cp /tmp/old_tiles.json $T; else cp $O/new_tiles.json $T; fi
    echo "== $which table (round $round)"; timeout -k 10 300 $B 2>/dev/null | tail -1 | cut -c1-110
  done
done
cp /tmp/old_tiles.json $T
