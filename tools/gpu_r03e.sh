# Round 3, fifth GPU call: HIP-graph plans + detect() (latency path), AMP model test.  -> gpurun_out/r03e/
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03e; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_amp_gpu.py -m gpu -q --timeout 600 -k "graph or detect or amp_train" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
L="python bench.py --size 512 --inflight 1 --steps 100 --warmup 20 --no-fast-mode --no-train-step --no-cpu-baseline"
for b in 1 2; do
  timeout -k 10 200 $L --batch $b > $O/lat_b${b}_eager.json 2>/dev/null; echo "eager b=$b: $(cut -c1-140 $O/lat_b${b}_eager.json)"
  timeout -k 10 200 $L --batch $b --graph > $O/lat_b${b}_graph.json 2>$O/lat_b${b}_graph.err; echo "graph b=$b: $(cut -c1-140 $O/lat_b${b}_graph.json)"; tail -2 $O/lat_b${b}_graph.err
done
timeout -k 10 300 python bench.py --graph --no-fast-mode --no-train-step --no-cpu-baseline 2>/dev/null | cut -c1-140
