# Round 3, sixth GPU call: 'mixed' precision plans (1x1 on f16x3, 3x3 on fp32 Winograd): parity + throughput.  -> gpurun_out/r03f/
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03f; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_model_gpu.py -m gpu -q --timeout 600 -k "mixed" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout -k 10 400 python bench.py --no-train-step --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03f/bench.json"))
print("headline", d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], "1x1", d.get("roofline_1x1", {}).get("frac"))
for k in ("fast_mode", "fast_mode_mixed"):
    print(k, json.dumps(d.get(k)))
PY
FD_CONV_PRECISION=mixed python bench.py --inflight 1 --layer-times $O/layer_times_mixed.tsv > /dev/null 2>&1; tail -1 $O/layer_times_mixed.tsv
