set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02r; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_postproc_gpu.py tests/test_model_gpu.py tests/test_effnet_gpu.py -q -m gpu > $O/t.log 2>&1; tail -6 $O/t.log
timeout -k 10 300 python bench.py --no-fast-mode --no-cpu-baseline --no-train-step > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02r/bench.json"))
print(d["value"], d["roofline"]["frac"], d["postproc"]["batch16"], d["postproc"]["batch1"], d["nms_micro"])
PY
