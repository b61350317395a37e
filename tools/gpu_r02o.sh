set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02o; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 1000 python -m pytest tests -q -m gpu > $O/t.log 2>&1; tail -8 $O/t.log
timeout -k 10 300 python tools/train_step_time.py 2>&1 | grep "train step"
timeout -k 10 300 python bench.py --no-fast-mode --no-cpu-baseline --no-train-step > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02o/bench.json"))
print(d["value"], d["roofline"]["frac"])
PY
