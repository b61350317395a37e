set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02m; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_postproc_gpu.py tests/test_layers_gpu.py tests/test_train_gpu.py tests/test_ddp_gpu.py -q -m gpu > $O/t.log 2>&1; tail -8 $O/t.log
timeout -k 10 300 python bench.py --no-fast-mode --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02m/bench.json"))
print(d["value"], d["roofline"]["frac"], d["postproc"]["batch16"], d["postproc"]["batch1"], d.get("train_step"))
PY
cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_train -o train -- python3 $GRAFT_REPO_ROOT/tools/train_step_time.py > $GRAFT_REPO_ROOT/$O/prof_train.log 2>&1; tail -3 $GRAFT_REPO_ROOT/$O/prof_train.log
ls $GRAFT_REPO_ROOT/$O/prof_train/*/ | head
