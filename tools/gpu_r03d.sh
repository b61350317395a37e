# Round 3, fourth GPU call: AMP (f16 MFMA forward / data gradient) tests and the train-step bench with and without --amp.  -> gpurun_out/r03d/
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_amp_gpu.py tests/test_train_gpu.py tests/test_train_nodes_gpu.py -m gpu -q -x --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
timeout -k 10 300 python bench.py --mode train > $O/bench_train.json 2> $O/bench_train.err; echo "train rc=$?"; cut -c1-250 $O/bench_train.json
timeout -k 10 300 python bench.py --mode train --amp > $O/bench_train_amp.json 2> $O/bench_train_amp.err; echo "train amp rc=$?"; cut -c1-250 $O/bench_train_amp.json; tail -2 $O/bench_train_amp.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03d/bench_train_amp.json"))
print(json.dumps(d.get("roofline_amp_f16")))
PY
