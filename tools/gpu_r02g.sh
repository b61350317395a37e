set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02g; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_postproc_gpu.py -x -q -m gpu > $O/t.log 2>&1; tail -3 $O/t.log
FD_AUTOTUNE=1 timeout -k 10 900 python bench.py --no-fast-mode --no-train-step --no-cpu-baseline --save-tuning > $O/bench_pair.json 2> $O/bench_pair.err; tail -3 $O/bench_pair.err
cp pytorch_object_detection_amd/tuned/gfx950_tiles.json $O/gfx950_tiles.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02g/bench_pair.json"))
print("pair-tuned:", d["value"], d["roofline"]["frac"], d["postproc"]["batch16"])
PY
timeout -k 10 300 python bench.py --no-fast-mode --no-train-step --no-cpu-baseline > $O/bench2.json 2> $O/bench2.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02g/bench2.json"))
print("second run:", d["value"], d["roofline"]["frac"])
PY
