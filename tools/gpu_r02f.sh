set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02f; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_postproc_gpu.py tests/test_model_gpu.py -x -q -m gpu > $O/t.log 2>&1; tail -5 $O/t.log
FD_AUTOTUNE=1 timeout -k 10 900 python bench.py --model FCOS-B3 --size 832x1344 --steps 5 --warmup 2 --no-fast-mode --no-train-step --save-tuning > $O/bench_b3_tuned.json 2> $O/bench_b3_tuned.err; cat $O/bench_b3_tuned.json | head -c 1500; tail -3 $O/bench_b3_tuned.err
cp pytorch_object_detection_amd/tuned/gfx950_tiles.json $O/gfx950_tiles.json
timeout -k 10 200 python bench.py --model FCOS-B3 --size 832x1344 --layer-times $O/layers_b3.tsv > /dev/null 2>&1; tail -3 $O/layers_b3.tsv
timeout -k 10 300 python bench.py --no-fast-mode --no-train-step --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r02f/bench.json"))
print(d["value"], d["roofline"]["frac"], d["postproc"])
PY
