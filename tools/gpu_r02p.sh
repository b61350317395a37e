set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02p; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
FD_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --no-fast-mode --no-cpu-baseline --no-train-step > $O/bench_dist1.json 2> $O/bench_dist1.err; tail -2 $O/bench_dist1.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02p/bench_dist1.json"))
print("forced RCCL world 1:", d["value"], d["roofline"]["frac"], d["detections_kept_rank0"])
PY
FD_AUTOTUNE=force timeout -k 10 900 python bench.py --inflight 1 --no-fast-mode --no-train-step --no-cpu-baseline --save-tuning > $O/bench_tune.json 2> $O/bench_tune.err; tail -2 $O/bench_tune.err
cp pytorch_object_detection_amd/tuned/gfx950_tiles.json $O/gfx950_tiles.json
timeout -k 10 300 python bench.py --no-fast-mode --no-cpu-baseline --no-train-step > $O/bench2.json 2> $O/bench2.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02p/bench2.json"))
print("after re-tune:", d["value"], d["roofline"]["frac"])
PY
timeout -k 10 300 python bench.py --inflight 1 --layer-times $O/layers.tsv > /dev/null 2>&1; tail -2 $O/layers.tsv
