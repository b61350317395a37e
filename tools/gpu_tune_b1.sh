# the latency path (batch 1, 512x512, test.py:202-223): time the conv shapes of its plan the committed tile table does not know (FD_AUTOTUNE=1: misses only),
# bring the merged table back, and time the plan before / after plus its per-layer breakdown
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-tune_b1}; mkdir -p $O
L="--batch 1 --size 512 --inflight 1 --no-fast-mode --no-train-step --no-cpu-baseline"
timeout -k 10 200 python bench.py $L --steps 100 --warmup 20 > $O/bench_before.json 2>/dev/null
timeout -k 10 200 python bench.py $L --layer-times $O/layer_times_before.tsv > /dev/null 2>&1
FD_AUTOTUNE=1 timeout -k 10 600 python bench.py $L --save-tuning --steps 5 --warmup 3 > $O/bench_tune.json 2> $O/bench_tune.err; tail -2 $O/bench_tune.err
cp pytorch_object_detection_amd/tuned/gfx950_tiles.json $O/gfx950_tiles.json
timeout -k 10 200 python bench.py $L --steps 100 --warmup 20 > $O/bench_after.json 2>/dev/null
timeout -k 10 200 python bench.py $L --graph --steps 100 --warmup 20 > $O/bench_after_graph.json 2>/dev/null
timeout -k 10 200 python bench.py $L --layer-times $O/layer_times_after.tsv > /dev/null 2>&1
python - $O <<'PY'
import json, sys
for n in ("before", "after", "after_graph"):
    for l in open(f"{sys.argv[1]}/bench_{n}.json"):
        if l.startswith("{"):
            d = json.loads(l); print(n, d["value"], d["ms_per_step"])
PY
