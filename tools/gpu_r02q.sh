set -x
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build_q.log 2>&1 || { tail -20 gpurun_out/build_q.log; exit 1; }
bash tools/prof.sh r02q 2>&1 | tail -3
bash tools/pmc.sh r02q 2>&1 | tail -3
bash tools/prof_train.sh r02q_train 2>&1 | tail -3
python bench.py --inflight 1 --layer-times gpurun_out/layers_r02q.tsv > /dev/null 2>&1; tail -1 gpurun_out/layers_r02q.tsv
timeout -k 10 400 python bench.py > gpurun_out/bench_r02q.json 2> gpurun_out/bench_r02q.err; tail -2 gpurun_out/bench_r02q.err
timeout -k 10 300 python bench.py --model FCOS-B3 --size 832x1344 --steps 10 --warmup 3 --no-fast-mode --no-train-step > gpurun_out/bench_r02q_b3.json 2> gpurun_out/bench_r02q_b3.err
timeout -k 10 300 python bench.py --model FCOS --no-fast-mode --no-train-step > gpurun_out/bench_r02q_fcos.json 2> gpurun_out/bench_r02q_fcos.err
timeout -k 10 300 python bench.py --mode train > gpurun_out/bench_r02q_train.json 2> gpurun_out/bench_r02q_train.err
python - <<'PY'
import json
for f in ("bench_r02q", "bench_r02q_b3", "bench_r02q_fcos", "bench_r02q_train"):
    try:
        d=json.load(open("gpurun_out/%s.json" % f)); print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"])
    except Exception as e: print(f, "ERR", e)
PY
