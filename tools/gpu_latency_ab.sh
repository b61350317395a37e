# the latency path (batch 1 and 2, 512 x 512, test.py:202-223), eager and as a HIP graph, with NAME=VALUE settings given on the command line against the defaults, same box,
# two alternating rounds.   usage (GPU box, repo root): bash tools/gpu_latency_ab.sh <tag> NAME=VALUE [NAME=VALUE ...]   -> gpurun_out/<tag>/
tag=$1; shift
cd $GRAFT_REPO_ROOT
O=gpurun_out/$tag; mkdir -p $O
L="--size 512 --steps 200 --warmup 30 --no-fast-mode --no-train-step --no-cpu-baseline"
for round in 1 2; do
  for setting in default "$@"; do
    for b in 1 2; do
      for g in "--inflight 1" "--graph"; do
        if [ "$setting" = default ]; then r=$(timeout -k 10 200 python bench.py --batch $b $g $L 2>/dev/null | tail -1); else r=$(env $setting timeout -k 10 200 python bench.py --batch $b $g $L 2>/dev/null | tail -1); fi
        echo "$setting batch=$b $g round=$round: $(echo "$r" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms", d["value"], "img/s")')" | tee -a $O/latency_ab.txt
      done
    done
  done
done
