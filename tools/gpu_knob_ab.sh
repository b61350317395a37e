# same-box A/B of environment knobs of the plan builder (engine.py): for each NAME given, the headline bench line and the one-lane layer times with NAME=0 and with
# NAME=1 (everything else at its default), two alternating rounds.   usage (GPU box, repo root): bash tools/gpu_knob_ab.sh <tag> NAME [NAME ...] -> gpurun_out/<tag>/
tag=$1; shift
cd $GRAFT_REPO_ROOT
O=gpurun_out/$tag; mkdir -p $O
for round in 1 2; do
  for knob in "$@"; do
    for v in 0 1; do
      echo "== $knob=$v (round $round)"
      env $knob=$v timeout -k 10 300 python bench.py --no-fast-mode --no-train-step --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-110
      if [ $round = 1 ]; then
        env $knob=$v timeout -k 10 300 python bench.py --inflight 1 --layer-times $O/layers_${knob}_$v.tsv > /dev/null 2>&1; tail -1 $O/layers_${knob}_$v.tsv
      fi
    done
  done
done
