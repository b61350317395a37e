cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1 || exit 1
for w in 0 1; do for b in 1 2 4; do
FD_WINOGRAD=$w timeout -k 10 200 python bench.py --batch $b --size 512 --inflight 1 --steps 50 --warmup 10 --no-fast-mode --no-train-step --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('WINOGRAD=$w batch $b 512x512:', d['ms_per_step'], 'ms/step', d['value'], 'img/s')"
done; done
FD_WINOGRAD=1 timeout -k 10 200 python bench.py --batch 1 --size 512 --inflight 1 --layer-times gpurun_out/layers_b1_wino.tsv > /dev/null 2>&1
FD_WINOGRAD=0 timeout -k 10 200 python bench.py --batch 1 --size 512 --inflight 1 --layer-times gpurun_out/layers_b1_direct.tsv > /dev/null 2>&1
