cd $GRAFT_REPO_ROOT
O=gpurun_out/r02sk; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 300 python -m pytest tests/test_layers_gpu.py -q -m gpu -k "winograd" 2>&1 | tail -3
for b in 1 2; do
timeout -k 10 200 python bench.py --batch $b --size 512 --inflight 1 --steps 50 --warmup 10 --no-fast-mode --no-train-step --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('batch $b 512x512 inflight 1:', d['ms_per_step'], 'ms/step', d['value'], 'img/s')"
done
timeout -k 10 200 python bench.py --batch 1 --size 512 --inflight 1 --layer-times $O/layers_b1.tsv > /dev/null 2>&1
timeout -k 10 300 python bench.py --inflight 1 --layer-times $O/layers_b16.tsv > /dev/null 2>&1
timeout -k 10 400 python bench.py --no-cpu-baseline --no-fast-mode --no-train-step 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench:', d['value'], 'img/s', d['ms_per_step'], 'ms; tower', d['roofline']['avg_launch_ms'], 'exec frac', d['roofline'].get('mfma_executed_frac'))"
