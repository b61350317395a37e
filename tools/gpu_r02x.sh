cd $GRAFT_REPO_ROOT
O=gpurun_out/r02x; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
for v in 2 0 2 0; do
  FD_WINO_NCH=$v timeout -k 10 300 python bench.py --no-fast-mode --no-train-step --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('NCH=$v', d['value'], 'img/s', d['ms_per_step'], 'ms  tower', d['roofline']['avg_launch_ms'], 'ms exec frac', d['roofline'].get('mfma_executed_frac'))"
done
