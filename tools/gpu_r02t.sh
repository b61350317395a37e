set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02t; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_effnet_gpu.py -q -m gpu > $O/t.log 2>&1; tail -6 $O/t.log
timeout -k 10 300 python bench.py --model FCOS-B3 --size 832x1344 --steps 10 --warmup 3 --no-fast-mode --no-train-step 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B3', d['value'], d['ms_per_step'])"
timeout -k 10 300 python bench.py --model FCOS-B3 --size 832x1344 --layer-times $O/layers_b3.tsv > /dev/null 2>&1; tail -1 $O/layers_b3.tsv
