cd $GRAFT_REPO_ROOT
O=gpurun_out/r05h; mkdir -p $O
timeout -k 10 200 python -m pytest tests/test_layers_gpu.py -x -q -m gpu -k "winograd_f4x4" 2>&1 | tail -3
FD_W4_SK_TRACE=1 FD_AUTOTUNE=1 timeout -k 10 200 python bench.py --inflight 1 --save-tuning --steps 5 --warmup 3 --no-fast-mode --no-train-step --no-cpu-baseline > $O/bench_tune.json 2> $O/bench_tune.err || { tail -5 $O/bench_tune.err; exit 1; }
grep -c "w4sk" $O/bench_tune.err
cp pytorch_object_detection_amd/tuned/gfx950_tiles.json $O/gfx950_tiles.json
grep w4sk $O/gfx950_tiles.json
timeout -k 10 300 python bench.py --inflight 1 --layer-times $O/layer_times.tsv > /dev/null 2>&1; tail -1 $O/layer_times.tsv
grep -E "tower|cls_logits|conv2|conv4|HisBlock3.conv3" $O/layer_times.tsv | cut -c1-100
timeout -k 10 600 python bench.py --no-train-step --no-fast-mode --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; cut -c1-1800 $O/bench.json
FD_W4_SK=0 timeout -k 10 600 python bench.py --no-train-step --no-fast-mode --no-cpu-baseline > $O/bench_nosk.json 2> $O/bench_nosk.err; cut -c1-200 $O/bench_nosk.json
