# time the conv shapes the committed tile table does not know (FD_AUTOTUNE=1: misses only) inside the bench plan and bring the merged table back
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-tune}; mkdir -p $O
FD_AUTOTUNE=1 timeout -k 10 600 python bench.py --inflight 1 --save-tuning --steps 5 --warmup 3 --no-fast-mode --no-train-step --no-cpu-baseline > $O/bench_tune.json 2> $O/bench_tune.err; tail -2 $O/bench_tune.err
cp pytorch_object_detection_amd/tuned/gfx950_tiles.json $O/gfx950_tiles.json
timeout -k 10 300 python bench.py --inflight 1 --layer-times $O/layer_times.tsv > /dev/null 2>&1; tail -1 $O/layer_times.tsv
grep -E "downsample|\+layer" $O/layer_times.tsv
