cd $GRAFT_REPO_ROOT
bash tools/gpu_tune_misses.sh r05f
O=gpurun_out/r05f
grep -E "tower|cls_logits|conv2|conv4|HisBlock3.conv3" $O/layer_times.tsv | cut -c1-100
timeout -k 10 600 python bench.py --no-train-step --no-fast-mode --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; cut -c1-1500 $O/bench.json
FD_W4_SK=0 timeout -k 10 600 python bench.py --no-train-step --no-fast-mode --no-cpu-baseline > $O/bench_nosk.json 2> $O/bench_nosk.err; cut -c1-200 $O/bench_nosk.json
