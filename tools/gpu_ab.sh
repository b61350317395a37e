# same-box A/B of two builds of the library: tools/_ab/libold.so (a previous build, copied there by hand: `cp csrc/libfcosdet_hip.so tools/_ab/libold.so` before the
# change) against the current one.  The variants are selected through FD_LIB (pytorch_object_detection_amd/_lib.py): the product library is never overwritten.
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for which in new old; do
    if [ $which = old ]; then export FD_LIB=$PWD/tools/_ab/libold.so; else unset FD_LIB; fi
    echo "== $which (round $round)"
    timeout -k 10 300 python bench.py --no-fast-mode --no-train-step --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-110
    if [ $round = 1 ]; then
      timeout -k 10 300 python bench.py --model FCOS-B3 --size 832x1344 --steps 10 --warmup 3 --no-fast-mode --no-train-step --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-110
      timeout -k 10 300 python bench.py --inflight 1 --layer-times gpurun_out/layers_$which.tsv > /dev/null 2>&1; tail -1 gpurun_out/layers_$which.tsv
    fi
  done
done
