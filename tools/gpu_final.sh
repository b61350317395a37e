# A round's final measurement set in ONE gpurun call: full -m gpu suite, smoke, rocprof kernel stats, PMC passes, layer times, every bench line.
#   usage (GPU box, repo root):  bash tools/gpu_final.sh <tag>      -> gpurun_out/<tag>/...
#   afterwards (here):           bash tools/gpu_final.sh --collect <tag>   copies the summaries into profiles/<tag>_*
if [ "$1" = "--collect" ]; then
  tag=$2; O=gpurun_out/$tag
  for f in $O/*.json $O/*.tsv $O/*.csv $O/smoke.log; do [ -f "$f" ] && cp "$f" profiles/${tag}_$(basename $f); done
  cp $O/pytest.log profiles/${tag}_pytest_gpu.log; [ -f $O/pmc_traffic.json ] && cp $O/pmc_traffic.json profiles/pmc_traffic.json; rm -f profiles/${tag}_pmc_traffic.json
  exit 0
fi
set -x
tag=${1:-final}
cd $GRAFT_REPO_ROOT
export FD_COMMIT=${FD_COMMIT:-unknown}
O=gpurun_out/$tag; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
PART=${FD_FINAL_PART:-all}     # A: tests, smoke, rocprof, PMC, layer times, the headline line; B: every other line; all: both (needs > 20 min)
if [ "$PART" != "B" ]; then
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
bash tools/prof.sh $tag 2>&1 | tail -3
cp gpurun_out/prof_$tag/trace_kernel_stats.csv $O/rocprofv3_kernel_stats.csv; rm -f gpurun_out/prof_$tag/trace_kernel_trace.csv
bash tools/pmc.sh $tag 2>&1 | tail -3
python tools/pmc_summary.py $tag > $O/pmc_summary.log 2>&1; tail -3 $O/pmc_summary.log
cp profiles/${tag}_pmc_summary.json $O/pmc_summary.json; cp profiles/pmc_traffic.json $O/pmc_traffic.json
rm -rf gpurun_out/pmc_$tag/*/pmc_kernel_trace.csv gpurun_out/pmc_$tag/*/pmc_counter_collection.csv
python bench.py --inflight 1 --layer-times $O/layer_times.tsv > /dev/null 2>&1; tail -1 $O/layer_times.tsv
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; cut -c1-200 $O/bench.json
fi
if [ "$PART" = "A" ]; then du -sh gpurun_out; exit 0; fi
timeout -k 10 300 python bench.py --model FCOS-B3 --size 832x1344 --steps 10 --warmup 3 --no-fast-mode --no-train-step > $O/bench_fcos_b3_832x1344.json 2> $O/bench_b3.err
timeout -k 10 300 python bench.py --model FCOS-B3 --size 832x1344 --layer-times $O/layer_times_fcos_b3.tsv > /dev/null 2>&1
timeout -k 10 300 python bench.py --model FCOS --no-fast-mode --no-train-step > $O/bench_fcos_r50.json 2> $O/bench_fcos.err
timeout -k 10 300 python bench.py --model MNFCOS --no-fast-mode --no-train-step --no-cpu-baseline > $O/bench_mnfcos.json 2> $O/bench_mn.err
timeout -k 10 300 python bench.py --mode train > $O/bench_train.json 2> $O/bench_train.err
timeout -k 10 300 python bench.py --mode train --amp > $O/bench_train_amp.json 2> $O/bench_train_amp.err
bash tools/prof_train.sh ${tag}_train 2>&1 | tail -3
cp gpurun_out/prof_${tag}_train/last_step_kernel_stats.csv $O/train_step_kernel_stats.csv; rm -f gpurun_out/prof_${tag}_train/trace_kernel_trace.csv
FD_AMP=1 bash tools/prof_train.sh ${tag}_train_amp 2>&1 | tail -3
cp gpurun_out/prof_${tag}_train_amp/last_step_kernel_stats.csv $O/train_step_amp_kernel_stats.csv; rm -f gpurun_out/prof_${tag}_train_amp/trace_kernel_trace.csv
for b in 1 2; do
  timeout -k 10 200 python bench.py --batch $b --size 512 --inflight 1 --steps 100 --warmup 20 --no-fast-mode --no-train-step --no-cpu-baseline > $O/bench_latency_b${b}_512.json 2>/dev/null
  timeout -k 10 200 python bench.py --batch $b --size 512 --graph --steps 100 --warmup 20 --no-fast-mode --no-train-step --no-cpu-baseline > $O/bench_latency_b${b}_512_graph.json 2>/dev/null
done
FD_WINOGRAD=0 timeout -k 10 300 python bench.py --no-fast-mode --no-train-step --no-cpu-baseline > $O/bench_direct_kernels_only.json 2> $O/bench_direct.err
# N = 2 control flow on ONE GPU (gloo carries the collectives; never a measurement): bench.py launches its own ranks
FD_BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-fast-mode --no-train-step > $O/rehearsal_gloo2_infer.json 2> $O/rehearsal_gloo2_infer.err; echo "gloo2 rc=$?"
FD_BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --mode train --steps 3 --warmup 1 > $O/rehearsal_gloo2_train.json 2> $O/rehearsal_gloo2_train.err; echo "gloo2 train rc=$?"
FD_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fast-mode --no-train-step > $O/rccl_world1.json 2> /dev/null; echo "rccl world1 rc=$?"
du -sh gpurun_out
