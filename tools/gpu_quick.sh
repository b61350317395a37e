# One gpurun call during development: build, the -m gpu suite (or a -k subset), layer times and one short bench line.
#   usage (GPU box, repo root): bash tools/gpu_quick.sh <tag> ["pytest -k expression" | "none"]   -> gpurun_out/<tag>/
tag=${1:-q}; kexpr=${2:-}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$tag; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
if [ "$kexpr" != "none" ]; then
  if [ -n "$kexpr" ]; then timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "$kexpr" -s > $O/pytest.log 2>&1; else timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 -s > $O/pytest.log 2>&1; fi
  echo "pytest rc=$?"; grep -E "max \|err\||passed|failed|error" $O/pytest.log | tail -25
fi
timeout -k 10 300 python bench.py --inflight 1 --layer-times $O/layer_times.tsv > /dev/null 2> $O/layer_times.err; tail -1 $O/layer_times.tsv
timeout -k 10 600 python bench.py --no-train-step --no-fast-mode > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; cut -c1-300 $O/bench.json
