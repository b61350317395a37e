set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02k; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 120 python tools/dec_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/dec.log
timeout -k 10 1100 python -m pytest tests -q -m gpu > $O/t.log 2>&1; tail -8 $O/t.log
