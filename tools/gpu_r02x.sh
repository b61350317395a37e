cd $GRAFT_REPO_ROOT
O=gpurun_out/r02x; mkdir -p $O; rm -f $O/var.log
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
for v in ${VARS:-0 8 16 2}; do
  echo "== FD_WINO_VAR=$v" | tee -a $O/var.log
  FD_WINO_VAR=$v WINO_SHORT=1 timeout -k 10 200 python tools/time_wino.py 2>&1 | grep -v amdgpu.ids | tee -a $O/var.log || exit 1
done
rocprofv3 -L 2>/dev/null | grep -o "TCP_[A-Z_0-9a-z]*\|TCC_[A-Z_0-9a-z]*\|TA_[A-Z_0-9a-z]*" | sort -u > $O/counters.txt; wc -l $O/counters.txt
