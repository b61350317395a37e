# F(4x4) kernel development: stand-alone layer times of several builds of the library (tools/_ab/lib<name>.so, selected through FD_LIB) on the same box, two rounds
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-w4var}; mkdir -p $O; shift
for round in 1 2; do
for v in "$@"; do
  echo "== $v (round $round)"; FD_LIB=$PWD/tools/_ab/lib$v.so timeout -k 10 200 python tools/time_wino4.py 2>&1 | grep -E "ms" | sed 's/F(2x2).*| F(4x4)/F(4x4)/' | sed 's/TF\/s-eq.*max/max/'
done
done | tee $O/time_wino4_var.txt
