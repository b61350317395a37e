# F(4x4) kernel development: stand-alone layer times of several builds of the library (csrc/_build/lib_<name>.so) on the same box, two rounds
cd $GRAFT_REPO_ROOT
L=pytorch_object_detection_amd/csrc
O=gpurun_out/${1:-w4var}; mkdir -p $O; shift
cp $L/libfcosdet_hip.so /tmp/libcur.so
for round in 1 2; do
for v in "$@"; do
  cp $L/_build/lib$v.so $L/libfcosdet_hip.so
  echo "== $v (round $round)"; timeout -k 10 200 python tools/time_wino4.py 2>&1 | grep -E "ms" | sed 's/F(2x2).*| F(4x4)/F(4x4)/' | sed 's/TF\/s-eq.*max/max/'
done
done | tee $O/time_wino4_var.txt
cp /tmp/libcur.so $L/libfcosdet_hip.so
