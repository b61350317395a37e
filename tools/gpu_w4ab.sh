# F(4x4) kernel development: parity tests of the current build, then stand-alone layer times of csrc/_build/libold.so against the current library on the same box
cd $GRAFT_REPO_ROOT
L=pytorch_object_detection_amd/csrc
O=gpurun_out/${1:-w4ab}; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_layers_gpu.py tests/test_model_gpu.py -q -m gpu -x -k "winograd or wino or f4x4 or shipped_plan or full_hisfcos" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
cp $L/libfcosdet_hip.so /tmp/libnew.so
for which in old new old new; do
  if [ $which = old ]; then cp $L/_build/libold.so $L/libfcosdet_hip.so; else cp /tmp/libnew.so $L/libfcosdet_hip.so; fi
  echo "== $which"; timeout -k 10 200 python tools/time_wino4.py 2>&1 | grep -E "ms" | sed 's/F(2x2).*| F(4x4)/F(4x4)/'
done | tee $O/time_wino4_ab.txt
cp /tmp/libnew.so $L/libfcosdet_hip.so
