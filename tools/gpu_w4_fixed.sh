# the F(4x4) kernel's fixed per-workgroup cost, split into its parts.  Timing build = tools/_ab/libw4timing.so (built in the container:
# csrc objects + fd_conv_wino4.hip with -DFD_W4_TIMING), selected through FD_LIB.   usage: bash tools/gpu_w4_fixed.sh <tag>   -> gpurun_out/<tag>/w4_fixed.txt
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/${1:-w4fixed}; mkdir -p $O
export FD_LIB=$PWD/tools/_ab/libw4timing.so
{
echo "== shipped library"; FD_LIB= timeout -k 10 200 python tools/time_wino4_fixed.py
for d in 0 1 2 4 5 6 7; do echo "== timing build FD_W4_DBG=$d (1 = no loader, 2 = no MFMAs, 4 = no epilogue)"; FD_W4_DBG=$d timeout -k 10 200 python tools/time_wino4_fixed.py; done
} > $O/w4_fixed.txt 2>&1
rm -f $O/w4_ts.txt
for d in 0 1 2 4; do FD_W4_DBG=$d FD_W4_TS=$O/w4_ts.txt timeout -k 10 200 python tools/time_wino4_fixed.py >> $O/w4_fixed.txt 2>&1; done
cat $O/w4_fixed.txt; cat $O/w4_ts.txt
