cd $GRAFT_REPO_ROOT
O=gpurun_out/r05g; mkdir -p $O
FD_W4_SK_TRACE=1 FD_AUTOTUNE=1 timeout -k 10 120 python bench.py --inflight 1 --layer-times $O/lt.tsv > $O/out.txt 2> $O/err.txt
echo rc=$?; grep "w4sk" $O/err.txt | tail -12 | cut -c1-250; tail -3 $O/err.txt | cut -c1-200
