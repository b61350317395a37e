# experiment set: narrow-conv weight delivery (DPP vs LDS broadcast), back-to-back GEMM wave height and layers -- per-layer times of each variant on one box
cd $GRAFT_REPO_ROOT
O=gpurun_out/exp1; mkdir -p $O
run() { name=$1; shift; env "$@" timeout -k 10 300 python bench.py --inflight 1 --layer-times $O/lt_$name.tsv > /dev/null 2> $O/lt_$name.err; echo "== $name: $(tail -1 $O/lt_$name.tsv)"; grep -E "cnt_reg|layer1\.[012]\.conv3|layer1\.[12]\.conv1|layer2\.[0123]\.conv(1|3)|layer3\.0\.conv1" $O/lt_$name.tsv | cut -f2,3 | tr '\n' ' '; echo; }
run base FD_B2B= FD_NARROW=0
run dpp FD_B2B=1
run ldsb FD_B2B=1 FD_NARROW_MODE=1
run tm1 FD_B2B=1 FD_B2B_TM1=1
run l12 FD_B2B=12
run l12tm1 FD_B2B=12 FD_B2B_TM1=1
