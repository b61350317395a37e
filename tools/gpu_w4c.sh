cd $GRAFT_REPO_ROOT
O=gpurun_out/w4c; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 300 python bench.py --inflight 1 --layer-times $O/layer_times_w4.tsv > /dev/null 2>&1
FD_WINOGRAD4=0 timeout -k 10 300 python bench.py --inflight 1 --layer-times $O/layer_times_w2.tsv > /dev/null 2>&1
tail -1 $O/layer_times_w4.tsv; tail -1 $O/layer_times_w2.tsv
