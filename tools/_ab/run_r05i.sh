cd $GRAFT_REPO_ROOT
O=gpurun_out/r05i; mkdir -p $O
for st in 1 0 1 0; do
  FD_AMP_F16_STORE=$st timeout -k 10 300 python bench.py --mode train --amp --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/train_amp_store$st.json
  echo "store=$st $(cut -c1-180 $O/train_amp_store$st.json)"
done
