cd $GRAFT_REPO_ROOT
O=gpurun_out/r03h; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
B="python bench.py --no-fast-mode --no-train-step --no-cpu-baseline"
for rule in "" "256,200000" "512,100000" "2048,20000"; do
  echo "== FD_WAVE_RULE=$rule"; FD_WAVE_RULE=$rule timeout -k 10 300 $B 2>/dev/null | tail -1 | cut -c1-110
  FD_WAVE_RULE=$rule python bench.py --inflight 1 --layer-times $O/lt_$rule.tsv > /dev/null 2>&1; tail -1 $O/lt_$rule.tsv
done
echo "== again baseline"; timeout -k 10 300 $B 2>/dev/null | tail -1 | cut -c1-110
paste <(cut -f2,3,6 $O/lt_.tsv) <(cut -f3,6 $O/lt_256,200000.tsv) <(cut -f3,6 $O/lt_512,100000.tsv) <(cut -f3,6 "$O/lt_2048,20000.tsv") | grep -E "layer1|layer2.0|layer2.1|layer3.1|head.pw1|tf" | head -30
