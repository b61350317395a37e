cd $GRAFT_REPO_ROOT
O=gpurun_out/r05l; mkdir -p $O
for f in 1 0 1 0; do
  FD_BENCH_FUSED_OPT=$f timeout -k 10 300 python bench.py --mode train --amp --steps 20 --warmup 5 2>$O/err_$f.txt | tail -1 > $O/train_amp_fused$f.json
  echo "fused=$f $(cut -c75-175 $O/train_amp_fused$f.json)"; tail -2 $O/err_$f.txt | cut -c1-200
done
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05l/train_amp_fused1.json")); print(json.dumps(d["roofline_amp_f16"])[:400]); print(d["final_loss"])
d=json.load(open("gpurun_out/r05l/train_amp_fused0.json")); print(d["final_loss"])
PY
