# ring depth of the fused update (P2PFL_UPDATE_DEPTH) across the configurations that take it
run() { timeout -k 10 300 python bench.py $2 --steps ${3:-10} --warmup 3 --no-reference-run > gpurun_out/d.json 2> gpurun_out/d.err || { tail -3 gpurun_out/d.err; exit 1; }
  echo "depth=${P2PFL_UPDATE_DEPTH:-8} $1 $(python3 -c "import json; d=json.loads(open('gpurun_out/d.json').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1))")"; }
for rep in 1 2; do for d in 8 4; do
  export P2PFL_UPDATE_DEPTH=$d
  run churn "--config cnn100-encrypted-churn"
  run mfma "--dtype fp32-mfma"
  run n200 "--nodes-per-gpu 200"
  run n1000 "--nodes-per-gpu 1000" 3
done; done
