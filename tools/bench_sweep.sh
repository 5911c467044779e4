#!/bin/bash
# Every bench preset in every precision mode, one JSON line per run under <out>/ (GPU box, repo root).
#   usage: bash tools/bench_sweep.sh gpurun_out/sweep
out=${1:-gpurun_out/sweep}
mkdir -p "$out"
for d in fp32 fp32-mfma bf16; do
  for p in headline cnn10-fully femnist50-ring cnn100-encrypted-churn femnist1000-star mlp100-random; do
    cfg=""; [ "$p" != headline ] && cfg="--config $p"
    timeout -k 10 200 python bench.py $cfg --dtype $d --steps 10 --warmup 3 --no-reference-run > "$out/${p}_$d.json" 2> "$out/${p}_$d.err" || { echo "FAILED $p $d"; tail -3 "$out/${p}_$d.err"; exit 1; }
    echo "$p $d $(python3 -c "import json,sys; d=json.loads(open('$out/${p}_$d.json').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1),'ms', round(d['value']), d['unit'])")"
  done
done
