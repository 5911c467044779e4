#!/usr/bin/env bash
# Scaling curve of the headline benchmark on ONE node: N = 1, 2, 4, 8 ranks (one per GPU, RCCL over xGMI), the same flags
# for every N. bench.py --gpus N starts its ranks itself; every run prints one JSON line, collected in $OUT.
#   tools/scale_curve.sh [--steps 20 --warmup 5 ...]      (needs an 8-GPU box: a rank exits non-zero if GPUs are missing)
set -euo pipefail
cd "$(dirname "$0")/.."
OUT=${OUT:-gpurun_out/scale_curve.jsonl}
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
for n in ${NS:-1 2 4 8}; do
    echo "== $n rank(s)" >&2
    timeout -k 10 900 python bench.py --gpus "$n" --no-reference-run "$@" | tee -a "$OUT"
done
python - "$OUT" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
base = rows[0]["value"] / rows[0]["n_gpus"]
for r in rows:
    print(f'{r["n_gpus"]} GPU: {r["ms_per_step"]:.1f} ms/round  {r["value"]:.0f} samples/s  '
          f'efficiency {r["value"] / (base * r["n_gpus"]):.3f}')
PY
