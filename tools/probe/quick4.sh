set -e
timeout -k 10 300 python -u -m pytest tests/test_hip_kernels.py -x -q -k "conv1" --timeout 120 --timeout-method thread 2>&1 | tail -3
for p in headline cnn10-fully femnist50-ring femnist1000-star; do
  cfg=""; [ "$p" != headline ] && cfg="--config $p"
  timeout -k 10 200 python bench.py $cfg --steps 10 --warmup 3 --no-reference-run > gpurun_out/q_$p.json 2> gpurun_out/q_$p.err
  echo "$p $(python3 -c "import json; d=json.loads(open('gpurun_out/q_$p.json').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1))")"
done
