# fp32-mfma policy: presets in one call
for p in headline cnn10-fully femnist50-ring femnist1000-star; do
  cfg=""; [ "$p" != headline ] && cfg="--config $p"
  timeout -k 10 200 python bench.py $cfg --dtype fp32-mfma --steps 10 --warmup 3 > gpurun_out/qm.json 2> gpurun_out/qm.err || { tail -3 gpurun_out/qm.err; exit 1; }
  echo "$p fp32-mfma $(python3 -c "import json; d=json.loads(open('gpurun_out/qm.json').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1))")"
done
