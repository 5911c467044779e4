# shard sizes on one card (the first bench of a fresh box is slow: one throw-away run first)
mkdir -p gpurun_out/cap
timeout -k 10 300 python bench.py --nodes-per-gpu 500 --steps 2 --warmup 1 --no-reference-run > /dev/null 2>&1
for n in 1000 2000 2400; do
  timeout -k 10 600 python bench.py --nodes-per-gpu $n --steps 3 --warmup 1 --no-reference-run > gpurun_out/cap/cnn$n.json 2> gpurun_out/cap/cnn$n.err || { tail -3 gpurun_out/cap/cnn$n.err; exit 1; }
  echo "$n $(python3 -c "import json; d=json.loads(open('gpurun_out/cap/cnn$n.json').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), round(d['value']))")"
done
