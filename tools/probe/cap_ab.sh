# 1000 nodes on the card: which of the round-3 kernel changes pay there (two passes, the first run of a box is slow)
run() { timeout -k 10 300 python bench.py --nodes-per-gpu 1000 --steps 3 --warmup 1 --no-reference-run > gpurun_out/cap.json 2> gpurun_out/cap.err || { tail -3 gpurun_out/cap.err; exit 1; }
  echo "$1 $(python3 -c "import json; d=json.loads(open('gpurun_out/cap.json').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1))")"; }
for rep in 1 2; do
run default
P2PFL_HEAD_SPLIT=0 run head-one-kernel
P2PFL_CONV2_DENSE=0 run conv2-window-row
done
