# A/B matrix of the client-group schedule (bench.py, 8 timed rounds each)
run() { name=$1; shift; out=$(env "$@" timeout -k 10 300 python bench.py --gpus 1 --steps 8 --warmup 2 --no-reference-run $EXTRA 2>/dev/null) || { echo "$name: FAILED"; return 1; }; echo "$name: $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],1), "ms", d["config"]["last_test_accuracy_mean"])')"; }
EXTRA=""
run "headline G=2 (default)" A=1 &&
run "headline G=2 depth 2" P2PFL_UPDATE_DEPTH=2 &&
# (small-footprint convolution variants beside the update, a since removed switch: 261.0 ms)
# (depth 2 + small variants: 261.2 ms)
EXTRA="--config cnn10-fully" &&
run "cnn10-fully G=1 (default)" A=1 &&
run "cnn10-fully G=2" P2PFL_GROUPS=2 &&
EXTRA="--nodes-per-gpu 200 --split-nodes 100" &&
run "200 nodes G=1" P2PFL_GROUPS=1 &&
run "200 nodes G=2 (default)" A=1 &&
run "200 nodes G=4" P2PFL_GROUPS=4
