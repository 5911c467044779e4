# fused update + data gradient below its client threshold (P2PFL_FUSE_DGRAD_FORCE), 8-row ring = one workgroup per CU
run() { name=$1; shift; out=$(env "$@" timeout -k 10 300 python bench.py --gpus 1 --steps 6 --warmup 2 --no-reference-run $EXTRA 2>/dev/null) || { echo "$name: FAILED"; return 1; }; echo "$name: $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],1), "ms", d["config"]["last_test_accuracy_mean"])')"; }
EXTRA="--config cnn10-fully"
run "cnn10-fully default" A=1 &&
run "cnn10-fully forced fused dgrad" P2PFL_FUSE_DGRAD_FORCE=1 &&
for k in 20 30 40; do
EXTRA="--nodes-per-gpu $k --split-nodes 100"
run "$k clients default" A=1 &&
run "$k clients forced fused dgrad" P2PFL_FUSE_DGRAD_FORCE=1 || exit 1
done
