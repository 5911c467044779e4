# client groups against the client count (two groups need >= 31 clients each for the fused update + data gradient)
run() { name=$1; shift; out=$(env "$@" timeout -k 10 300 python bench.py --gpus 1 --steps 6 --warmup 2 --no-reference-run $EXTRA 2>/dev/null) || { echo "$name: FAILED"; return 1; }; echo "$name: $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],1), "ms", d["config"]["last_test_accuracy_mean"])')"; }
for k in 64 72 80; do
EXTRA="--nodes-per-gpu $k --split-nodes 100"
run "$k clients G=1" P2PFL_GROUPS=1 &&
run "$k clients G=2" P2PFL_GROUPS=2 || exit 1
done
