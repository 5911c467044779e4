# more than two client groups (each >= 31 clients keeps the fused update + data gradient)
run() { name=$1; shift; out=$(env "$@" timeout -k 10 300 python bench.py --gpus 1 --steps 6 --warmup 2 --no-reference-run $EXTRA 2>/dev/null) || { echo "$name: FAILED"; return 1; }; echo "$name: $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],1), "ms", d["config"]["last_test_accuracy_mean"])')"; }
EXTRA=""
run "100 clients G=2" P2PFL_GROUPS=2 &&
run "100 clients G=3" P2PFL_GROUPS=3 &&
EXTRA="--nodes-per-gpu 200 --split-nodes 100" &&
run "200 clients G=2" P2PFL_GROUPS=2 &&
run "200 clients G=3" P2PFL_GROUPS=3 &&
run "200 clients G=6" P2PFL_GROUPS=6
