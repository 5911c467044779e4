# client groups under the other precision policies (bench.py, 8 timed rounds each)
run() { name=$1; shift; out=$(env "$@" timeout -k 10 300 python bench.py --gpus 1 --steps 8 --warmup 2 --no-reference-run $EXTRA 2>/dev/null) || { echo "$name: FAILED"; return 1; }; echo "$name: $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],1), "ms", d["config"]["last_test_accuracy_mean"])')"; }
EXTRA="--dtype fp32-mfma"
run "fp32-mfma G=1 (default)" A=1 &&
run "fp32-mfma G=2" P2PFL_GROUPS=2 &&
run "fp32-mfma G=2 + fused dgrad" P2PFL_GROUPS=2 P2PFL_FUSE_DGRAD=1 &&
EXTRA="--dtype bf16" &&
run "bf16 G=1 (default)" A=1 &&
run "bf16 G=2" P2PFL_GROUPS=2
