#!/bin/bash
# Counter passes over the headline step (fp32 by default): one rocprofv3 --pmc run per pass (the profiler does not
# split counters over passes), every run under its own time limit, the program directly after "--". A last pass
# takes a kernel trace of the same in-order eager step so the summary can turn bytes into GB/s.
#   usage (GPU box, repo root): bash tools/pmc_run.sh <out dir under gpurun_out> [bench flags ...]
set -o pipefail
out=${1:-gpurun_out/pmc}; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export P2PFL_INORDER=${PMC_INORDER-1}      # PMC_INORDER= (empty): the shipped schedule (client groups: 50-client launches)
run() {  # name, counters...
  name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-graph $EXTRA > "$out/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$out/$name.log"; return 1; }
  echo "pass $name ok"
}
EXTRA="$*"
run rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum GRBM_GUI_ACTIVE &&
run wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum &&
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_ANY SQ_INSTS_LDS &&
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY &&
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o t -- python3 bench.py --steps 2 --warmup 1 --no-graph $EXTRA > "$out/trace.log" 2>&1 &&
python3 tools/pmc_summary.py "$out/summary.csv" "$out"/trace/*kernel_stats.csv "$out"/*/*counter_collection.csv > /dev/null &&
rm -rf "$out/rd" "$out/wr" "$out/sq" "$out/lds" "$out/trace" && echo "summary in $out/summary.csv"
