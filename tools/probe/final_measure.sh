# The measurements the result tables quote, in one call on one box (GPU box, repo root):
#   driver-style headline bench, kernel stats + round timeline of the headline, every preset in every precision mode
set -e
mkdir -p gpurun_out/final
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
tail -c 400 gpurun_out/final/bench.json | head -c 200; echo
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/prof -o t -- python3 bench.py --gpus 1 --steps 5 --warmup 2 --no-reference-run > gpurun_out/final/prof.log 2>&1
python3 tools/round_timeline.py gpurun_out/final/prof/*kernel_trace.csv > gpurun_out/final/round_timeline.txt
cp gpurun_out/final/prof/*kernel_stats.csv gpurun_out/final/kernel_stats.csv
rm -rf gpurun_out/final/prof
bash tools/bench_sweep.sh gpurun_out/final/sweep
