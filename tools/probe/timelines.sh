set -e
mkdir -p gpurun_out/r3_tl
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for p in femnist50-ring femnist1000-star; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_tl/$p -o t -- python3 bench.py --config $p --gpus 1 --steps 3 --warmup 1 --no-reference-run > gpurun_out/r3_tl/$p.log 2>&1
  python3 tools/round_timeline.py gpurun_out/r3_tl/$p/*kernel_trace.csv > gpurun_out/r3_tl/${p}_timeline.txt
  rm -rf gpurun_out/r3_tl/$p
done
