# round timeline of the headline under the fp32-mfma policy
set -e
mkdir -p gpurun_out/tl
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tl/mfma -o t -- python3 bench.py --gpus 1 --steps 3 --warmup 1 --dtype fp32-mfma > gpurun_out/tl/mfma.log 2>&1
python3 tools/round_timeline.py gpurun_out/tl/mfma/*kernel_trace.csv > gpurun_out/tl/headline_mfma_timeline.txt
rm -rf gpurun_out/tl/mfma
