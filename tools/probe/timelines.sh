# per-kernel round timelines (tools/round_timeline.py) of bench presets:  bash tools/probe/timelines.sh <tag> <preset>...
# (environment variables of the caller reach bench.py: A/B runs of one switch get two tags)
set -e
tag=$1; shift
mkdir -p gpurun_out/tl
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for p in "$@"; do
  cfg=""; [ "$p" != headline ] && cfg="--config $p"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tl/$p.$tag -o t -- python3 bench.py $cfg --gpus 1 --steps 3 --warmup 1 --no-reference-run > gpurun_out/tl/$p.$tag.log 2>&1
  python3 tools/round_timeline.py gpurun_out/tl/$p.$tag/*kernel_trace.csv > gpurun_out/tl/${p}_${tag}_timeline.txt
  rm -rf gpurun_out/tl/$p.$tag
done
