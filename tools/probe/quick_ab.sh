# A/B of one environment variable over bench presets in ONE call (boxes differ by a few per cent):
#   bash tools/probe/quick_ab.sh VAR "preset:valueA:valueB ..."     ("-" = unset)
var=$1; shift
for spec in $@; do
  p=${spec%%:*}; rest=${spec#*:}
  cfg=""; [ "$p" != headline ] && cfg="--config $p"
  for rep in 1 2; do for v in ${rest//:/ }; do
    if [ "$v" = "-" ]; then unset $var; else export $var=$v; fi
    timeout -k 10 200 python bench.py $cfg --steps 10 --warmup 3 --no-reference-run > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; exit 1; }
    echo "$p $var=$v $(python3 -c "import json; d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1))")"
  done; done
done
