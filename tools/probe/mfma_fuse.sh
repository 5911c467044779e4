# fp32-mfma policy: fused update + data gradient with the two-group schedule (P2PFL_FUSE_DGRAD=1), with the small-footprint
# convolution variants in it (P2PFL_GROUPS_CORES=1), against the one-group pipeline
run() { timeout -k 10 200 python bench.py $2 --dtype fp32-mfma --steps 10 --warmup 3 > gpurun_out/qm.json 2> gpurun_out/qm.err || { tail -3 gpurun_out/qm.err; exit 1; }
  echo "$1 $2 $(python3 -c "import json; d=json.loads(open('gpurun_out/qm.json').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1))")"; }
for rep in 1 2; do
  P2PFL_FUSE_DGRAD=0 run "pipeline" ""
  P2PFL_FUSE_DGRAD=1 run "groups" ""
  P2PFL_FUSE_DGRAD=1 P2PFL_GROUPS_CORES=1 run "groups+cores" ""
done
P2PFL_FUSE_DGRAD=0 run "pipeline" "--config cnn10-fully"
P2PFL_FUSE_DGRAD=1 run "groups" "--config cnn10-fully"
P2PFL_FUSE_DGRAD=0 run "pipeline" "--config cnn100-encrypted-churn"
P2PFL_FUSE_DGRAD=1 run "groups" "--config cnn100-encrypted-churn"
