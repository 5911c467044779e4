# 64-channel blocks of the exact-split stage-2 weight gradient against the two-row-tile form, batch splits swept
for cfg in "50 5 5" "50 7 3" "1000 7 3" "100 5 5" "10 5 5" "25 5 5"; do for w in 1 0; do for s in 0 1 2 4 5 8; do echo "cfg=$cfg wide=$w ns=$s"; P2PFL_WGRAD_WIDE=$w P2PFL_WGRAD_NS=$s timeout -k 10 60 python tools/bench_conv.py $cfg | grep conv2_wgrad || exit 1; done; done; done
