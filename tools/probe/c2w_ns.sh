# sweep of the batch splits of conv2_wgrad_split (P2PFL_WGRAD_NS forces them; 0 = the built-in rule)
for K in 50 100 25; do for ks in 3 5; do for s in 0 1 2 3 4 5 6 8; do echo "K=$K ks=$ks ns=$s"; P2PFL_WGRAD_NS=$s timeout -k 10 60 python tools/bench_conv.py $K 5 $ks | grep conv2_wgrad || exit 1; done; done; done
