# sweep of the blocks-per-client of conv1_wgrad_f32 (P2PFL_C1W_SPLITS forces it; 0 = the built-in rule)
for K in 100 20 10; do for ks in 5 7; do for s in 0 2 4 5 6 8 12 16 25 32; do echo "K=$K ks=$ks splits=$s"; P2PFL_C1W_SPLITS=$s timeout -k 10 60 python tools/bench_conv.py $K $ks 3 | grep conv1_wgrad || exit 1; done; done; done
