# samples per block of the dense stage-2 forward (P2PFL_CONV2_IPB), stand-alone
for cfg in "1000 7 3" "50 7 3" "50 5 5" "100 5 5" "10 5 5"; do for i in 1 2 4 8 16; do echo "cfg=$cfg ipb=$i"; P2PFL_CONV2_IPB=$i timeout -k 10 60 python tools/bench_conv.py $cfg | grep conv2_fwd || exit 1; done; done
