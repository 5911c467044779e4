export P2PFL_HEAD_MIN_K=1000
bash tools/probe/quick_ab.sh P2PFL_HEAD_RC headline:4:8:16 cnn10-fully:4:8:16
