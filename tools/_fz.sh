set -e
cd /root/repo
for dbg in 0 2; do RT2D_FUSE_DBG=$dbg timeout -k 10 120 python3 tools/_stream_k.py; done
for b in 512; do RT2D_GH_BLOCK=$b timeout -k 10 120 python3 tools/_stream_k.py; done
