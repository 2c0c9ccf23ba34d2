set -e
cd /root/repo
python tools/time_short_calls.py > gpurun_out/short_calls.txt 2>&1
python tools/dispatch_stages.py 300 > gpurun_out/dispatch_stages.txt 2>&1
NPBNN_CHAIN_TIMING=1 python tools/profile_dispatch.py 30 > gpurun_out/chain_timing1.txt 2>&1
NPBNN_CHAIN_TIMING=2 python tools/profile_dispatch.py 30 > gpurun_out/chain_timing2.txt 2>&1
