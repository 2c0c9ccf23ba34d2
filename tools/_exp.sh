set -e
cd /root/repo
python tools/switch_sweep.py > gpurun_out/switch_sweep_auto.csv 2> gpurun_out/switch_sweep_auto.err
NPBNN_FORCE_WIDE=1 python tools/switch_sweep.py 256 384 512 576 640 672 > gpurun_out/switch_sweep_forced.csv 2> gpurun_out/switch_sweep_forced.err
NPBNN_SWEEP_HIDDEN=32-8 python tools/switch_sweep.py 256 512 768 1024 1280 1536 2048 > gpurun_out/switch_sweep_32_8.csv 2> gpurun_out/switch_sweep_32_8.err
