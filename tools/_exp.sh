timeout -k 10 600 python -m pytest tests/test_hip_wide.py -q -m gpu --maxfail=8 2>&1 | tail -4
NPBNN_CHAIN=0 timeout -k 10 300 python tools/time_wide.py 100000,1024,50-5,10 100000,704,50-5,10 2>&1 | grep pass | cut -c1-100
