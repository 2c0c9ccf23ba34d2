timeout -k 10 900 python -m pytest tests/test_hip_wide.py -q -m gpu -x 2>&1 | tail -4
for c in 3 2 1; do echo "== max cand $c"; NPBNN_WIDE_MAX_CAND=$c $( [ $c = 1 ] && echo "env NPBNN_WIDE_ONE_CAND=1" ) timeout -k 10 300 python tools/time_wide.py 100000,1024,50-5,10 100000,704,50-5,10 400000,1024,32-8,10 2>&1 | grep "chain" | cut -c1-140; done
