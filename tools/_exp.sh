export NPBNN_CHAIN=0
for cfg in 0 5; do for sl in 1 2 3 4; do echo "== cfg $cfg slices $sl"; NPBNN_WIDE_NO_TAIL=1 NPBNN_WIDE_CFG=$cfg NPBNN_WIDE_SLICES=$sl timeout -k 10 100 python tools/time_wide.py 20000,4096,256-64,10 2>&1 | grep pass; done; done
