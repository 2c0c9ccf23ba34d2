#!/bin/bash
# Same-box A/B of two builds of the library: bash tools/ab_libs.sh <libA.so> <libB.so> [config]  (run on the GPU box from the repo root)
# Pass-kernel time (npbnn_time_pass), a chain that barely moves (default proposals) and one that moves, per library, twice over.
A=$1; B=$2; C=${3:-2}
for rep in 1 2; do
  for L in $A $B; do
    echo "== $L"
    NPBNN_HIP_LIB=$L timeout -k 10 200 python tools/time_rows_sweep.py 100000 2>&1 | tail -1
    NPBNN_HIP_LIB=$L NPBNN_DEFAULT_PROPOSALS=1 timeout -k 10 200 python tools/time_moving_chain.py $C 4 2>&1 | tail -1
    NPBNN_HIP_LIB=$L timeout -k 10 200 python tools/time_moving_chain.py $C 5 2>&1 | tail -1
  done
done
