python3 -m pytest tests/test_hip_sampler.py tests/test_hip_exchange.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2; do
for L in base new; do
  echo "== $L"
  if [ $L = base ]; then export NPBNN_HIP_LIB=$PWD/npbnn_amd/lib/variants/libbase.so; else unset NPBNN_HIP_LIB; fi
  python3 tools/step_stamps.py 2 4 2>&1 | grep "step stamps" | sed -n 5,6p | cut -c1-200
  NPBNN_DEFAULT_PROPOSALS=1 timeout -k 10 200 python tools/time_moving_chain.py 2 4 2>&1 | tail -1
  timeout -k 10 200 python tools/time_moving_chain.py 2 5 2>&1 | tail -1
  NPBNN_DEFAULT_PROPOSALS=1 timeout -k 10 200 python tools/time_moving_chain.py 4 0 2>&1 | tail -1
  NPBNN_DEFAULT_PROPOSALS=1 timeout -k 10 200 python tools/time_moving_chain.py 5 0 2>&1 | tail -1
done
done
