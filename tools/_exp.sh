for f in 704; do
NPBNN_WIDE_MIN_WAVES=1 python tools/_exp.py $f mh
NPBNN_WIDE_MIN_WAVES=1 python tools/_exp.py $f run 0
NPBNN_WIDE_MIN_WAVES=1 python tools/_exp.py $f run 1
NPBNN_WIDE_MIN_WAVES=1 python tools/_exp.py $f run 2
NPBNN_FORCE_WIDE=1 python tools/_exp.py $f mh
NPBNN_FORCE_WIDE=1 python tools/_exp.py $f run 0
done
