#!/usr/bin/env python3
"""What stands between two operations of the chain's stream: kernels and copies of a `rocprofv3 --kernel-trace --memory-copy-trace
--output-format csv -d DIR -- python3 tools/time_moving_chain.py 2 4` run in time order, each with its duration and the gap before it.
   python3 tools/trace_gaps.py DIR        (NOTES.md, "running one batch ahead of the GPU": ~93 us of a 960-us dispatch are not the pass kernel)"""
import csv, glob, sys
d = sys.argv[1]
rows = []
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K:' + r['Kernel_Name'][:40]))
for f in glob.glob(d + '/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'C:' + r.get('Direction', r.get('Kind', '?'))))
rows.sort()
print(len(rows), 'ops')
# last 60 ops before the long call: find the region of 100-iteration batches: print a window from the middle
n = len(rows)
lo = max(0, n // 2 - 30)
prev_end = None
for s, e, name in rows[lo:lo + 60]:
    gap = (s - prev_end) / 1e3 if prev_end else 0
    print('%-44s dur %8.1f us  gap before %7.1f us' % (name, (e - s) / 1e3, gap))
    prev_end = e
