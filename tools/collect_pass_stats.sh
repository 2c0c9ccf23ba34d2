#!/bin/bash
# Only the pass-kernel statistics of tools/collect_profiles.sh (rocprofv3 --kernel-trace --stats over tools/profile_eval.py with a
# spin-up), for a quick look at what a box gives:   gpurun -- 'bash tools/collect_pass_stats.sh r04b'
tag=${1:-r04b}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in 2 4 5 0; do
  cands="3 1"; [ $c = 0 ] && cands="1"
  for cand in $cands; do
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/pass_c${c}_d${cand} -o k -- python3 tools/profile_eval.py --config $c --cand $cand --iters 2000 --spin 0.25 > $out/pass_c${c}_d${cand}.log 2>&1 || echo "pass stats of config $c D=$cand failed"
    f=$(find $out/pass_c${c}_d${cand} -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && grep -h "eval_kernel" "$f" | cut -c1-150
  done
done
