#!/bin/bash
# The rocprofv3 evidence of a round, collected on the GPU box from the repository root:
#   gpurun -- 'bash tools/collect_profiles.sh r02'            (a second argument names the configs: "5", "2 4" ... default all)
# kernel-trace statistics of the default bench run per config, and FETCH_SIZE / WRITE_SIZE of the pass kernel (separate --pmc
# passes, nothing else traced with them).  Everything lands under gpurun_out/prof_<tag>/; copy what is to be kept into profiles/.
tag=${1:-r04}
configs=${2:-"2 4 5 0"}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in $configs; do      # (0: config 2's data under the reference's default network [50,5])
  # one launch per pass (schedule 2): a launch's duration in the trace is the pass kernel's own, comparable with roofline.kernel_ms
  NPBNN_BENCH_SCHEDULE=2 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_c$c -o b -- python3 bench.py --config $c --no-cpu-baseline > $out/bench_c${c}_under_rocprof.json 2> $out/bench_c$c.err || echo "bench profile of config $c failed"
  # the default command (persistent launch: one kernel per run_steps call loops over the passes)
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_c${c}_persistent -o b -- python3 bench.py --config $c --no-cpu-baseline > $out/bench_c${c}_persistent_under_rocprof.json 2> $out/bench_c${c}_persistent.err || echo "persistent bench profile of config $c failed"
  cands="3 1"; [ $c = 0 ] && cands="2 1"      # (the default network: two candidates per pass fit since its image holds 13 rows per tile)
  for cand in $cands; do
    # full passes only (a quarter second of spin-up, then 2000 launches of the pass kernel, nothing idle or void among them): the mean to
    # hold against roofline.kernel_ms
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/pass_c${c}_d${cand} -o k -- python3 tools/profile_eval.py --config $c --cand $cand --iters 2000 --spin 0.25 > $out/pass_c${c}_d${cand}.log 2>&1 || echo "pass stats of config $c D=$cand failed"
    for ctr in FETCH_SIZE WRITE_SIZE; do
      timeout -k 10 200 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/pmc_c${c}_d${cand}_$ctr -o p -- python3 tools/profile_eval.py --config $c --cand $cand --iters 3 > $out/pmc_c${c}_d${cand}_$ctr.log 2>&1 || echo "pmc $ctr of config $c D=$cand failed"
    done
  done
done
find $out -name "*.csv" | head -50
# the summaries under the names profiles/ keeps them by (bench.py reads the newest r*_cfg<c>_pass<d>_kernel_stats.csv / _pmc_*.csv)
keep=$out/keep
mkdir -p $keep
first() { find "$1" -name "$2" 2>/dev/null | head -1; }
for c in $configs; do
  f=$(first $out/bench_c$c "*kernel_stats.csv"); [ -n "$f" ] && cp "$f" $keep/${tag}_bench_cfg${c}_kernel_stats.csv
  f=$(first $out/bench_c${c}_persistent "*kernel_stats.csv"); [ -n "$f" ] && cp "$f" $keep/${tag}_bench_cfg${c}_persistent_kernel_stats.csv
  cp $out/bench_c${c}_under_rocprof.json $keep/${tag}_bench_cfg${c}_under_rocprof.json 2>/dev/null
  cp $out/bench_c${c}_persistent_under_rocprof.json $keep/${tag}_bench_cfg${c}_persistent_under_rocprof.json 2>/dev/null
  for cand in 3 2 1; do
    f=$(first $out/pass_c${c}_d${cand} "*kernel_stats.csv"); [ -n "$f" ] && cp "$f" $keep/${tag}_cfg${c}_pass${cand}_kernel_stats.csv
    for ctr in FETCH_SIZE WRITE_SIZE; do
      f=$(first $out/pmc_c${c}_d${cand}_$ctr "*counter_collection.csv"); [ -n "$f" ] && cp "$f" $keep/${tag}_cfg${c}_pass${cand}_pmc_$ctr.csv
    done
  done
done
ls $keep
