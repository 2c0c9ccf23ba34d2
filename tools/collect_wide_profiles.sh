#!/bin/bash
# rocprofv3 evidence of the weight-streamed leg (tools/profile_wide.py), on the GPU box from the repository root:
#   gpurun -- 'bash tools/collect_wide_profiles.sh r05'
# kernel-trace statistics, then one --pmc pass per counter group (nothing else traced with them); summaries under gpurun_out/prof_wide_<tag>/keep
tag=${1:-r05}
out=gpurun_out/prof_wide_$tag
mkdir -p $out/keep
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o k -- python3 tools/profile_wide.py --iters 300 > $out/stats.log 2>&1 < /dev/null || echo "kernel stats failed"
f=$(find $out/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $out/keep/${tag}_wide_kernel_stats.csv
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU" "GRBM_GUI_ACTIVE"; do
  n=$(echo $c | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$n -o p -- python3 tools/profile_wide.py --iters 3 > $out/pmc_$n.log 2>&1 < /dev/null || echo "pmc $n failed"
  f=$(find $out/pmc_$n -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $out/keep/${tag}_wide_pmc_$n.csv
done
# the fused pass of the default network on 1024 features (bench leg "default network on 1024 features"), three candidates and one
for cand in 3 1; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/fused_stats_d$cand -o k -- python3 tools/profile_wide.py --config 8 --cand $cand --iters 300 > $out/fused_stats_d$cand.log 2>&1 < /dev/null || echo "fused kernel stats failed"
  f=$(find $out/fused_stats_d$cand -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $out/keep/${tag}_widefused_pass${cand}_kernel_stats.csv
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
    n=$(echo $c | tr ' ' '_')
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/fused_pmc_d${cand}_$n -o p -- python3 tools/profile_wide.py --config 8 --cand $cand --iters 3 > $out/fused_pmc_d${cand}_$n.log 2>&1 < /dev/null || echo "fused pmc $n failed"
    f=$(find $out/fused_pmc_d${cand}_$n -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $out/keep/${tag}_widefused_pass${cand}_pmc_$n.csv
  done
done
python3 tools/read_wide_pmc.py $out/keep $tag > $out/keep/${tag}_wide_layer0_counters.json 2> $out/read.err
for cand in 3 1; do python3 tools/read_wide_pmc.py $out/keep $tag $cand > $out/keep/${tag}_widefused_pass${cand}_counters.json 2>> $out/read.err; done
cat $out/keep/${tag}_wide_layer0_counters.json $out/keep/${tag}_widefused_pass3_counters.json; cat $out/read.err | tail -5; ls $out/keep
