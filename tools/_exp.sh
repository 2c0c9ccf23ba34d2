cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 60 rocprofv3 -L 2>/dev/null | grep -i "mall\|dram\|EA0_RDREQ\|TCC_EA\|HBM\|_MISS\b" | head -40 > gpurun_out/counters_list.txt
timeout -k 10 500 python tools/dram_vs_mall.py > gpurun_out/dram_vs_mall.csv 2> gpurun_out/dram_vs_mall.err
cat gpurun_out/dram_vs_mall.csv; tail -3 gpurun_out/dram_vs_mall.err; cat gpurun_out/counters_list.txt | cut -c1-200
