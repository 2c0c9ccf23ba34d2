cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_wide4
mkdir -p $out
for spec in 20000,4096,256-64,10 100000,1024,50-5,10; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$spec -o k -- python3 tools/time_wide.py $spec > $out/$spec.log 2>&1 < /dev/null
  grep -v "^  *last" $out/$spec.log | grep "pass\|chain"
done
