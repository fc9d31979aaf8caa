cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in plain pack both both_res; do
  BN_CASE=$c rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bn_time_$c -- python tools/bn_time.py > gpurun_out/bn_time_$c.log 2>&1 || exit 1
  f=$(ls gpurun_out/bn_time_$c/*/*kernel_stats.csv | head -1)
  echo "== $c"; grep -i "bn_\|elementwise" $f | awk -F, '{print $1, $2, $4}'
done
