#!/bin/bash
# per-K kernel statistics of the shared-A batched CGNR (run on the GPU box): tools/prof_batched_k.sh 8 16
# EXTRA="gram=1 resident=0" tools/prof_batched_k.sh 8   passes bench_batched.py's switches; the stats csv is kept in gpurun_out/
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
for k in "$@"; do
  rm -rf /tmp/pb_$k
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb_$k -o b -- python3 "$GRAFT_REPO_ROOT/tools/bench_batched.py" $k $EXTRA > /tmp/pb_$k.log 2>&1)
  cp /tmp/pb_$k/*kernel_stats.csv "gpurun_out/batched_k${k}_$(echo $EXTRA | tr ' =' '__')_kernel_stats.csv" 2>/dev/null || cp $(find /tmp/pb_$k -name '*kernel_stats.csv' | head -1) "gpurun_out/batched_k${k}_$(echo $EXTRA | tr ' =' '__')_kernel_stats.csv"
  grep -i "us\|it/s" /tmp/pb_$k.log | tail -3
  python3 - <<PY
import csv,glob
f=glob.glob('/tmp/pb_$k/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r['Percentage'])>1: print('K=$k', r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,2),'us')
PY
done
