#!/bin/bash
# usage: kstats.sh <tag> <python script + args...>   -> prints the top kernel stats
tag=$1; shift
(cd /tmp && export TMPDIR=/tmp && timeout 280 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- python3 "$@" > /tmp/prof_$tag.log 2>&1)
tail -4 /tmp/prof_$tag.log
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/prof_$tag/**/${tag}_kernel_stats.csv",recursive=True)
for r in list(csv.DictReader(open(f[0])))[:14]: print(r["Name"][:100], r["Calls"], r["AverageNs"])
PY
