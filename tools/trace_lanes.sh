#!/bin/bash
# rocprofv3 kernel trace of a short bench run -> per-queue (lane) busy / gap summary of the last steps + per-kernel in-step averages.
# tools/trace_lanes.sh <tag> [ENV=..]      (run through gpurun)
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace -d $O/${tag}_trace -o t -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $O/${tag}_trace.log 2>&1
cd $R
db=$(find $O/${tag}_trace -name "*.db" | head -1)
python tools/rocpd_lanes.py $db --steps 2 --gaps 8 > $O/${tag}_lanes.txt 2>&1
python tools/rocpd_stats.py $db --csv $O/${tag}_kstats.csv > /dev/null 2>&1
rm -rf $O/${tag}_trace
tail -30 $O/${tag}_lanes.txt
