# per-kernel times of the batched NMS launches (rocprofv3 kernel trace); run through gpurun
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_nms -o n -- python3 $R/tools/bench_nms.py > $R/gpurun_out/prof_nms.log 2>&1
python tools/rocpd_stats.py "$(find gpurun_out/prof_nms -name '*.db' | head -1)" --csv gpurun_out/prof_nms.csv > /dev/null
grep -E "k_|nms|Nms" gpurun_out/prof_nms.csv | awk -F, '{printf "%-60s calls %5d avg %8.1f us\n", substr($1,1,60), $2, $4/1000}'
rm -rf gpurun_out/prof_nms
