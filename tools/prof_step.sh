# rocprofv3 kernel trace of a short bench run: per-kernel averages of the kernels named in $1 (egrep pattern); run through gpurun
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_step -o s -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_step.log 2>&1
cd $R
python tools/rocpd_stats.py "$(find gpurun_out/prof_step -name "*.db" | head -1)" --csv gpurun_out/prof_step.csv > /dev/null; grep -E "${1:-pack}" gpurun_out/prof_step.csv | awk -F, "{printf \"%-60s calls %5d avg %8.1f us\\n\", substr(\$1,1,60), \$2, \$4/1000}"
rm -rf gpurun_out/prof_step
