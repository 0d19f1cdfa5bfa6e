# rocprofv3 kernel trace of a short YOLOv7 bench run: per-kernel averages of the kernels matching $1; run through gpurun
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_step7 -o s -- python3 $R/bench.py --model yolov7 --steps 4 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_step7.log 2>&1
cd $R
python tools/rocpd_stats.py "$(find gpurun_out/prof_step7 -name "*.db" | head -1)" --csv gpurun_out/prof_step7.csv > /dev/null; grep -E "${1:-v7}" gpurun_out/prof_step7.csv | awk -F, "{printf \"%-64s calls %5d avg %8.1f us\\n\", substr(\$1,1,64), \$2, \$4/1000}"
rm -rf gpurun_out/prof_step7
