# per-kernel times of the YOLOX loss launches (rocprofv3 kernel trace); run through gpurun
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_loss -o l -- python3 $R/tools/bench_loss.py 20 > $R/gpurun_out/prof_loss.log 2>&1
cd $R; tail -1 gpurun_out/prof_loss.log
python tools/rocpd_stats.py "$(find gpurun_out/prof_loss -name '*.db' | head -1)" | grep -E "k_prep|k_topk|k_resolve|k_loss|k_final|k_bwd|Name" | head -12
rm -rf gpurun_out/prof_loss
