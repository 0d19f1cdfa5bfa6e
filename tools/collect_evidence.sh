#!/bin/bash
# Round evidence on a GPU box (run through gpurun): rocprofv3 kernel trace + the two HBM PMC passes + the full
# default bench line + the per-launch plan profile + lane end times.  Outputs under gpurun_out/<tag>_*.
tag=${1:-fin}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/${tag}_trace -o t -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/${tag}_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/${tag}_pmc_fetch -o p -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/${tag}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/${tag}_pmc_write -o p -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/${tag}_pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/${tag}_pmc_mfma -o p -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/${tag}_pmc_mfma.log 2>&1
cd $R
python tools/rocpd_mfma_busy.py $(find gpurun_out/${tag}_pmc_mfma -name "*.db" | head -1) gpurun_out/${tag}_mfma_busy.json | head -4
rm -rf gpurun_out/${tag}_pmc_mfma
python tools/rocpd_stats.py $(find gpurun_out/${tag}_trace -name "*.db" | head -1) --csv gpurun_out/${tag}_kstats.csv | tail -3
python tools/rocpd_pmc.py $(find gpurun_out/${tag}_pmc_fetch -name "*.db" | head -1) $(find gpurun_out/${tag}_pmc_write -name "*.db" | head -1) gpurun_out/${tag}_pmc.json | tail -3
rm -rf gpurun_out/${tag}_trace gpurun_out/${tag}_pmc_fetch gpurun_out/${tag}_pmc_write
python bench.py --profile-out gpurun_out/${tag}_plan_profile.json > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -1 gpurun_out/${tag}_bench.json | cut -c1-400
python tools/lane_times.py 2>/dev/null | tail -2 | tee gpurun_out/${tag}_lane_times.txt
