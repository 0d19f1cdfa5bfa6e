#!/bin/bash
# Round evidence on a GPU box (run through gpurun): rocprofv3 kernel trace + the two HBM PMC passes + the MFMA-busy
# pass + the full default bench line + the per-launch plan profile + lane end times.  Outputs: gpurun_out/<tag>_*.
# Every profiling pass is checked: a pass that produced no rocpd database stops the script (no partial evidence).
set -euo pipefail
tag=${1:-fin}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp

db_of() {  # the rocpd database a pass left under $1, or fail
  local f
  f=$(find "$1" -name "*.db" | head -1)
  if [ -z "$f" ]; then echo "collect_evidence: no rocpd database under $1 (see $1.log)" >&2; exit 1; fi
  echo "$f"
}

# the program itself follows `--` (no env / bash -c hop: the profiler's preloaded library has initialised the GPU)
rocprofv3 --kernel-trace -d "$O/${tag}_trace" -o t -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$O/${tag}_trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$O/${tag}_pmc_fetch" -o p -- python3 "$R/bench.py" --steps 3 --warmup 2 --no-cpu-baseline > "$O/${tag}_pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$O/${tag}_pmc_write" -o p -- python3 "$R/bench.py" --steps 3 --warmup 2 --no-cpu-baseline > "$O/${tag}_pmc_write.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d "$O/${tag}_pmc_mfma" -o p -- python3 "$R/bench.py" --steps 3 --warmup 2 --no-cpu-baseline > "$O/${tag}_pmc_mfma.log" 2>&1
cd "$R"
python tools/rocpd_mfma_busy.py "$(db_of "$O/${tag}_pmc_mfma")" "$O/${tag}_mfma_busy.json" | head -4
python tools/rocpd_stats.py "$(db_of "$O/${tag}_trace")" --csv "$O/${tag}_kstats.csv" | tail -3
python tools/rocpd_pmc.py "$(db_of "$O/${tag}_pmc_fetch")" "$(db_of "$O/${tag}_pmc_write")" "$O/${tag}_pmc.json" | tail -3
# stamp the summary with the identity of the library it was measured on (bench.py prints the same lib_md5)
python - "$O/${tag}_pmc.json" "$tag" <<'PY'
import hashlib, json, sys
p, tag = sys.argv[1], sys.argv[2]
d = json.load(open(p))
d["build"] = {"tag": tag, "lib_md5": hashlib.md5(open("pl_yolo_amd/libplyolo_hip.so", "rb").read()).hexdigest()}
json.dump(d, open(p, "w"), indent=1)
PY
rm -rf "$O/${tag}_trace" "$O/${tag}_pmc_fetch" "$O/${tag}_pmc_write" "$O/${tag}_pmc_mfma"
# the bench line below reads its `roofline.traffic` from profiles/ and refuses a summary of another build: hand it this one
cp "$O/${tag}_pmc.json" "$R/profiles/${tag}_pmc_hbm_traffic.json"
python bench.py --profile-out "$O/${tag}_plan_profile.json" > "$O/${tag}_bench.json" 2> "$O/${tag}_bench.err"
tail -1 "$O/${tag}_bench.json" | cut -c1-400
python tools/lane_times.py 2>/dev/null | tail -2 | tee "$O/${tag}_lane_times.txt"
