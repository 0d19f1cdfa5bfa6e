# FETCH_SIZE of the weight-gradient launches under the two workgroup orders (PLYOLO_WG_XCD)
set -euo pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for x in 0 1; do
  export PLYOLO_WG_XCD=$x
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$O/wgx${x}_f" -o p -- python3 "$R/bench.py" --steps 3 --warmup 2 --no-cpu-baseline > "$O/wgx${x}_f.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$O/wgx${x}_w" -o p -- python3 "$R/bench.py" --steps 3 --warmup 2 --no-cpu-baseline > "$O/wgx${x}_w.log" 2>&1
  cd "$R"
  echo "PLYOLO_WG_XCD=$x"
  python tools/rocpd_pmc.py "$(find "$O/wgx${x}_f" -name '*.db' | head -1)" "$(find "$O/wgx${x}_w" -name '*.db' | head -1)" | grep -E "conv_wgrad|reduce_slabs|conv_mfma |conv_pw"
  rm -rf "$O/wgx${x}_f" "$O/wgx${x}_w"
  cd /tmp
done
