#!/bin/bash
# same-box, single-stream rocprofv3 kernel statistics of two builds: tools/probe/ab_kernels.sh OUT pkgdirA pkgdirB ("cur" = the tree)
out=$1; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp; cd - > /dev/null
for side in A B; do
  dir=$2; [ $side = B ] && dir=$3
  if [ "$dir" = cur ]; then unset WTPSE_PKG_DIR; else export WTPSE_PKG_DIR="$dir"; fi
  WTPSE_WGRAD_STREAM=0 WTPSE_TEACHER_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/raw$side" -o step -- python3 bench.py --steps 7 --warmup 2 --no-cpu-baseline --no-kernel-roofline > "$out/$side.log" 2>&1 || exit 1
  cp $(find "$out/raw$side" -name "*kernel_stats.csv" | head -1) "$out/${side}_kernel_stats.csv" && rm -rf "$out/raw$side"
done
python3 tools/compare_kernel_stats.py "$out/A_kernel_stats.csv" "$out/B_kernel_stats.csv" 8 | tee "$out/compare.txt"
