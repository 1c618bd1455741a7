#!/bin/bash
# gpurun_out/<dir> of tools/profile_round.sh -> the tracked summaries under profiles/ (what the judge reads):
#   bash tools/collect_profiles.sh gpurun_out/r05prof r05
set -e
SRC=${1:?profile_round output dir}; TAG=${2:?round tag, e.g. r05}
P=profiles
cp "$SRC/STAMP.json" $P/${TAG}_STAMP.json
cp "$SRC/step/step_kernel_stats.csv" $P/${TAG}_bench_b32_kernel_stats.csv
cp "$SRC/single/step_kernel_stats.csv" $P/${TAG}_bench_b32_single_stream_kernel_stats.csv
cp "$SRC/ko/ko_kernel_stats.csv" $P/${TAG}_kernels_only_kernel_stats.csv
python3 tools/condense_pmc.py "$SRC/pmc_f/f_counter_collection.csv" > $P/${TAG}_pmc_fetch_size.csv
python3 tools/condense_pmc.py "$SRC/pmc_w/w_counter_collection.csv" > $P/${TAG}_pmc_write_size.csv
python3 tools/condense_pmc.py "$SRC/pmc_a/a_counter_collection.csv" > $P/${TAG}_pmc_sq_mfma_busy.csv
python3 tools/condense_pmc.py "$SRC/pmc_sf/f_counter_collection.csv" > $P/${TAG}_pmc_step_fetch_size.csv
python3 tools/condense_pmc.py "$SRC/pmc_sw/w_counter_collection.csv" > $P/${TAG}_pmc_step_write_size.csv
cp "$SRC/pmc_traffic.json" $P/pmc_traffic.json
cp "$SRC/step_traffic.json" $P/${TAG}_step_traffic.json
for m in x3 wgrad c16 dwt rest; do grep -v "amdgpu.ids" "$SRC/microbench_$m.log" > $P/${TAG}_microbench_$m.txt; done
if [ -f "$SRC/bench_b32.json" ]; then cp "$SRC/bench_b32.json" $P/${TAG}_bench_b32.json; fi
ls -la $P/${TAG}_* | awk '{print $5, $9}'
