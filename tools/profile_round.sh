#!/bin/bash
# One gpurun call's worth of profiling for a round (run from the repo root on the GPU box):
#   WTPSE_GIT_HEAD=$(git rev-parse --short HEAD) gpurun -- bash tools/profile_round.sh gpurun_out/r04prof      (the box has no .git)
# in-step (three streams, and back to back on one stream: the profile to read kernel costs from) and kernels-only rocprofv3 kernel
# statistics, HBM-traffic and MFMA-busy PMC passes (each in its own run, as the MI355X guide prescribes), the per-layer
# microbenchmarks.  Every run is stamped ($OUT/STAMP.json: git revision + source hash of the library the numbers belong to; the
# traffic file carries the same stamp and bench.py refuses it on any other library).  Copy what should be judged into profiles/.
set -o pipefail
OUT=${1:-gpurun_out/prof}
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 - "$OUT" <<'PY'
import json, os, sys, time
sys.path[:0] = [os.path.join(os.getcwd(), "wt-pse-code_amd")]
from wtpse_hip import build
json.dump({"git_head": os.environ.get("WTPSE_GIT_HEAD", "unknown"), "source_hash": build.source_hash(),
           "library_hash": build.built_hash(), "utc": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime())}, open(os.path.join(sys.argv[1], "STAMP.json"), "w"), indent=1)
PY
B="python3 bench.py"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/step" -o step -- $B --steps 7 --warmup 2 --no-cpu-baseline --no-kernel-roofline > "$OUT/step.log" 2>&1 &&
WTPSE_WGRAD_STREAM=0 WTPSE_TEACHER_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/single" -o step -- $B --steps 7 --warmup 2 --no-cpu-baseline --no-kernel-roofline > "$OUT/single.log" 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ko" -o ko -- $B --kernels-only > "$OUT/ko.log" 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_f" -o f -- $B --kernels-only > "$OUT/pmc_f.log" 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_w" -o w -- $B --kernels-only > "$OUT/pmc_w.log" 2>&1 &&
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_a" -o a -- $B --kernels-only > "$OUT/pmc_a.log" 2>&1 &&
python3 tools/pmc_traffic.py "$(ls $OUT/pmc_f/*/f_counter_collection.csv $OUT/pmc_f/f_counter_collection.csv 2>/dev/null | head -1)" "$(ls $OUT/pmc_w/*/w_counter_collection.csv $OUT/pmc_w/w_counter_collection.csv 2>/dev/null | head -1)" > "$OUT/pmc_traffic.log" 2>&1 &&
cp profiles/pmc_traffic.json "$OUT/pmc_traffic.json" &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_sf" -o f -- $B --steps 2 --warmup 1 --roi-presteps 0 --no-cpu-baseline --no-kernel-roofline > "$OUT/pmc_sf.log" 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_sw" -o w -- $B --steps 2 --warmup 1 --roi-presteps 0 --no-cpu-baseline --no-kernel-roofline > "$OUT/pmc_sw.log" 2>&1 &&
python3 tools/pmc_step_traffic.py "$(ls $OUT/pmc_sf/*/f_counter_collection.csv $OUT/pmc_sf/f_counter_collection.csv 2>/dev/null | head -1)" "$(ls $OUT/pmc_sw/*/w_counter_collection.csv $OUT/pmc_sw/w_counter_collection.csv 2>/dev/null | head -1)" 5 > "$OUT/step_traffic.json" 2> "$OUT/step_traffic.err" &&
python3 tools/microbench_x3.py > "$OUT/microbench_x3.log" 2>&1 &&
python3 tools/microbench_wgrad.py 32 > "$OUT/microbench_wgrad.log" 2>&1 &&
python3 tools/microbench_c16.py 32 > "$OUT/microbench_c16.log" 2>&1 &&
python3 tools/bench_dwt.py > "$OUT/microbench_dwt.log" 2>&1 &&
python3 tools/microbench.py --only wt > "$OUT/microbench_rest.log" 2>&1 &&
python3 tools/microbench.py --only head >> "$OUT/microbench_rest.log" 2>&1 &&
python3 tools/microbench.py --only bn >> "$OUT/microbench_rest.log" 2>&1 &&
python3 tools/microbench.py --only pw >> "$OUT/microbench_rest.log" 2>&1
echo "profile_round rc $?"
