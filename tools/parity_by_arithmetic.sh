#!/bin/bash
# The parity tests that are sensitive to WHICH fp32-accurate arithmetic the MFMA-bound layers run in, under each of them:
#   x2h (two fp16 terms, default) | x3 (three bf16 terms) | fp32-input MFMA (WTPSE_X3=0, round 1's arithmetic: exact fp32 products)
# -> gpurun_out/parity_arith/{x2h,x3,fp32}.log    (run on the GPU box from the repo root)
OUT=gpurun_out/parity_arith
mkdir -p $OUT
T="tests/test_parity_gpu.py tests/test_conv_x3_gpu.py tests/test_determinism_gpu.py tests/test_dp_gpu.py"
WTPSE_X3_TERMS=2 timeout -k 10 900 python -m pytest $T -m gpu -q -rA --timeout 600 > $OUT/x2h.log 2>&1; echo "x2h rc $?"
WTPSE_X3_TERMS=3 timeout -k 10 900 python -m pytest $T -m gpu -q -rA --timeout 600 > $OUT/x3.log 2>&1; echo "x3 rc $?"
WTPSE_X3=0 timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -rA --timeout 600 > $OUT/fp32.log 2>&1; echo "fp32 rc $?"
for f in x2h x3 fp32; do echo "== $f"; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/$f.log | cut -c1-200; done
