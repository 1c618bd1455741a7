#!/bin/bash
run() { echo -n "$1: "; env $1 timeout -k 10 200 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-kernel-roofline 2>&1 | grep -m1 "timed region:" | sed 's/.*timed region: //'; }
run "WTPSE_DUMMY=0"
run "WTPSE_TAIL_MAX_WGS=1024"
run "WTPSE_TAIL_MAX_WGS=4096"
run "WTPSE_WGRAD_R_WAVES=768"
run "WTPSE_WGRAD_R_WAVES=1536"
run "WTPSE_X3_HALF_MIN=128"
run "WTPSE_X3_HALF_MIN=512"
run "WTPSE_X3R=2"
run "WTPSE_X3R=0"
run "WTPSE_DUMMY=1"
