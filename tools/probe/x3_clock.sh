# clock and throughput of the x3 forward kernel on random and on all-zero operands (same binary, same launches)
for z in 0 1; do for shape in "64 64 128 32" "128 128 64 32"; do ZEROS=$z python tools/probe/x3_stamps.py $shape 2>&1 | grep -E "operands|in-kernel clock|matrix pipe"; done; done
