export NOSTAMPS=1
for d in "" "-DEXP_NODATA" "-DEXP_NODATA -DEXP_NOBAR"; do PROBE_DEFS="$d" python tools/probe/x3_stamps.py 64 64 128 32 2>&1 | grep -E "unstamped"; done
WTPSE_X3_PP=0 python tools/probe/x3_stamps.py 64 64 128 32 2>&1 | grep -E "unstamped"
