#!/bin/bash
# same-box A/B of whole-step time between builds of the package: tools/probe/ab_builds.sh OUT ROUNDS name=pkgdir ...
# ("cur" = the tree's own build).  One bench.py per build and round, alternating; prints ms_per_step per build.
out=$1; rounds=$2; shift 2
mkdir -p "$out"
for r in $(seq 1 "$rounds"); do
  for spec in "$@"; do
    name=${spec%%=*}; dir=${spec#*=}
    if [ "$name" = cur ]; then env -u WTPSE_PKG_DIR python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-roofline > "$out/${name}_$r.json" 2> "$out/${name}_$r.err" || exit 1
    else WTPSE_PKG_DIR="$dir" python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-roofline > "$out/${name}_$r.json" 2> "$out/${name}_$r.err" || exit 1; fi
    python3 -c "import json,sys; d=json.loads(open('$out/${name}_$r.json').read().strip().splitlines()[-1]); print('$name', $r, d['ms_per_step'], d['value'])"
  done
done
