#!/bin/bash
# Headline frames/s of several builds of the library on the same box, alternating, three rounds.
# usage (GPU box, repository root): bash tools/ab_bench.sh a.so b.so ...
for rnd in 1 2 3; do
  for lib in "$@"; do
    ORBX_LIB_PATH="$GRAFT_REPO_ROOT/$lib" timeout -k 10 120 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-configs --no-hamming 2>/dev/null \
      | python3 -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('$(basename $lib)', round(d['value']), 'frames/s', d['ms_per_step'], 'ms', {k: v['ms'] for k, v in d['stages'].items()})" || exit 1
  done
done
