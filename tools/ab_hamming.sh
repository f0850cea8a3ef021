#!/bin/bash
# Brute-force matcher leg of bench.py for several builds of the library on the same box, alternating, three rounds.
# usage (GPU box, repository root): bash tools/ab_hamming.sh a.so b.so ...
for rnd in 1 2 3; do
  for lib in "$@"; do
    ORBX_LIB_PATH="$GRAFT_REPO_ROOT/$lib" timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs 2>/dev/null \
      | python3 -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); h = d['hamming']; print('$(basename $lib)', '%.3e pairs/s' % h['value'], h['ms_per_step'], 'ms', 'frac', h['roofline']['frac'])" || exit 1
  done
done
