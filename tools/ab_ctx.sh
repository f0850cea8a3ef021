#!/bin/bash
# Headline frames/s against the number of extractor contexts, for several builds. usage: ab_ctx.sh "1 2 4 8" a.so b.so
ctxs=$1; shift
for c in $ctxs; do
  for lib in "$@"; do
    ORBX_LIB_PATH="$GRAFT_REPO_ROOT/$lib" timeout -k 10 120 python3 bench.py --steps 100 --warmup 10 --contexts $c --no-cpu-baseline --no-configs --no-hamming 2>/dev/null \
      | python3 -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('ctx $c $(basename $lib)', round(d['value']), 'frames/s', round(d['ms_per_step'], 4), 'ms')" || exit 1
  done
done
