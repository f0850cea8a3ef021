#!/bin/bash
# Headline frames/s against batch size and context count. usage: ab_batch.sh "256 512 1024" "4" lib.so
batches=$1; ctxs=$2; lib=$3
for b in $batches; do for c in $ctxs; do
  ORBX_LIB_PATH="$GRAFT_REPO_ROOT/$lib" timeout -k 10 200 python3 bench.py --steps 40 --warmup 5 --batch $b --contexts $c --no-cpu-baseline --no-configs --no-hamming 2>/dev/null \
    | python3 -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('batch $b ctx $c', round(d['value']), 'frames/s', round(d['ms_per_step'], 4), 'ms')" || exit 1
done; done
