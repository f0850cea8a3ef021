#!/bin/bash
# Headline frames/s against the number of hardware queues of the process (GPU_MAX_HW_QUEUES) and of contexts.
lib=$1
for q in 4 8 16; do for c in 4 6 8 12 16; do
  [ $c -lt $q ] && [ $c -ne 4 ] && continue
  GPU_MAX_HW_QUEUES=$q ORBX_LIB_PATH="$GRAFT_REPO_ROOT/$lib" timeout -k 10 200 python3 bench.py --steps 60 --warmup 8 --contexts $c --no-cpu-baseline --no-configs --no-hamming 2>/dev/null \
    | python3 -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('queues $q ctx $c', round(d['value']), 'frames/s', round(d['ms_per_step'], 4), 'ms')" || exit 1
done; done
