#!/bin/bash
# 3840x2160 / 8000 features / 12 levels: frames/s for several (contexts, frames per launch); two rounds
for rnd in 1 2; do
for cb in "1 64" "2 64" "4 32" "4 64" "2 32" "3 64"; do
  set -- $cb
  timeout -k 10 200 python3 bench.py --contexts $1 --batch $(( $1 * $2 )) --width 3840 --height 2160 --nfeatures 8000 --nlevels 12 --steps 6 --warmup 2 --no-cpu-baseline --no-configs --no-hamming --no-extra-legs 2>/dev/null \
    | python3 -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('contexts $1 x $2 frames:', round(d['value']), 'frames/s')" || exit 1
done
done
