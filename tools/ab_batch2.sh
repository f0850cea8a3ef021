#!/bin/bash
# Headline for several (contexts, frames per step) of the default shape; two rounds
for rnd in 1 2; do
for cb in "4 512" "4 1024" "4 2048" "3 768" "4 1536" "2 1024"; do
  set -- $cb
  timeout -k 10 200 python3 bench.py --contexts $1 --batch $2 --steps 60 --warmup 6 --no-cpu-baseline --no-configs --no-hamming --no-extra-legs 2>/dev/null \
    | python3 -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('contexts $1, $2 frames per step:', round(d['value']), 'frames/s')" || exit 1
done
done
