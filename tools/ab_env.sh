#!/bin/bash
# Headline frames/s of the current library with and without one environment switch, alternating, three rounds.
# usage (GPU box, repository root): bash tools/ab_env.sh VAR [bench args...]
var=$1; shift
for rnd in 1 2 3; do
  for on in 0 1; do
    if [ $on = 1 ]; then export $var=1; else unset $var; fi
    timeout -k 10 120 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-configs --no-hamming "$@" 2>/dev/null \
      | python3 -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('$var=$on', round(d['value']), 'frames/s', 'one context', round(d['config'].get('one_context_value', 0)), {k: v['ms'] for k, v in d['stages'].items()})" || exit 1
  done
done
