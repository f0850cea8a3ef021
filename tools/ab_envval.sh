#!/bin/bash
# Headline frames/s of the current library for several values of one environment variable ("-" = unset), two rounds.
# usage (GPU box, repository root): bash tools/ab_envval.sh VAR v1 v2 ... [-- bench args]
var=$1; shift
vals=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done; [ "$1" = "--" ] && shift
for rnd in 1 2; do
  for v in "${vals[@]}"; do
    if [ "$v" = "-" ]; then unset $var; else export $var=$v; fi
    timeout -k 10 120 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-configs --no-hamming "$@" 2>/dev/null \
      | python3 -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('$var=$v', round(d['value']), 'frames/s', 'one context', round(d['config'].get('one_context_value', 0)), {k: v['ms'] for k, v in d['stages'].items()})" || exit 1
  done
done
