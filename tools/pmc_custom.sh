#!/bin/bash
# One rocprofv3 --pmc pass with a caller-given counter list.  usage: pmc_custom.sh <outdir> "<counters>" [bench args]
set -u
out=$1; counters=$2; shift 2
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 180 rocprofv3 --pmc $counters --output-format csv -d "$out/p" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --contexts 1 $* > "$out/p.log" 2>&1 && echo ok || echo failed
