#!/bin/bash
# Frames/s of the extraction at one shape against the frames per step (four contexts).
# usage: ab_shape.sh W H NFEATURES NLEVELS "batches" lib.so
W=$1; H=$2; NF=$3; NL=$4; batches=$5; lib=$6
for b in $batches; do
  ORBX_LIB_PATH="$GRAFT_REPO_ROOT/$lib" timeout -k 10 300 python3 bench.py --steps 12 --warmup 3 --width $W --height $H --nfeatures $NF --nlevels $NL --batch $b --contexts 4 --no-cpu-baseline --no-configs --no-hamming 2>/dev/null \
    | python3 -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('$W x $H batch $b', round(d['value']), 'frames/s', round(d['ms_per_step'], 3), 'ms')" || exit 1
done
