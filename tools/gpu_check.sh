#!/bin/bash
# Round-trip check on the GPU box: GPU test suite, then one bench run; logs under gpurun_out/<tag>/.
# usage: bash tools/gpu_check.sh <tag> [extra pytest files...]
set -u
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 900 python -u -m pytest tests "$@" -m gpu -x -q --timeout 300 --timeout-method thread > $out/pytest.log 2>&1
rc=$?; tail -5 $out/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python bench.py --steps 40 --warmup 5 > $out/bench.json 2> $out/bench.err
rc=$?; tail -c 3000 $out/bench.json; tail -3 $out/bench.err
exit $rc
