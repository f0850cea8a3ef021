#!/bin/bash
# SQ instruction counters of several builds of the library, one rocprofv3 --pmc pass each.
# usage (GPU box, repository root): bash tools/pmc_ab.sh <outdir> a.so b.so ...
set -u
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  name=$(basename "$lib" .so)
  export ORBX_LIB_PATH="$GRAFT_REPO_ROOT/$lib"
  timeout -s KILL 180 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d "$out/$name" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --contexts 1 --no-configs --no-hamming > "$out/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$out/$name.log"; exit 1; }
  echo "== $name"; python3 tools/pmc_summary.py "$out/$name" | grep -E "kernel|k_fast|k_resize|k_blur|k_describe|k_octree|k_level0"
done
