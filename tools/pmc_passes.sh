#!/bin/bash
# Collect rocprofv3 PMC counters for one short bench run, one pass per counter group (the profiler does
# not split counters over passes).  Run on the GPU box from the repository root:
#   bash tools/pmc_passes.sh <outdir> [bench args...]
set -u
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
args="--steps 2 --warmup 1 --no-cpu-baseline --contexts 1 $*"
run() { # name counters...
  name=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python3 bench.py $args > "$out/$name.log" 2>&1 || { echo "pass $name failed"; return 1; }
  echo "pass $name ok"
}
run fetch FETCH_SIZE && run write WRITE_SIZE && \
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES && \
run sq2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
