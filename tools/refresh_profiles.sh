#!/bin/bash
# Takes every profile the documentation quotes, on the GPU box, from the repository root:
#   bash tools/refresh_profiles.sh <tag>        (writes gpurun_out/<tag>/; copy what should be kept into profiles/)
# One rocprofv3 run per pass (kernel trace, then each --pmc counter group on its own).  Every summary is stamped with
# the library's build stamp (orbx_build_stamp: a hash of its sources), which bench.py compares with the library it runs.
set -u
tag=$1
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
stamp=$(python3 -c "import importlib; print(importlib.import_module('visual-slam_amd').build_stamp())") || exit 1
echo "build stamp $stamp" | tee $out/STAMP
B="--contexts 1 --batch 256 --no-cpu-baseline --no-configs --no-extra-legs"
kt() { # name, title, command...
  name=$1; title=$2; shift 2
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name -- "$@" > $out/$name.json 2> $out/$name.err || { echo "trace $name failed"; tail -3 $out/$name.err; return 1; }
  ks=$(ls $out/$name/*/*kernel_stats.csv | tail -1); tr=$(ls $out/$name/*/*kernel_trace.csv | tail -1)
  python3 tools/summarize_prof.py $ks $out/${name}_kernel_stats.md "$tag (library build $stamp): $title" > /dev/null || return 1
  python3 tools/trace_by_grid.py $tr > $out/${name}_by_grid.md || return 1
  echo "trace $name ok"
}
pmc() { # name, counters, bench args...
  name=$1; counters=$2; shift 2
  timeout -s KILL 240 rocprofv3 --pmc $counters --output-format csv -d $out/$name -- python3 bench.py --steps 2 --warmup 1 $B "$@" > $out/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $out/$name.log; return 1; }
  echo "pass $name ok"
}
kt kt "bench.py $B --steps 10 (every extraction launch covers 256 frames of 640x480; k_best2_mfma: 4096 queries x 2^24 rows)" python3 bench.py $B --steps 10 || exit 1
kt kt4ctx "bench.py --no-cpu-baseline --no-configs --no-hamming --steps 10 (the default: 1024 frames per step on four contexts, 256 frames per launch)" python3 bench.py --no-cpu-baseline --no-configs --no-hamming --no-extra-legs --steps 10 || exit 1
kt kt_uhd "bench.py $B --no-hamming --width 3840 --height 2160 --nfeatures 8000 --nlevels 12 --batch 64 --steps 4" python3 bench.py --contexts 1 --no-cpu-baseline --no-configs --no-hamming --no-extra-legs --width 3840 --height 2160 --nfeatures 8000 --nlevels 12 --batch 64 --steps 4 --warmup 1 || exit 1
kt kt_stereo "tools/stereo_bench.py (128 pairs of 752x480, 1200 features, left and right contexts, row search + SAD refinement)" python3 tools/stereo_bench.py || exit 1
pmc pmc_rd "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" --no-hamming || exit 1
pmc pmc_wr "TCC_EA0_WRREQ_WRITE_DRAM_32B_sum" --no-hamming || exit 1
pmc pmc_sq "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" --no-hamming || exit 1
pmc pmc_sq2 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_ANY" --no-hamming || exit 1
pmc pmc_ta "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" --no-hamming || exit 1
pmc pmc_mfma "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" || exit 1
for g in "extraction pmc_rd pmc_wr pmc_sq pmc_sq2 pmc_ta" "matcher pmc_mfma"; do
  set -- $g; name=$1; shift
  rm -rf $out/_sel; mkdir -p $out/_sel; for d in "$@"; do cp -r $out/$d $out/_sel/; done
  python3 tools/pmc_summary.py $out/_sel $out/pmc_$name.md > /dev/null
  sed -i "1s/.*/# $tag (library build $stamp): rocprofv3 --pmc over bench.py --steps 2 $B, mean per launch, one run per counter group ($*)/" $out/pmc_$name.md
done
rm -rf $out/_sel
python3 tools/pmc_traffic.py $out $out/traffic.json 7 $stamp
echo "refresh done"
