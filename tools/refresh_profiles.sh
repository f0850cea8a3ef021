set -u
mkdir -p gpurun_out/r01e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01e/kt -- python3 bench.py --contexts 1 --steps 10 --no-cpu-baseline > gpurun_out/r01e/bench_under_rocprof.json 2> gpurun_out/r01e/kt.err || exit 1
ks=$(ls gpurun_out/r01e/kt/*/*kernel_stats.csv | tail -1); tr=$(ls gpurun_out/r01e/kt/*/*kernel_trace.csv | tail -1)
python3 tools/summarize_prof.py $ks gpurun_out/r01e/kernel_stats.md "r01e kernel stats: bench.py --no-cpu-baseline --contexts 1 --steps 10 (every extraction launch covers 256 frames)" || exit 1
python3 tools/trace_by_grid.py $tr > gpurun_out/r01e/by_grid.md || exit 1
bash tools/pmc_custom.sh gpurun_out/r01e/pmc_rd "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" --no-hamming || exit 1
bash tools/pmc_custom.sh gpurun_out/r01e/pmc_wr "TCC_EA0_WRREQ_WRITE_DRAM_32B_sum" --no-hamming || exit 1
bash tools/pmc_custom.sh gpurun_out/r01e/pmc_sq "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" --no-hamming || exit 1
python3 tools/pmc_summary.py gpurun_out/r01e gpurun_out/r01e/pmc.md
python3 tools/pmc_traffic.py gpurun_out/r01e gpurun_out/r01e/traffic.json
