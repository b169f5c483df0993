#!/bin/bash
# HBM traffic + issue counters of the dominant kernel over the bench workload: SEPARATE --pmc passes (FETCH_SIZE,
# WRITE_SIZE, SQ/GRBM) with --kernel-trace only (gpurun refuses --pmc combined with sys/hip/hsa trace domains).
#   pmc_bench.sh [outdir = gpurun_out/pmc_bench] [kernel substring = "gemm_nt_f32<64, 64"]   (+ any SET_* switches in the env)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=${1:-gpurun_out/pmc_bench}
KERNEL=${2:-"gemm_nt_f32<64, 64|gemm_nt_f32_asm<"}     # the 64x64 GEMM, compiler-scheduled and hand-written k-loop
rm -rf $OUT; mkdir -p $OUT
BENCH="python bench.py --steps 3 --warmup 1 --repeat 1 --streams ${PMC_STREAMS:-3} --no-cpu-baseline --no-profile --no-train --no-secondary"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- $BENCH > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- $BENCH > $OUT/write.log 2>&1
if [ -z "$PMC_NO_SQ" ]; then
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/sq -o sq -- $BENCH > $OUT/sq.log 2>&1
python tools/pmc_table.py $OUT/sq sq > $OUT/sq_table.txt 2>&1 || true
fi
python tools/pmc_traffic.py $OUT "$KERNEL" $OUT/traffic.json
