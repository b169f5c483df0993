#!/bin/bash
# HBM traffic of the dominant kernel over the bench workload: separate --pmc passes (FETCH_SIZE, WRITE_SIZE)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_bench
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-experimental > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-experimental > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/sq -o sq -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-experimental > $OUT/sq.log 2>&1
ls $OUT/*
