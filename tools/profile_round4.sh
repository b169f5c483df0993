# one gpurun call that regenerates everything under profiles/r04_* : bench lines, rocprofv3 kernel stats of the bench and of the
# training step, PMC traffic per launch shape, the GEMM step microbenchmark for the default kernel and the round-4
# weights-to-registers variant with SQ counters of both, small-batch profiles, the persistent small-batch decode
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04f
mkdir -p $O
python -m show_edit_tell_amd.build > /dev/null 2>&1
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-secondary > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
rocprofv3 --kernel-trace --stats -d $O/prof_bench -o bench -- python bench.py --steps 40 --warmup 5 --repeat 1 --no-cpu-baseline --no-train --no-secondary > $O/prof_bench.log 2>&1
python tools/rocprof_summary.py $O/prof_bench/bench_results.db > $O/bench_kernel_stats.txt
rocprofv3 --kernel-trace --stats -d $O/prof_train -o train -- python tools/bench_train.py --steps 7 --warmup 0 > $O/prof_train.log 2>&1
python tools/rocprof_summary.py $O/prof_train/train_results.db > $O/train_kernel_stats.txt
PMC_STREAMS=3 bash tools/pmc_bench.sh gpurun_out/pmc_bench > $O/pmc.log 2>&1
cp gpurun_out/pmc_bench/traffic.json $O/pmc_traffic.json; cp gpurun_out/pmc_bench/sq_table.txt $O/sq_table.txt
bash tools/ubench/build_gemm_steps.sh r4 > /dev/null 2>&1
( cd tools/ubench; echo "== hand-written k-loop (default since round 4) gemm_steps 3000"; ./gemm_steps_r4 3000 | grep -v "^     "; echo "== SET_GEMM_ASM=0: compiler-scheduled k-loop (the round-3 kernel) gemm_steps 3000"; SET_GEMM_ASM=0 ./gemm_steps_r4 3000 | grep -v "^     " | sed -n 2,13p; echo "== SET_GEMM_ASM=0 SET_GEMM_WREG=1 gemm_steps 3000"; SET_GEMM_ASM=0 SET_GEMM_WREG=1 ./gemm_steps_r4 3000 | grep -v "^     " | sed -n 2,13p; echo "== parity: hand-written loop vs compiler-scheduled (slab sums, fp64 on the host; bit-identical by construction)"; CHECK_ASM=1 ./gemm_steps_r4 20 | grep "^check"; echo "== parity: weights-to-registers vs compiler-scheduled"; SET_GEMM_ASM=0 CHECK_WREG=1 ./gemm_steps_r4 20 | grep "^check" ) > $O/gemm_steps.txt 2>&1
# SQ counters of the step launches: compiler-scheduled (wreg0), weights-to-registers (wreg1), hand-written loop (asm) — two
# passes each: issue / wait, then LDS / VMEM
for v in 0 1 asm; do
  P=$O/pmc_wreg$v; rm -rf $P; mkdir -p $P
  if [ $v = asm ]; then export SET_GEMM_ASM=1 SET_GEMM_WREG=0; else export SET_GEMM_ASM=0 SET_GEMM_WREG=$v; fi
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $P/p1 -o p1 -- tools/ubench/gemm_steps_r4 200 > $P/p1.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $P/p2 -o p2 -- tools/ubench/gemm_steps_r4 200 > $P/p2.log 2>&1
  for p in p1 p2; do
    d=$(dirname $(find $P/$p -name "${p}_counter_collection.csv" | head -1))
    python tools/pmc_table.py $d $p | grep -B1 -A9 "gemm_nt_f32" > $O/pmc_wreg${v}_$p.txt 2>&1
  done
  rm -rf $P
done
unset SET_GEMM_ASM SET_GEMM_WREG
( for b in 4 16 128; do python tools/profile_small_batch.py $b 2>&1 | grep -v amdgpu.ids; done ) > $O/small_batch.txt 2>&1
bash tools/profile_persistent_decode.sh $O/pdec > /dev/null 2>&1
cp $O/pdec/persistent_decode.txt $O/persistent_decode.txt; cp $O/pdec/persistent_decode_kernel_stats.txt $O/persistent_decode_kernel_stats.txt; rm -rf $O/pdec
rm -rf $O/prof_bench $O/prof_train gpurun_out/pmc_bench/fetch gpurun_out/pmc_bench/write gpurun_out/pmc_bench/sq
ls -la $O
