# one gpurun call that regenerates everything under profiles/r03_* (bench line, rocprofv3 kernel stats of bench and training
# step, PMC traffic per launch shape, GEMM microbenchmark incl. the LDS-DMA variant, small-batch profiles)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03f
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
PMC_NO_SQ=1 PMC_STREAMS=1 SET_GEMM_BN32=1 bash tools/pmc_bench.sh gpurun_out/pmc_bn32 "gemm_nt_f32<128, 32" > /dev/null 2>&1
cp gpurun_out/pmc_bn32/traffic.json $O/pmc_traffic_bn32.json
bash tools/ubench/build_gemm_steps.sh r3 > /dev/null 2>&1
( cd tools/ubench; echo "== register-staged (default) gemm_steps 3000"; ./gemm_steps_r3 3000 | grep -v "^     "; echo "== SET_GEMM_DMA=1 gemm_steps 3000"; SET_GEMM_DMA=1 ./gemm_steps_r3 3000 | grep -v "^     " | sed -n 2,13p; echo "== SET_GEMM_BN32=1 gemm_steps 3000"; SET_GEMM_BN32=1 ./gemm_steps_r3 3000 | grep -v "^     " | sed -n 1,6p ) > $O/gemm_steps.txt 2>&1
( for b in 4 16 32 128; do python tools/profile_small_batch.py $b 2>&1 | grep -v amdgpu.ids; done; echo "== B=4, round-2 paths (SET_GEMM_BM16_UPTO=0 SET_ATT_V2=0 SET_ATT_SMALL_SLICES=0 SET_ENC_PERSISTENT=0)"; SET_GEMM_BM16_UPTO=0 SET_ATT_V2=0 SET_ATT_SMALL_SLICES=0 SET_ENC_PERSISTENT=0 python tools/profile_small_batch.py 4 2>&1 | grep -v amdgpu.ids ) > $O/small_batch.txt 2>&1
rm -rf $O/prof_bench $O/prof_train gpurun_out/pmc_bench/fetch gpurun_out/pmc_bench/write gpurun_out/pmc_bench/sq gpurun_out/pmc_bn32/fetch gpurun_out/pmc_bn32/write
ls -la $O
