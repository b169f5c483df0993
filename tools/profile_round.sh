cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02f
python bench.py > gpurun_out/r02f/bench.json 2> gpurun_out/r02f/bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/r02f/prof_bench -o bench -- python bench.py --steps 40 --warmup 5 --repeat 1 --no-cpu-baseline --no-train --no-secondary > gpurun_out/r02f/prof_bench.log 2>&1
python tools/rocprof_summary.py gpurun_out/r02f/prof_bench/bench_results.db > gpurun_out/r02f/bench_kernel_stats.txt
rocprofv3 --kernel-trace --stats -d gpurun_out/r02f/prof_train -o train -- python tools/bench_train.py --steps 7 --warmup 0 > gpurun_out/r02f/prof_train.log 2>&1
python tools/rocprof_summary.py gpurun_out/r02f/prof_train/train_results.db > gpurun_out/r02f/train_kernel_stats.txt
bash tools/pmc_bench.sh > gpurun_out/r02f/pmc.log 2>&1
cp gpurun_out/pmc_bench/traffic.json gpurun_out/r02f/pmc_traffic.json; cp gpurun_out/pmc_bench/sq_table.txt gpurun_out/r02f/sq_table.txt
( cd tools/ubench; echo "== gemm_steps_base 3000"; ./gemm_steps_base 3000; echo "== gemm_steps_stamps 300"; ./gemm_steps_stamps 300 | head -7; echo "== gemm_steps_noepi 2000 (diagnostic: epilogue stores skipped)"; ./gemm_steps_noepi 2000 | sed -n 2,13p; echo "== gemm_steps_noload 2000 (diagnostic: no global loads after the first three k-tiles)"; ./gemm_steps_noload 2000 | sed -n 7,13p; echo "== gemm_steps_samew 2000 (diagnostic: all workgroups stream the same weight rows)"; ./gemm_steps_samew 2000 | sed -n 7,13p ) > gpurun_out/r02f/gemm_steps.txt 2>&1
rm -rf gpurun_out/r02f/prof_bench gpurun_out/r02f/prof_train gpurun_out/pmc_bench/fetch gpurun_out/pmc_bench/write gpurun_out/pmc_bench/sq
ls -la gpurun_out/r02f
