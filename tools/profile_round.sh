#!/bin/bash
# ONE gpurun call that regenerates a round's evidence under gpurun_out/<tag>f/ (copy what is to be judged into
# profiles/<tag>_*): tools/profile_round.sh r06
#   bench line (default + the driver's command), rocprofv3 kernel stats of the bench / the XE training step / the SCST step,
#   SCST host-device split, PMC traffic of the bench's launch shapes (separate --pmc passes: tools/pmc_bench.sh), the large
#   products of the training step, per-kernel breakdown at B = 4 .. 128 with the persistent kernel's phase stamps.
# (Rounds 2-5 each had their own copy of this script; they differed in the tag and in one-off lines.)
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG}f
mkdir -p $O
python -m show_edit_tell_amd.build > /dev/null 2>&1
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
rocprofv3 --kernel-trace --stats -d $O/prof_bench -o bench -- python bench.py --steps 40 --warmup 5 --repeat 1 --no-cpu-baseline --no-train --no-secondary > $O/prof_bench.log 2>&1
python tools/rocprof_summary.py $O/prof_bench/bench_results.db > $O/bench_kernel_stats.txt
rocprofv3 --kernel-trace --stats -d $O/prof_train -o train -- python tools/bench_train.py --steps 7 --warmup 0 > $O/prof_train.log 2>&1
python tools/rocprof_summary.py $O/prof_train/train_results.db > $O/train_kernel_stats.txt
rocprofv3 --kernel-trace --stats -d $O/prof_scst -o scst -- python tools/bench_scst.py > $O/prof_scst.log 2>&1
python tools/rocprof_summary.py $O/prof_scst/scst_results.db > $O/scst_kernel_stats.txt
python tools/scst_breakdown.py > $O/scst_breakdown.txt 2>&1
PMC_STREAMS=3 bash tools/pmc_bench.sh gpurun_out/pmc_bench > $O/pmc.log 2>&1
cp gpurun_out/pmc_bench/traffic.json $O/pmc_traffic.json; cp gpurun_out/pmc_bench/sq_table.txt $O/sq_table.txt
python tools/bench_wgrad_shapes.py 2>&1 | grep -v amdgpu.ids > $O/train_large_products.txt
( for b in 4 8 16 32 64 128; do SET_PDEC_STAMPS=2 python tools/profile_small_batch.py $b 2>&1 | grep -v amdgpu.ids | awk '/pdec stamps/ {last=$0; next} {print} END {if (last) print last}'; done ) > $O/small_batch.txt 2>&1
rm -rf $O/prof_bench $O/prof_train $O/prof_scst gpurun_out/pmc_bench/fetch gpurun_out/pmc_bench/write gpurun_out/pmc_bench/sq
ls -la $O
