#!/bin/bash
# Persistent small-batch decode (csrc/decode_persistent*.hip): B = 1..8 against the per-step loop, the phase time stamps of one
# workgroup, the per-kernel breakdown of both paths and the rocprofv3 kernel summary.   profile_persistent_decode.sh [outdir]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/pdec}; mkdir -p $O
{
echo "== DCNet greedy (BASELINE.json configs[0]: B = 4), persistent launch vs per-step loop"
PROBE_BS=4,1,2,3,8 python tools/pdec_probe.py 2>&1 | grep "B=\|golden"
echo "== EditNet greedy"
PROBE_BS=4,1,2,3,5,6 python tools/pdec_probe_editnet.py 2>&1 | grep "B=\|golden"
echo "== phase stamps of workgroup 0 (100-MHz clock), DCNet B = 4: 0 token gather | 1 reduce+cell1+put h1 | 2 request next tiles | 3 poll h1 |"
echo "   4 S2 | 5 put projection | 6 poll + attention | 7 gates + cell2 + put h2 | 8 poll h2 | 9 fc | 10 stats + put | 11 S1' (next timestep's gate products) | 12 poll + combine"
PROBE_BS=4 SET_PDEC_STAMPS=1 python tools/pdec_probe.py 2>&1 | grep stamps | tail -1
echo "== phase stamps, EditNet B = 4: 0 token gather | 1 cell1+put | 2 poll h1 | 3 S2 | 4 put proj | 5 poll | 6 caption attention + visual score | 7 context gate + put |"
echo "   8 poll | 9 S4 + visual softmax | 10 c_new put | 11 poll | 12 S5 + copy gate + put h2 | 13 poll | 14 fc | 15 stats + put + S1' (next timestep's gate products) | 16 poll | 17 combine"
PROBE_BS=4 SET_PDEC_STAMPS=1 python tools/pdec_probe_editnet.py 2>&1 | grep stamps | tail -1
echo "== per-kernel breakdown, EditNet B = 4 (persistent launch)"
python tools/profile_small_batch.py 4
echo "== per-kernel breakdown, EditNet B = 4 (per-step loop, SET_DEC_PERSISTENT=0)"
SET_DEC_PERSISTENT=0 python tools/profile_small_batch.py 4
} > $O/persistent_decode.txt 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof -o pd -- python tools/profile_small_batch.py 4 > $O/prof.log 2>&1
python tools/rocprof_summary.py $O/prof/pd_results.db > $O/persistent_decode_kernel_stats.txt 2>&1
rm -rf $O/prof
cat $O/persistent_decode.txt
head -12 $O/persistent_decode_kernel_stats.txt | cut -c1-150
