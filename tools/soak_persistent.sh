#!/bin/bash
# stability soak of the persistent decode kernels (rounds 5-6): the persistent-decode, beam and boundary tests 12 times each
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/soak
fail=0
for i in $(seq 1 12); do
  timeout 600 python -m pytest tests/test_hip_persistent_decode.py tests/test_hip_beam.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -1 | tee -a gpurun_out/soak/soak.log | grep -q "passed" || fail=$((fail+1))
  timeout 300 python -m pytest tests/test_hip_boundary.py -m gpu -q -x -p no:cacheprovider -k "persistent or concurrent or stream" 2>&1 | tail -1 >> gpurun_out/soak/soak.log
done
echo "rounds with a failure: $fail" | tee -a gpurun_out/soak/soak.log
