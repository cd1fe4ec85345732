#!/bin/bash
# the three randomised parity sweeps against the oracle on the final build of round 4 (bit-exact / tolerance per op)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04_fuzz; mkdir -p $O
timeout 400 python tools/fuzz_parity.py 120 41 2>&1 | grep -v amdgpu | tail -4 | tee $O/fuzz_emd_chamfer.txt
timeout 400 python tools/fuzz_parity2.py 120 42 2>&1 | grep -v amdgpu | tail -4 | tee $O/fuzz_mds_expansion_p2i.txt
timeout 300 python tools/fuzz_parity3.py 60 43 2>&1 | grep -v amdgpu | tail -4 | tee $O/fuzz_backward_grnet.txt
