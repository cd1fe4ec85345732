#!/bin/bash
# the three randomised parity sweeps against the oracle on the final build of round 6 (bit-exact / tolerance per op);
# the EMD sweep a second time with the data-dependent paths forced on (SN_EMD_SKIP=2 SN_EMD_SPREAD=2)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r06_fuzz; mkdir -p $O
timeout 400 python tools/fuzz_parity.py 100 71 2>&1 | grep -v amdgpu | tail -4 | tee $O/fuzz_emd_chamfer.txt
SN_EMD_SKIP=2 SN_EMD_SPREAD=2 timeout 400 python tools/fuzz_parity.py 80 72 2>&1 | grep -v amdgpu | tail -4 | tee $O/fuzz_emd_chamfer_forced_paths.txt
timeout 400 python tools/fuzz_parity2.py 100 73 2>&1 | grep -v amdgpu | tail -4 | tee $O/fuzz_mds_expansion_p2i.txt
timeout 300 python tools/fuzz_parity3.py 60 74 2>&1 | grep -v amdgpu | tail -4 | tee $O/fuzz_backward_grnet.txt
