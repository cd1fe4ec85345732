#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r03_fuzz
timeout 400 python tools/fuzz_parity.py 150 31 2>&1 | grep -v amdgpu | tail -4 | tee gpurun_out/r03_fuzz/fuzz1.txt
timeout 400 python tools/fuzz_parity2.py 120 32 2>&1 | grep -v amdgpu | tail -4 | tee gpurun_out/r03_fuzz/fuzz2.txt
timeout 300 python tools/fuzz_parity3.py 60 33 2>&1 | grep -v amdgpu | tail -4 | tee gpurun_out/r03_fuzz/fuzz3.txt
