#!/bin/bash
O=gpurun_out/r4c6; mkdir -p $O
export TMPDIR=/tmp
{
  echo "== surface data, per call"; AB_DATA=surface AB_BS=32,4 timeout 300 python tools/emd_ab.py --parity 2>&1 | grep "parity\|per call"
  echo "== surface data, B=4 phases"; AB_DATA=surface SN_EMD_DIAG=2 AB_DIAG_B=4 AB_BS=4 timeout 300 python tools/emd_ab.py 2>&1 | grep -v "amdgpu.ids"
  echo "== surface data, B=32 phases"; AB_DATA=surface SN_EMD_DIAG=2 AB_DIAG_B=32 AB_BS=32 timeout 300 python tools/emd_ab.py 2>&1 | grep -v "amdgpu.ids"
  echo "== surface data, noise 0.05, per call"; AB_DATA=surface AB_NOISE=0.05 AB_BS=32,4 timeout 300 python tools/emd_ab.py 2>&1 | grep "per call"
  echo "== surface data, SN_EMD_SCAN=0 (matrix-core search everywhere), per call"; AB_DATA=surface SN_EMD_SCAN=0 AB_BS=32,4 timeout 300 python tools/emd_ab.py 2>&1 | grep "per call"
  echo "== 4 clouds uniform, team size"; for gg in 32 16 8; do echo "SN_EMD_G=$gg"; SN_EMD_G=$gg AB_BS=4,8 timeout 300 python tools/emd_ab.py 2>&1 | grep "per call"; done
} > $O/emd_surface.txt 2>&1
timeout 600 python tools/net_host_profile.py config5 trained_stand_in > $O/host_profile_config5.txt 2>&1
echo done > $O/done.txt
