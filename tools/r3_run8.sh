#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03g; mkdir -p $O
timeout 600 python tools/mds_ab.py --parity 2>&1 | grep parity
for spec in "G8_r0.5:SN_MDS_G=8 SN_MDS_RATIO=0.5" "G8_r0.1:SN_MDS_G=8 SN_MDS_RATIO=0.1" "G4_r0.1:SN_MDS_G=4 SN_MDS_RATIO=0.1" "G16_r0.1:SN_MDS_G=16 SN_MDS_RATIO=0.1" "G2_r0.1:SN_MDS_G=2 SN_MDS_RATIO=0.1"; do
  name=${spec%%:*}; envs=${spec#*:}
  echo "-- $name"
  env $envs timeout 900 python tools/mds_ab.py 2>&1 | grep -v amdgpu | tee $O/mds2_$name.txt
done
