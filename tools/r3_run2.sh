#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03b; mkdir -p $O
export TMPDIR=/tmp
echo "== gpu tests"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/tests.log; tail -12 $O/tests.log
echo "== A/B per batch size"
for spec in "new::" "bw8:tools/ab/lib_bw8.so:" "bw8_G32:tools/ab/lib_bw8.so:SN_EMD_G=32" "bw8_G16:tools/ab/lib_bw8.so:SN_EMD_G=16"; do
  name=${spec%%:*}; rest=${spec#*:}; lib=${rest%%:*}; envs=${rest#*:}
  echo "-- $name"
  env $envs AB_LIB=$lib AB_BS=32,16,8,4,2,1 timeout 600 python tools/emd_ab.py --parity 2>&1 | grep "ms per call\|parity" | tee $O/ab_$name.txt
done
echo "== phases"
for spec in "b32::32" "b4::4" "b32_bw8:tools/ab/lib_bw8.so:32" "b4_bw8:tools/ab/lib_bw8.so:4"; do
  name=${spec%%:*}; rest=${spec#*:}; lib=${rest%%:*}; bb=${rest#*:}
  echo "-- $name"
  SN_EMD_DIAG=2 AB_LIB=$lib AB_BS=$bb AB_DIAG_B=$bb timeout 600 python tools/emd_ab.py > $O/phases_$name.txt 2>&1; tail -16 $O/phases_$name.txt
done
echo "== bid stamps"
for spec in "b32:tools/ab/lib_stamps.so:32" "b4:tools/ab/lib_stamps.so:4" "b32_bw8:tools/ab/lib_bw8stamps.so:32" "b4_bw8:tools/ab/lib_bw8stamps.so:4"; do
  name=${spec%%:*}; rest=${spec#*:}; lib=${rest%%:*}; bb=${rest#*:}
  echo "-- $name"
  BID_STAMPS=1 SN_EMD_DIAG=1 AB_LIB=$lib AB_BS=$bb AB_DIAG_B=$bb timeout 600 python tools/emd_ab.py 2>&1 | tail -20 | tee $O/stamps_$name.txt
done
echo "== strong share"
timeout 600 python tools/strong_share.py 2>&1 | tee $O/strong_share.txt
AB_LIB=tools/ab/lib_bw8.so timeout 600 python tools/strong_share.py 2>&1 | tee $O/strong_share_bw8.txt
