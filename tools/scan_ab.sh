#!/bin/bash
# usage (GPU box): tools/scan_ab.sh [lib ...] -- the per-bidder scan of sparse iterations (SN_EMD_SCAN = bidders per
# workgroup up to which an iteration takes it; 0: the matrix-core group search everywhere)
for v in "" "$@"; do
echo "=== lib ${v:-default}"
AB_LIB=$v python tools/emd_ab.py --parity --parity32 2>&1 | grep -i parity | grep -v "True True True"
for s in ${SCANS:-0 256}; do
  echo "== SN_EMD_SCAN=$s"
  AB_LIB=$v SN_EMD_SCAN=$s AB_BS=32,16,8,4,1 python tools/emd_ab.py 2>&1 | grep "per call"
done
done
for s in ${DIAGS:-256}; do
echo "== phases, SN_EMD_SCAN=$s"
SN_EMD_SCAN=$s SN_EMD_DIAG=2 AB_BS=32 AB_DIAG_B=32 python tools/emd_ab.py 2>&1 | grep -E "it +[0-9]+:|tail mean"
SN_EMD_SCAN=$s SN_EMD_DIAG=2 AB_BS=4 AB_DIAG_B=4 python tools/emd_ab.py 2>&1 | grep -E "it +(20|40|49):|tail mean"
if [ -f tools/ab/lib_stamps.so ]; then
SN_EMD_SCAN=$s SN_EMD_DIAG=2 BID_STAMPS=1 AB_LIB=tools/ab/lib_stamps.so AB_BS=32 AB_DIAG_B=32 python tools/emd_ab.py 2>&1 | grep -A20 "bid_group, tail"
fi
done
