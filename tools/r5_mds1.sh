#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5_mds1; mkdir -p $O
{
echo "== default (picks 4)"; timeout 900 python tools/mds_surface.py --parity 2>&1 | grep mds
for k in 1 2 3; do echo "== picks $k"; AB_LIB=tools/ab/lib_picks$k.so timeout 600 python tools/mds_surface.py 2>&1 | grep mds; done
echo "== mds_ab parity"; timeout 900 python tools/mds_ab.py --parity 2>&1 | grep "parity\|B=32\|B=4 "
} > $O/mds.txt 2>&1
timeout 900 python -m pytest tests/test_mds.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
cat $O/mds.txt
