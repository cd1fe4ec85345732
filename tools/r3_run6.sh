#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03e; mkdir -p $O
export TMPDIR=/tmp
echo "== p2i tests"; timeout 1500 python -m pytest tests/test_p2i.py tests/test_fullsize.py tests/test_dropin.py -m gpu -q -k "p2i or dropin or depth" 2>&1 | tail -5
echo "== render probe new / old"
for i in 1 2; do python tools/render_probe.py; AB_LIB=tools/ab/lib_p2iold.so python tools/render_probe.py; done 2>&1 | grep render
echo "== kernel stats new"; KTOP=8 tools/kstats.sh tools/render_probe.py 2>&1 | tail -9
echo "== kernel stats old"; AB_LIB=tools/ab/lib_p2iold.so KTOP=8 tools/kstats.sh tools/render_probe.py 2>&1 | tail -9
