#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_networks.py -m gpu -q -k "blocks" 2>&1 | grep -E "Mismatch|Max abs|Max rel|assert|Error|passed|failed" | head -40
