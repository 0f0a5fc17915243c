#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 1500 python -m pytest tests/test_cli_gpu.py -q -x -k "bench" 2>&1 | tail -8
