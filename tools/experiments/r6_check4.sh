#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 1500 python -m pytest tests/test_spectral_gpu.py -q -x -k "coefficients_vs_oracle" 2>&1 | tail -4
