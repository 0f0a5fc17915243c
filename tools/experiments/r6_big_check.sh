#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
python tools/big_anyN_check.py 2>&1 | grep -v amdgpu.ids | grep -v "waveletFamily" | tee gpurun_out/r6_big_check.txt
