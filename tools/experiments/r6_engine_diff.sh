#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
TSPWS_ENGINE=fir python tools/engine_diff.py run 128 86400 /tmp/a.npz 2>&1 | grep -v amdgpu
TSPWS_ENGINE=spectral python tools/engine_diff.py run 128 86400 /tmp/b.npz 2>&1 | grep -v amdgpu
python tools/engine_diff.py cmp /tmp/a.npz /tmp/b.npz 128 86400 2>&1 | grep -v amdgpu | tee gpurun_out/r6_engine_diff.txt
