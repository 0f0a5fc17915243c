#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
for e in fir spectral; do TSPWS_ENGINE=$e python tools/noise_floor_probe.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r6_noise_floor.txt
