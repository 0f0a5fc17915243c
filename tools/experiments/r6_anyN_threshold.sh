#!/bin/bash
# round 6: from which batch size on the spectral engine beats the FIR kernels when N is not a power of two (the window is up to twice the trace)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
{
for shape in 64:16501 96:16501 128:16501 192:16501 256:16501 64:20000 128:20000 64:5000 128:5000 256:5000 256:3000 512:3000 1024:3000 256:1500 1024:1500 64:86400 128:86400 64:40000 128:40000; do
  for e in fir spectral; do
    printf "%-9s" $e; TSPWS_ENGINE=$e python tools/cfg_bench.py c:$shape 30 2>&1 | grep -v amdgpu
  done
done
} | tee gpurun_out/r6_anyN_threshold.txt
