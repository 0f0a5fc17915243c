#!/bin/bash
# round 6, last sweeps on the final library (new seeds): the masked / jackknife / feature / device-shard / command-line sweeps after the asynchronous return of the
# stack + jackknife call, and one more any-N run
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
f() { grep -E "MISMATCH|cases|mismatch|Traceback|Error" | grep -v "waveletFamily"; }
{
python tools/random_sweep_masked.py 75000 60 2>&1 | f
python tools/random_sweep_jackknife.py 76000 40 2>&1 | f
python tools/random_sweep_features.py 77000 60 2>&1 | f
python tools/random_sweep_device.py 78000 40 2>&1 | f
python tools/random_sweep_cli.py 79000 20 2>&1 | f
SWEEP_ANYN=1 python tools/random_sweep_spectral.py 80000 40 2>&1 | f
} 2>&1 | tee gpurun_out/r6_sweeps3.txt
