#!/bin/bash
# round 6: whole GPU suite on the any-N build + cfg1 / other sizes timings + timeline
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
S=$PWD/ts-pws_amd/lib/libtspws_hip_sweeps.so
{
python tools/cfg_bench.py cfg1 20
python tools/cfg_bench.py cfg2 20
python tools/cfg_bench.py c:499:16384 20
python tools/cfg_bench.py c:500:20000 20
python tools/cfg_bench.py c:256:86400 10
for nsmax in 512 1024 2048; do
  echo "== sweeps NSMAX=$nsmax"
  TSPWS_LIB_PATH=$S TSPWS_SPEC_NSMAX=$nsmax python tools/cfg_bench.py cfg1 20
done
} > gpurun_out/r6_anyN_bench.log 2>&1
grep -v amdgpu.ids gpurun_out/r6_anyN_bench.log
bash tools/gpu_timeline_cfg.sh r6cfg1 24 tools/cfg1s_run.py | grep -v amdgpu.ids | tail -26
timeout 3000 python -m pytest tests -q -m gpu -x > gpurun_out/r6_gpu_suite.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6_gpu_suite.log
tail -12 gpurun_out/r6_gpu_suite.log
