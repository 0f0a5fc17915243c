#!/bin/bash
mkdir -p gpurun_out
timeout 900 python3 tools/spectral_check.py quick > gpurun_out/spec_check.log 2>&1
grep -c "e-1[0-9]\|0.00e+00" gpurun_out/spec_check.log; grep -i "error\|assert\|Traceback" gpurun_out/spec_check.log | head
export TSPWS_ENGINE=spectral TSPWS_SPEC_SERIAL=1 TSPWS_SPEC_NSMAX=2048
bash tools/gpu_prof_cfg.sh r05f tools/cfg2_run.py 2>&1 | grep "k_spec\|k_fwd_tl"
unset TSPWS_SPEC_SERIAL TSPWS_SPEC_NSMAX
echo "== size sweep: fir | spectral nsmax 512 1024 2048 4096"
for sz in 128:32768 256:32768 512:32768 1024:32768 2048:32768 256:8192 1024:8192 4096:8192 512:16384 2048:16384 128:65536 512:65536 256:131072 1024:131072; do
  line="$sz:"
  r=$(TSPWS_ENGINE=fir python3 tools/cfg_bench.py c:$sz 5 2>/dev/null | grep -o "[0-9.]* ms/call, digest [0-9a-f]*"); line="$line fir $r |"
  for ns in 512 1024 2048 4096; do
    r=$(TSPWS_ENGINE=spectral TSPWS_SPEC_NSMAX=$ns python3 tools/cfg_bench.py c:$sz 5 2>/dev/null | grep -o "[0-9.]* ms/call, digest [0-9a-f]*"); line="$line $ns: $r |"
  done
  echo "$line"
done
