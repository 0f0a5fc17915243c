#!/bin/bash
unset TSPWS_ENGINE
bash tools/gpu_timeline_cfg.sh r05c4s 30 tools/cfg4_run.py 2>&1 | grep -v "^W2026\|amdgpu.ids" | tail -34
