#!/bin/bash
# round 6: the N > 1 bench line at FULL size with two ranks on the one GPU (collectives over gloo): flow, legs, deadlines, JSON -- and the N = 1 line's wall time
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
SECONDS=0; BENCH_BACKEND=gloo python bench.py --gpus 2 > gpurun_out/r6_bench_n2.json 2> gpurun_out/r6_bench_n2.err
echo "rc=$? wall=${SECONDS}s"; SECONDS=0; tail -c 3000 gpurun_out/r6_bench_n2.json; true
python bench.py > gpurun_out/r6_bench_n1.json 2> gpurun_out/r6_bench_n1.err
echo "rc=$? wall=${SECONDS}s"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6_bench_n1.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["stage_ms"]["median"], [ (k, v.get("ms_per_call")) for k,v in d["other_configs"].items()], d["other_configs"]["cfg1_example_shape_499x16501"]["single_stage"]["ms_per_call"])
PY
