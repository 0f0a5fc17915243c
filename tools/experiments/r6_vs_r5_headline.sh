#!/bin/bash
# round 6: the headline call of the round-5 tree (git worktree _r5, built in place) and of this tree, alternating on ONE box: is the box-to-box spread of the
# streaming stage (0.73 vs 0.76 ms) a property of the box or of this round's library?
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
{
for i in 1 2 3; do
  echo -n "r5: "; python _r5/tools/finish_ab.py 60
  echo -n "r6: "; python tools/finish_ab.py 60
done
echo "bench r5:"; python _r5/bench.py --no-cpu | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['stage_ms']['median'], d['step_ms_gpu']['median'])"
echo "bench r6:"; python bench.py --no-cpu | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['stage_ms']['median'], d['step_ms_gpu']['median'])"
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6_vs_r5_headline.txt
