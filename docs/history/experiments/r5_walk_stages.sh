export TSPWS_LIB_PATH=${GRAFT_REPO_ROOT:-$PWD}/ts-pws_amd/lib/libtspws_hip_sweeps.so
for st in 1 2 5 10; do
TSPWS_JK_STAGES=$st bash tools/gpu_timeline_cfg.sh ws$st 120 tools/cfg4_run.py > /dev/null 2>&1
echo "stages $st"; grep "k_rows_walk\|k_seg_fix" gpurun_out/timeline_ws$st.txt | tail -22 | awk '{print $3, $5, $8}' | tr '\n' ';'; echo
done
