cd $GRAFT_REPO_ROOT
bash tools/variants_cmd.sh "tools/cfg4_run.py" libtspws_hip.so variant_qt32.so variant_qt32b.so > gpurun_out/qt32_cfg4.txt 2>&1
for so in variant_qt32.so variant_qt32b.so; do
TSPWS_LIB_PATH=$PWD/ts-pws_amd/lib/$so timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "forward_inverse or jackknife or many_trace" 2>&1 | tail -5 >> gpurun_out/qt32_cfg4.txt
done
cat gpurun_out/qt32_cfg4.txt
