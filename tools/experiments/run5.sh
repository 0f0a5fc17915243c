cd $GRAFT_REPO_ROOT
for w in 256 384 512; do for pr in -1 1; do echo walk $w xfprio $pr; TSPWS_JK_XFPRIO=$pr TSPWS_JK_WALK=$w python tools/cfg4_run.py | tail -1; done; done
echo wave prio 3; for w in 256 512; do TSPWS_LIB_PATH=$PWD/ts-pws_amd/lib/variant_wprio.so TSPWS_JK_WALK=$w python tools/cfg4_run.py | tail -1; done
