#!/bin/bash
# What could a trace PAIR per thread in k_fwd_lds save?  Its shareable parts are the two workgroup barriers per trace (the tap reads are already shared by
# the 16-output window, the window addresses are scalar).  Timing ablation (-DFL_ABLATE=1 build, results wrong): barriers only in front of even traces.
export TSPWS_LIB_PATH=ts-pws_amd/lib/variant_abl.so
for m in 7f 107f 7f 107f 40 1040 3f 103f; do echo "mask $m"; TSPWS_FWD_CLASSES=$m python tools/fwd_bench.py 131072 10 200; done
