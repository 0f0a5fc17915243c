#!/bin/bash
# usage: bash tools/build_variant.sh NAME "-DFL_BATCH=2 ..."  -> ts-pws_amd/lib/variant_NAME.so (use with TSPWS_LIB_PATH)
set -e
NAME=$1; FLAGS=$2
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/ts-pws_amd/build
/opt/rocm/bin/hipcc $FLAGS --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -Wall -Wno-unused-result -c $R/ts-pws_amd/csrc/tspws_hip.hip -o $R/ts-pws_amd/build/v_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ts-pws_amd/lib/variant_$NAME.so $R/ts-pws_amd/build/v_$NAME.o $R/ts-pws_amd/build/fwd_mfma_spec.o $R/ts-pws_amd/build/tspws_main.o -lm
echo built variant_$NAME.so
