#!/bin/bash
# usage: bash tools/build_variant.sh NAME "-DFL_BATCH=2 ..."  -> ts-pws_amd/lib/variant_NAME.so (use with TSPWS_LIB_PATH)
# every unit is compiled with the extra flags into its own build directory; the default library is left alone
set -e
NAME=$1; FLAGS=$2
R=$(cd "$(dirname "$0")/.." && pwd)
make -C $R/ts-pws_amd -j6 BUILD=build_v_$NAME LIB=$R/ts-pws_amd/lib/variant_$NAME.so EXTRA_HIPFLAGS="$FLAGS" lib
echo built variant_$NAME.so
