#!/bin/bash
# usage: bash tools/build_base.sh [REV]  -> ts-pws_amd/lib/variant_base.so = the library of REV (default HEAD), for
# side-by-side timing against the working tree's build (tools/variants.sh)
set -e
REV=${1:-HEAD}
R=$(cd "$(dirname "$0")/.." && pwd)
rm -rf /tmp/tspws_base && git -C $R worktree add -f /tmp/tspws_base $REV -q
make -C /tmp/tspws_base/ts-pws_amd -j6 lib > /dev/null
cp /tmp/tspws_base/ts-pws_amd/lib/libtspws_hip.so $R/ts-pws_amd/lib/variant_base.so
git -C $R worktree remove --force /tmp/tspws_base
echo built variant_base.so from $REV
