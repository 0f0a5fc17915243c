#!/bin/bash
# Steady-state ablation of the opt-in matrix-pipe forward kernel (run on the GPU box from the repo root).
# Build the variants first, in-tree so that they travel with the snapshot:
#   for v in NOX NOMULT NOLDSREAD NOSTAGE NOSTORE; do make -C ts-pws_amd clean; make -C ts-pws_amd EXTRA_HIPFLAGS=-DFM_ABL_$v; \
#     cp ts-pws_amd/lib/libtspws_hip.so ts-pws_amd/lib/abl_$v.so; done; make -C ts-pws_amd clean; make -C ts-pws_amd
export MG_TRACES=256 TSPWS_DEBUG_NOGATHER=1
for lib in libtspws_hip abl_NOX abl_NOSTAGE abl_NOLDSREAD abl_NOMULT abl_NOSTORE; do
  [ -f ts-pws_amd/lib/$lib.so ] || continue
  echo "== $lib: $(TSPWS_LIB_PATH=$PWD/ts-pws_amd/lib/$lib.so python tools/mfma_groups.py 1 6 11 | grep '^group' | awk '{printf "g%s %s | ", $2, $NF}')"
done
echo "== lds kernel, all: $(TSPWS_FWD_KERNEL=lds python tools/mfma_groups.py -1 | grep ^group)"
echo "== mfma kernel, all: $(python tools/mfma_groups.py -1 | grep ^group)"
