#!/usr/bin/env python3
"""usage: pmc_record_call.py gpurun_out/pmc_TAG k_rows_walk > profiles/pmc_cfg4_call.json
Whole-call HBM traffic of a multi-kernel call from the csv passes of profiles/collect_pmc.sh (separate rocprofv3 --pmc FETCH_SIZE /
WRITE_SIZE runs over tools/cfg4_run.py): sum over ALL kernels of the process (plan-time kernels and k_synth excluded) of
2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE counts 64 B per 128-B request on wide coalesced
streams), divided by the number of calls = dispatches of the marker kernel (second argument: one per call).  Tagged with the sha256 of
the sources of the call's kernels: bench.py quotes the record only while they are unchanged."""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict

root, marker = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SKIP = ("k_synth", "k_gen_taps", "k_spec_twiddle", "k_spec_wnorm", "k_spec_place", "k_spec_permute", "k_spec_mid<true>")
tot = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE") and not k.startswith(SKIP):
            tot[row["Counter_Name"]][k] += float(row["Counter_Value"])
            cnt[row["Counter_Name"]][k] += 1
calls = max((n for k, n in cnt["FETCH_SIZE"].items() if k.startswith(marker)), default=0)
if not calls:
    sys.exit(f"no {marker} dispatches under {root}")
per = {k: {"FETCH_SIZE_KB": round(tot["FETCH_SIZE"][k] / calls, 1), "WRITE_SIZE_KB": round(tot["WRITE_SIZE"].get(k, 0.0) / calls, 1)} for k in sorted(tot["FETCH_SIZE"])}
fetch = sum(v["FETCH_SIZE_KB"] for v in per.values())
write = sum(v["WRITE_SIZE_KB"] for v in per.values())
srcs = ["stream.hip", "resample.hip", "spectral.hip", "forward.hip", "inverse.hip", "fwd_lds.h", "fwd_poly.h", "inv_poly.h"]
h = hashlib.sha256()
for s in srcs:
    h.update(open(os.path.join(here, "ts-pws_amd", "csrc", s), "rb").read())
print(json.dumps({
    "source": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes (profiles/collect_pmc.sh, PMC_CMD=tools/cfg4_run.py): 10000 x 131072, Mexican hat, "
              f"two-stage K = 10 + jackknife n = 10, d = 1; totals over every kernel of the process / {calls} calls ({marker} dispatches)",
    "calls": calls,
    "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request on wide coalesced streams -> x2 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported",
    "hbm_bytes_per_call": int(round((2 * fetch + write) * 1024)),
    "FETCH_SIZE_KB_per_call": round(fetch, 1), "WRITE_SIZE_KB_per_call": round(write, 1),
    "per_kernel_per_call": per,
    "sources_sha256": h.hexdigest(), "sources": srcs,
}, indent=1))
