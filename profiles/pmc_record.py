#!/usr/bin/env python3
"""usage: pmc_record.py gpurun_out/pmc_TAG > profiles/pmc_partial_stacks.json
The HBM-traffic record bench.py quotes in roofline.traffic: mean FETCH_SIZE / WRITE_SIZE of the k_partial launches in the csv passes
of profiles/collect_pmc.sh (separate rocprofv3 --pmc runs), with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts
64 B per 128-B request on wide coalesced streams: x2) and the sha256 of the streaming kernel's source the counters were taken on --
bench.py reports traffic only while csrc/stream.hip still has that hash."""
import csv
import glob
import hashlib
import json
import os
import sys

root = sys.argv[1]
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
vals = {"FETCH_SIZE": [], "WRITE_SIZE": []}
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Kernel_Name"].startswith("void k_partial<true>") or row["Kernel_Name"].startswith("k_partial<true>"):
            if row["Counter_Name"] in vals:
                vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
if not vals["FETCH_SIZE"] or not vals["WRITE_SIZE"]:
    sys.exit("no k_partial counters under " + root)
fetch = sum(vals["FETCH_SIZE"]) / len(vals["FETCH_SIZE"])
write = sum(vals["WRITE_SIZE"]) / len(vals["WRITE_SIZE"])
src = os.path.join(here, "ts-pws_amd", "csrc", "stream.hip")
rec = {
    "kernel": "k_partial<true>",
    "source": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes (profiles/collect_pmc.sh), bench.py --steps 2 --warmup 1: 10000x131072, K=10; mean over {len(vals['FETCH_SIZE'])} k_partial launches",
    "launch": "grid 128 x 2 (two 1000-trace groups, one workgroup per CU), 256 threads; five such launches per call",
    "launches_per_call": 5,
    "FETCH_SIZE_KB": round(fetch, 1),
    "WRITE_SIZE_KB": round(write, 1),
    "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request on wide coalesced streams -> x2 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported",
    "hbm_bytes_per_launch": int(round((2 * fetch + write) * 1024)),
    "algorithmic_bytes_per_launch": 2 * 1000 * 131072 * 4 + 2 * 131072 * 8,
    "stream_hip_sha256": hashlib.sha256(open(src, "rb").read()).hexdigest(),
}
print(json.dumps(rec, indent=1))
