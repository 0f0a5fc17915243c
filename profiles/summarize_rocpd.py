#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (…_results.db from `rocprofv3 --kernel-trace --stats`) into the
per-kernel summary text committed under profiles/.   usage: summarize_rocpd.py in.db > out.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
print(f"{'calls':>6} {'total_us':>12} {'avg_us':>10} {'%':>6}  kernel")
for name, calls, tot, avg, pct in rows:
    short = name.split("(")[0].replace("void ", "")
    print(f"{calls:6d} {tot / 1e3 if tot > 1e6 else tot:12.1f} {avg / 1e3 if tot > 1e6 else avg:10.2f} {pct:6.2f}  {short}")
