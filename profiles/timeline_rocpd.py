#!/usr/bin/env python3
"""Kernel timeline of the LAST step in a rocprofv3 rocpd database: start (us, relative), duration, queue, name.
usage: timeline_rocpd.py in.db [n_last_kernels]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = db.execute("select start, end, queue_id, name, grid_x, grid_y, workgroup_x, vgpr_count, lds_size from kernels order by start").fetchall()
rows = rows[-n:]
t0 = rows[0][0]
print(f"{'start_us':>10} {'end_us':>10} {'dur_us':>9} {'q':>3} {'grid':>12} {'vgpr':>5} {'lds':>7}  kernel")
for s, e, q, name, gx, gy, wx, vg, lds in rows:
    short = name.split("(")[0].replace("void ", "")[:60]
    print(f"{(s - t0) / 1e3:10.1f} {(e - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f} {q:3d} {gx // max(wx, 1):7d}x{gy:<4d} {vg:5d} {lds:7d}  {short}")
