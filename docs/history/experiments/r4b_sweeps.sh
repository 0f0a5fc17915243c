#!/bin/bash
# round 4, second half (GPU box): the seeded sweeps against the oracle with NEW seeds on the final library (short-frame / many-trace rules in force),
# and once more with the rules forced the other way
cd $GRAFT_REPO_ROOT
f() { grep -i "mismatch" | tail -1; }
python tools/random_sweep_features.py 11000 120 2>&1 | f
python tools/random_sweep.py 11500 160 2>&1 | f
python tools/random_sweep_jackknife.py 11400 40 2>&1 | f
python tools/random_sweep_masked.py 11600 60 2>&1 | f
python tools/random_sweep_cli.py 11700 30 2>&1 | f
python tools/random_sweep_device.py 11800 80 2>&1 | f
python tools/random_sweep_large.py 11900 8 2>&1 | f
TSPWS_INV_SPLIT=1 TSPWS_FUSE_WGS=2048 TSPWS_FUSE_MINTPS=1 TSPWS_FWD_STEPS=16 python tools/random_sweep.py 12000 30 2>&1 | f
TSPWS_INV_SPLIT=0 TSPWS_FUSE_WGS=1 TSPWS_FWD_STEPS=96 TSPWS_TL_MIN=40 python tools/random_sweep_features.py 12200 30 2>&1 | f
TSPWS_TL_MIN=33 TSPWS_TL_PICK=0 python tools/random_sweep.py 12400 30 2>&1 | f
TSPWS_TL_MIN=33 TSPWS_TL_PICK=1 TSPWS_TL_MINNS1=17 TSPWS_TLSTEPS=8 python tools/random_sweep.py 12500 30 2>&1 | f
