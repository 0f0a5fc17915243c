"""Child process of test_masked_replica_engines_agree: the masked-replica calls of tspws_main against the oracle under the
environment switches of the parent (read once per process by the library).  Prints MASKED_ENGINES <worst relative error>."""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import abi

tspws = importlib.import_module("ts-pws_amd")
lib = tspws.load()
rng = np.random.default_rng(11)
worst = 0.0


def check(kw, X, times):
    global worst
    p = abi.default_params(**kw)
    a = abi.run_main(lib.tspws_main, p, X, times=times)
    b = abi.run_main(abi.oracle().orc_tspws_main, p, X, times=times)
    assert a["rc"] == 0 and b["rc"] == 0
    worst = max(worst, abi.relerr(a["ls"], b["ls"]), abi.relerr(a["tsPWS"], b["tsPWS"]))
    if "jk_mtr" in a:
        assert np.array_equal(a["jk_mtr"], b["jk_mtr"]), (a["jk_mtr"], b["jk_mtr"])
        for c in range(len(a["jk_mtr"])):
            worst = max(worst, abi.relerr(a["jk_ls"][c], b["jk_ls"][c]), abi.relerr(a["jk_ts"][c], b["jk_ts"][c]))
    if "sub_ls" in a:
        for c in range(a["sub_ls"].shape[0]):
            worst = max(worst, abi.relerr(a["sub_ls"][c], b["sub_ls"][c]), abi.relerr(a["sub_ts"][c], b["sub_ts"][c]))


mtr, N = 500, 2048
X = abi.synth_traces(mtr, N, seed=21)
times = 1262304000 + 86400 * np.sort(rng.integers(0, 4 * 365, mtr))
check(dict(Kmax=10, unbiased=1, jackknife_n=10, jackknife_d=1), X, times)              # 11 columns: rows straight from the walk
check(dict(Kmax=6, jackknife_n=7, jackknife_d=2, type=-3), X, times)                    # 22 columns: snapshots
check(dict(Kmax=7, unbiased=1, jackknife_n=3, jackknife_d=1), X[:61], times[:61])       # odd sizes: ragged groups
# one bin holds most of the traces: the replica that deletes it has fewer traces than some groups of the others
t2 = times.copy()
t2[50:450] = 1262304000 + 86400 * 200
check(dict(Kmax=8, jackknife_n=5, jackknife_d=1), X, np.sort(t2))
# a replica with fewer traces than groups (3 traces, 8 groups): most of its groups are empty -- rows nobody stores
t3 = np.concatenate([np.full(17, 1262304000 + 86400 * 10), np.full(3, 1262304000 + 86400 * 300)]).astype(np.int64)
check(dict(Kmax=8, jackknife_n=2, jackknife_d=1), X[:20], t3)
# N not a multiple of 4: the scalar form of the walk
check(dict(Kmax=4, jackknife_n=4, jackknife_d=1), abi.synth_traces(90, 1501, seed=5), times[:90])
# two-stage random subsampling runs on the same engine (masks from libc rand(): same seed on both sides)
p = abi.default_params(subsmpl_N=5, subsmpl_p=0.6, Kmax=5, unbiased=1)
abi.srand(3)
a = abi.run_main(lib.tspws_main, p, X[:200], times=None)
abi.srand(3)
b = abi.run_main(abi.oracle().orc_tspws_main, p, X[:200], times=None)
for c in range(5):
    worst = max(worst, abi.relerr(a["sub_ls"][c], b["sub_ls"][c]), abi.relerr(a["sub_ts"][c], b["sub_ts"][c]))
print("MASKED_ENGINES", worst)
