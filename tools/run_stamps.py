#!/usr/bin/env python3
"""Phase shares of the step kernel from a -DFCM_STAMP diagnostic build (tools/run_stamps.sh builds it
into tools/libfcm_stamp.so).  Shares only: the stamps drain every counter, so the run time means nothing."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import flag_complex_mcmc_amd as fcm
from flag_complex_mcmc_amd import _ffi
n, chains, props = 1000, int(os.environ.get("FCM_STAMP_CHAINS", "4096")), 512
CLIQUE = len(sys.argv) > 1 and sys.argv[1] == "clique"   # clique moves only: slots 1..5 then mean the clique phases
e = fcm.graphs.random_with_p(n, 0.10, 0)
FLIPS = len(sys.argv) > 1 and sys.argv[1] == "flips"    # flips only: slots 3..5 then split the evaluation
if FLIPS:
    g = fcm.Graph.from_edges(n, e)
    fc = g.flagser_count(0)
    b = fcm.Bounds.calculate(g, fc, fcm.Bounds.target(fc, 0.01), 0)
    s = fcm.MCMCSampler(g, b, n_chains=chains, seed=0, move_weights=[1.0, 0.0, 0.0, 0.0])
elif CLIQUE:
    props = 64
    g = fcm.Graph.from_edges(n, e)
    fc = g.flagser_count(0)
    b = fcm.Bounds.calculate(g, fc, fcm.Bounds.target(fc, 0.01), 0)
    s = fcm.MCMCSampler(g, b, n_chains=chains, seed=0, move_weights=[0.0, 0.0, 0.75, 0.25])
else:
    s = fcm.initialize_new_sampler(fcm.Graph.from_edges(n, e), n_chains=chains, seed=0)
s.step(props)
out = np.zeros((chains, 8), np.uint64)
_ffi.check(_ffi.lib().fcm_sampler_debug_stamps(s._h, out.ctypes.data_as(_ffi.u64p)))
tot = out.sum(axis=0).astype(float)
names = ["decode/other", "flip: list round trip", "flip: build", "flip: 2 evaluations", "dmove: lists + candidate build",
         "dmove: build + 2 evaluations", "reduce + bounds + commit", "batch draw"]
if FLIPS:
    names = ["decode/other", "flip: list round trip", "flip: build", "eval: classes, seating, split rows (+ tail of slot 5)",
             "eval: arc scan + scatter", "eval: arcs and deeper levels", "reduce + bounds + commit", "batch draw"]
if CLIQUE:
    names = ["decode/other", "clique: pick, d, permutations", "clique: OLD gather, NEW, pair list", "clique: pair ids + table entries",
             "clique: per pair list + build", "clique: per pair evaluations + stores", "reduce + bounds + slots/revert", "batch draw"]
st = s.stats()
nf, nd = st["n_flip"].sum(), st["n_dmove"].sum()
print("cycles per proposal (all phases): %.0f" % (tot.sum() / (chains * props)))
for nm, t in zip(names, tot):
    print("%-34s %5.1f %%   %8.0f cycles per proposal" % (nm, 100 * t / tot.sum(), t / (chains * props)))
if CLIQUE:
    nc = st["n_cperm"].sum() + st["n_cswap"].sum()
    print("non-empty clique moves %d of %d, changed directed edges per move %.2f" % (nc, chains * props, st["n_changes"].sum() / max(nc, 1)))
    sys.exit(0)
print("per flip: list %.0f, build %.0f, evals %.0f;  per dmove: lists+cand build %.0f, build+evals %.0f" %
      (tot[1] / nf, tot[2] / nf, tot[3] / nf, tot[4] / nd, tot[5] / nd))
