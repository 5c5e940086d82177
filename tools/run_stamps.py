#!/usr/bin/env python3
"""Phase shares of the step kernel from a -DFCM_STAMP diagnostic build (tools/run_stamps.sh builds it
into tools/libfcm_stamp.so).  Shares only: the stamps drain every counter, so the run time means nothing."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import flag_complex_mcmc_amd as fcm
from flag_complex_mcmc_amd import _ffi
n, chains, props = 1000, 4096, 512
e = fcm.graphs.random_with_p(n, 0.10, 0)
s = fcm.initialize_new_sampler(fcm.Graph.from_edges(n, e), n_chains=chains, seed=0)
s.step(props)
out = np.zeros((chains, 8), np.uint64)
_ffi.check(_ffi.lib().fcm_sampler_debug_stamps(s._h, out.ctypes.data_as(_ffi.u64p)))
tot = out.sum(axis=0).astype(float)
names = ["decode/other", "flip: list round trip", "flip: build", "flip: 2 evaluations", "dmove: lists + candidate build",
         "dmove: build + 2 evaluations", "reduce + bounds + commit", "batch draw"]
st = s.stats()
nf, nd = st["n_flip"].sum(), st["n_dmove"].sum()
print("cycles per proposal (all phases): %.0f" % (tot.sum() / (chains * props)))
for nm, t in zip(names, tot):
    print("%-34s %5.1f %%   %8.0f cycles per proposal" % (nm, 100 * t / tot.sum(), t / (chains * props)))
print("per flip: list %.0f, build %.0f, evals %.0f;  per dmove: lists+cand build %.0f, build+evals %.0f" %
      (tot[1] / nf, tot[2] / nf, tot[3] / nf, tot[4] / nd, tot[5] / nd))
