#!/usr/bin/env python3
"""Where a wave's time goes in the multi-wave kernel, from a -DMW_STAMP diagnostic build (tools/mw_stamps.sh builds it
and points FCM_LIB_PATH at it).  usage: mw_stamps.py <config 2|3|4> <chains> [W]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import flag_complex_mcmc_amd as fcm
from flag_complex_mcmc_amd import _ffi
from bench import build_workload
cfg, chains = int(sys.argv[1]), int(sys.argv[2])
if len(sys.argv) > 3:
    os.environ["FCM_MW"] = sys.argv[3]
n, e = build_workload(fcm, cfg, 1000, 0.10, 0)
s = fcm.initialize_new_sampler(fcm.Graph.from_edges(n, e), n_chains=chains, seed=0)
props = 4096
s.step(props)
out = np.zeros((chains, 8), np.uint64)
_ffi.check(_ffi.lib().fcm_sampler_debug_stamps(s._h, out.ctypes.data_as(_ffi.u64p)))
tot = out.sum(axis=0).astype(float)
names = ["table entry, vis publish (own stores), snap", "proposal: lists, builds, evaluations", "staging, checks, wait for the token",
         "hand-over (head store -> next holder has the token)", "decision under the token"]
cnt = tot[5]
raw = out[:, 6].astype(np.uint64)
d1 = float((raw & np.uint64(0xFFFFFFFF)).sum()) / cnt
d2 = float((raw >> np.uint64(32)).sum()) / cnt
r7 = out[:, 7].astype(np.uint64)
print("  after the stores: head store %.0f, commit (token already passed on) %.0f cycles" % (float((r7 >> np.uint64(32)).sum()) / cnt, float((r7 & np.uint64(0xFFFFFFFF)).sum()) / cnt))
print("config %d, %d chains, W=%d: %d proposals, %.1f polls per proposal; decision = %.0f (counts, bounds) + %.0f (stores) + rest (state word, head)" % (cfg, chains, s.info["waves_per_chain"], cnt, tot[7] / cnt, d1, d2))
for i, nm in enumerate(names):
    print("  %-50s %9.0f cycles per proposal" % (nm, tot[i] / cnt))
print("  %-50s %9.0f" % ("sum (per wave per proposal)", tot[:5].sum() / cnt))
