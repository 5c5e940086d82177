#!/usr/bin/env python3
"""Timeline of one chain of the multi-wave kernel from a -DMW_TRACE=<first proposal> diagnostic build (tools/variant_lib.sh
trace -DMW_TRACE=2048; FCM_LIB_PATH=tools/_stamp/trace/libfcm.so).  usage: mw_trace.py <config> <chains> [W]
Per proposal: start, end of the run (with q - snap), end of the wait (token, or token and run again), head store."""
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
W = s.info["waves_per_chain"]
s.step(4096)
out = np.zeros((chains, 8), np.uint64)
_ffi.check(_ffi.lib().fcm_sampler_debug_stamps(s._h, out.ctypes.data_as(_ffi.u64p)))
flat = out.reshape(-1)
ev = flat[flat != 0]
nev = ev.size
t = (ev >> np.uint64(24)).astype(np.int64); q = ((ev >> np.uint64(8)) & np.uint64(0xFFFF)).astype(np.int64)
x = ((ev >> np.uint64(4)) & np.uint64(15)).astype(np.int64); ty = (ev & np.uint64(15)).astype(np.int64)
t0 = t.min()
rows = {}
for ti, qi, xi, yi in zip(t, q, x, ty):
    rows.setdefault(qi, {})[yi] = (ti - t0, xi)
print("W = %d; %d events; columns: proposal wave | start | run done (+q-snap) | wait over (T token, R token and run again) | head store" % (W, nev))
prev_dec = None
for qi in sorted(rows):
    r = rows[qi]
    g = lambda k: r.get(k, (None, 0))
    st, rn, wo, hd = g(1), g(2), g(3), g(4)
    f = lambda v: "%8d" % v if v is not None else "       -"
    kind = "-" if wo[0] is None else ("R" if wo[1] & 2 else "T")
    print("%5d w%-2d | %s | %s +%-2d | %s %s | %s" % (qi, qi % W, f(st[0]), f(rn[0]), rn[1], f(wo[0]), kind, f(hd[0])))
