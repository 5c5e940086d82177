#!/usr/bin/env python3
"""Kernel time of a step with a probe library whose results are not meant to be right (FCM_LIB_PATH = a
tools/variant_lib.sh build with -DMW_PROBE=...): no verification, the launch time only.
usage: time_probe.py <config> <chains> [proposals]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flag_complex_mcmc_amd as fcm
from bench import build_workload
cfg, chains = int(sys.argv[1]), int(sys.argv[2])
props = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
n, e = build_workload(fcm, cfg, 1000, 0.10, 0)
s = fcm.initialize_new_sampler(fcm.Graph.from_edges(n, e), n_chains=chains, seed=0)
ms = []
for i in range(3):
    try:
        s.step(props)
    except fcm.FcmError as ex:   # a probe build may trip the consistency checks
        pass
    ms.append(s.last_step_ms())
print("config %d chains %d W %d: %.3f ms per launch, %.4g proposals/s" % (cfg, chains, s.info["waves_per_chain"], min(ms[1:]), chains * props / (min(ms[1:]) * 1e-3)))
