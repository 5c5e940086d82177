import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'oracle'); sys.path.insert(0, 'tests')
import numpy as np
import flag_complex_mcmc_amd as fcm, oracle_ffi as oracle
import importlib.util
spec=importlib.util.spec_from_file_location('tg','tests/test_gpu_parity.py'); tg=importlib.util.module_from_spec(spec); spec.loader.exec_module(tg)
from helpers import setup_pair
for t,pp in [(63,0.04),(64,0.04),(65,0.04),(66,0.04),(67,0.04),(127,0.02),(128,0.02),(130,0.02)]:
    e=tg._book_graph(t,pp,t)
    gg,go,bg,bo=setup_pair(fcm,oracle,t,e,0.3)
    s=fcm.MCMCSampler(gg,bg,n_chains=4,seed=1)
    tw=[oracle.Chain(go,bo,seed=1,chain_id=c) for c in range(4)]
    bad=None
    for it in range(40):
        s.step(100)
        for c in range(4):
            tw[c].step(100)
            if s.flag_count(c)!=tw[c].state.flag_count or int(s.stats()['accepted'][c])!=tw[c].stats()['accepted']:
                bad=(it,c,s.flag_count(c),tw[c].state.flag_count); break
        if bad: break
    print(t, 'kmax',s.info['k_max'], 'BAD' if bad else 'ok', bad)
