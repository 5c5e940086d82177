"""Find the first proposal at which a GPU chain and its oracle twin diverge on config 3."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'oracle'); sys.path.insert(0, 'tests')
import numpy as np
import flag_complex_mcmc_amd as fcm, oracle_ffi as oracle
from helpers import setup_pair
n = 1000
e = fcm.graphs.random_with_p(n, 0.10, 0)
gg, go, bg, bo = setup_pair(fcm, oracle, n, e, 0.01)
NC = int(sys.argv[1]) if len(sys.argv) > 1 else 8
CH = int(sys.argv[2]) if len(sys.argv) > 2 else 256
NG = int(sys.argv[3]) if len(sys.argv) > 3 else NC
s = fcm.MCMCSampler(gg, bg, n_chains=NG, seed=0)
tw = [oracle.Chain(go, bo, seed=0, chain_id=c) for c in range(NC)]
alive = set(range(NC))
for it in range(16384 // CH):
    s.step(CH)
    st = s.stats()
    for c in sorted(alive):
        tw[c].step(CH)
        o = tw[c].stats()
        g = {k: int(st[k][c]) for k in o}
        if s.flag_count(c) != tw[c].state.flag_count or g != o:
            print("chain", c, "diverged in proposals", it * CH, "..", (it + 1) * CH)
            print("  gpu", s.flag_count(c), g)
            print("  cpu", tw[c].state.flag_count, o)
            ge = set(map(tuple, s.edges(c).tolist())); oe = set(map(tuple, tw[c].state.graph_edges().tolist()))
            print("  edge diff gpu-cpu", sorted(ge - oe)[:10], "cpu-gpu", sorted(oe - ge)[:10])
            alive.discard(c)
print("still in parity:", sorted(alive))
