import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
from oracle import pyoracle as o
P = g.load_package()
for logn, cfg in [(21, dict(bits1=5,bits2=4)), (22, None), (23, None), (24, None), (24, dict(bits1=8,bits2=8)), (25, None)]:
    n = 1 << logn
    rng = np.random.default_rng(logn)
    R = rng.permutation(n).astype(np.int32); S = rng.permutation(n).astype(np.int32)
    Pr = np.arange(n, dtype=np.int32)
    with P.HashJoin(0) as hj:
        if cfg: hj.configure(**cfg)
        hj.load_host(0, R, Pr); hj.load_host(1, S, Pr)
        m, agg = hj.join()
        k, pr, ps = hj.join_materialize()
        c = hj.config()
    okR = np.array_equal(R[pr], k); okS = np.array_equal(S[ps], k)
    uniq = len(np.unique(pr)) 
    print(logn, c['bits1'], c['bits2'], 'm', m == n, 'R[pr]==k', okR, 'S[ps]==k', okS, 'unique pr', uniq == n, uniq, flush=True)
    if not (okR and okS and uniq == n):
        bad = np.nonzero(R[pr] != k)[0]
        print('  bad count', len(bad), 'first', bad[:10], 'k', k[bad[:5]], 'pr', pr[bad[:5]])
        # duplicates / missing
        cnt = np.bincount(pr, minlength=n)
        print('  missing', int((cnt == 0).sum()), 'dups', int((cnt > 1).sum()))
        z = np.nonzero(cnt == 0)[0][:10]
        print('  missing rowids', z, 'their keys', R[z], 'part', R[z] & ((1 << (c['bits1']+c['bits2'])) - 1))
