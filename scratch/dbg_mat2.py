import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
P = g.load_package()
dev = torch.device("cuda:0")
for logn in (24, 26, 27):
    n = 1 << logn
    Rk, Sk = torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev)
    Rp, Sp = torch.empty_like(Rk), torch.empty_like(Sk)
    with P.HashJoin(0, stream=torch.cuda.current_stream().cuda_stream) as hj:
        hj.gen_unique(Rk, n, 0, n, 1); hj.gen_unique(Sk, n, 0, n, 2)
        hj.fill_payload(Rp, n, "rowid"); hj.fill_payload(Sp, n, "rowid"); hj.sync()
        hj.bind_device(0, Rk, Rp); hj.bind_device(1, Sk, Sp)
        m, agg = hj.join()
        ok, opr, ops = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(3))
        nout = hj.join_materialize_into(ok, opr, ops, n)
        torch.cuda.synchronize()
        inv = torch.empty(n, dtype=torch.int64, device=dev); inv[Sk.long()] = torch.arange(n, device=dev)
        e1 = torch.equal(Rk[opr.long()], ok); e2 = torch.equal(Sk[ops.long()], ok)
        uq = int(torch.unique(opr).numel())
        exp_ps = inv[Rk.long()].int()
        torch.cuda.synchronize()
        d_exp = hj.digest_triples(Rk, Rp, exp_ps, n)
        d_got = hj.digest_triples(ok, opr, ops, n)
        # digest of expected, via outputs permuted back: ops should equal exp_ps[opr]
        e3 = torch.equal(exp_ps[opr.long()], ops)
        d_exp2 = hj.digest_triples(Rk[opr.long()].contiguous(), opr, exp_ps[opr.long()].contiguous(), n)
        print(logn, hj.config()['bits1'], hj.config()['bits2'], m == n, nout == n, 'R[pr]==k', e1, 'S[ps]==k', e2, 'uniq', uq == n, 'ps ok', e3, 'digest', d_exp == d_got, d_exp2 == d_got, flush=True)
