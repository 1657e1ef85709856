"""How should the radix bits of a size be split between the two passes?  (choose_bits: 9 + the rest since round 2's sweep; the kernels have changed since.)
One context per size, unique uniform keys from the device generator, every split of the size's total bits in turn, three rounds; wall clock around 20
synchronous hj_join calls after 5 warm-up calls (the split changes the partition layout: the first call after configure() re-plans)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from hjtest import pkg
P = pkg()
dev = torch.device("cuda:0")
# arguments: sizes "L" (2^L x 2^L unique) or "LR:LS" (2^LR unique x 2^LS foreign keys, every key of R 2^(LS-LR) times)
sizes = sys.argv[1:] or ["25", "26", "27", "28", "29"]
for spec in sizes:
    a, b = spec.split(":") if ":" in spec else (spec, spec)
    l = spec
    nR, n = ((int(x[1:]) if x[0] == "n" else 1 << int(x)) for x in (a, b))   # "27" = 2^27 tuples, "n100000000" = that many
    Rk, Rp = (torch.empty(nR, dtype=torch.int32, device=dev) for _ in range(2))
    Sk, Sp = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(2))
    with P.HashJoin(0) as hj:
        hj.gen_unique(Rk, nR, 0, nR, 1); hj.gen_unique(Sk, n, 0, nR, 2); hj.fill_payload(Rp, nR, "ones"); hj.fill_payload(Sp, n, "ones"); hj.sync()
        hj.bind_device(P.REL_R, Rk, Rp, nR); hj.bind_device(P.REL_S, Sk, Sp, n)
        assert hj.join()[0] == n
        c = hj.config(); total = c["bits1"] + c["bits2"]
        splits = [(b1, total - b1) for b1 in range(9, 5, -1) if 0 < total - b1 <= 9]
        res = {s: [] for s in splits}
        for rnd in range(3):
            for s in splits:
                hj.configure(bits1=s[0], bits2=s[1])
                for _ in range(5): assert hj.join()[0] == n
                t0 = time.perf_counter()
                for _ in range(20): hj.join()
                res[s].append((time.perf_counter() - t0) * 50)
        if os.environ.get("KT"):   # per-kernel times of one instrumented step per split (each kernel alone on the chip)
            for sp in splits:
                hj.configure(bits1=sp[0], bits2=sp[1])
                for _ in range(3): hj.join()
                hj.enable_timings(1); hj.timings_reset()
                for _ in range(5): hj.join()
                tm = hj.timings(); hj.enable_timings(0)
                print("   %d+%d kernels (ms per launch): " % sp + "  ".join("%s %.4f x%d" % (k, v["total_ms"] / v["launches"], v["launches"] // 5) for k, v in tm.items() if v["launches"]), flush=True)
        print("2^%s (default %d+%d): " % (l, c["bits1"], c["bits2"]) + "   ".join("%d+%d %s" % (s[0], s[1], "/".join("%.3f" % x for x in res[s])) for s in splits) + " ms", flush=True)
    del Rk, Rp, Sk, Sp
