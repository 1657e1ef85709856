#!/usr/bin/env python3
"""bench.py — throughput of the radix-partitioned hash join on MI355X.

A "step" = one full pass of the hot path over one batch of synthetic input already resident in HBM:
radix-partition R, radix-partition S, build+probe (count-only), read the count back
(the reference's timed region, src/hash_join_clustered_probe.cu:881-933 / 953-980).
Workload at N=1: BASELINE.json configs[2] — 2^30 ⋈ 2^30 unique uniform int32 keys, payload = 1.
For N>1 each rank holds 2^30 tuples of R and of S (weak scaling, config 5 shape): level-0 shard
split → all-to-all over xGMI (RCCL) → local partition + build/probe → all-reduce of the count.

Prints ONE JSON line (rank 0).  See DESIGN.md §Measurement for every field.
"""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cpu_baseline(pkg, hj, torch, dev, max_log2n, budget_s=20.0):
    """The oracle's OpenMP radix join ("port") on a bounded sample of the same workload shape, on
    this box's host cores.  Only this leg touches oracle/.  The sample size is calibrated so that the
    join takes about budget_s seconds (at most the full 2^max_log2n workload)."""
    import math
    import psutil
    from oracle import pyoracle as o

    def usable_cpus():
        """Cores this process may actually use: OpenMP's maximum, the affinity mask and the cgroup CPU
        quota (the GPU boxes show 256 logical CPUs under a 16-CPU quota: 128 threads run slower than 16)."""
        n = min(o.max_threads(), len(os.sched_getaffinity(0)))
        try:
            quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]          # cgroup v2
            if quota != "max":
                n = min(n, max(1, math.ceil(int(quota) / int(period))))
        except Exception:
            try:
                q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())           # cgroup v1
                p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, math.ceil(q / p)))
            except Exception:
                pass
        return n

    threads = usable_cpus()

    def sample(log2n):
        n = 1 << log2n
        k = torch.empty(n, dtype=torch.int32, device=dev)
        hj.gen_unique(k, n, 0, n, 1)
        hj.sync()
        R = k.cpu().numpy()
        hj.gen_unique(k, n, 0, n, 2)
        hj.sync()
        S = k.cpu().numpy()
        del k
        return R, S

    best = {}

    def run(log2n):
        R, S = sample(log2n)
        bits = max(0, log2n - 12)
        b2 = bits // 2
        t0 = time.perf_counter()
        m, _ = o.radix_join_omp(R, None, S, None, bits - b2, b2, threads)
        dt = time.perf_counter() - t0
        assert m == (1 << log2n), (m, log2n)
        # beside the port of the reference's scheme: the best this repository's own host code does with the same input on the same
        # threads (hj_host_join: the one-pass block split of the co-processing path, then cache-sized chained tables) — product host
        # code, not the oracle; the count is checked against the known answer, which is also what the GPU step returns
        bm, _, bdt = pkg.host_join(R, None, S, None, threads)
        assert bm == (1 << log2n), (bm, log2n)
        best.update({"value": round(2 * (1 << log2n) / bdt / 1e9, 4), "unit": "billion tuples/s", "cores": threads, "kind": "own host code",
                     "what": "hj_host_join on the same 2^%d x 2^%d input and threads: hj_host_split_blocks (one pass, <= 4096 partitions, "
                             "streaming stores) on both relations, then per partition pair a counting sort into ~1024-tuple pieces and a "
                             "bucket-chained table per piece (partition-primitives.cu:40-125, hash_join_clustered_probe.cu:2013-2059); "
                             "count checked; %.2f s" % (log2n, log2n, bdt)})
        return dt

    dt = run(24)                                   # calibration
    rate = 2 * (1 << 24) / dt
    log2n = 24
    while log2n < max_log2n and 2 * (1 << (log2n + 1)) / rate <= budget_s * 1.5:
        log2n += 1
    while log2n > 24 and 48 * (1 << log2n) > psutil.virtual_memory().available:  # oracle peak ~ 48 B/tuple
        log2n -= 1
    if log2n > 24:
        dt = run(log2n)
    n = 1 << log2n
    return {"value": round(2 * n / dt / 1e9, 4), "unit": "billion tuples/s", "cores": threads, "kind": "port", "best_effort": dict(best),
            "sample": "2^%d x 2^%d unique uniform int32 (the GPU workload's generator and shape%s), oracle "
                      "o_radix_join_omp: two-pass OpenMP radix partition + per-partition chained build/probe, "
                      "%d threads (= min of OpenMP max, affinity mask and cgroup CPU quota), %.1f s" %
                      (log2n, log2n, ", full size" if log2n == max_log2n else ", bounded sample", threads, dt)}


def zipf_cpu_baseline(hj, torch, dev, budget_log2=24):
    """The oracle's OpenMP radix join on a bounded sample of the SAME shape: PK-FK 1:16 with Zipf(1.0) foreign keys from the
    device generator.  Only this leg touches oracle/."""
    import math
    from oracle import pyoracle as o
    nR, nS = 1 << (budget_log2 - 4), 1 << budget_log2
    k = torch.empty(nR, dtype=torch.int32, device=dev)
    hj.gen_unique(k, nR, 0, nR, 3)
    hj.sync()
    R = k.cpu().numpy()
    k = torch.empty(nS, dtype=torch.int32, device=dev)
    hj.gen_zipf(k, nS, 0, nR, 1.0, 4)
    hj.sync()
    S = k.cpu().numpy()
    expect = nS - int((S == nR).sum())
    threads = min(o.max_threads(), len(os.sched_getaffinity(0)))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            threads = min(threads, max(1, math.ceil(int(quota) / int(period))))
    except Exception:
        pass
    bits = max(0, (budget_log2 - 4) - 12)
    t0 = time.perf_counter()
    m, _ = o.radix_join_omp(R, None, S, None, bits - bits // 2, bits // 2, threads)
    dt = time.perf_counter() - t0
    assert m == expect, (m, expect)
    return {"value": round((nR + nS) / dt / 1e9, 4), "unit": "billion tuples/s", "cores": threads, "kind": "port", "cpu_model": cpu_model(),
            "sample": "PK-FK 2^%d x 2^%d, Zipf(1.0) foreign keys (the GPU workload's generators and 1:16 shape, bounded sample), oracle "
                      "o_radix_join_omp on %d threads, %.1f s" % (budget_log2 - 4, budget_log2, threads, dt)}


def bench_zipf(a, pkg, torch, dev, local):
    """BASELINE configs[3]: PK-FK 2^27 x 2^31, Zipf(1.0) foreign keys, one GPU.  Reported in DESIGN.md; not the headline line.
    The first join on a fresh binding is timed on its own: it finds S's slots overflowing, samples the key distribution and
    builds the capacity tables; the steady-state steps (what `value` is) reuse them."""
    nR, nS = 1 << a.zipf_sizes[0], 1 << a.zipf_sizes[1]
    hj = pkg.HashJoin(local, stream=torch.cuda.current_stream().cuda_stream)
    if a.probe_chunk or a.exact_only or a.build_side or a.bits:
        hj.configure(probe_chunk=a.probe_chunk, exact_only=a.exact_only, build_side=a.build_side,
                     bits1=a.bits[0] if a.bits else 0, bits2=a.bits[1] if a.bits else 0)
    Rk, Rp = (torch.empty(nR, dtype=torch.int32, device=dev) for _ in range(2))
    Sk, Sp = (torch.empty(nS, dtype=torch.int32, device=dev) for _ in range(2))
    hj.gen_unique(Rk, nR, 0, nR, 3)
    hj.gen_zipf(Sk, nS, 0, nR, a.zipf_theta, 4)
    hj.fill_payload(Rp, nR, "ones")
    hj.fill_payload(Sp, nS, "ones")
    hj.sync()
    expect = nS - int((Sk == nR).sum().item())
    hj.bind_device(pkg.REL_R, Rk, Rp)
    hj.bind_device(pkg.REL_S, Sk, Sp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    assert hj.join()[0] == expect          # first contact: optimistic attempt -> overflow -> sample -> tables -> sampled passes -> join
    torch.cuda.synchronize()
    first_ms = (time.perf_counter() - t0) * 1e3
    first_split = hj.last_call_breakdown()   # allocation / failed optimistic attempt / sample + plan / the step that answered
    for _ in range(a.warmup):
        assert hj.join()[0] == expect
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        got = hj.join()[0]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert got == expect
    hot = hj.hot_stats()   # the heavy-hitter bypass of the timed steps: keys, sampled share of S, matches pass 1 counted itself
    layout = [hj.partition_layout(pkg.REL_R), hj.partition_layout(pkg.REL_S)]
    # instrumented steps (kernel events on), outside the timed region
    hj.enable_timings(1)
    hj.timings_reset()
    isteps = max(1, min(a.steps, 3))
    for _ in range(isteps):
        assert hj.join()[0] == expect
    kt = hj.timings()
    hj.enable_timings(0)
    kernels = {k: {"launches_per_step": v["launches"] / isteps, "ms_per_step": round(v["total_ms"] / isteps, 4)} for k, v in kt.items() if v["launches"]}
    # roofline of the dominant kernel: a radix pass over S = 2^31 tuples, 16 algorithmic bytes per tuple
    passes = [k for k in kt if k.startswith("k_part") or k.startswith("k_scatter")]
    dom = max(passes, key=lambda k: kt[k]["total_ms"])
    avg_ms = kt[dom]["total_ms"] / kt[dom]["launches"]
    # launches of that kernel per step: the sampled passes run over S only; the exact scatter runs twice over R and twice over S
    tuples = nS if dom in ("k_part1_var", "k_part2_var") else ((nR + nS) / 2.0 if dom == "k_scatter_wc" else nR)
    # The heavy-hitter bypass (round 6): pass 1 READS every tuple of S but writes only those it does not join itself; pass 2 never sees the
    # others.  `achieved` prices the bytes the launch really moves (so the fraction cannot pass 1); the pass's nominal 16 B x |S| over the
    # same time is given beside it as what the kernel is worth to the step.
    bypassed = float(hot["matches"]) if (hot["mode"] == 1 and dom in ("k_part1_var", "k_part2_var")) else 0.0
    read_b = 8.0 * (tuples - (bypassed if dom == "k_part2_var" else 0.0))
    write_b = 8.0 * (tuples - bypassed)
    achieved = (read_b + write_b) / (avg_ms * 1e-3) / 1e9
    traffic = None
    try:
        pmf = json.load(open(os.path.join(ROOT, "profiles", "r6_pmc_zipf.json")))
        key = [k for k in pmf["kernels"] if k.startswith("hj::" + dom + "<")]
        if key and pmf.get("lib_sha256") == lib_sha256():
            traffic = pmf["kernels"][key[0]]["hbm_bytes_per_launch"]
    except Exception:
        pass
    roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "avg_launch_ms": round(avg_ms, 4),
            "algorithmic_bytes_per_launch": read_b + write_b,
            "algorithmic_bytes_without_bypass": 16.0 * tuples, "tuples_the_bypass_took": int(bypassed),
            "equivalent_GBs_at_16B_per_tuple": round(16.0 * tuples / (avg_ms * 1e-3) / 1e9, 1)}
    if not a.no_extras:
        nub = min(nS, 1 << 30)
        tk, tp = (torch.empty(nub, dtype=torch.int32, device=dev) for _ in range(2))
        roof.update(mix_ceiling(hj, Sk, Sp, tk, tp, nub, read_b, write_b, achieved))
        del tk, tp
    jc = kt.get("k_join_count", {"launches": 0, "total_ms": 0.0})
    probe = None
    if jc["launches"]:
        avg = jc["total_ms"] / jc["launches"]
        left = nS - (float(hot["matches"]) if hot["mode"] == 1 else 0.0)   # the probe only sees what pass 1 did not join itself
        probe = {"kernel": "k_join_count", "avg_launch_ms": round(avg, 4), "achieved_GBs": round(8.0 * (nR + left) / (avg * 1e-3) / 1e9, 1),
                 "frac_of_8TBs": round(8.0 * (nR + left) / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "probe_side_tuples": int(left)}
    # the materialising variant (north_star: count AND materialised (key,payR,payS) tuples; the reference's lead timed run,
    # hjcp.cu:913,937-940): partition both, then ONE probe writing ~2^31 output tuples (24 GiB) — the sampled probe side makes
    # the items LIST items (k_join_mat_reg<.,LISTS>)
    mat = None
    if not a.no_materialize:
        cap = expect
        ok, opr, ops = (torch.empty(cap, dtype=torch.int32, device=dev) for _ in range(3))

        def mat_step():   # partition both + ONE probe in one call: pass 1 of the skewed side writes its heavy hitters' tuples itself
            return hj.join_and_materialize_into(ok, opr, ops, cap)

        assert mat_step() == expect   # warm-up (first touch of the output columns)
        hot_mat = hj.hot_stats()
        torch.cuda.synchronize()
        reps = max(1, a.steps // 2)
        t0 = time.perf_counter()
        for _ in range(reps):
            nout = mat_step()
        torch.cuda.synchronize()
        dtm = (time.perf_counter() - t0) / reps
        assert nout == expect
        # full-size property check, outside the timed region: R's keys are unique and every payload is 1, so the output multiset
        # is {(k,1,1) : S tuples whose key occurs in R} = all of S except the tuples with key nR (the generator's alphabet is
        # 1..nR, R holds 0..nR-1); the digest is a sum of per-tuple mixes mod 2^64, so the missing tuples are subtracted
        miss = nS - expect
        want = (hj.digest_triples(Sk, Sp, Sp, nS) - miss * mix_triple(nR, 1, 1)) % (1 << 64)
        assert hj.digest_triples(ok, opr, ops, nout) == want, "materialised output digest"
        hj.enable_timings(2)
        hj.timings_reset()
        assert mat_step() == expect
        km = hj.timings()
        hj.enable_timings(0)
        mk = km.get("k_join_materialize", {"launches": 0, "total_ms": 0.0})
        mat = {"value": round((nR + nS) / dtm / 1e9, 3), "unit": "billion tuples/s", "ms_per_step": round(dtm * 1e3, 3),
               "output_tuples": int(nout), "digest_checked": True, "heavy_hitter_bypass": hot_mat,
               "probes_per_step": sum(v["launches"] for k, v in km.items() if k.startswith("k_join_count") or k.startswith("k_join_mat")),
               "launches_of_one_step": {k: v["launches"] for k, v in km.items() if v["launches"]},
               "kernel_ms_of_one_step": {k: round(v["total_ms"], 4) for k, v in km.items() if v["launches"] and v["total_ms"] > 0.05}}
        if mk["launches"]:
            avg = mk["total_ms"] / mk["launches"]
            # (the probe neither reads nor writes the tuples pass 1 wrote itself: the same tuples it counts itself in a count-only step)
            hot_n = float(hot["matches"]) if (hot_mat["mode"] == 2 and hot["mode"] == 1) else 0.0
            rd_b, wr_b = 8.0 * (nR + nS - hot_n), 12.0 * (nout - hot_n)
            gbs = (rd_b + wr_b) / (avg * 1e-3) / 1e9
            mat.update({"k_join_materialize_ms": round(avg, 4), "k_join_materialize_GBs": round(gbs, 1),
                        "k_join_materialize_frac_of_8TBs": round(gbs / HBM_PEAK_GBS, 4),
                        "algorithmic_bytes_per_launch": rd_b + wr_b, "tuples_written_by_pass_1": int(hot_n)})
            mat.update(mix_ceiling(hj, Sk, Sp, ok, opr, min(nS, nout), rd_b, wr_b, gbs))
        del ok, opr, ops
    cpu = None if a.no_cpu_baseline else zipf_cpu_baseline(hj, torch, dev)
    print(json.dumps({"metric": "billion tuples/sec (build+probe), PK-FK 2^%d x 2^%d Zipf theta=%.1f, 1 GPU%s" % (a.zipf_sizes[0], a.zipf_sizes[1], a.zipf_theta, ", the Zipf side BUILDS" if a.build_side == 2 else ""),
                      "value": round((nR + nS) * a.steps / dt / 1e9, 3), "unit": "billion tuples/s", "n_gpus": 1,
                      "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
                      "dtype": "int32", "data": "synthetic", "vs_baseline": None,
                      "first_call_ms": round(first_ms, 3), "first_call_split_ms": first_split,
                      "first_call": "optimistic histogram-free attempt on S (overflows) + sampling pass + host-side capacity tables + the step itself; "
                                    "later steps on the same binding reuse the tables",
                      "config": {"workload": "PK-FK 2^%d x 2^%d, Zipf(1.0) foreign keys (device generator), payload=1, count-only" % tuple(a.zipf_sizes),
                                 "build_side": hj.config()["build_side"],
                                 "matches": int(got), "radix_bits": [hj.config()["bits1"], hj.config()["bits2"]],
                                 "partition_layout_R_S": layout, "heavy_hitter_bypass": hot},
                      "roofline": roof, "probe_phase": probe, "kernels": kernels, "materialize": mat, "cpu_baseline": cpu, "lib_sha256": lib_sha256()}))


def bench_baselines(a, pkg, torch, dev, local):
    """the partitioned join against the reference's non-partitioned baselines
    (perfect array jp.cu:628-668, global chained table jp.cu:681-742) on the same unique uniform input."""
    n = 1 << a.log2n
    hj = pkg.HashJoin(local, stream=torch.cuda.current_stream().cuda_stream)
    Rk, Rp, Sk, Sp = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(4))
    hj.gen_unique(Rk, n, 0, n, 1)
    hj.gen_unique(Sk, n, 0, n, 2)
    hj.fill_payload(Rp, n, "ones")
    hj.fill_payload(Sp, n, "ones")
    hj.sync()
    hj.bind_device(pkg.REL_R, Rk, Rp)
    hj.bind_device(pkg.REL_S, Sk, Sp)
    out = {}
    for name, fn in (("partitioned (radix + LDS tables)", hj.join), ("perfect array", lambda: hj.join_nonpartitioned(0)),
                     ("global chained table", lambda: hj.join_nonpartitioned(1))):
        for _ in range(a.warmup):
            assert fn()[0] == n
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            assert fn()[0] == n
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        out[name] = {"ms": round(dt * 1e3, 3), "Gtuples_per_s": round(2 * n / dt / 1e9, 2)}
    print(json.dumps({"metric": "partitioned vs non-partitioned join, 2^%d x 2^%d unique uniform int32, 1 GPU" % (a.log2n, a.log2n),
                      "results": out, "lib_sha256": lib_sha256()}))


def bench_stream(a, pkg, torch, dev, local):
    """SURVEY §8(f) rank 1 (outOfGPU_Join3_payload): R = 2^27 resident, S = 2^30 in pinned HOST memory, streamed
    through HBM in segments.  PCIe-bound by construction; reported in DESIGN.md, not the headline."""
    import numpy as np
    nR, nS = 1 << 27, 1 << 30
    hj = pkg.HashJoin(local, stream=torch.cuda.current_stream().cuda_stream)
    Rk, Rp = (torch.empty(nR, dtype=torch.int32, device=dev) for _ in range(2))
    Sk = torch.empty(nS, dtype=torch.int32, device=dev)
    hj.gen_unique(Rk, nR, 0, nR, 3)
    hj.fill_payload(Rp, nR, "ones")
    hj.gen_unique(Sk, nS, 0, nR, 5)   # foreign keys: every R key 8 times
    hj.sync()
    S_host = torch.empty(nS, dtype=torch.int32).pin_memory()
    S_host.copy_(Sk)
    del Sk
    S_np = S_host.numpy()
    hj.bind_device(pkg.REL_R, Rk, Rp)
    times = []
    for i in range(a.warmup + a.steps):
        t0 = time.perf_counter()
        m, _ = hj.join_stream_probe(S_np, None, "ones")
        torch.cuda.synchronize()
        if i >= a.warmup:
            times.append(time.perf_counter() - t0)
        assert m == nS, (m, nS)
    dt = sum(times) / len(times)
    # the materialising form (hjcp.cu:1917-1961): every segment's (key, payR, payS) tuples go back to pinned host columns on a third
    # stream; one probe per segment, the host reads a segment's output size one segment later.  A quarter of S: 2^28 output tuples.
    mat = None
    if not a.no_materialize:
        nM = nS // 4
        outs = [torch.empty(nM, dtype=torch.int32).pin_memory() for _ in range(3)]
        outs_np = [x.numpy() for x in outs]
        mt = []
        for i in range(1 + max(1, a.steps // 2)):
            t0 = time.perf_counter()
            (k, pr, ps), _ = hj.join_stream_probe_materialize(S_np[:nM], None, "ones", cap=nM, out=outs_np)
            torch.cuda.synchronize()
            if i:
                mt.append(time.perf_counter() - t0)
            assert len(k) == nM
        dm = sum(mt) / len(mt)
        mat = {"probe_tuples": nM, "output_tuples": nM, "ms": round(dm * 1e3, 2), "value": round((nR + nM) / dm / 1e9, 3), "unit": "billion tuples/s",
               "h2d_GBs": round(nM * 4 / dm / 1e9, 1), "d2h_GBs": round(nM * 12 / dm / 1e9, 1)}
    print(json.dumps({"metric": "billion tuples/sec, streaming probe side: R 2^27 in HBM, S 2^30 in pinned host memory",
                      "value": round((nR + nS) / dt / 1e9, 3), "unit": "billion tuples/s", "n_gpus": 1,
                      "ms_per_step": round(dt * 1e3, 2), "h2d_GBs": round(nS * 4 / dt / 1e9, 1), "materialize": mat,
                      "config": {"workload": "PK-FK 2^27 x 2^30, S streamed from pinned host memory in segments of max(|R|/4, 2^24)"},
                      "lib_sha256": lib_sha256()}))


def bench_coprocess(a, pkg, torch, dev, local):
    """SURVEY §8(f) rank 2 (outOfGPU_Join2_payload): both relations in HOST memory; host level-0 split (software
    write-combining, non-temporal stores) + per-partition upload and GPU join.  Host- and PCIe-bound by construction."""
    n = 1 << min(a.log2n, 28)
    hj = pkg.HashJoin(local, stream=torch.cuda.current_stream().cuda_stream)
    k = torch.empty(n, dtype=torch.int32, device=dev)
    hj.gen_unique(k, n, 0, n, 1)
    hj.sync()
    R = k.cpu().numpy()
    hj.gen_unique(k, n, 0, n, 2)
    hj.sync()
    S = k.cpu().numpy()
    del k
    times, gbs = [], []
    for i in range(a.warmup + a.steps):
        t0 = time.perf_counter()
        m, _ = hj.join_coprocess(R, None, S, None)
        dt = time.perf_counter() - t0
        assert m == n, (m, n)
        if i >= a.warmup:
            times.append(dt)
            gbs.append(hj.host_split_throughput())
    dt = sorted(times)[len(times) // 2]   # the median call: the host is shared, its neighbours move a call by +-20 %
    print(json.dumps({"metric": "billion tuples/sec, CPU-GPU co-processing: R and S (2^%d each) in host memory" % (n.bit_length() - 1),
                      "value": round(2 * n / dt / 1e9, 3), "unit": "billion tuples/s", "n_gpus": 1, "ms_per_step": round(dt * 1e3, 2),
                      "ms_of_every_call": [round(t * 1e3, 2) for t in times], "value_is": "median call",
                      "host_split_GBs": round(sorted(gbs)[len(gbs) // 2], 2), "host_split_GBs_of_every_call": [round(g, 1) for g in gbs],
                      "cpu_model": cpu_model(),
                      "numa": dict(zip(("nodes", "gpu_node", "workers_bound_to_cpus_of_that_node"), hj.coprocess_numa())),
                      "config": {"workload": "unique uniform int32, 16 level-0 partitions, one-pass host split on the box's CPU quota, uploads "
                                             "beside the split, one residency group, one GPU join"}, "lib_sha256": lib_sha256()}))


def launch_ranks(n):
    """Re-run this script under torch.distributed.run with n ranks on this node (child process, never an exec:
    this process has not touched the GPU and stays alive to forward the result)."""
    import socket
    import subprocess
    with socket.socket() as sk:                      # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # HSA_ENABLE_IPC_MODE_LEGACY=0: the host driver of this pool only supports dmabuf IPC; with the legacy mode RCCL's
    # peer-buffer registration fails in every rank with `hipIpcGetMemHandle: invalid argument` (the image exports the
    # variable already; it is set here only when the caller's environment does not have it, never overridden)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    if p.returncode != 0 or not lines:
        sys.stderr.write(p.stdout[-4000:] + p.stderr[-8000:])
        sys.exit(p.returncode or 1)
    print(lines[-1])
    sys.exit(0)


# Stated probe-phase targets (north_star: "probe-phase achieved HBM bandwidth >= a stated fraction of the MI355X roofline on
# 2^30 x 2^30"): k_join_count's 8 B x (|R|+|S|) over its launch time, as a fraction of 8 TB/s.  Round 3 raises the 2^30 target
# from 0.60 to 0.70 (measured 0.72-0.74) and states one for the smaller configurations, where the launch is too short to
# amortise its ramp-up and drain (2^27: measured 0.60-0.63, target 0.68 NOT met).
PROBE_TARGET_FRAC = 0.70
# Round 5 (profiles/r5_launch_structure_ab.txt): below 2^30 the target is the 2^30 target diluted by a FIXED cost of 60 us per launch —
# the ramp-up and tail of a short kernel, measured as k_join_count 0.429 ms at 2^27 against 2.78 / 8 = 0.348 ms; removing launches from
# the step (14 -> 6) did not move it.  At 2^27: 0.70 x 0.3835 / (0.3835 + 0.060) = 0.605.
# Round 6 (profiles/r6_fixed_cost_2p27.txt): per-workgroup timelines put the fixed cost at 20-25 us per launch (first wave 7-10 us slower than a
# steady workgroup, 4-11 us of tail, ~8 us between the start event and the first workgroup's work); the rest of round 5's 60 us was the
# steady rate of the full-key tables below 16 radix bits, which the tag kernels now replace.  At 2^27: 0.70 x 0.3835 / (0.3835 + 0.025) = 0.657.
PROBE_FIXED_COST_MS = 0.025


def probe_target_frac(log2n):
    if log2n >= 30:
        return PROBE_TARGET_FRAC
    t_ms = 8.0 * 2 * (1 << log2n) / (PROBE_TARGET_FRAC * HBM_PEAK_GBS * 1e9) * 1e3
    return round(PROBE_TARGET_FRAC * t_ms / (t_ms + PROBE_FIXED_COST_MS), 4)


def mix_ceiling(hj, in_k, in_p, out_k, out_p, n, read_bytes, write_bytes, achieved_gbs):
    """Same-run ceiling of a kernel that reads read_bytes and writes write_bytes: hj_ubench streams two columns in only and out only
    (what this box's HBM gives pure reads / pure writes of 16 B per lane); no kernel with that mix can beat
    (R + W) / (R / read_rate + W / write_rate)."""
    rd = hj.ubench("read", in_k, in_p, out_k, out_p, n)
    wr = hj.ubench("write", in_k, in_p, out_k, out_p, n)
    ceil = (read_bytes + write_bytes) / (read_bytes / rd + write_bytes / wr)
    return {"read_only_ceiling_GBs": round(rd, 1), "write_only_ceiling_GBs": round(wr, 1), "write_share_of_bytes": round(write_bytes / (read_bytes + write_bytes), 3),
            "mix_ceiling_GBs": round(ceil, 1), "frac_of_mix_ceiling": round(achieved_gbs / ceil, 4)}


def _fmix64(x):
    m = (1 << 64) - 1
    x ^= x >> 33
    x = (x * 0xff51afd7ed558ccd) & m
    x ^= x >> 33
    x = (x * 0xc4ceb9fe1a85ec53) & m
    x ^= x >> 33
    return x


def mix_triple(key, pr, ps):
    """hj_digest_triples' per-tuple mix (csrc/hj_device.h: mix_triple), for digest arithmetic on the host."""
    m = (1 << 64) - 1
    pair = _fmix64((((key & 0xFFFFFFFF) << 32) | (pr & 0xFFFFFFFF)) & m)
    return _fmix64(pair ^ (((ps & 0xFFFFFFFF) * 0x9E3779B97F4A7C15) & m))


def lib_sha256():
    """sha256 of the libhj.so this process runs: ties a bench line to the profiles collected from the same binary."""
    import hashlib
    try:
        return hashlib.sha256(open(os.path.join(ROOT, "icde2019-gpu-join_amd", "libhj.so"), "rb").read()).hexdigest()
    except Exception:
        return None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def join_cpu_baseline(hj, torch, dev, threads, log2n=22):
    """The reference's own CPU join, restated (oracle o_joinCpu = joinCpu + h_hashMurmur, hjcp.cu:2013-2059: one
    2^20-slot chained table, serial build, OpenMP probe), at a size where it is meaningful (chains of 2^log2n / 2^20)."""
    from oracle import pyoracle as o
    n = 1 << log2n
    k = torch.empty(n, dtype=torch.int32, device=dev)
    hj.gen_unique(k, n, 0, n, 1)
    hj.sync()
    R = k.cpu().numpy()
    hj.gen_unique(k, n, 0, n, 2)
    hj.sync()
    S = k.cpu().numpy()
    t0 = time.perf_counter()
    m, _ = o.joinCpu(R, S, threads=threads)
    dt = time.perf_counter() - t0
    assert m == n, (m, n)
    return {"value": round(2 * n / dt / 1e9, 4), "unit": "billion tuples/s", "cores": threads, "kind": "port",
            "sample": "2^%d x 2^%d unique uniform int32, oracle o_joinCpu (restatement of the reference's joinCpu, "
                      "hash_join_clustered_probe.cu:2013-2059: 2^20-slot chained table, serial build, %d-thread probe), %.2f s"
                      % (log2n, log2n, threads, dt)}


def dist_materialize_leg(a, pkg, torch, dist, hj, dj, cols, n, world, rank, expect, dup, cdev, barrier):
    """N > 1: the sharded MATERIALISING join (hj_dist_rank_join_materialize): every rank writes the (key, payR, payS) tuples of the
    partitions it owns into its own device columns; timed like the headline (barrier + synchronize on both sides, max over ranks).
    Output columns sized at the even share + 10 % (hash sharding of uniform keys is even to a fraction of a percent); a rank whose
    share does not fit makes every rank raise (HJ_ECAPACITY).  Full-size property check outside the timed region: the sum over the
    ranks of the order-independent digest of each share equals the digest of {(k,1,1) : k in R} (unique keys, payloads 1)."""
    Rk, Rp, Sk, Sp = cols
    G = a.phantom if (world == 1 and a.phantom > 1) else world
    share = expect if world == 1 else (expect // world + expect // (10 * world) + 4096)
    dev = Rk.device
    ok, opr, ops = (torch.empty(share, dtype=torch.int32, device=dev) for _ in range(3))

    def mstep():
        return dj.join_materialize(Rk, Rp, Sk, Sp, ok, opr, ops, cap=share)

    m, _, nloc, nall = mstep()                       # warm-up: first touch of the output columns
    assert m == expect and sum(nall) == expect, (m, expect, nall)
    barrier()
    reps = max(1, a.steps // 2)
    t0 = time.perf_counter()
    for _ in range(reps):
        m, _, nloc, nall = mstep()
    barrier()
    dtm = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dtm], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dtm = float(t.item())
    dtm /= reps
    assert m == expect and sum(nall) == expect and nall[rank] == nloc
    digest_checked = False
    if dup == 1:
        got_d, want_d = hj.digest_triples(ok, opr, ops, nloc), hj.digest_triples(Rk, Rp, Sp, n)
        if world > 1:
            both = [None] * world
            dist.all_gather_object(both, (got_d, want_d))
            got_d, want_d = sum(x for x, _ in both) % (1 << 64), sum(y for _, y in both) % (1 << 64)
        assert got_d == want_d, "materialised output digest (union of the ranks' shares)"
        digest_checked = True
    st = dj.stats()
    out = {"value": round(2.0 * n * world / dtm / 1e9, 3), "unit": "billion tuples/s", "ms_per_step": round(dtm * 1e3, 3), "scaling": "weak",
           "output_tuples_total": int(m), "output_tuples_per_rank": [int(x) for x in nall], "output_capacity_per_rank": int(share),
           "digest_checked": digest_checked, "path": st["path"], "probe_groups": st["probe_groups"],
           "output": "sharded: every GPU keeps the (key,payR,payS) tuples of the partitions it owns (SURVEY §8(e)); no tuple crosses a link twice",
           "rank0_stage_ms": {k: st[k] for k in ("split_ms", "pass1_ms", "pass2_join_ms", "early_pass2_join_ms", "first_split_ms", "last_pass1_ms", "exchange_ms", "wall_ms")}}
    if st["path"] == "sliced" and G > 1:
        LINK_GBS = 76.8
        link_ms = st["link_bytes"] / (G - 1) / (LINK_GBS * 1e9) * 1e3
        local_ms = sum(st["split_ms"]) + sum(st["pass1_ms"]) + st["pass2_join_ms"] + st["early_pass2_join_ms"]
        exposed = st["first_split_ms"] + st["last_pass1_ms"] + st["pass2_join_ms"]
        out["model"] = {"gpus": G, "phantom": bool(world == 1), "link_ms": round(link_ms, 3), "local_ms_total": round(local_ms, 3),
                        "exposed_local_ms": round(exposed, 3), "modelled_step_ms": round(max(link_ms, local_ms - exposed) + exposed, 3),
                        "modelled_Gtuples_per_s_per_gpu": round(2.0 * n / ((max(link_ms, local_ms - exposed) + exposed) * 1e-3) / 1e9, 2),
                        "note": "as dist.model, with the materialising probe in place of the count (the earlier probe-side group writes its "
                                "output under the exchange, the last group in the tail)"}
    del ok, opr, ops
    return out


def strong_leg(a, pkg, torch, dist, hj, dj, n, world, rank, cdev, barrier):
    """N > 1: the strong-scaling point of BASELINE.json's metric read as "2^30 ⋈ 2^30 over 1/2/4/8 GPUs": 2^log2n tuples per relation
    in TOTAL, 1/N of each on every GPU (slices of the same two permutations of [0, 2^log2n) the N=1 run joins)."""
    ns = n // world
    dev = torch.device("cuda", torch.cuda.current_device())
    Rk, Rp, Sk, Sp = (torch.empty(ns, dtype=torch.int32, device=dev) for _ in range(4))
    hj.gen_unique(Rk, ns, rank * ns, n, 1)
    hj.gen_unique(Sk, ns, rank * ns, n, 2)
    hj.fill_payload(Rp, ns, "ones")
    hj.fill_payload(Sp, ns, "ones")
    hj.sync()
    for _ in range(max(1, a.warmup)):
        assert dj.join(Rk, Rp, Sk, Sp)[0] == n
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        got = dj.join(Rk, Rp, Sk, Sp)[0]
    barrier()
    dt = time.perf_counter() - t0
    assert got == n
    t = torch.tensor([dt], dtype=torch.float64, device=cdev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    st = dj.stats()
    return {"value": round(2.0 * n * a.steps / dt / 1e9, 3), "unit": "billion tuples/s", "ms_per_step": round(dt / a.steps * 1e3, 3),
            "scaling": "strong", "steps": a.steps, "tuples_per_relation_total": n, "tuples_per_relation_per_gpu": ns, "matches": int(got),
            "path": st["path"], "slices": st["slices"], "exchange_ms": round(st["exchange_ms"], 3),
            "workload": "2^%d ⋈ 2^%d unique uniform int32 in total, 1/%d of each relation per GPU, count-only" % (a.log2n, a.log2n, world)}


def alt_transport_leg(a, pkg, torch, world, n, domain, expect):
    """N > 1, rank 0 only, after every other rank has released its GPU: the SAME workload driven by ONE process (hj_dist: one host
    thread per GPU) over both transports of one group — RCCL's send/recv kernels, then the copy engines (hipMemcpyPeerAsync over
    xGMI; hj_dist_set_transport keeps contexts, inputs and buffers) — so that the first multi-GPU session answers DESIGN §7's two open
    questions in one run: what RCCL's kernels cost beside 155-KiB-LDS pass workgroups, and what a link sustains at these message
    sizes.  Recorded under dist.alt_transport; the headline stays the one-process-per-GPU RCCL run above."""
    from importlib import import_module
    D = import_module(pkg.__name__ + ".dist")
    out = {"driver": "hj_dist (ONE process, one host thread per GPU; inputs regenerated per GPU with the headline's generator and seeds)"}
    slices_list = [a.slices] + ([8] if os.environ.get("HJ_BENCH_SLICE_SWEEP") and a.slices != 8 else [])
    with D.GroupJoin(list(range(world)), transport="rccl") as g:
        keep = []
        for r in range(world):
            dev = torch.device("cuda", r)
            cols = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(4)]
            hj = g.context(r)
            hj.gen_unique(cols[0], n, r * n, domain, 1)
            hj.gen_unique(cols[2], n, r * n, domain, 2)
            hj.fill_payload(cols[1], n, "ones")
            hj.fill_payload(cols[3], n, "ones")
            hj.sync()
            keep.append(cols)
            g.bind(r, pkg.REL_R, cols[0], cols[1])
            g.bind(r, pkg.REL_S, cols[2], cols[3])
        for transport in ("rccl", "copy"):
            for sl in slices_list:
                g.set_transport(transport)
                g.configure(slices=sl, exact_only=a.exact_only, single_group=a.single_group)
                for _ in range(max(1, a.warmup)):
                    assert g.join()[0] == expect
                t0 = time.perf_counter()
                for _ in range(a.steps):
                    got = g.join()[0]
                dt = (time.perf_counter() - t0) / a.steps
                assert got == expect
                st = g.stats(0)
                per_peer = st["link_bytes"] / max(1, world - 1)
                key = g.transport + ("" if sl == a.slices else "_slices%d" % sl)
                out[key] = {"ms_per_step": round(dt * 1e3, 3), "value": round(2.0 * n * world / dt / 1e9, 3), "unit": "billion tuples/s",
                            "path": st["path"], "slices": st["slices"], "exchange_ms": round(st["exchange_ms"], 3),
                            "link_GBs_per_direction": round(per_peer / (st["exchange_ms"] * 1e-3) / 1e9, 2) if st["exchange_ms"] > 0 else None,
                            "rank0_stage_ms": {k: st[k] for k in ("split_ms", "pass1_ms", "pass2_join_ms", "early_pass2_join_ms", "first_split_ms", "last_pass1_ms")}}
        del keep
    return out


# main() is a sequence of legs over one namespace B (round 6: one function per leg, no behaviour change).  A leg takes what it reads
# from B into locals and keeps what later legs read (a name a leg did not assign — e.g. the N > 1 fields at N = 1 — is left alone).
def _take(B, *names):
    return [getattr(B, x, None) for x in names]


def _keep(B, loc, *names):
    for x in names:
        if x in loc:
            setattr(B, x, loc[x])


def leg_inputs_and_driver(B):
    """the context, the synthetic inputs of this rank and (N > 1) the multi-GPU driver; step() = one pass of the hot path, barrier()"""
    a, backend, cdev, dev, dist, domain, local, n, pkg, rank = _take(B, "a", "backend", "cdev", "dev", "dist", "domain", "local", "n", "pkg", "rank")
    torch, total_n, use_dist, world = _take(B, "torch", "total_n", "use_dist", "world")
    if use_dist:
        # a stream of our own for N>1: the all-to-alls run asynchronously next to local kernels, and HIP's
        # legacy default stream would add implicit synchronisation with other blocking streams
        torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    hj = pkg.HashJoin(local, stream=torch.cuda.current_stream().cuda_stream)
    if a.bits or a.probe_chunk or a.lds or a.exact_only:
        hj.configure(bits1=a.bits[0] if a.bits else 0, bits2=a.bits[1] if a.bits else 0, probe_chunk=a.probe_chunk,
                     lds_capacity=a.lds[0] if a.lds else 0, lds_heads=a.lds[1] if a.lds else 0, exact_only=a.exact_only)
    # inputs: rank r holds slice r of two independent pseudo-random permutations of the global key
    # domain [0, min(total_n, 2^32)) (beyond 2^32 tuples keys repeat: int32 keys cannot be unique)
    domain = min(total_n, 1 << 32)
    Rk = torch.empty(n, dtype=torch.int32, device=dev)
    Sk = torch.empty(n, dtype=torch.int32, device=dev)
    Rp = torch.empty(n, dtype=torch.int32, device=dev)
    Sp = torch.empty(n, dtype=torch.int32, device=dev)
    hj.gen_unique(Rk, n, rank * n, domain, 1)
    hj.gen_unique(Sk, n, rank * n, domain, 2)
    hj.fill_payload(Rp, n, "ones")
    hj.fill_payload(Sp, n, "ones")
    hj.sync()
    dup = max(1, total_n // domain)
    expect = total_n * dup  # every key occurs dup times in R and in S

    dj = None
    c_impl = False
    if use_dist:
        from importlib import import_module
        dmod = import_module(pkg.__name__ + ".dist")
        c_impl = a.dist_impl == "c" and backend != "gloo" and a.balance == "hash"
        dist_fallback = None
        if c_impl:
            # the exchange behind the C ABI: C++ host code over RCCL (ncclCommInitRank with an id broadcast over the control plane).
            # If the communicator cannot be made (every rank agrees on that through an all-reduce), the torch.distributed driver of
            # rounds 1-2 takes over and the line says so.
            try:
                dj = dmod.RankJoin(hj, rank, world)
                dj.configure(slices=a.slices, exact_only=a.exact_only, phantom_world=a.phantom if world == 1 else 0, single_group=a.single_group)
                ok = 1
            except Exception as e:   # noqa: BLE001
                ok, dist_fallback = 0, repr(e)
            t_ok = torch.tensor([ok], dtype=torch.int32, device=cdev)
            dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)
            if int(t_ok.item()) == 0:
                c_impl, dj = False, None
                dist_fallback = dist_fallback or "another rank could not create its hj_dist_rank"
        if not c_impl:
            dj = dmod.ShardedJoin(hj, pkg, dev, balance=a.balance)
            dj.force_exchange = a.force_dist

    def step(verify=False):
        if not use_dist:
            hj.bind_device(pkg.REL_R, Rk, Rp)
            hj.bind_device(pkg.REL_S, Sk, Sp)
            return hj.join()[0]
        # verify: the digest of everything sent must equal the digest of everything received (dist.ShardedJoin.join)
        return dj.join(Rk, Rp, Sk, Sp, verify=verify)[0]

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
    _keep(B, locals(), "Rk", "Rp", "Sk", "Sp", "barrier", "c_impl", "dist_fallback", "dj", "dup", "expect", "hj", "step")


def leg_timed_headline(B):
    """W warm-up steps (each checks the exchange), then EXACTLY K timed steps between barriers: value / ms_per_step"""
    a, barrier, cdev, dist, expect, hj, step, torch, total_n, world = _take(B, "a", "barrier", "cdev", "dist", "expect", "hj", "step", "torch", "total_n", "world")
    for _ in range(a.warmup):
        got = step(verify=True)   # every warm-up step checks the exchange
        assert got == expect, (got, expect)
    hj.timings_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        got = step()
    barrier()
    dt = time.perf_counter() - t0
    assert got == expect, (got, expect)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / a.steps * 1e3
    value = 2.0 * total_n * a.steps / dt / 1e9
    _keep(B, locals(), "got", "ms_per_step", "value")


def leg_instrumented_steps(B):
    """further steps with HIP events around the data-moving kernels, outside the timed region: kernels / roofline / probe_phase come from them"""
    a, barrier, expect, hj, pkg, step = _take(B, "a", "barrier", "expect", "hj", "pkg", "step")
    # The headline loop above ran with the library's default: no HIP events around the kernels.  The per-kernel figures
    # (roofline, probe phase, kernels) come from extra, instrumented steps of the same workload, outside the timed region.
    hj.enable_timings(1)
    hj.timings_reset()
    isteps = max(1, min(a.steps, 5))
    for _ in range(isteps):
        assert step() == expect
    barrier()
    kt = hj.timings()
    hj.enable_timings(0)
    layout = [hj.partition_layout(pkg.REL_R), hj.partition_layout(pkg.REL_S)]
    hj_cfg_bits = [hj.config()["bits1"], hj.config()["bits2"]]
    _keep(B, locals(), "hj_cfg_bits", "isteps", "kt", "layout")


def leg_dist_info(B):
    """N > 1: who received what, the driver that ran, the timeline model of the sliced exchange"""
    a, c_impl, cdev, dist, dist_fallback, dj, n, torch, use_dist, world = _take(B, "a", "c_impl", "cdev", "dist", "dist_fallback", "dj", "n", "torch", "use_dist", "world")
    dist_info = None
    if use_dist:
        recv = torch.tensor(list(dj.last_received), dtype=torch.int64, device=cdev)
        allrecv = [torch.empty_like(recv) for _ in range(world)]
        dist.all_gather(allrecv, recv)
        dist_info = {"world": world, "rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(),
                     "driver": "hj_dist (C++ over RCCL, include/hj_dist.h)" if c_impl else "dist.py (torch.distributed)",
                     "driver_fallback_reason": dist_fallback,
                     "received_tuples_per_rank_R_S": [[int(x) for x in t.tolist()] for t in allrecv]}
        dist_info["transport"] = "rccl (grouped ncclSend/ncclRecv per slice, hj_dist)" if c_impl else "torch.distributed all_to_all_single (%s)" % dist.get_backend()
        if c_impl:
            st = dj.stats()
            G = a.phantom if (world == 1 and a.phantom > 1) else world
            dist_info["rank0"] = st
            if st["path"] == "sliced" and G > 1:
                # Timeline model: every ordered pair of GPUs has its own xGMI link; a rank's bytes to ONE peer
                # cross ONE link direction at LINK_GBS.  Local stages measured by HIP events in this run; everything except the
                # first split, the last pass 1, and pass 2 + join is enqueued to run under the exchange.
                LINK_GBS = 76.8
                per_peer = st["link_bytes"] / (G - 1)
                link_ms = per_peer / (LINK_GBS * 1e9) * 1e3
                local_ms = sum(st["split_ms"]) + sum(st["pass1_ms"]) + st["pass2_join_ms"] + st["early_pass2_join_ms"]
                exposed = st["first_split_ms"] + st["last_pass1_ms"] + st["pass2_join_ms"]
                dist_info["model"] = {"gpus": G, "phantom": bool(world == 1), "link_GBs_per_direction": LINK_GBS,
                                      "bytes_per_link_direction": per_peer, "link_ms": round(link_ms, 3),
                                      "local_ms_total": round(local_ms, 3), "exposed_local_ms": round(exposed, 3),
                                      "exposed_over_link": round(exposed / link_ms, 4),
                                      "modelled_step_ms": round(max(link_ms, local_ms - exposed) + exposed, 3),
                                      "modelled_Gtuples_per_s_per_gpu": round(2.0 * n / ((max(link_ms, local_ms - exposed) + exposed) * 1e-3) / 1e9, 2),
                                      "note": "split(i+1) || exchange(i) || pass-1(i-1); exposed = first split + last pass 1 + pass 2 and join of the probe side's last group of slices"}
    _keep(B, locals(), "dist_info")


def leg_dist_materialize_and_strong(B):
    """N > 1 (or the multi-GPU path on one GPU): the sharded MATERIALISING join and the strong-scaling point, each recorded as {error} if it fails"""
    Rk, Rp, Sk, Sp, a, barrier, c_impl, cdev, dist, dj = _take(B, "Rk", "Rp", "Sk", "Sp", "a", "barrier", "c_impl", "cdev", "dist", "dj")
    dup, expect, hj, n, pkg, rank, torch, use_dist, world = _take(B, "dup", "expect", "hj", "n", "pkg", "rank", "torch", "use_dist", "world")
    # ---- N > 1 (or the multi-GPU path on one GPU): the materialising sharded join and the strong-scaling point ----
    dist_mat, strong = None, None
    # (the headline above is measured: a failure in one of the extra legs is recorded in the line, it does not take the line down —
    # a rank that fails leaves its peers to the deadline of their next collective, after which they fail into the same handler)
    if use_dist and c_impl and not a.no_materialize:
        try:
            dist_mat = dist_materialize_leg(a, pkg, torch, dist, hj, dj, (Rk, Rp, Sk, Sp), n, world, rank, expect, dup, cdev, barrier)
        except Exception as e:   # noqa: BLE001
            dist_mat = {"error": repr(e)}
    if use_dist and c_impl and world > 1 and not a.no_strong and not (dist_mat or {}).get("error"):
        try:
            strong = strong_leg(a, pkg, torch, dist, hj, dj, n, world, rank, cdev, barrier)
        except Exception as e:   # noqa: BLE001
            strong = {"error": repr(e)}
    _keep(B, locals(), "dist_mat", "strong")


def leg_roofline(B):
    """roofline of the dominant kernel (N = 1: a radix pass against HBM, same-run ceilings; N > 1: the exchange against one xGMI link direction)"""
    Rk, Rp, a, c_impl, dist_info, hj, isteps, kt, n, torch = _take(B, "Rk", "Rp", "a", "c_impl", "dist_info", "hj", "isteps", "kt", "n", "torch")
    use_dist, world = _take(B, "use_dist", "world")
    # roofline of the dominant kernel: a radix pass over one relation (4 launches per step at N=1: 2 passes x 2
    # relations), 16 algorithmic bytes per tuple per launch (8 B read + 8 B written)
    # (up to 2^29 tuples in all the two relations' passes are ONE launch per pass: k_part1_fast2 / k_part2_fast2 move both relations)
    passes = ("k_part1_fast", "k_part2_fast", "k_part1_fast2", "k_part2_fast2", "k_scatter_wc", "k_scatter")
    dom = max(passes, key=lambda k: kt.get(k, {}).get("total_ms", 0.0))
    sc = kt.get(dom, {"launches": 0, "total_ms": 0.0})
    roof = None
    if sc["launches"] and not use_dist:
        launches_per_step = sc["launches"] / isteps
        tuples_per_launch = float(n) * (2 if dom.endswith("fast2") else 1)  # a pass launch moves one whole relation (keys + payloads); a merged one, both
        avg_ms = sc["total_ms"] / sc["launches"]
        achieved = 16.0 * tuples_per_launch / (avg_ms * 1e-3) / 1e9
        # HBM bytes per launch from the committed PMC passes of this same command (profiles/): separate
        # --pmc runs for FETCH_SIZE and WRITE_SIZE, KB units, FETCH_SIZE doubled (gfx950 note, MI355X_MICROARCH §HBM)
        # ... and only if that file was collected from THIS build of libhj.so (its sha256 is stored in the file)
        traffic, src = None, None
        try:
            src = "profiles/r6_pmc_2p%d%s.json" % (a.log2n, "_exact" if dom.startswith("k_scatter") else "")
            pmf = json.load(open(os.path.join(ROOT, src)))
            key = [k for k in pmf["kernels"] if k.startswith("hj::" + dom + "<")]
            if key and pmf.get("lib_sha256") == lib_sha256():
                traffic = pmf["kernels"][key[0]]["hbm_bytes_per_launch"]
        except Exception:
            traffic = None
        roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_source": (src + " (rocprofv3 --pmc passes of this command)") if traffic else None,
                "avg_launch_ms": round(avg_ms, 4), "launches_per_step": launches_per_step,
                "algorithmic_bytes_per_launch": 16.0 * tuples_per_launch,
                "measured": ("instrumented steps: kernel events on; up to 2^29 tuples in all, one launch per pass moves BOTH relations (the same "
                             "launches as in the timed steps)") if dom.endswith("fast2") else
                            ("instrumented steps: kernel events on, the passes of R and S on ONE stream (a kernel alone on the chip); the timed "
                             "steps run S's passes on a second stream beside R's (rocprofv3 of those shows overlapped kernel durations; the "
                             "committed kernel stats are taken with HJ_FORK_LOG2=0)")}
        if not a.no_extras:
            # The bound of THIS box, same run: a pass reads 8 B and writes 8 B per tuple, and no kernel with that mix can beat
            # (R + W) / (R / read_only + W / write_only), the two one-way streams measured by hj_ubench kinds 2 / 3 on the same columns.
            # Two further micro-benchmarks are kept as named REFERENCE POINTS, not ceilings (a naive two-column copy and the same copy
            # with every 128-B line stored at a pseudo-random aligned position: the passes beat both, round 4's line said 1.03 / 1.07).
            tk, tp = torch.empty_like(Rk), torch.empty_like(Rp)
            roof.update(mix_ceiling(hj, Rk, Rp, tk, tp, n, 8.0 * tuples_per_launch, 8.0 * tuples_per_launch, achieved))
            copy = hj.ubench("copy", Rk, Rp, tk, tp, n)
            scat = hj.ubench("line_scatter", Rk, Rp, tk, tp, n)
            del tk, tp
            roof.update({"reference_points": {"stream_copy_GBs": round(copy, 1), "line_scatter_GBs": round(scat, 1),
                                              "what": "hj_ubench kinds 0 / 1, same run: a plain 16 B/lane copy of a 2^%d-tuple column pair; the same "
                                                      "reads with every 128-B line stored at a pseudo-random aligned line position.  Not bounds." % a.log2n}})
    if use_dist and c_impl and dist_info and dist_info["rank0"]["path"] == "sliced" and (world > 1 or a.phantom > 1):
        # N > 1: the step is bound by the links, not by HBM: every ordered pair of GPUs has its own xGMI link, and a
        # rank's bytes to ONE peer cross ONE link direction.  achieved = those bytes over the device time the exchange was in
        # flight on the communication stream (HIP events, first slice's exchange start to last slice's end) — on one GPU in the
        # shape of a G-GPU job (--phantom) nothing crosses a link and achieved is null, the model stands in.
        st = dist_info["rank0"]
        G = a.phantom if world == 1 else world
        per_peer = st["link_bytes"] / (G - 1)
        LINK_GBS = 76.8
        achieved = (per_peer / (st["exchange_ms"] * 1e-3) / 1e9) if (world > 1 and st["exchange_ms"] > 0) else None
        roof = {"bound": "xgmi", "kernel": "exchange (one grouped send/recv per slice, %d slices per relation)" % st["slices"],
                "achieved": round(achieved, 2) if achieved else None, "peak": LINK_GBS, "unit": "GB/s per link direction",
                "frac": round(achieved / LINK_GBS, 4) if achieved else None, "traffic": per_peer,
                "bytes_per_link_direction": per_peer, "payload_bytes_per_link_direction": st["payload_bytes"] / (G - 1),
                "exchange_ms": round(st["exchange_ms"], 3),
                "exposed_local_ms": round(st["first_split_ms"] + st["last_pass1_ms"] + st["pass2_join_ms"], 3),
                "local_ms_total": round(sum(st["split_ms"]) + sum(st["pass1_ms"]) + st["pass2_join_ms"] + st["early_pass2_join_ms"], 3),
                "note": "rank 0's view; peak = one xGMI link direction (7 links x 153.6 GB/s bidirectional per GPU); traffic = bytes rank 0 "
                        "sends to ONE peer per step, padding of the fixed-size regions included"}
    _keep(B, locals(), "roof")


def leg_probe_phase(B):
    """per-kernel times of a step and the probe phase against its stated target"""
    a, isteps, kt, n, use_dist = _take(B, "a", "isteps", "kt", "n", "use_dist")
    kernels = {k: {"launches_per_step": v["launches"] / isteps, "ms_per_step": round(v["total_ms"] / isteps, 4)}
               for k, v in kt.items() if v["launches"]}
    jc = kt.get("k_join_count", {"launches": 0, "total_ms": 0.0})
    probe = None
    if jc["launches"] and not use_dist:
        avg = jc["total_ms"] / jc["launches"]
        frac = 8.0 * 2 * n / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS
        probe = {"kernel": "k_join_count", "avg_launch_ms": round(avg, 4),
                 "achieved_GBs": round(8.0 * 2 * n / (avg * 1e-3) / 1e9, 1), "frac_of_8TBs": round(frac, 4),
                 "target_frac": probe_target_frac(a.log2n),
                 "target_model": "0.70 of 8 TB/s at 2^30; below: the same with the measured 25 us of fixed cost per launch — first wave, tail, launch (profiles/r6_fixed_cost_2p27.txt)",
                 "meets_target": bool(frac >= probe_target_frac(a.log2n))}
    _keep(B, locals(), "kernels", "probe")


def leg_reference_phase_split(B):
    """the reference's own phase split (Partition / Joins / Total, hjcp.cu:938-940)"""
    Rk, Rp, Sk, Sp, a, expect, hj, n, pkg, torch = _take(B, "Rk", "Rp", "Sk", "Sp", "a", "expect", "hj", "n", "pkg", "torch")
    use_dist, = _take(B, "use_dist")
    # the reference's phase split (hjcp.cu:938-940: Partition / Joins / Total throughput in MB/s of 2*(|R|+|S|)*4 bytes)
    phase = None
    if not use_dist and not a.no_extras:
        reps = max(2, a.steps // 2)
        tp_, tj_ = 0.0, 0.0
        for _ in range(reps):
            hj.bind_device(pkg.REL_R, Rk, Rp)
            hj.bind_device(pkg.REL_S, Sk, Sp)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            hj.partition_both()
            hj.sync()
            t3 = time.perf_counter()
            assert hj.join_count()[0] == expect
            t2 = time.perf_counter()
            tp_ += t3 - t1
            tj_ += t2 - t3
        nbytes = 2.0 * (2 * n) * 4
        phase = {"partition_ms": round(tp_ / reps * 1e3, 3), "join_ms": round(tj_ / reps * 1e3, 3),
                 "partition_MBps": round(nbytes / (tp_ / reps) / 1e6, 0), "joins_MBps": round(nbytes / (tj_ / reps) / 1e6, 0),
                 "total_MBps": round(nbytes / ((tp_ + tj_) / reps) / 1e6, 0),
                 "units": "the reference's printed lines (hjcp.cu:938-940): 2*(|R|+|S|)*sizeof(int) bytes / seconds / 10^6"}
    _keep(B, locals(), "phase")


def leg_materialize(B):
    """N = 1: the materialising variant (partition both + ONE probe writing (key, payR, payS)), timed the same way"""
    Rk, Rp, Sk, Sp, a, dev, dup, expect, hj, n = _take(B, "Rk", "Rp", "Sk", "Sp", "a", "dev", "dup", "expect", "hj", "n")
    pkg, torch, use_dist = _take(B, "pkg", "torch", "use_dist")
    # secondary: the materialising variant — partition both relations, then build+probe writing (key,payR,payS) in the
    # same probe (the reference's lead timed run, hjcp.cu:881-940), N=1 only
    mat = None
    if not use_dist and not a.no_materialize:
        cap = expect
        ok, opr, ops = (torch.empty(cap, dtype=torch.int32, device=dev) for _ in range(3))

        def mat_step():
            hj.bind_device(pkg.REL_R, Rk, Rp)
            hj.bind_device(pkg.REL_S, Sk, Sp)
            hj.partition_both()
            return hj.join_materialize_into(ok, opr, ops, cap)

        assert mat_step() == expect   # warm-up (first touch of the output columns)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = max(1, a.steps // 2)
        for _ in range(reps):
            nout = mat_step()
        torch.cuda.synchronize()
        dtm = (time.perf_counter() - t0) / reps
        assert nout == expect
        if dup == 1:
            # full-size property check (outside the timed region): with unique keys and payloads = 1 the output
            # multiset is {(k,1,1) : k in R}; its order-independent digest must equal that of (R keys, 1, 1)
            assert hj.digest_triples(ok, opr, ops, nout) == hj.digest_triples(Rk, Rp, Sp, n), "materialised output digest"
        hj.enable_timings(2)   # one instrumented step: every launch of a materialising step, by name
        hj.timings_reset()
        assert mat_step() == expect
        km = hj.timings()
        hj.enable_timings(0)
        mk = km.get("k_join_materialize", {"launches": 0, "total_ms": 0.0})
        mat = {"value": round(2.0 * n / dtm / 1e9, 3), "unit": "billion tuples/s", "ms_per_step": round(dtm * 1e3, 3),
               "output_tuples": int(nout), "probes_per_step": sum(v["launches"] for k, v in km.items() if k.startswith("k_join_count") or k.startswith("k_join_mat")),
               "launches_of_one_step": {k: v["launches"] for k, v in km.items() if v["launches"]}}
        if mk["launches"]:
            avg = mk["total_ms"] / mk["launches"]
            mat["k_join_materialize_ms"] = round(avg, 4)
            mat["k_join_materialize_GBs"] = round((8.0 * 2 * n + 12.0 * nout) / (avg * 1e-3) / 1e9, 1)
            mat["k_join_materialize_frac_of_8TBs"] = round((8.0 * 2 * n + 12.0 * nout) / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            if not a.no_extras:
                mat.update(mix_ceiling(hj, Rk, Rp, ok, opr, n, 8.0 * 2 * n, 12.0 * nout, mat["k_join_materialize_GBs"]))
        del ok, opr, ops
    _keep(B, locals(), "mat")


def leg_config2_as_stated(B):
    """2^27 only: BASELINE configs[1] as written (ONE 9-bit pass) beside the default split"""
    a, expect, hj, hj_cfg_bits, ms_per_step, n, step, torch, use_dist = _take(B, "a", "expect", "hj", "hj_cfg_bits", "ms_per_step", "n", "step", "torch", "use_dist")
    # BASELINE configs[1] as stated — 2^27 x 2^27 with a SINGLE radix pass of 9 bits (2^18-tuple partitions, the LDS table rebuilt
    # ~60 times per partition) — timed beside the default two-pass split of the same size, so that the choice of 9+6 bits is visible
    # where the config is quoted
    as_stated = None
    if a.log2n == 27 and not use_dist and not a.no_extras and not a.bits:
        hj.configure(bits1=9, force_bits=True)
        assert step() == expect
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            assert step() == expect
        torch.cuda.synchronize()
        ms1 = (time.perf_counter() - t0) / 2 * 1e3
        as_stated = {"radix_bits": [9, 0], "ms_per_step": round(ms1, 3), "value": round(2.0 * n / (ms1 * 1e-3) / 1e9, 3), "unit": "billion tuples/s",
                     "note": "configs[1] as stated: single-pass radix (9 bits); the default for this size is two passes (%d+%d bits): %.3f ms"
                             % (hj_cfg_bits[0], hj_cfg_bits[1], ms_per_step)}
        hj.configure()
    _keep(B, locals(), "as_stated")


def leg_alt_transport(B):
    """N > 1: the same workload driven by ONE process over RCCL and over the copy engines (rank 0, in a child process with a deadline)"""
    a, c_impl, dist, dist_info, dj, hj, local, pkg, quiet, rank = _take(B, "a", "c_impl", "dist", "dist_info", "dj", "hj", "local", "pkg", "quiet", "rank")
    torch, use_dist, world = _take(B, "torch", "use_dist", "world")
    # ---- N > 1: both transports on the same workload, one process driving every GPU (rank 0), the others quiet ----
    alt = None
    final_cfg = hj.config()
    if use_dist and c_impl and world > 1 and quiet is not None and not a.no_alt_transport:
        if dj is not None:
            dj.close()
        hj.close()
        del Rk, Rp, Sk, Sp
        B.Rk = B.Rp = B.Sk = B.Sp = B.dj = None   # (the namespace must not keep the columns alive either)
        torch.cuda.empty_cache()
        dist.barrier(group=quiet)           # every rank has released its GPU
        if rank == 0:
            # in a CHILD process with a deadline: the headline is already measured, and a communicator that does not come up (or a
            # crash) in this extra leg must cost a note in the line, not the line
            import subprocess
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK",
                                                                   "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
            cmd = [sys.executable, os.path.abspath(__file__), "--alt-child", "--gpus", str(world), "--log2n", str(a.log2n), "--steps", str(a.steps),
                   "--warmup", str(a.warmup), "--slices", str(a.slices)] + (["--exact-only"] if a.exact_only else []) + (["--single-group"] if a.single_group else [])
            try:
                p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=float(os.environ.get("HJ_BENCH_ALT_TIMEOUT_S", "600")))
                lines = [l for l in p.stdout.splitlines() if l.startswith("{") and '"alt_transport"' in l]
                alt = json.loads(lines[-1])["alt_transport"] if (p.returncode == 0 and lines) else {"error": "child exited with %d: %s" % (p.returncode, p.stderr[-600:])}
            except subprocess.TimeoutExpired:
                alt = {"error": "the one-process leg did not finish within its deadline"}
            except Exception as e:   # noqa: BLE001
                alt = {"error": repr(e)}
        dist.barrier(group=quiet)
        hj = pkg.HashJoin(local, stream=torch.cuda.current_stream().cuda_stream)   # (the CPU baseline below generates its sample on the GPU)
    if dist_info is not None:
        dist_info["alt_transport"] = alt
    _keep(B, locals(), "final_cfg", "hj")


def leg_cpu_baseline(B):
    """rank 0: the CPU baselines (the port of the reference's scheme, the library's own host join, the reference's joinCpu)"""
    a, dev, hj, pkg, rank, torch = _take(B, "a", "dev", "hj", "pkg", "rank", "torch")
    cpu = None
    if rank == 0 and not a.no_cpu_baseline and not (a.phantom > 1):
        # rank 0 at every N (the other ranks wait at the final barrier): the same bounded sample of the per-GPU workload
        cpu = cpu_baseline(pkg, hj, torch, dev, a.log2n)
        cpu["cpu_model"] = cpu_model()
        cpu["joinCpu"] = join_cpu_baseline(hj, torch, dev, cpu["cores"])
    _keep(B, locals(), "cpu")


def leg_print_line(B):
    """rank 0 prints the ONE JSON line; the process group goes down"""
    a, as_stated, backend, c_impl, cpu, dist, dist_info, dist_mat, final_cfg, got = _take(B, "a", "as_stated", "backend", "c_impl", "cpu", "dist", "dist_info", "dist_mat", "final_cfg", "got")
    hj_cfg_bits, isteps, kernels, layout, mat, ms_per_step, n, phase, probe, rank = _take(B, "hj_cfg_bits", "isteps", "kernels", "layout", "mat", "ms_per_step", "n", "phase", "probe", "rank")
    roof, strong, use_dist, value, world = _take(B, "roof", "strong", "use_dist", "value", "world")
    if rank == 0:
        cfg = final_cfg
        cfg["bits1"], cfg["bits2"] = hj_cfg_bits
        line = {
            "metric": ("billion tuples/sec (build+probe), 2^%d⋈2^%d int32 uniform, %d GPU" % (a.log2n, a.log2n, world)
                       if world == 1 else
                       "billion tuples/sec (build+probe), 2^%d⋈2^%d int32 uniform per GPU, %d GPUs" % (a.log2n, a.log2n, world))
                      + (" [PLUMBING RUN over gloo on ONE GPU: not a measurement]" if backend == "gloo" else "")
                      + (" [PHANTOM: ONE GPU running the local stages in the shape of a %d-GPU job, nothing crosses a link; value = this "
                         "GPU's local work only, the modelled step is dist.model: not a measurement of the metric]" % a.phantom
                         if (use_dist and world == 1 and a.phantom > 1) else "")
                      + (" [multi-GPU code path forced at world size 1: not the headline]" if (use_dist and world == 1 and a.phantom <= 1) else ""),
            # not a measurement: the gloo plumbing mode, and any N > 1 line produced by the torch.distributed FALLBACK driver when
            # hj_dist was asked for (a communicator could not be made): a scaling number from it must not pass for hj_dist's
            "is_measurement": backend != "gloo" and not (use_dist and a.dist_impl == "c" and not c_impl and backend != "gloo" and a.balance == "hash")
                              and not (use_dist and world == 1 and a.phantom > 1),
            "value": round(value, 3), "unit": "billion tuples/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "2^%d ⋈ 2^%d unique uniform int32 keys per GPU, payload=1, count-only "
                                   "build+probe after %d-pass radix partition (%d+%d bits)%s" %
                                   (a.log2n, a.log2n, 2 if cfg["bits2"] else 1, cfg["bits1"], cfg["bits2"],
                                    "" if not use_dist else "; level-0 shard split + RCCL all-to-all over %d GPUs" % world),
                       "tuples_per_relation_per_gpu": n, "radix_bits": [cfg["bits1"], cfg["bits2"]],
                       "partition_layout_R_S": layout, "matches": int(got)},
            "roofline": roof, "probe_phase": probe, "phase": phase, "kernels": kernels, "materialize": mat if not use_dist else dist_mat,
            "strong_scaling": strong, "config2_as_stated": as_stated,
            "cpu_baseline": cpu, "dist": dist_info, "lib_sha256": lib_sha256(),
            "timing": "value/ms_per_step: %d steps with no kernel events (library default); kernels/roofline/probe_phase: %d "
                      "further steps with HIP events around the data-moving kernels" % (a.steps, isteps),
        }
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs of this node; default: WORLD_SIZE under a launcher, else 1")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log2n", type=int, default=30, help="tuples per relation per GPU = 2^log2n")
    ap.add_argument("--workload", choices=["uniform", "zipf", "stream", "coprocess", "baselines"], default="uniform",
                    help="uniform = BASELINE configs[2] (the headline); zipf = configs[3]: 2^27 x 2^31 PK-FK, Zipf theta 1.0 (N=1 only)")
    ap.add_argument("--zipf-sizes", type=int, nargs=2, default=[27, 31], help="--workload zipf: log2 of |R| (unique keys) and |S| (Zipf foreign keys); default = BASELINE configs[3]")
    ap.add_argument("--zipf-theta", type=float, default=1.0, help="--workload zipf: skew of the foreign keys (0 = uniform: experiments)")
    ap.add_argument("--build-side", type=int, default=0, help="hj_config.build_side: 0 = the smaller relation, 1 = R, 2 = S (--workload zipf: the skewed side builds)")
    ap.add_argument("--probe-chunk", type=int, default=0, help="experiment knob: hj_config.probe_chunk")
    ap.add_argument("--bits", type=int, nargs=2, default=None, help="experiment knob: radix bits of pass 1 and 2")
    ap.add_argument("--lds", type=int, nargs=2, default=None, help="experiment knob: LDS table capacity and heads of the join kernel")
    ap.add_argument("--exact-only", action="store_true", help="histogram + scan + scatter passes only (no histogram-free passes)")
    ap.add_argument("--balance", choices=["hash", "size"], default="hash",
                    help="multi-GPU level-0 assignment: hash shard g -> GPU g, or 8 virtual shards per GPU assigned by size")
    ap.add_argument("--force-dist", action="store_true", help="run the multi-GPU code path even at world size 1 (sanity runs)")
    ap.add_argument("--dist-impl", choices=["c", "torch"], default="c",
                    help="multi-GPU driver: c = hj_dist (C++ over RCCL behind the C ABI: sliced fixed-size exchange), torch = dist.py over torch.distributed")
    ap.add_argument("--slices", type=int, default=0, help="hj_dist: slices per relation (0 = default)")
    ap.add_argument("--single-group", action="store_true", help="hj_dist A/B: one pass 2 + join per relation instead of joining the probe side's last slice separately")
    ap.add_argument("--phantom", type=int, default=0,
                    help="with --force-dist on ONE GPU: run the sliced pipeline in the shape of a job of this many GPUs (every region copied "
                         "locally instead of crossing a link) and model the link time beside the measured local stages")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-materialize", action="store_true")
    ap.add_argument("--alt-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-alt-transport", action="store_true", help="N > 1: skip the one-process leg that times the same workload over RCCL and over the copy engines")
    ap.add_argument("--no-strong", action="store_true", help="N > 1: skip the strong-scaling leg (2^log2n tuples per relation in TOTAL, 1/N per GPU)")
    ap.add_argument("--no-extras", action="store_true", help="skip the HBM ceilings and the phase split (profiling runs)")
    a = ap.parse_args()
    if a.gpus is None:   # `torchrun ... bench.py` without --gpus: the launcher's world size is the GPU count
        a.gpus = int(os.environ.get("WORLD_SIZE", "1"))

    if a.alt_child:   # the one-process two-transport leg of an N > 1 run (started by rank 0 of that run, see below)
        import torch
        pkg = graft.load_package()
        n = 1 << a.log2n
        total_n = n * a.gpus
        domain = min(total_n, 1 << 32)
        expect = total_n * max(1, total_n // domain)
        print(json.dumps({"alt_transport": alt_transport_leg(a, pkg, torch, a.gpus, n, domain, expect)}))
        return
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: start the N ranks as CHILD processes (one per GPU, RCCL over
        # xGMI) before anything here touches the GPU, forward rank 0's JSON line and the launcher's exit code.
        return launch_ranks(a.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node equal to --gpus" % (a.gpus, world))

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # HJ_BENCH_BACKEND=gloo: plumbing check of the N>1 code path on a one-GPU box — every rank on cuda:0, columns staged
    # through host memory (dist.ShardedJoin.staged).  Not a measurement.
    backend = os.environ.get("HJ_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local = 0
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or a.force_dist
    if use_dist:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29599")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        elif backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    cdev = torch.device("cpu") if backend == "gloo" else dev   # where tensors of small collectives live
    # a CPU-side group for the end of the run: while rank 0 drives every GPU from one process (alt_transport_leg) the other ranks
    # must wait WITHOUT a collective kernel spinning on their GPU
    quiet = None
    if use_dist and world > 1 and backend != "gloo":
        import datetime
        quiet = dist.new_group(backend="gloo", timeout=datetime.timedelta(minutes=40))   # (longer than the child's deadline)
    pkg = graft.load_package()
    n = 1 << a.log2n
    total_n = n * world
    if a.workload == "zipf":
        return bench_zipf(a, pkg, torch, dev, local)
    if a.workload == "stream":
        return bench_stream(a, pkg, torch, dev, local)
    if a.workload == "baselines":
        return bench_baselines(a, pkg, torch, dev, local)
    if a.workload == "coprocess":
        return bench_coprocess(a, pkg, torch, dev, local)

    B = types.SimpleNamespace(**{k: v for k, v in locals().items() if k != 'ap'})
    for leg in (leg_inputs_and_driver,
                leg_timed_headline,
                leg_instrumented_steps,
                leg_dist_info,
                leg_dist_materialize_and_strong,
                leg_roofline,
                leg_probe_phase,
                leg_reference_phase_split,
                leg_materialize,
                leg_config2_as_stated,
                leg_alt_transport,
                leg_cpu_baseline,
                leg_print_line):
        leg(B)


if __name__ == "__main__":
    main()
