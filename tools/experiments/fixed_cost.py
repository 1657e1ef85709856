#!/usr/bin/env python3
"""tools/experiments/fixed_cost.py [log2n] [steps] — where the time of k_join_count and k_part2_fast goes at 2^27 beyond bytes / bandwidth
(VERDICT r5 item 5): per-workgroup start / end stamps (s_memrealtime, 10-ns ticks) and hardware ids written by a library built with
-DHJ_STAMPS (`make -C icde2019-gpu-join_amd/csrc stamps`; gpu_fixed_cost.sh copies libhj_stamps.so over libhj.so on the box's scratch
copy).  Prints one JSON object per kernel: the launch as a timeline — how long until every slot of the chip holds a workgroup (ramp-up),
the steady state, and the tail in which slots run dry one by one — and the idle slot-time of each part."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

TICK_US = 0.01  # s_memrealtime: 100 MHz


def analyse(name, st, slots_per_cu, bytes_moved):
    """st: [n, 4] uint64 {start, mid, end, hw}; rows with start == 0 did not run."""
    st = st[st[:, 0] > 0]
    n = len(st)
    t0, t1 = st[:, 0].astype(np.int64), st[:, 2].astype(np.int64)
    hw = st[:, 3]
    xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xF
    cu = ((hw & np.uint64(0xFFFF)) >> np.uint64(8)).astype(np.int64) & 0xFF   # cu_id[11:8] sh_id[12] se_id[15:13]
    unit = xcc * 256 + cu
    begin, end = int(t0.min()), int(t1.max())
    span = (end - begin) * TICK_US
    dur = (t1 - t0) * TICK_US
    units = np.unique(unit)
    # concurrency over time (workgroups resident), in 1-us bins
    nb = int(span) + 2
    conc = np.zeros(nb)
    for a, b in zip((t0 - begin) * TICK_US, (t1 - begin) * TICK_US):
        ia, ib = int(a), int(b)
        if ia == ib:
            conc[ia] += b - a
        else:
            conc[ia] += ia + 1 - a
            conc[ia + 1:ib] += 1
            conc[ib] += b - ib
    peak = conc.max()
    full = np.nonzero(conc >= 0.9 * peak)[0]
    ramp_us, tail_start = float(full[0]), float(full[-1] + 1)
    slot_time = peak * span                       # what the launch had: peak slots x its duration
    busy = dur.sum()
    idle_ramp = peak * ramp_us - conc[: int(ramp_us)].sum()
    idle_tail = peak * (span - tail_start) - conc[int(tail_start):].sum()
    starts = np.sort((t0 - begin) * TICK_US)
    first_wave = starts[: int(peak)] if len(starts) >= int(peak) else starts
    # per-CU: when did each CU get its last workgroup, when did it finish
    last_end = np.array([t1[unit == u].max() for u in units])
    out = {"kernel": name, "workgroups": n, "cus_seen": int(len(units)), "xcds_seen": int(len(np.unique(xcc))), "peak_resident": round(float(peak), 1),
           "slots_expected": int(len(units)) * slots_per_cu, "span_us": round(span, 2),
           "ramp_up_us": round(ramp_us, 2), "tail_us": round(span - tail_start, 2), "steady_us": round(tail_start - ramp_us, 2),
           "first_wave_start_spread_us": [round(float(first_wave[0]), 2), round(float(np.median(first_wave)), 2), round(float(first_wave[-1]), 2)],
           "workgroup_us": {"median": round(float(np.median(dur)), 2), "p10": round(float(np.percentile(dur, 10)), 2), "p90": round(float(np.percentile(dur, 90)), 2),
                            "max": round(float(dur.max()), 2),
                            "first_wave_median": round(float(np.median(dur[np.argsort(t0)[: int(peak)]])), 2),
                            "later_median": round(float(np.median(dur[np.argsort(t0)[int(peak):]])) if n > peak else 0.0, 2)},
           "slot_time_us": round(float(slot_time), 1), "busy_frac": round(float(busy / slot_time), 4),
           "idle_frac_ramp": round(float(idle_ramp / slot_time), 4), "idle_frac_tail": round(float(idle_tail / slot_time), 4),
           "cu_finish_spread_us": round(float((last_end.max() - last_end.min()) * TICK_US), 2),
           "bytes_over_span_GBs": round(bytes_moved / (span * 1e-6) / 1e9, 1),
           "bytes_over_steady_rate_GBs": None}
    # rate in the steady part: bytes of the workgroups that both start and end inside it are not separable; use busy-weighted share instead
    steady_conc = conc[int(ramp_us): int(tail_start)].sum()
    if steady_conc > 0:
        out["bytes_over_steady_rate_GBs"] = round(bytes_moved * (steady_conc / busy) / ((tail_start - ramp_us) * 1e-6) / 1e9, 1)
    if st[:, 1].max() > 0:
        built = (st[:, 1].astype(np.int64) - t0) * TICK_US
        out["table_built_after_us"] = {"median": round(float(np.median(built)), 2), "first_wave_median": round(float(np.median(built[np.argsort(t0)[: int(peak)]])), 2)}
    return out


def analyse_ends(name, st, event_ms, bytes_moved):
    """The count kernel: END stamps only (a start stamp costs the kernel its third workgroup per CU).  Per CU the sorted end times ARE the
    timeline: a CU holds S workgroups at a time, so its first S ends are the first wave, and after that every end starts a workgroup.
    The launch begins event_ms before the last end (HIP events around the launch)."""
    st = st[st[:, 2] > 0]
    n = len(st)
    t1 = st[:, 2].astype(np.int64)
    hw = st[:, 3]
    unit = ((hw >> np.uint64(32)).astype(np.int64) & 0xF) * 256 + (((hw & np.uint64(0xFFFF)) >> np.uint64(8)).astype(np.int64) & 0xFF)
    end = int(t1.max())
    span = event_ms * 1e3
    begin = end - int(span / TICK_US)
    rel = (t1 - begin) * TICK_US                  # us since the launch began
    units = np.unique(unit)
    per = {u: np.sort(rel[unit == u]) for u in units}
    # residency S: ends per CU come in a steady stream of one per (duration / S); estimate S from the first wave — the number of ends on a
    # CU before the (S+1)-th workgroup could have finished, i.e. the ends closer together than half the first end's time
    first_end = np.array([v[0] for v in per.values()])
    S = int(np.median([int((v < 1.5 * v[0]).sum()) for v in per.values()]))
    gaps = np.concatenate([np.diff(v[S:-S]) for v in per.values() if len(v) > 3 * S])
    steady_wg_us = float(np.median(gaps)) * S if len(gaps) else 0.0      # a slot finishes one workgroup per S gaps
    last_end = np.array([v[-1] for v in per.values()])
    tail_idle = float((span - last_end).sum()) * S                # slot-time with nothing left to start, upper bound (slots of a CU end together)
    wgs_per_cu = np.array([len(v) for v in per.values()])
    # ends per 5-us bin: the throughput curve
    nb = int(span / 5) + 1
    hist = np.bincount(np.minimum((rel / 5).astype(int), nb - 1), minlength=nb)
    steady = hist[2:-2] if nb > 6 else hist
    ideal_us = bytes_moved / 6.1e12 * 1e6                         # at the read-only ceiling of this box class
    return {"kernel": name, "workgroups": n, "cus_seen": int(len(units)), "resident_per_cu": S, "event_us": round(span, 2),
            "first_end_us": {"min": round(float(first_end.min()), 2), "median": round(float(np.median(first_end)), 2), "max": round(float(first_end.max()), 2)},
            "steady_workgroup_us": round(steady_wg_us, 2),
            "first_wave_extra_us": round(float(np.median(first_end)) - steady_wg_us, 2),
            "last_end_per_cu_us": {"min": round(float(last_end.min()), 2), "median": round(float(np.median(last_end)), 2), "max": round(float(last_end.max()), 2)},
            "tail_us": round(float(span - np.median(last_end)), 2), "tail_idle_slot_frac": round(tail_idle / (span * S * len(units)), 4),
            "workgroups_per_cu": {"min": int(wgs_per_cu.min()), "median": int(np.median(wgs_per_cu)), "max": int(wgs_per_cu.max())},
            "ends_per_5us": {"steady_median": int(np.median(steady)), "first_bins": [int(x) for x in hist[:4]], "last_bins": [int(x) for x in hist[-4:]]},
            "bytes_over_event_GBs": round(bytes_moved / (span * 1e-6) / 1e9, 1),
            "steady_rate_GBs": round(float(np.median(steady)) / 5e-6 * (bytes_moved / n) / 1e9, 1),
            "us_at_6.1TBs": round(ideal_us, 2)}


def main():
    import torch
    log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 27
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    lds_cap = int(sys.argv[3]) if len(sys.argv) > 3 else 0      # hj_config.lds_capacity / lds_heads (0 = default): the residency sweep
    lds_heads = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    pkg = graft.load_package()
    n = 1 << log2n
    dev = torch.device("cuda", 0)
    hj = pkg.HashJoin(0, stream=torch.cuda.current_stream().cuda_stream)
    if lds_cap or lds_heads:
        hj.configure(lds_capacity=lds_cap, lds_heads=lds_heads)
    Rk, Rp, Sk, Sp = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(4))
    hj.gen_unique(Rk, n, 0, n, 1)
    hj.gen_unique(Sk, n, 0, n, 2)
    hj.fill_payload(Rp, n, "ones")
    hj.fill_payload(Sp, n, "ones")
    hj.sync()
    hj.bind_device(pkg.REL_R, Rk, Rp)
    hj.bind_device(pkg.REL_S, Sk, Sp)
    assert hj.join()[0] == n
    c = hj.config()
    nparts = 1 << (c["bits1"] + c["bits2"])
    items = nparts + (n >> 16) + 1
    sj = torch.zeros(items * 4, dtype=torch.int64, device=dev)
    sp2 = torch.zeros((1 << c["bits1"]) * 4, dtype=torch.int64, device=dev)
    hj.enable_timings(1)   # one relation per launch, one stream: a kernel alone on the chip
    res = []
    for s in range(steps):
        sj.zero_()
        sp2.zero_()
        hj.debug_set_stamps(sj, sp2)
        hj.timings_reset()
        hj.partition(pkg.REL_R)
        hj.sync()
        p2 = sp2.cpu().numpy().view(np.uint64).reshape(-1, 4).copy()   # R's pass 2 (S's would overwrite it)
        hj.partition(pkg.REL_S)
        assert hj.join_count()[0] == n
        kt = hj.timings()
        j = sj.cpu().numpy().view(np.uint64).reshape(-1, 4)
        if p2[:, 0].max() == 0 or j[:, 2].max() == 0:
            print(json.dumps({"error": "no stamps: the library was not built with -DHJ_STAMPS"}))
            return
        ev = kt["k_join_count"]["total_ms"] / kt["k_join_count"]["launches"]
        a = analyse_ends("k_join_count", j, ev, 8.0 * 2 * n)
        a["event_ms"] = round(ev, 4)
        b = analyse("k_part2_fast", p2, 1, 16.0 * n)
        b["event_ms"] = round(kt["k_part2_fast"]["total_ms"] / kt["k_part2_fast"]["launches"], 4)
        res.append((a, b))
    hj.debug_set_stamps(None, None)
    for a, b in res[1:] or res:
        print(json.dumps(dict(a, log2n=log2n, lds_capacity=c["lds_capacity"], lds_heads=c["lds_heads"])))
        if not (lds_cap or lds_heads):
            print(json.dumps(dict(b, log2n=log2n)))


if __name__ == "__main__":
    main()
