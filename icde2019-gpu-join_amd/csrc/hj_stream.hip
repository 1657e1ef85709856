// hj_stream.hip — host side of the two paths whose relations live in HOST memory: the streaming probe side (outOfGPU_Join3_payload,
// hjcp.cu:1684-1984: R resident, S in segments, optionally materialising back to the host) and CPU–GPU co-processing
// (outOfGPU_Join2_payload, hjcp.cu:1263-1618: the CPU splits both relations into level-0 partitions, the GPU joins the pairs).
// The in-HBM join and the C ABI around it: hj_api.hip.
#include <hip/hip_runtime.h>
#include <ctype.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <chrono>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "hj.h"
#include "hj_internal.h"
#include "hj_host.h"
#include "hj_ctx.h"

using namespace hj;
using namespace hjx;

namespace {

// S streamed from host memory against a resident R (outOfGPU_Join3_payload, hjcp.cu:1684-1984).  With output columns
// (h_out != nullptr) every segment's join also materialises (key,payR,payS) into double-buffered device columns that a
// third stream copies back to the host while the next segment is partitioned and joined (hjcp.cu:1917-1961).
int stream_probe(hj_ctx *c, const int32_t *h_keys, const int32_t *h_pays, uint64_t n, uint64_t segment_tuples,
                 int payload_mode, uint64_t *matches, uint64_t *agg, int32_t *const h_out[3], uint64_t out_cap) {
    if (!c) return HJ_EINVAL;
    if (n && !h_keys) return fail(c, HJ_EINVAL, "keys == NULL");
    if (payload_mode == HJ_PAYLOAD_GIVEN && n && !h_pays) return fail(c, HJ_EINVAL, "payload_mode GIVEN needs a payload column");
    if (payload_mode < HJ_PAYLOAD_ONES || payload_mode > HJ_PAYLOAD_GIVEN) return fail(c, HJ_EINVAL, "bad payload_mode");
    Rel &R = c->rel[HJ_REL_R];
    if (!R.bound) return fail(c, HJ_EINVAL, "load or bind R before streaming S");
    HIPCHK(c, hipSetDevice(c->device));
    // segment size: the reference cuts S into |R|/4 (hjcp.cu:1697-1698, an 8 GB card); with HBM to spare a
    // segment is at least 2^24 tuples so that the copies are long and the passes efficient
    uint64_t seg = segment_tuples ? segment_tuples : (R.n / 4 > ((uint64_t)1 << 24) ? R.n / 4 : ((uint64_t)1 << 24));
    if (seg > n && n) seg = n;
    if (seg == 0) seg = 1;
    if (!c->copy) HIPCHK(c, hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking));
    for (int i = 0; i < 2; i++) {
        if (!c->seg_ready[i]) HIPCHK(c, hipEventCreateWithFlags(&c->seg_ready[i], hipEventDisableTiming));
        RET(ensure(c, c->seg_k[i], (size_t)(seg + PAD) * 4));
        RET(ensure(c, c->seg_p[i], (size_t)(seg + PAD) * 4));
    }
    if (h_out) {
        if (!c->d2h) HIPCHK(c, hipStreamCreateWithFlags(&c->d2h, hipStreamNonBlocking));
        for (int i = 0; i < 2; i++) {
            if (!c->out_ready[i]) HIPCHK(c, hipEventCreateWithFlags(&c->out_ready[i], hipEventDisableTiming));
            if (!c->out_free[i]) HIPCHK(c, hipEventCreateWithFlags(&c->out_free[i], hipEventDisableTiming));
        }
    }
    bool out_used[2] = {false, false};
    c->rel[HJ_REL_S].prefer_exact = false; // every call streams new data: the histogram-free passes get their chance again
    c->rel[HJ_REL_S].sampled_failed = false; c->rel[HJ_REL_S].sp.valid = false;
    c->rel[HJ_REL_S].probed = true; // no look before the attempt for streamed segments: it reads back, and the segment loop never blocks the host
    const bool saved_force = c->force_build_r;
    c->force_build_r = true; // R builds, whatever the segment size; radix bits follow |R|
    // R is partitioned once (hjcp.cu:1874-1892), against an S stand-in of one segment so the bits are fixed
    c->rel[HJ_REL_S].in_k = (const int32_t *)c->seg_k[0].p;
    c->rel[HJ_REL_S].in_p = (const int32_t *)c->seg_p[0].p;
    c->rel[HJ_REL_S].n = seg; c->rel[HJ_REL_S].bound = true;
    invalidate(c, HJ_REL_S);
    int rc = 0;
    choose_bits(c);
    if (!R.partitioned || R.pb1 + R.pb2 != c->bits1 + c->bits2) rc = partition_rel(c, HJ_REL_R);
    uint64_t tot_m = 0, tot_a = 0;
    const uint64_t nseg = n ? (n + seg - 1) / seg : 0;
    auto issue_copy = [&](uint64_t i) -> int {
        const uint64_t off = i * seg, cnt = (off + seg <= n) ? seg : n - off;
        const int b = (int)(i & 1);
        HIPCHK(c, hipMemcpyAsync(c->seg_k[b].p, h_keys + off, cnt * 4, hipMemcpyHostToDevice, c->copy));
        if (payload_mode == HJ_PAYLOAD_GIVEN)
            HIPCHK(c, hipMemcpyAsync(c->seg_p[b].p, h_pays + off, cnt * 4, hipMemcpyHostToDevice, c->copy));
        HIPCHK(c, hipEventRecord(c->seg_ready[b], c->copy));
        return 0;
    };
    if (!h_out && !rc && nseg) {
        // ---- count-only: the segment loop never blocks the host (the reference chains its segments with events the same way,
        // hjcp.cu:1897-1965).  copy(i) waits for join(i-2) — the last reader of its staging buffer — by event; every segment's
        // (matches, aggregate, overflow flag of S) is parked in a device array; ONE read-back at the end.  A segment whose slots
        // overflowed (skew) contributed nothing: those few are redone afterwards through the blocking path. ----
        if ((rc = resolve_layout(c, R))) goto done; // R's flag is read once, before the loop [sync]
        for (int i = 0; i < 2; i++)
            if (!c->seg_joined[i] && hipEventCreateWithFlags(&c->seg_joined[i], hipEventDisableTiming) != hipSuccess) { rc = fail(c, HJ_EHIP, "event"); goto done; }
        if ((rc = ensure(c, c->seg_res, (size_t)nseg * 32))) goto done;
        uint64_t *sc = (uint64_t *)c->scalars.p;
        for (uint64_t i = 0; i < nseg && !rc; i++) {
            const uint64_t off = i * seg, cnt = (off + seg <= n) ? seg : n - off;
            const int b = (int)(i & 1);
            if (i >= 2 && hipStreamWaitEvent(c->copy, c->seg_joined[b], 0) != hipSuccess) { rc = fail(c, HJ_EHIP, "hipStreamWaitEvent"); break; }
            if ((rc = issue_copy(i))) break;
            if (hipStreamWaitEvent(c->stream, c->seg_ready[b], 0) != hipSuccess) { rc = fail(c, HJ_EHIP, "hipStreamWaitEvent"); break; }
            if (payload_mode != HJ_PAYLOAD_GIVEN && launch_fill(c->stream, (int32_t *)c->seg_p[b].p, cnt, payload_mode, off) != hipSuccess) { rc = fail(c, HJ_EHIP, "fill"); break; }
            Rel &S = c->rel[HJ_REL_S];
            S.in_k = (const int32_t *)c->seg_k[b].p; S.in_p = (const int32_t *)c->seg_p[b].p; S.n = cnt; S.bound = true;
            invalidate(c, HJ_REL_S);
            if ((rc = partition_rel(c, HJ_REL_S))) break;
            if ((rc = hj_join_count_enqueue(c))) break;
            uint64_t *slot = (uint64_t *)c->seg_res.p + 4 * i;
            if (hipMemcpyAsync(slot, sc + 1, 16, hipMemcpyDeviceToDevice, c->stream) != hipSuccess ||
                hipMemcpyAsync(slot + 2, sc + 8 + HJ_REL_S, 8, hipMemcpyDeviceToDevice, c->stream) != hipSuccess ||
                hipEventRecord(c->seg_joined[b], c->stream) != hipSuccess) { rc = fail(c, HJ_EHIP, "segment result"); break; }
        }
        std::vector<uint64_t> res((size_t)nseg * 4, 0);
        if (!rc && hipMemcpyAsync(res.data(), c->seg_res.p, (size_t)nseg * 32, hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = fail(c, HJ_EHIP, "result read-back");
        if (hipStreamSynchronize(c->stream) != hipSuccess && !rc) rc = fail(c, HJ_EHIP, "hipStreamSynchronize");
        (void)hipStreamSynchronize(c->copy);
        for (uint64_t i = 0; i < nseg && !rc; i++) {
            if (!(uint32_t)res[4 * i + 2]) { tot_m += res[4 * i]; tot_a += res[4 * i + 1]; continue; }
            // this segment's slots overflowed: blocking redo (hj_join_count re-partitions S along the skew ladder)
            const uint64_t off = i * seg, cnt = (off + seg <= n) ? seg : n - off;
            if (hipMemcpy(c->seg_k[0].p, h_keys + off, cnt * 4, hipMemcpyHostToDevice) != hipSuccess ||
                (payload_mode == HJ_PAYLOAD_GIVEN && hipMemcpy(c->seg_p[0].p, h_pays + off, cnt * 4, hipMemcpyHostToDevice) != hipSuccess)) { rc = fail(c, HJ_EHIP, "H2D"); break; }
            if (payload_mode != HJ_PAYLOAD_GIVEN && launch_fill(c->stream, (int32_t *)c->seg_p[0].p, cnt, payload_mode, off) != hipSuccess) { rc = fail(c, HJ_EHIP, "fill"); break; }
            Rel &S = c->rel[HJ_REL_S];
            S.in_k = (const int32_t *)c->seg_k[0].p; S.in_p = (const int32_t *)c->seg_p[0].p; S.n = cnt; S.bound = true;
            invalidate(c, HJ_REL_S);
            if ((rc = partition_rel(c, HJ_REL_S))) break;
            uint64_t m = 0, a = 0;
            if ((rc = hj_join_count(c, &m, &a))) break;
            tot_m += m; tot_a += a;
        }
        goto done;
    }
    if (!rc && nseg) {
        // ---- materialising (hjcp.cu:1917-1961): ONE probe per segment writes (key, payR, payS) into double-buffered device columns
        // (k_join_mat_reg; the two-probe form this replaces counted every segment, read the count back, then probed again).  The
        // size of a segment's output is known on the device only, and the D2H copy needs it on the host — so the host looks at
        // segment i's cursor ONE SEGMENT LATER: while it waits for it, the device already has segment i+1's copy, partition passes
        // and probe queued, and never idles (the reference chains its segments by events, hjcp.cu:1897-1965, and folds what does
        // not fit its ring, D6; here every tuple is kept).  Output columns are sized for one match per probe tuple; a segment that
        // produces more (duplicates in R), or whose slots overflowed (skew in S), wrote nothing usable: those few are redone at the
        // end through the blocking path, with exact sizes. ----
        if ((rc = resolve_layout(c, R))) goto done; // R's flag is read once, before the loop [sync]
        uint64_t *sc = (uint64_t *)c->scalars.p;
        hipEvent_t ev_done[2] = {nullptr, nullptr};
        uint64_t *h_seg = nullptr; // pinned: per parity {cursor, S's overflow flag, aggregate}
        std::vector<uint64_t> redo;
        Buf d_seg;
        const uint64_t ocap = seg + seg / 8 + 1024;
        if (hipHostMalloc((void **)&h_seg, 2 * 4 * 8, hipHostMallocDefault) != hipSuccess) { rc = fail(c, HJ_ENOMEM, "pinned result block"); goto done; }
        for (int b = 0; b < 2 && !rc; b++) {
            if (hipEventCreateWithFlags(&ev_done[b], hipEventDisableTiming) != hipSuccess) rc = fail(c, HJ_EHIP, "event");
            if (!rc) rc = ensure(c, c->out_k[b], (size_t)(ocap + PAD) * 4);
            if (!rc) rc = ensure(c, c->out_p1[b], (size_t)(ocap + PAD) * 4);
            if (!rc) rc = ensure(c, c->out_p2[b], (size_t)(ocap + PAD) * 4);
        }
        if (!rc) rc = ensure(c, d_seg, 2 * 4 * 8);
        // what segment j left behind: copy its output to the host columns, or put it on the redo list
        auto finalize = [&](uint64_t j) -> int {
            const int b = (int)(j & 1);
            if (hipEventSynchronize(ev_done[b]) != hipSuccess) return fail(c, HJ_EHIP, "hipEventSynchronize");
            const uint64_t m = h_seg[4 * b], flagged = h_seg[4 * b + 1] & 0xFFFFFFFFu, a = h_seg[4 * b + 2];
            if (flagged || m > ocap) { redo.push_back(j); return 0; }
            const uint64_t room = tot_m < out_cap ? out_cap - tot_m : 0, take = m < room ? m : room;
            if (take) {
                if (hipStreamWaitEvent(c->d2h, ev_done[b], 0) != hipSuccess) return fail(c, HJ_EHIP, "hipStreamWaitEvent");
                const void *src[3] = {c->out_k[b].p, c->out_p1[b].p, c->out_p2[b].p};
                for (int q = 0; q < 3; q++)
                    if (hipMemcpyAsync(h_out[q] + tot_m, src[q], take * 4, hipMemcpyDeviceToHost, c->d2h) != hipSuccess) return fail(c, HJ_EHIP, "D2H of the output");
            }
            if (hipEventRecord(c->out_free[b], c->d2h) != hipSuccess) return fail(c, HJ_EHIP, "event");
            out_used[b] = true;
            tot_m += m; tot_a += a;
            return 0;
        };
        if (!rc) rc = issue_copy(0);
        for (uint64_t i = 0; i < nseg && !rc; i++) {
            const uint64_t off = i * seg, cnt = (off + seg <= n) ? seg : n - off;
            const int b = (int)(i & 1);
            // staging buffer b^1 was last read by the probe of segment i-1 (still queued or running): the copy of segment i+1 waits for it
            if (i + 1 < nseg) {
                if (i >= 1 && hipStreamWaitEvent(c->copy, ev_done[b ^ 1], 0) != hipSuccess) { rc = fail(c, HJ_EHIP, "hipStreamWaitEvent"); break; }
                if ((rc = issue_copy(i + 1))) break;
            }
            if (hipStreamWaitEvent(c->stream, c->seg_ready[b], 0) != hipSuccess) { rc = fail(c, HJ_EHIP, "hipStreamWaitEvent"); break; }
            if (payload_mode != HJ_PAYLOAD_GIVEN && launch_fill(c->stream, (int32_t *)c->seg_p[b].p, cnt, payload_mode, off) != hipSuccess) { rc = fail(c, HJ_EHIP, "fill"); break; } // row ids are global
            Rel &S = c->rel[HJ_REL_S];
            S.in_k = (const int32_t *)c->seg_k[b].p; S.in_p = (const int32_t *)c->seg_p[b].p; S.n = cnt; S.bound = true;
            invalidate(c, HJ_REL_S);
            if ((rc = partition_rel(c, HJ_REL_S))) break;
            // the device output columns of this parity were last read by the D2H copies of segment i-2
            if (out_used[b] && hipStreamWaitEvent(c->stream, c->out_free[b], 0) != hipSuccess) { rc = fail(c, HJ_EHIP, "hipStreamWaitEvent"); break; }
            JoinArgs ja;
            bool tag16 = false;
            if ((rc = plan_join(c, ja, tag16))) break;
            ja.out_key = (int32_t *)c->out_k[b].p;
            ja.out_bpay = (int32_t *)c->out_p1[b].p; // R builds: build payload = payR
            ja.out_ppay = (int32_t *)c->out_p2[b].p;
            ja.out_cap = ocap;
            { Timed tm(c, "k_join_materialize"); if (launch_join_mat_reg(c->stream, ja, c->max_items, tag16) != hipSuccess) { rc = fail(c, HJ_EHIP, "materialise launch"); break; } }
            c->join_planned = false;
            uint64_t *slot = (uint64_t *)d_seg.p + 4 * b;
            if (hipMemsetAsync(slot + 2, 0, 8, c->stream) != hipSuccess ||
                launch_dot(c->stream, (const int32_t *)c->out_p1[b].p, (const int32_t *)c->out_p2[b].p, sc + SC_CURSOR, ocap, slot + 2) != hipSuccess ||
                hipMemcpyAsync(slot, sc + SC_CURSOR, 8, hipMemcpyDeviceToDevice, c->stream) != hipSuccess ||
                hipMemcpyAsync(slot + 1, sc + 8 + HJ_REL_S, 8, hipMemcpyDeviceToDevice, c->stream) != hipSuccess ||
                hipMemcpyAsync(h_seg + 4 * b, slot, 24, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                hipEventRecord(ev_done[b], c->stream) != hipSuccess) { rc = fail(c, HJ_EHIP, "segment result"); break; }
            if (i >= 1 && (rc = finalize(i - 1))) break; // one segment behind: the device has segment i queued while the host waits here
        }
        if (!rc && nseg) rc = finalize(nseg - 1);
        if (hipStreamSynchronize(c->stream) != hipSuccess && !rc) rc = fail(c, HJ_EHIP, "hipStreamSynchronize");
        (void)hipStreamSynchronize(c->copy);
        // the few segments that did not fit / overflowed their slots: blocking redo with exact sizes (count, then one probe)
        for (size_t r = 0; r < redo.size() && !rc; r++) {
            const uint64_t j = redo[r], off = j * seg, cnt = (off + seg <= n) ? seg : n - off;
            if (hipMemcpy(c->seg_k[0].p, h_keys + off, cnt * 4, hipMemcpyHostToDevice) != hipSuccess ||
                (payload_mode == HJ_PAYLOAD_GIVEN && hipMemcpy(c->seg_p[0].p, h_pays + off, cnt * 4, hipMemcpyHostToDevice) != hipSuccess)) { rc = fail(c, HJ_EHIP, "H2D"); break; }
            if (payload_mode != HJ_PAYLOAD_GIVEN && launch_fill(c->stream, (int32_t *)c->seg_p[0].p, cnt, payload_mode, off) != hipSuccess) { rc = fail(c, HJ_EHIP, "fill"); break; }
            Rel &S = c->rel[HJ_REL_S];
            S.in_k = (const int32_t *)c->seg_k[0].p; S.in_p = (const int32_t *)c->seg_p[0].p; S.n = cnt; S.bound = true;
            S.prefer_exact = false; S.sampled_failed = false; S.sp.valid = false;
            invalidate(c, HJ_REL_S);
            if ((rc = partition_rel(c, HJ_REL_S))) break;
            uint64_t m = 0, a = 0;
            if ((rc = hj_join_count(c, &m, &a))) break; // [sync]; re-partitions S along the skew ladder
            if (m) {
                if (hipStreamSynchronize(c->d2h) != hipSuccess) { rc = fail(c, HJ_EHIP, "hipStreamSynchronize"); break; } // the columns may grow: nothing may still read them
                if ((rc = ensure(c, c->out_k[0], (size_t)(m + PAD) * 4)) || (rc = ensure(c, c->out_p1[0], (size_t)(m + PAD) * 4)) ||
                    (rc = ensure(c, c->out_p2[0], (size_t)(m + PAD) * 4))) break;
                uint64_t nout = 0;
                if ((rc = hj_join_materialize(c, (int32_t *)c->out_k[0].p, (int32_t *)c->out_p1[0].p, (int32_t *)c->out_p2[0].p, m, &nout))) break;
                const uint64_t room = tot_m < out_cap ? out_cap - tot_m : 0, take = m < room ? m : room;
                const void *src[3] = {c->out_k[0].p, c->out_p1[0].p, c->out_p2[0].p};
                for (int q = 0; q < 3 && !rc && take; q++)
                    if (hipMemcpy(h_out[q] + tot_m, src[q], take * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(c, HJ_EHIP, "D2H of the output");
            }
            tot_m += m; tot_a += a;
        }
        for (int b = 0; b < 2; b++) if (ev_done[b]) (void)hipEventDestroy(ev_done[b]);
        if (c->d2h) (void)hipStreamSynchronize(c->d2h);
        if (h_seg) (void)hipHostFree(h_seg);
        release(d_seg);
    }
done:
    // on every exit path: the H2D copy of the next segment may still be reading the caller's columns
    (void)hipStreamSynchronize(c->copy);
    if (c->d2h) (void)hipStreamSynchronize(c->d2h);
    c->force_build_r = saved_force;
    c->rel[HJ_REL_S].bound = false; // the staging buffers are not a user relation: S is unbound afterwards (hj.h)
    c->rel[HJ_REL_S].n = 0;
    invalidate(c, HJ_REL_S);
    if (rc) return rc;
    if (matches) *matches = tot_m;
    if (agg) *agg = tot_a;
    if (h_out && tot_m > out_cap) return fail(c, HJ_ECAPACITY, "join produced %llu tuples, capacity %llu",
                                              (unsigned long long)tot_m, (unsigned long long)out_cap);
    return HJ_OK;
}

} // namespace

extern "C" {

int hj_join_stream_probe(hj_ctx *c, const int32_t *h_keys, const int32_t *h_pays, uint64_t n, uint64_t segment_tuples,
                         int payload_mode, uint64_t *matches, uint64_t *agg) {
    return stream_probe(c, h_keys, h_pays, n, segment_tuples, payload_mode, matches, agg, nullptr, 0);
}

int hj_join_stream_probe_materialize(hj_ctx *c, const int32_t *h_keys, const int32_t *h_pays, uint64_t n, uint64_t segment_tuples,
                                     int payload_mode, int32_t *h_out_key, int32_t *h_out_payR, int32_t *h_out_payS,
                                     uint64_t cap, uint64_t *n_out, uint64_t *agg) {
    if (!c) return HJ_EINVAL;
    if (cap && (!h_out_key || !h_out_payR || !h_out_payS)) return fail(c, HJ_EINVAL, "output columns == NULL");
    int32_t *const out[3] = {h_out_key, h_out_payR, h_out_payS};
    return stream_probe(c, h_keys, h_pays, n, segment_tuples, payload_mode, n_out, agg, out, cap);
}

// CPU-GPU co-processing (outOfGPU_Join2_payload, hjcp.cu:1000-1680): both relations live in HOST memory;
// the host splits them into level-0 partitions (16-way on 16 threads in the reference, hjcp.cu:1256-1266,
// pp.cuh:38-39), and every level-0 partition pair is an independent join (hjcp.cu:1503-1618): uploaded
// over PCIe into double-buffered staging while the previous pair is partitioned and joined on the GPU.
// Pairs that fit the device-memory budget together form a residency group (first-fit decreasing: the role of the reference's
// knapsack, groupOptimal2, pp.cu:307-468); with one group R is uploaded while the host still splits S.
int hj_join_coprocess(hj_ctx *c, const int32_t *h_R, const int32_t *h_Pr, uint64_t nR, const int32_t *h_S,
                      const int32_t *h_Ps, uint64_t nS, uint32_t level0_parts, uint32_t host_threads, uint64_t *matches,
                      uint64_t *agg) {
    if (!c) return HJ_EINVAL;
    if ((nR && !h_R) || (nS && !h_S)) return fail(c, HJ_EINVAL, "keys == NULL");
    if (level0_parts == 0) level0_parts = 16;
    if (level0_parts > 4096) return fail(c, HJ_EINVAL, "level0_parts out of range");
    const auto t_enter = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_enter).count() * 1e3; };
    double mark[6] = {0, 0, 0, 0, 0, 0}; // HJ_DEBUG: ms since entry at the end of each phase
    if (host_threads == 0) {
        host_threads = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
        // containers: honour the cgroup v2 CPU quota (oversubscribed threads partition slower, not faster)
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[32]; long period = 0;
            if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
                long cpus = (atol(q) + period - 1) / period;
                if (cpus >= 1 && (unsigned long)cpus < host_threads) host_threads = (uint32_t)cpus;
            }
            fclose(f);
        }
    }
    HIPCHK(c, hipSetDevice(c->device));
    // NUMA (partition-primitives.cu:129-253, hjcp.cu:1142-1149: partitions allocated per node, threads bound to sockets, staging
    // near the GPU): hipHostMalloc places pinned memory on the node closest to the current device unless told otherwise; on a
    // host with more than one node the split's workers are bound to that node's CPUs, so that the write-combining buffers,
    // the non-temporal stores and the DMA reads all stay on the GPU's socket.  HJ_NUMA=0 leaves the threads where they are.
    std::vector<int> pin;
    c->numa_nodes = host_numa_nodes();
    c->numa_gpu_node = -1;
    {
        int node = -1;
        if (hipDeviceGetAttribute(&node, hipDeviceAttributeHostNumaId, c->device) == hipSuccess) c->numa_gpu_node = node;
        else (void)hipGetLastError();
        if (c->numa_gpu_node < 0) { // the runtime does not say: ask the PCI device in sysfs
            char bus[32] = {0}, path[128];
            if (hipDeviceGetPCIBusId(bus, sizeof bus, c->device) == hipSuccess) {
                for (char *q = bus; *q; q++) *q = (char)tolower(*q);
                snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
                if (FILE *f = fopen(path, "r")) { if (fscanf(f, "%d", &node) == 1) c->numa_gpu_node = node; fclose(f); }
            } else (void)hipGetLastError();
        }
        const char *e = getenv("HJ_NUMA");
        if (c->numa_nodes > 1 && c->numa_gpu_node >= 0 && !(e && atoi(e) == 0)) pin = host_node_cpus(c->numa_gpu_node);
    }
    c->numa_pinned_cpus = (int)pin.size();
    // host split; the partitioned copies are pinned so that the uploads are asynchronous.  The staging buffers belong
    // to the context and only ever grow: a caller that joins in a loop pins host memory once.
    // Payload columns that are not given (= ones, what the reference synthesises, hjcp.cu:1994-1999) are neither split nor
    // uploaded: the device fills its staging payload columns with ones once per call — half the split's stores and half the
    // PCIe bytes of round 4's version.
    int32_t *pk[2] = {nullptr, nullptr}, *pp[2] = {nullptr, nullptr};
    const uint64_t nn[2] = {nR, nS};
    const int32_t *srcK[2] = {h_R, h_S}, *srcP[2] = {h_Pr, h_Ps};
    // The split is ONE pass over each input (round 5; the histogram pass of hj_host_split is gone from this path): a partition comes
    // out as a list of blocks of the staging columns, which is all an upload needs.
    // HJ_COPROCESS_SPLIT=2 (read per call: same-context A/B) runs the two-pass split instead.
    std::vector<HostBlock> blk[2];
    std::vector<uint64_t> psize[2];
    const char *split_env = getenv("HJ_COPROCESS_SPLIT");
    const bool two_pass = split_env && atoi(split_env) == 2, no_overlap = split_env && atoi(split_env) == 3; // 3: one pass, uploads after the split
    std::vector<uint64_t> arena_of[2], sent_upto[2]; // one group: what went to the device while the split was running
    uint64_t streamed[2] = {0, 0};
    int rc = 0;
    for (int r = 0; r < 2 && !rc; r++) {
        const uint64_t need = host_split_blocks_capacity(nn[r], level0_parts, host_threads);
        if (c->host_cap[r] < need + 16 || (srcP[r] && !c->host_p[r])) {
            if (c->host_k[r]) (void)hipHostFree(c->host_k[r]);
            if (c->host_p[r]) (void)hipHostFree(c->host_p[r]);
            c->host_k[r] = c->host_p[r] = nullptr; c->host_cap[r] = 0;
            if (hipHostMalloc((void **)&c->host_k[r], (size_t)(need + 16) * 4, hipHostMallocDefault) != hipSuccess ||
                (srcP[r] && hipHostMalloc((void **)&c->host_p[r], (size_t)(need + 16) * 4, hipHostMallocDefault) != hipSuccess))
                rc = fail(c, HJ_ENOMEM, "pinned host buffers for the level-0 split");
            else c->host_cap[r] = need + 16;
        }
        pk[r] = c->host_k[r]; pp[r] = srcP[r] ? c->host_p[r] : nullptr;
    }
    // Device-memory budget of a residency group: staging (double buffered) + both relations' partition buffers ~ 35 bytes per tuple,
    // half of what the card can give this context — the free memory PLUS what the context already holds in buffers this call reuses
    // (round 4 looked at the free memory alone: the second identical call saw less and could cut more groups).
    // HJ_COPROCESS_GROUP_TUPLES overrides (tests force small groups).  On a 288-GB card everything up to ~2^31 tuples is one group.
    uint64_t budget = 0;
    if (const char *e = getenv("HJ_COPROCESS_GROUP_TUPLES")) budget = strtoull(e, nullptr, 10);
    if (!budget) {
        size_t free_b = 0, total_b = 0, held = 0;
        for (int i = 0; i < 2; i++) held += c->cop_k[i].cap + c->cop_p[i].cap + c->seg_k[i].cap + c->seg_p[i].cap +
                                            c->rel[i].a_k.cap + c->rel[i].a_p.cap + c->rel[i].b_k.cap + c->rel[i].b_p.cap;
        budget = (hipMemGetInfo(&free_b, &total_b) == hipSuccess) ? ((uint64_t)free_b + held) / 2 / 35 : ((uint64_t)1 << 28);
    }
    if (!rc && !c->copy) { if (hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking) != hipSuccess) rc = fail(c, HJ_EHIP, "copy stream"); }
    for (int i = 0; i < 2 && !rc; i++)
        if (!c->seg_ready[i] && hipEventCreateWithFlags(&c->seg_ready[i], hipEventDisableTiming) != hipSuccess) rc = fail(c, HJ_EHIP, "event");
    const bool one_group = nR + nS <= budget; // known before anything is split: R's upload may then start while S is being split
    auto split = [&](int r) -> int {
        if (two_pass) { // the round-4 split (histogram pass + scatter into contiguous partitions): one block per partition
            std::vector<uint64_t> off;
            if (!host_level0_split(srcK[r], srcP[r], nn[r], level0_parts, host_threads, pk[r], pp[r], off, pin.empty() ? nullptr : &pin))
                return fail(c, HJ_ENOMEM, "the two-pass level-0 split of %llu tuples could not allocate or start its %u host threads", (unsigned long long)nn[r], host_threads);
            blk[r].clear(); psize[r].assign(level0_parts, 0);
            for (uint32_t p = 0; p < level0_parts; p++)
                if (off[p + 1] > off[p]) { blk[r].push_back(HostBlock{p, off[p], off[p + 1] - off[p]}); psize[r][p] = off[p + 1] - off[p]; }
            return 0;
        }
        // One residency group: which partition a tuple belongs to does not matter to the upload, so the complete part of every worker's
        // arena crosses PCIe WHILE the split is still running (the reference overlaps staging, H2D and GPU work per batch, hjcp.cu:
        // 1477-1618); the calling thread issues the copies and takes no share of the split.  What is left when the workers have
        // finished — the partly filled blocks and the last few full ones — follows block by block (upload_rest).
        std::vector<uint64_t> sent; // per worker: its arena is uploaded up to here
        int up_rc = 0;
        std::function<void(const HostSplitProgress &)> pump = [&](const HostSplitProgress &pg) {
            const uint32_t W = (uint32_t)pg.arena.size() - 1;
            if (sent.empty()) { sent.assign(pg.arena.begin(), pg.arena.end() - 1); arena_of[r] = pg.arena; }
            const uint64_t chunk = std::max<uint64_t>(host_split_block_size(nn[r], level0_parts, W), std::min<uint64_t>((uint64_t)1 << 20, nn[r] / W / 8));
            Buf &dk = r ? c->seg_k[0] : c->cop_k[0], &dp = r ? c->seg_p[0] : c->cop_p[0];
            for (uint32_t t = 0; t < W && !up_rc; t++) {
                const uint64_t d = pg.done(t);
                if (d < sent[t] + chunk) continue;
                const uint64_t cnt = d - sent[t];
                if (hipMemcpyAsync((int32_t *)dk.p + streamed[r], pk[r] + sent[t], cnt * 4, hipMemcpyHostToDevice, c->copy) != hipSuccess ||
                    (pp[r] && hipMemcpyAsync((int32_t *)dp.p + streamed[r], pp[r] + sent[t], cnt * 4, hipMemcpyHostToDevice, c->copy) != hipSuccess))
                    up_rc = fail(c, HJ_EHIP, "H2D");
                sent[t] = d; streamed[r] += cnt;
            }
        };
        const bool streaming = one_group && !no_overlap && nn[r] >= ((uint64_t)1 << 16);
        if (!host_level0_split_blocks(srcK[r], srcP[r], nn[r], level0_parts, host_threads, pk[r], pp[r], blk[r], psize[r], pin.empty() ? nullptr : &pin,
                                      streaming ? &pump : nullptr))
            return fail(c, HJ_ENOMEM, "the level-0 split of %llu tuples could not allocate its write-combining lines or start its %u host threads",
                        (unsigned long long)nn[r], host_threads);
        if (up_rc) return up_rc;
        if (streaming) { sent.push_back(~(uint64_t)0); sent_upto[r] = sent; }
        return 0;
    };
    // staging columns of one residency group: R's partitions of the group in cop_k/p[b], S's in seg_k/p[b], each a contiguous run
    auto ensure_staging = [&](uint64_t maxR, uint64_t maxS) -> int {
        for (int i = 0; i < (one_group ? 1 : 2); i++) { // one group: nothing to double-buffer
            RET(ensure(c, c->cop_k[i], (size_t)(maxR + PAD) * 4)); RET(ensure(c, c->cop_p[i], (size_t)(maxR + PAD) * 4));
            RET(ensure(c, c->seg_k[i], (size_t)(maxS + PAD) * 4)); RET(ensure(c, c->seg_p[i], (size_t)(maxS + PAD) * 4));
            // payloads that are not given: ones, filled on the device (the buffers keep them for every group of the call)
            if (!srcP[0]) HIPCHK(c, launch_fill(c->stream, (int32_t *)c->cop_p[i].p, maxR, HJ_PAYLOAD_ONES, 0));
            if (!srcP[1]) HIPCHK(c, launch_fill(c->stream, (int32_t *)c->seg_p[i].p, maxS, HJ_PAYLOAD_ONES, 0));
        }
        return 0;
    };
    // blocks -> one staging column pair: the blocks (any order: a group is joined as one relation) sorted by address, neighbours
    // merged into one copy.  Returns the tuples uploaded through *at.
    auto upload_blocks = [&](int r, int b, std::vector<const HostBlock *> &list, uint64_t *at) -> int {
        std::sort(list.begin(), list.end(), [](const HostBlock *x, const HostBlock *y) { return x->start < y->start; });
        Buf &dk = r ? c->seg_k[b] : c->cop_k[b], &dp = r ? c->seg_p[b] : c->cop_p[b];
        for (size_t i = 0; i < list.size();) {
            const uint64_t o = list[i]->start;
            uint64_t cnt = list[i]->count;
            for (i++; i < list.size() && list[i]->start == o + cnt; i++) cnt += list[i]->count;
            HIPCHK(c, hipMemcpyAsync((int32_t *)dk.p + *at, pk[r] + o, cnt * 4, hipMemcpyHostToDevice, c->copy));
            if (pp[r]) HIPCHK(c, hipMemcpyAsync((int32_t *)dp.p + *at, pp[r] + o, cnt * 4, hipMemcpyHostToDevice, c->copy));
            *at += cnt;
        }
        return 0;
    };
    // one group: the blocks of relation r that did not cross while its split was running (all of them when nothing was streamed)
    auto upload_rest = [&](int r) -> int {
        std::vector<const HostBlock *> rest;
        for (const HostBlock &hb : blk[r]) {
            bool gone = false;
            if (!sent_upto[r].empty()) {
                const size_t t = (size_t)(std::upper_bound(arena_of[r].begin(), arena_of[r].end(), hb.start) - arena_of[r].begin()) - 1;
                gone = hb.start < sent_upto[r][t];
            }
            if (!gone) rest.push_back(&hb);
        }
        uint64_t at = streamed[r];
        RET(upload_blocks(r, 0, rest, &at));
        if (at != nn[r]) return fail(c, HJ_EINVAL, "internal: %llu of %llu tuples of relation %d were uploaded", (unsigned long long)at, (unsigned long long)nn[r], r);
        return 0;
    };
    // ---- the pipeline: with one residency group both relations cross PCIe while the host is still splitting ----
    bool r_uploaded = false;
    if (!rc && one_group) rc = ensure_staging(nR, nS);
    mark[0] = since();
    const auto t_split0 = std::chrono::steady_clock::now();
    double split_s = 0;
    if (!rc) rc = split(0);
    mark[1] = since();
    split_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_split0).count();
    if (!rc && one_group && nR) {
        rc = upload_rest(0);
        r_uploaded = !rc;
    }
    mark[2] = since();
    const auto t_split1 = std::chrono::steady_clock::now();
    if (!rc) rc = split(1);
    split_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_split1).count();
    mark[3] = since();
    // bytes read + written by the scatter (keys, + payloads where given), like partition-primitives.cu:218
    c->host_split_gbs = split_s > 0 ? ((srcP[0] ? 16.0 : 8.0) * (double)nR + (srcP[1] ? 16.0 : 8.0) * (double)nS) / split_s / 1e9 : 0;
    // Residency groups (the reference decides which level-0 partitions are resident together with a knapsack over PARTS_RESIDENT
    // slots, groupOptimal2, pp.cu:307-468 / hjcp.cu:1357-1363): first-fit decreasing over the pairs' sizes — the big pairs open the
    // groups, the small ones fill them up — so a skewed split needs no more groups than its bytes ask for.  Every group is uploaded
    // pair by pair into contiguous staging columns and joined by ONE hj_join.
    std::vector<std::vector<uint32_t>> groups;
    if (!rc) {
        std::vector<uint32_t> by(level0_parts);
        for (uint32_t p = 0; p < level0_parts; p++) by[p] = p;
        auto size_of = [&](uint32_t p) { return psize[0][p] + psize[1][p]; };
        if (!one_group) std::stable_sort(by.begin(), by.end(), [&](uint32_t x, uint32_t y) { return size_of(x) > size_of(y); });
        std::vector<uint64_t> load;
        for (uint32_t p : by) {
            size_t g = 0;
            while (g < groups.size() && load[g] + size_of(p) > budget) g++;
            if (g == groups.size()) { groups.emplace_back(); load.push_back(0); }
            groups[g].push_back(p); load[g] += size_of(p);
        }
    }
    const uint32_t ngroups = (uint32_t)groups.size();
    c->coprocess_groups = ngroups;
    uint64_t maxp[2] = {0, 0};
    std::vector<uint64_t> gsz[2];
    for (int r = 0; r < 2 && !rc; r++) {
        gsz[r].assign(ngroups, 0);
        for (uint32_t g = 0; g < ngroups; g++) {
            for (uint32_t p : groups[g]) gsz[r][g] += psize[r][p];
            maxp[r] = std::max(maxp[r], gsz[r][g]);
        }
    }
    if (!rc && !r_uploaded) rc = ensure_staging(maxp[0], maxp[1]);
    // the blocks of partition p are blk[r][first[r][p] .. first[r][p + 1]) (the split returns them sorted by partition)
    std::vector<size_t> first[2];
    for (int r = 0; r < 2 && !rc; r++) {
        first[r].assign(level0_parts + 1, 0);
        for (const HostBlock &hb : blk[r]) first[r][hb.part + 1]++;
        for (uint32_t p = 0; p < level0_parts; p++) first[r][p + 1] += first[r][p];
    }
    auto upload = [&](uint32_t g) -> int {
        const int b = (int)(g & 1);
        for (int r = 0; r < 2; r++) {
            if (r == 0 && g == 0 && r_uploaded) continue;
            if (one_group) { RET(upload_rest(r)); continue; } // (S, or an empty R)
            std::vector<const HostBlock *> list;
            for (uint32_t p : groups[g])
                for (size_t i = first[r][p]; i < first[r][p + 1]; i++) list.push_back(&blk[r][i]);
            uint64_t at = 0;
            RET(upload_blocks(r, b, list, &at));
            if (at != gsz[r][g]) return fail(c, HJ_EINVAL, "internal: the blocks of residency group %u hold %llu tuples, its partitions %llu", g,
                                             (unsigned long long)at, (unsigned long long)gsz[r][g]);
        }
        HIPCHK(c, hipEventRecord(c->seg_ready[b], c->copy));
        return 0;
    };
    uint64_t tot_m = 0, tot_a = 0;
    if (!rc && ngroups) rc = upload(0);
    mark[4] = since();
    for (uint32_t g = 0; g < ngroups && !rc; g++) {
        const int b = (int)(g & 1);
        if (g + 1 < ngroups && (rc = upload(g + 1))) break; // the other pair of buffers was joined + synchronised last round
        hipError_t e = hipStreamWaitEvent(c->stream, c->seg_ready[b], 0);
        if (e != hipSuccess) { rc = fail(c, HJ_EHIP, "hipStreamWaitEvent: %s", hipGetErrorString(e)); break; }
        if ((rc = hj_bind_device(c, HJ_REL_R, (const int32_t *)c->cop_k[b].p, (const int32_t *)c->cop_p[b].p, gsz[0][g]))) break;
        if ((rc = hj_bind_device(c, HJ_REL_S, (const int32_t *)c->seg_k[b].p, (const int32_t *)c->seg_p[b].p, gsz[1][g]))) break;
        uint64_t m = 0, a = 0;
        if ((rc = hj_join(c, &m, &a))) break; // [sync]; the next group is already on its way
        tot_m += m; tot_a += a;
    }
    (void)hipStreamSynchronize(c->copy);
    mark[5] = since();
    if (getenv("HJ_DEBUG"))
        fprintf(stderr, "[hj] coprocess: setup %.2f ms, split R %.2f, R's rest queued %.2f, split S %.2f, groups + rest queued %.2f, joins done %.2f (streamed while splitting: %llu + %llu tuples)\n",
                mark[0], mark[1] - mark[0], mark[2] - mark[1], mark[3] - mark[2], mark[4] - mark[3], mark[5] - mark[4],
                (unsigned long long)streamed[0], (unsigned long long)streamed[1]);
    c->rel[0].bound = c->rel[1].bound = false; // the staging buffers are not user relations
    invalidate(c);
    if (rc) return rc;
    if (matches) *matches = tot_m;
    if (agg) *agg = tot_a;
    return HJ_OK;
}

int hj_coprocess_numa(const hj_ctx *c, int *nodes, int *gpu_node, int *pinned_cpus) {
    if (!c) return HJ_EINVAL;
    if (nodes) *nodes = c->numa_nodes;
    if (gpu_node) *gpu_node = c->numa_gpu_node;
    if (pinned_cpus) *pinned_cpus = c->numa_pinned_cpus;
    return HJ_OK;
}

int hj_coprocess_groups(const hj_ctx *c, uint32_t *groups) {
    if (!c) return HJ_EINVAL;
    if (groups) *groups = c->coprocess_groups;
    return HJ_OK;
}

int hj_host_split_throughput(const hj_ctx *c, double *gbs) {
    if (!c || !gbs) return HJ_EINVAL;
    *gbs = c->host_split_gbs;
    return HJ_OK;
}

} // extern "C"
