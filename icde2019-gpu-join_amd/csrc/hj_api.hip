// hj_api.hip — host side of libhj.so: context, HBM buffers, pass orchestration, the C ABI of
// include/hj.h (the two paths whose relations live in host memory: hj_stream.hip).  Replaces the host steps of outOfGPU_Join1_payload (hjcp.cu:802-994) and
// prepare_Relation_payload (jp.cu:1582-1613).  Everything between hj_partition and the final
// result copy is enqueued on one HIP stream with no host read of device data (same discipline as
// the reference's timed region, hjcp.cu:881-933: *buckets_used is only dereferenced on device).
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

namespace hjx {

int fail(hj_ctx *c, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int ensure(hj_ctx *c, Buf &b, size_t bytes) {
    if (bytes <= b.cap && b.p) return 0;
    const double t0 = now_ms();
    if (b.p) {
        HIPCHK(c, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    if (bytes == 0) bytes = 256;
    HIPCHK(c, hipMalloc(&b.p, bytes));
    b.cap = bytes;
    if (c) { c->prof.alloc_ms += now_ms() - t0; c->prof.allocs++; }
    static const bool dbg = getenv("HJ_DEBUG") != nullptr;
    if (dbg && bytes >= ((size_t)64 << 20)) fprintf(stderr, "[hj] (re)allocated %.2f GiB in %.1f ms\n", bytes / 1073741824.0, now_ms() - t0);
    return 0;
}

void release(Buf &b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

int kid_of(hj_ctx *c, const char *name) {
    for (size_t i = 0; i < c->kstats.size(); i++)
        if (c->kstats[i].name == name) return (int)i;
    KStat k;
    k.name = name;
    c->kstats.push_back(k);
    return (int)c->kstats.size() - 1;
}

hipEvent_t get_event(hj_ctx *c) {
    if (!c->pool.empty()) {
        hipEvent_t e = c->pool.back();
        c->pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr; // the caller skips timing this launch
    return e;
}

void account(hj_ctx *c, const Stamp &s) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
        KStat &k = c->kstats[s.kid];
        k.launches++;
        k.total_ms += ms;
        k.last_ms = ms;
    }
    c->pool.push_back(s.a);
    c->pool.push_back(s.b);
}

// Stamps whose end event has completed are folded into the per-kernel statistics and their events go back
// to the pool.  Called from every launch once the backlog is long, so a caller that never asks for
// hj_timings (or never synchronises) holds a bounded number of HIP events.
void resolve_completed(hj_ctx *c) {
    size_t done = 0;
    while (done < c->stamps.size() && hipEventQuery(c->stamps[done].b) == hipSuccess) account(c, c->stamps[done++]);
    if (done) c->stamps.erase(c->stamps.begin(), c->stamps.begin() + (long)done);
}

// RAII: HIP events on the context stream around one kernel launch
bool Timed::is_main(const char *n) {
    return !strncmp(n, "k_hist", 6) || !strncmp(n, "k_scatter", 9) || !strncmp(n, "k_part", 6) || !strncmp(n, "k_join_count", 12) ||
           !strncmp(n, "k_join_mat", 10) || !strncmp(n, "k_join_late", 11) || !strncmp(n, "k_np_", 5) || !strncmp(n, "k_split", 7) || !strncmp(n, "k_hot_build", 11);
}
// Each timed launch costs two event records on the stream; timing all ~35 launches of a step costs 4 %
// at 2^30 x 2^30 and 27 % at 2^24 (measured), so by default only the kernels that move data are timed.
Timed::Timed(hj_ctx *ctx, const char *name, hipStream_t stream, bool use_given)
    : c(ctx), on(ctx->events == 2 || (ctx->events == 1 && is_main(name))), st(use_given ? stream : ctx->stream) {
    if (!on) return;
    if (c->stamps.size() >= 256) resolve_completed(c);
    s.kid = kid_of(c, name);
    s.a = get_event(c);
    s.b = get_event(c);
    if (!s.a || !s.b) { // out of events: run untimed rather than record a null event
        if (s.a) c->pool.push_back(s.a);
        if (s.b) c->pool.push_back(s.b);
        on = false;
        return;
    }
    (void)hipEventRecord(s.a, st);
}
Timed::~Timed() {
    if (!on) return;
    (void)hipEventRecord(s.b, st);
    c->stamps.push_back(s);
}

// after a synchronisation of every stream that carries stamps
void resolve_stamps(hj_ctx *c) {
    for (auto &s : c->stamps) account(c, s);
    c->stamps.clear();
}

uint32_t ceil_log2(uint64_t x) {
    uint32_t k = 0;
    while (((uint64_t)1 << k) < x) k++;
    return k;
}

// radix bits from the relation sizes (replaces the constants of common.h:51-52)
void choose_bits(hj_ctx *c) {
    const hj_config &g = c->cfg;
    uint64_t nR = c->rel[0].n, nS = c->rel[1].n;
    if (c->force_build_r || g.build_side == 1) c->build = HJ_REL_R;
    else if (g.build_side == 2) c->build = HJ_REL_S;
    else c->build = (nS < nR) ? HJ_REL_S : HJ_REL_R;
    c->cap = g.lds_capacity ? g.lds_capacity : DEFAULT_CAP;
    if (c->cap > 65535) c->cap = 65535;
    c->nh = g.lds_heads ? g.lds_heads : DEFAULT_HEADS;
    while (c->nh & (c->nh - 1)) c->nh &= c->nh - 1; // round down to a power of two
    if (!c->nh) c->nh = 1;
    c->chunk = g.probe_chunk ? g.probe_chunk : DEFAULT_CHUNK;
    if (g.force_bits || g.bits1) {
        c->bits1 = g.bits1 > 9 ? 9 : g.bits1;
        c->bits2 = g.bits2 > 9 ? 9 : g.bits2;
        if (!c->bits1) { c->bits1 = c->bits2; c->bits2 = 0; }
        return;
    }
    // The partition count follows the SMALLER relation — which is the build relation unless hj_config.build_side names the larger one:
    // then the table side is chosen per partition (general items, plan_join), which makes the smaller partition build after all.
    // (The streaming probe path partitions R once for segments of any size: its bits follow |R|.)
    uint64_t nb = c->force_build_r ? c->rel[c->build].n : std::min(nR, nS);
    uint32_t total = nb > TARGET_PART ? ceil_log2((nb + TARGET_PART - 1) / TARGET_PART) : 0;
    if (total > 18) total = 18;
    if (total <= 9) {
        c->bits1 = total; c->bits2 = 0;
    } else {
        // two passes: the first is always 9 bits (512-way: one LDS line per digit), the second takes the rest, so that
        // a build partition averages 2048-4096 tuples — the join kernel's sweet spot (bit sweep at 2^26-2^28,
        // profiles/r2_bits_sweep.txt: a 7-bit floor for the second pass, round 1's rule, now costs 5-11 %)
        c->bits1 = 9;
        // Round 6 (profiles/r6_bits_split.txt): for two relations of SIMILAR size — the larger at most 1.5 x the smaller — up to 16 radix
        // bits, a first pass of 7 bits: their passes run side by side on two streams, and 128 + 128 pass-2 workgroups (one per parent) are
        // resident on the 256 CUs at once and finish together, where 512 + 512 take four turns with a prologue and a partial-line
        // epilogue each.  2^22-2^28 tuples a side: -1 ... -27 % per step (2^27: -1.3 %, 2^26: -7 %, 2^24: -10 %, 96 M: -19 %); never
        // more than 0.7 % behind 9 bits in the sweep.  With sizes apart the larger relation's pass 2 would be alone on the chip with 128
        // workgroups (2^27 x 2^31: +22 %): 9 bits stay; so do the streaming probe (segments of any size against one R) and the ranks
        // of the multi-GPU join (slices, not relations, run side by side there).
        const uint64_t nl = std::max(nR, nS), nsm = std::min(nR, nS);
        const uint32_t bs = c->bits1_similar;
        if (bs && bs < 9 && !c->force_build_r && !c->keep_nine && total > bs && total - bs <= 9 && nsm && nl <= nsm + nsm / 2) c->bits1 = bs;
        c->bits2 = total - c->bits1;
    }
    if (!g.lds_heads) { // hash-table heads ~ 2x the average build partition, power of two in [256, 4096]
        uint64_t avg = nb >> (c->bits1 + c->bits2);
        uint32_t nh = 256;
        while (nh < 2 * avg && nh < DEFAULT_HEADS) nh <<= 1;
        c->nh = nh;
    }
    // fewer than 16 radix bits: the LDS table stores full 4-byte keys (no 16-bit tags), 10 instead of 8 bytes per build
    // tuple.  With the default shape that is 62 KiB = 2 workgroups per CU; 4352 tuples + 2048 heads is 51.5 KiB = 3 per CU
    // (measured at 2^26-2^27: k_join_count 0.476 -> 0.439 ms, -8 %).
    // (round 6: 16-bit tags are exact whenever bits + log2(heads) >= 16 — plan_join — so this shape is left for the few-heads tables only)
    if ((c->tags_legacy ? c->bits1 + c->bits2 : c->bits1 + c->bits2 + ceil_log2(c->nh)) < 16 && !g.lds_capacity && !g.lds_heads) {
        c->cap = 4352;
        if (c->nh > 2048) c->nh = 2048;
    }
}

// one exact radix pass: in(keys,pays), parents = contiguous ranges poff[0..nparents] → out, child offsets → coff
int pass_prep(hj_ctx *c, int wsid, const int32_t *in_k, const int32_t *in_p, uint64_t n, const uint64_t *poff,
              uint32_t nparents, uint32_t shift, uint32_t P, uint32_t mask_or_n, int32_t *out_k, int32_t *out_p,
              PassArgs &pa) {
    if (nparents > (uint32_t)MAX_SEGS || P > (uint32_t)MAX_PARTS || P == 0) return fail(c, HJ_EINVAL, "pass fan-out out of range");
    if (n > ((uint64_t)1 << 34)) return fail(c, HJ_EINVAL, "relation too large for one GPU pass (n <= 2^34 tuples)");
    const uint32_t target_spans = c->target_spans ? c->target_spans : TARGET_SPANS;
    uint64_t span64 = (n + target_spans - 1) / target_spans;
    span64 = ((span64 + TILE - 1) / TILE) * TILE;
    if (span64 < (uint64_t)TILE) span64 = TILE;
    pa = PassArgs{};
    pa.keys = in_k; pa.pays = in_p; pa.nalloc = n;
    pa.sbeg = poff; pa.send = poff + 1; pa.nseg = nparents; pa.spp = 1; pa.nparents = nparents;
    pa.span = (uint32_t)span64;
    pa.max_spans = (uint32_t)((n + span64 - 1) / span64) + nparents;
    pa.shift = shift; pa.P = P; pa.mask_or_n = mask_or_n;
    pa.n_out = n;
    const uint64_t max_len = (uint64_t)pa.max_spans * P;
    const uint64_t nchunks = (max_len + SCAN_CHUNK - 1) / SCAN_CHUNK + 2;
    hj_ctx::PassWs &w = c->ws[wsid];
    RET(ensure(c, w.span_start, (size_t)(MAX_SEGS + 1) * 4));
    RET(ensure(c, w.hist, (size_t)max_len * 4));
    RET(ensure(c, w.chunk_sums, (size_t)nchunks * 8));
    RET(ensure(c, w.chunk_prefix, (size_t)nchunks * 8));
    pa.span_start = (uint32_t *)w.span_start.p;
    pa.hist = (uint32_t *)w.hist.p;
    pa.chunk_sums = (uint64_t *)w.chunk_sums.p;
    pa.chunk_prefix = (uint64_t *)w.chunk_prefix.p;
    pa.out_keys = out_k; pa.out_pays = out_p;
    return 0;
}

int pass_hist(hj_ctx *c, hipStream_t st, int mode, const PassArgs &pa, uint64_t n, uint64_t *coff) {
    const uint64_t max_len = (uint64_t)pa.max_spans * pa.P;
    { Timed t(c, "k_plan", st, true); HIPCHK(c, launch_plan(st, pa)); }
    { Timed t(c, "k_hist", st, true); HIPCHK(c, launch_hist(st, mode, pa)); }
    { Timed t(c, "k_scan", st, true); HIPCHK(c, launch_scan_u32(st, pa.hist, pa.span_start + pa.nseg, pa.P, max_len, pa.chunk_sums, pa.chunk_prefix, nullptr)); }
    { Timed t(c, "k_offsets", st, true); HIPCHK(c, launch_offsets(st, pa, n, coff)); }
    return 0;
}

int pass_scatter(hj_ctx *c, hipStream_t st, int mode, const PassArgs &pa) {
    { Timed t(c, "k_scatter_wc", st, true); HIPCHK(c, launch_scatter(st, mode, pa)); }
    return 0;
}

// beg/end: optional ranges of the child partitions for the join.
int run_pass(hj_ctx *c, int wsid, int mode, const int32_t *in_k, const int32_t *in_p, uint64_t n, const uint64_t *poff,
             uint32_t nparents, uint32_t shift, uint32_t P, uint32_t mask_or_n, int32_t *out_k, int32_t *out_p,
             uint64_t *coff, uint64_t *beg = nullptr, uint64_t *end = nullptr, const uint32_t *remap = nullptr) {
    PassArgs pa;
    RET(pass_prep(c, wsid, in_k, in_p, n, poff, nparents, shift, P, mask_or_n, out_k, out_p, pa));
    pa.beg = beg; pa.end = end; pa.remap = remap;
    RET(pass_hist(c, c->stream, mode, pa, n, coff));
    return pass_scatter(c, c->stream, mode, pa);
}

int check_rel(hj_ctx *c, int rel) {
    if (!c) return HJ_EINVAL;
    if (rel != HJ_REL_R && rel != HJ_REL_S) return fail(c, HJ_EINVAL, "rel must be HJ_REL_R or HJ_REL_S");
    return 0;
}

void invalidate(hj_ctx *c, int rel) {
    if (rel < 0) c->rel[0].partitioned = c->rel[1].partitioned = false;
    else c->rel[rel].partitioned = false;
    c->join_planned = false;
}

// Geometry of the histogram-free passes for a relation of n tuples (see hj_part.hip): spans of pass 1, slot
// capacities of both passes.  false when the slotted layout would not fit 32-bit positions.
bool plan_fast(const hj_ctx *c, uint64_t n, uint32_t P1, uint32_t P2, FastPlan &f) {
    if (n == 0) return false;
    // one workgroup per CU up to 2^28 tuples (fewer, longer spans: the per-span prologue and partial-line epilogue weigh
    // less: pass 1 -4..-12 % at 2^26-2^27), two per CU beyond (no difference measured at 2^30)
    const uint32_t target = c->target_spans ? c->target_spans : (n >= ((uint64_t)1 << 29) ? 512 : 256);
    uint64_t span = (n + target - 1) / target;
    span = ((span + TILE - 1) / TILE) * TILE;
    uint64_t nspans = (n + span - 1) / span;
    while (nspans > 1024) { span += TILE; nspans = (n + span - 1) / span; } // one LDS table entry per span in pass 2
    f.span = (uint32_t)span; f.nspans = (uint32_t)nspans;
    f.cap1 = fast_slot_cap((span + P1 - 1) / P1, P1);
    f.cap2 = fast_slot_cap((n + (uint64_t)P1 * P2 - 1) / ((uint64_t)P1 * P2), P2);
    f.sizeA = (uint64_t)P1 * nspans * f.cap1;
    f.sizeB = (uint64_t)P1 * P2 * f.cap2;
    const uint64_t lim = (uint64_t)1 << 36; // 32-bit line numbers inside the kernels: 2^37 tuples; half of that here
    return f.sizeA < lim && f.sizeB < lim && f.cap1 < ((uint32_t)1 << 31) && f.cap2 < ((uint32_t)1 << 31);
}

// ---- the sampled path: a relation known to be skewed, on the probe side ----
// One keys-only pass over every 8th 4096-tuple block gives a joint histogram of the final partition ids (the low b1+b2 key
// bits); the host turns it into per-digit slot capacities (expected count + sampling error + 8 sigma), per-digit LDS lines
// (k_scatter_wc's dealing rule, from the sample instead of an exact histogram), and — for pass 2 — a workgroup table that
// cuts every pass-1 digit into pieces of about one span, so that the digit holding a heavy hitter is spread over many
// workgroups.  Final partition p is then a LIST of ranges (one per piece of its parent): the join reads the probe side as
// ranges (JoinArgs.rpart).  Once per binding: later partition calls reuse the tables, with no histogram and no host read.
constexpr uint32_t SAMPLE_STRIDE = 8;

// ---- the heavy-hitter bypass: candidates ----
// Once per binding (first call of hj_join / hj_join_and_materialize that finds the relation skewed): up to 2^20 keys of the relation are
// counted in a device hash table, the keys seen at least 8 times come back to the host, which puts them — most frequent first — into the
// direct-mapped candidate table (a key whose slot is taken is left out: it takes the ordinary path).  k_hot_build then looks
// at the other relation as it is now: only candidates it holds exactly once are joined by pass 1.  Returns 0 with sp.hot_ready set, or
// 1: the bypass does not pay here (less than hot_min_share of the relation).
int hot_prepare(hj_ctx *c, Rel &R, const Rel &O, Rel::Sampled &sp) {
    const double t0 = now_ms(), a0 = c->prof.alloc_ms;
    sp.hot_ready = false; sp.hot_keys = 0; sp.hot_share = 0;
    uint32_t nsamp = 1u << 20;
    while ((uint64_t)nsamp * 4 > R.n && nsamp > 4096) nsamp >>= 1;
    if (R.n < 16 * (uint64_t)(nsamp >> 4)) return 1;
    const uint32_t slots = nsamp * 2, cap = nsamp / 8, thr = 8;
    hipStream_t st = c->stream;
    Buf tk, tc, outp;
    std::vector<uint2> pairs;
    uint32_t nout = 0;
    int rc = 0;
    do {
        if ((rc = ensure(c, tk, (size_t)slots * 4)) || (rc = ensure(c, tc, (size_t)slots * 4 + 16)) || (rc = ensure(c, outp, (size_t)cap * 8))) break;
        uint32_t *d_nout = (uint32_t *)tc.p + slots;
        hipError_t e = hipMemsetD32Async((hipDeviceptr_t)tk.p, (int)(uint32_t)HOT_NEVER, slots, st);
        if (e == hipSuccess) e = hipMemsetAsync(tc.p, 0, (size_t)slots * 4 + 16, st);
        if (e == hipSuccess) { Timed t(c, "k_hot_sample"); e = launch_hot_sample(st, R.in_k, R.n, nsamp, (uint32_t *)tk.p, (uint32_t *)tc.p, slots); }
        if (e == hipSuccess) { Timed t(c, "k_hot_collect"); e = launch_hot_collect(st, (const uint32_t *)tk.p, (const uint32_t *)tc.p, slots, thr, (uint2 *)outp.p, d_nout, cap); }
        if (e == hipSuccess) e = hipMemcpyAsync(&nout, d_nout, 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e == hipSuccess && nout) {
            pairs.resize(std::min(nout, cap));
            e = hipMemcpyAsync(pairs.data(), outp.p, pairs.size() * 8, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
        }
        if (e != hipSuccess) rc = fail(c, HJ_EHIP, "hot-key sample: %s", hipGetErrorString(e));
    } while (0);
    release(tk); release(tc); release(outp);
    if (rc) return rc;
    std::sort(pairs.begin(), pairs.end(), [](const uint2 &a, const uint2 &b) { return a.y != b.y ? a.y > b.y : a.x < b.x; });
    std::vector<uint32_t> cand(HOT_SLOTS), ccount(HOT_SLOTS, 0);
    for (uint32_t i = 0; i < HOT_SLOTS; i++) cand[i] = hot_filler(i);
    uint32_t nkeys = 0;
    uint64_t covered = 0;
    for (const uint2 &kc : pairs) {
        if (nkeys == HOT_SLOTS) break;
        const uint32_t sl = hot_slot(kc.x);
        if (ccount[sl]) continue;
        cand[sl] = kc.x; ccount[sl] = kc.y;
        nkeys++; covered += kc.y;
    }
    int verdict = 1;
    if (nkeys && (double)covered / nsamp >= c->hot_min_share) {
        // what the other relation holds for them right now
        RET(ensure(c, sp.hot_tab, (size_t)HOT_SLOTS * 4 * 3));
        uint32_t *d_cand = (uint32_t *)sp.hot_tab.p, *d_cnt = d_cand + HOT_SLOTS;
        int32_t *d_pay = (int32_t *)(d_cnt + HOT_SLOTS);
        std::vector<uint32_t> cnt(HOT_SLOTS, 0);
        HIPCHK(c, hipMemcpyAsync(d_cand, cand.data(), HOT_SLOTS * 4, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemsetAsync(d_cnt, 0, (size_t)HOT_SLOTS * 4 * 2, st));
        { Timed t(c, "k_hot_build"); HIPCHK(c, launch_hot_build(st, O.in_k, O.in_p, O.n, d_cand, d_cnt, d_pay, nullptr)); }
        HIPCHK(c, hipMemcpyAsync(cnt.data(), d_cnt, HOT_SLOTS * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st)); // (cand, cnt: pageable host vectors)
        uint64_t taken = 0;
        uint32_t unique = 0;
        for (uint32_t i = 0; i < HOT_SLOTS; i++) if (ccount[i] && cnt[i] == 1) { taken += ccount[i]; unique++; }
        sp.hot_keys = unique; sp.hot_share = (double)taken / nsamp;
        if (sp.hot_share >= c->hot_min_share) { sp.hot_ready = true; verdict = 0; }
        if (c->debug) fprintf(stderr, "[hj] hot keys: %u candidates (%.3f of the sample), %u unique in the other relation (%.3f): %s\n", nkeys, (double)covered / nsamp, unique, sp.hot_share, verdict ? "no bypass" : "bypass");
    }
    c->prof.plan_ms += (now_ms() - t0) - (c->prof.alloc_ms - a0);
    return verdict;
}

// rc 0: planned; 1: not plannable (positions); 2: planned for the bypass, but the rest of the relation still has a dominant digit
int plan_sampled_impl(hj_ctx *c, Rel &R, uint32_t b1, uint32_t b2, Rel::Sampled &sp);
int plan_sampled(hj_ctx *c, Rel &R, uint32_t b1, uint32_t b2, Rel::Sampled &sp) {
    const double t0 = now_ms(), a0 = c->prof.alloc_ms;
    const int rc = plan_sampled_impl(c, R, b1, b2, sp);
    c->prof.plan_ms += (now_ms() - t0) - (c->prof.alloc_ms - a0); // sampling kernel + read-back + host planning + table upload
    return rc;
}
int plan_sampled_impl(hj_ctx *c, Rel &R, uint32_t b1, uint32_t b2, Rel::Sampled &sp) {
    const uint32_t P1 = 1u << b1, P2 = 1u << b2, NP = P1 * P2;
    hipStream_t st = c->stream;
    // sample: device histogram -> host
    Buf hist;
    RET(ensure(c, hist, (size_t)NP * 4 + 8));
    std::vector<uint32_t> h(NP);
    uint64_t ns = 0;
    int rc = 0;
    do {
        hipError_t e = hipMemsetAsync(hist.p, 0, (size_t)NP * 4 + 8, st);
        const uint32_t *hc = sp.hot_ready ? (const uint32_t *)sp.hot_tab.p : nullptr; // the bypass takes these keys' tuples out in pass 1: they enter no bin
        if (e == hipSuccess) { Timed t(c, "k_sample_joint"); e = launch_sample_joint(st, R.in_k, R.n, b1 + b2, SAMPLE_STRIDE, (uint32_t *)hist.p, (uint64_t *)((uint32_t *)hist.p + NP), hc, hc ? hc + HOT_SLOTS : nullptr); }
        if (e == hipSuccess) e = hipMemcpyAsync(h.data(), hist.p, (size_t)NP * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(&ns, (uint32_t *)hist.p + NP, 8, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) rc = fail(c, HJ_EHIP, "sampling pass: %s", hipGetErrorString(e));
    } while (0);
    release(hist);
    if (rc) return rc;
    if (ns == 0) return fail(c, HJ_EHIP, "empty sample");
    // ---- geometry, all on the host ----
    // shares: (count + 1/2) / samples — the half count keeps unseen partitions alive.  NOT renormalised to sum 1: with few samples
    // per partition (small inputs) that would scale every observed share down by NP / (2 * samples), 4 % at 3 * 2^20 tuples and
    // 2^15 partitions — more than the margins of a large piece.  The shares sum to slightly more than 1: capacities err upwards.
    const double n = (double)R.n, tot = (double)ns;
    std::vector<double> f(NP), fd(P1, 0.0);
    for (uint32_t i = 0; i < NP; i++) { f[i] = ((double)h[i] + 0.5) / tot; fd[i / P2] += f[i]; }
    auto relerr = [](double samples) { return 4.0 / std::sqrt(std::max(1.0, samples)); }; // 4 sigma of the sampled share
    const double ROUNDT = 8192.0; // tuples per write-combining round (WC_THREADS * 4 * U)
    // lines of the 512 dealt to `cnt` digits with shares sh[]: a digit wants its expected arrivals per round + 30 %
    auto deal = [&](const double *sh, uint32_t cnt, std::vector<uint32_t> &lt, std::vector<uint32_t> &own, std::vector<uint32_t> &lines) {
        // every digit starts from what the uniform kernels give it (512/cnt lines) or what it needs, whichever is more; when
        // that is more than 512 lines the lightest digits give theirs back first (down to one), then everybody scales
        const uint32_t kdef = std::max<uint32_t>(1u, 512u / cnt);
        std::vector<uint32_t> need(cnt);
        uint32_t total = 0;
        lines.assign(cnt, 0);
        for (uint32_t d = 0; d < cnt; d++) {
            need[d] = std::min<uint32_t>((uint32_t)(1.3 * ROUNDT * sh[d] / 32.0) + 1, 512u);
            lines[d] = std::max(need[d], kdef);
            total += lines[d];
        }
        if (total > 512) {
            std::vector<uint32_t> by(cnt);
            for (uint32_t d = 0; d < cnt; d++) by[d] = d;
            std::stable_sort(by.begin(), by.end(), [&](uint32_t x, uint32_t y) { return sh[x] < sh[y]; });
            for (uint32_t round = 0; round < kdef && total > 512; round++)
                for (uint32_t i = 0; i < cnt && total > 512; i++) {
                    const uint32_t d = by[i];
                    if (lines[d] > need[d] && lines[d] > 1) { lines[d]--; total--; }
                }
        }
        if (total > 512) { // the needs alone exceed the lines: one each, the rest in proportion to the wish beyond one
            uint32_t tn = 0;
            for (uint32_t d = 0; d < cnt; d++) tn += need[d];
            for (uint32_t d = 0; d < cnt; d++) lines[d] = 1 + (uint32_t)((uint64_t)(need[d] - 1) * (512 - cnt) / (tn - cnt));
        }
        uint32_t first = 0;
        for (uint32_t d = 0; d < cnt; d++) {
            lt[d] = (lines[d] << 16) | first;
            for (uint32_t j = 0; j < lines[d]; j++) own[first + j] = d;
            first += lines[d];
        }
    };
    // pass 1
    // spans like the exact passes' (four per CU): the pieces are uneven under skew, many of them balance better than few
    const uint32_t target = c->target_spans ? c->target_spans : TARGET_SPANS;
    uint64_t span = (R.n + target - 1) / target;
    span = ((span + TILE - 1) / TILE) * TILE;
    uint64_t nspans = (R.n + span - 1) / span;
    while (nspans > 1024) { span += TILE; nspans = (R.n + span - 1) / span; }
    std::vector<uint32_t> lt1(P1), own1(512, 0xFFFFu), lines1, vbase1(P1), vcap1(P1);
    deal(fd.data(), P1, lt1, own1, lines1);
    uint64_t posA = 0;
    double maxfd = 0;
    for (uint32_t d = 0; d < P1; d++) {
        double hd = 0;
        for (uint32_t q = 0; q < P2; q++) hd += h[d * P2 + q];
        const double E = (double)span * fd[d];
        const uint64_t gran = (uint64_t)lines1[d] * 32;
        uint64_t cap = (uint64_t)(E * (1.0 + relerr(hd)) + 8.0 * std::sqrt(E) + 64.0);
        cap = ((cap + gran - 1) / gran) * gran + gran;
        vbase1[d] = (uint32_t)posA; vcap1[d] = (uint32_t)cap;
        posA += cap * nspans;
        maxfd = std::max(maxfd, fd[d]);
        if (posA >= ((uint64_t)1 << 32) - ((uint64_t)1 << 20)) return 1; // does not fit 32-bit positions: not plannable
    }
    if (sp.hot_ready && maxfd > 0.25) return 2; // (the bypass kernels have no wave-aggregated ranking: the plain sampled path takes such a relation)
    // pass 2
    std::vector<uint32_t> cbase2(NP), cap2(NP), lt2(NP), own2((size_t)P1 * 512, 0xFFFFu), heavy2(P1, 0), wg, rpart, pr0(NP), pnr(NP);
    uint64_t posB = 0;
    bool any_heavy = false, any_light = false;
    // parents in the order their workgroups should start: the slow ones (one dominant child: wave-aggregated ranking) and the
    // big ones first, so that the tail of the launch is made of small pieces
    std::vector<uint32_t> porder(P1);
    for (uint32_t d = 0; d < P1; d++) porder[d] = d;
    {
        std::vector<double> mxs(P1, 0.0);
        for (uint32_t d = 0; d < P1; d++) for (uint32_t q = 0; q < P2; q++) mxs[d] = std::max(mxs[d], f[d * P2 + q] / fd[d]);
        std::stable_sort(porder.begin(), porder.end(), [&](uint32_t x, uint32_t y) {
            const bool hx = mxs[x] > 0.25, hy = mxs[y] > 0.25;
            if (hx != hy) return hx;
            return fd[x] > fd[y];
        });
    }
    // Pieces: a pass-2 workgroup takes one parent's slots of a run of spans, and the launch runs ONE workgroup per CU (128 KiB of
    // LDS lines), handed out in index order.  Equal pieces of ~one span (the first version) left the launch at the mercy of
    // 803 pieces / 256 CUs = 3.14 rounds (78% busy).  Guided sizing instead: a piece is (work still to hand out) / (guide * CUs),
    // never below 2^18 tuples — large pieces first, small ones to level the tail (list scheduling: the makespan exceeds the ideal
    // by at most one of the last pieces).
    double remaining = n;
    const double guide = c->var_guide, cus = (double)std::max(1, c->ncu);
    for (uint32_t di = 0; di < P1; di++) {
        const uint32_t d = porder[di];
        const double Ed = n * fd[d];
        double target = (double)span;
        if (guide > 0) {
            target = std::max(262144.0, remaining / (guide * cus));
            double mxs = 0;
            for (uint32_t q = 0; q < P2; q++) mxs = std::max(mxs, f[d * P2 + q] / fd[d]);
            if (mxs > 0.25) target *= 0.5; // wave-aggregated ranking: slower per tuple
        }
        remaining -= Ed;
        uint32_t J = (uint32_t)std::min<double>((double)nspans, std::max(1.0, std::ceil(Ed / target)));
        const double maxshare = (double)((nspans + J - 1) / J) / (double)nspans;
        std::vector<double> sh(P2);
        double mx = 0;
        for (uint32_t q = 0; q < P2; q++) { sh[q] = f[d * P2 + q] / fd[d]; mx = std::max(mx, sh[q]); }
        std::vector<uint32_t> lt(P2), own(512, 0xFFFFu), lines;
        deal(sh.data(), P2, lt, own, lines);
        heavy2[d] = mx > 0.25;
        (heavy2[d] ? any_heavy : any_light) = true;
        uint64_t W = 0;
        for (uint32_t q = 0; q < P2; q++) {
            const double E = maxshare * Ed * sh[q];
            const uint64_t gran = (uint64_t)lines[q] * 32;
            uint64_t cap = (uint64_t)(E * (1.0 + relerr((double)h[d * P2 + q])) + 8.0 * std::sqrt(E) + 64.0);
            cap = ((cap + gran - 1) / gran) * gran + gran;
            cbase2[d * P2 + q] = (uint32_t)W; cap2[d * P2 + q] = (uint32_t)cap; lt2[d * P2 + q] = lt[q];
            W += cap;
        }
        for (uint32_t i = 0; i < 512; i++) own2[(size_t)d * 512 + i] = own[i];
        if (c->debug && J > 1) {
            uint32_t qm = 0;
            for (uint32_t q = 0; q < P2; q++) if (sh[q] > sh[qm]) qm = q;
            fprintf(stderr, "[hj] parent %u Ed %.0f J %u maxshare %.4f first wg %zu heavy %u child %u share %.4f cap %u lines %u W %llu\n", d, Ed, J, maxshare, wg.size() / 4, heavy2[d], qm, sh[qm], cap2[d * P2 + qm], lines[qm], (unsigned long long)W);
        }
        for (uint32_t q = 0; q < P2; q++) { pr0[d * P2 + q] = (uint32_t)(wg.size() / 4) * P2 + q; pnr[d * P2 + q] = J; } // the partition's ranges: stride P2
        for (uint32_t j = 0; j < J; j++) {
            const uint32_t s0 = (uint32_t)((uint64_t)j * nspans / J), s1 = (uint32_t)((uint64_t)(j + 1) * nspans / J);
            wg.push_back(d); wg.push_back(s0); wg.push_back(s1 - s0); wg.push_back((uint32_t)posB);
            for (uint32_t q = 0; q < P2; q++) rpart.push_back(d * P2 + q);
            posB += W;
            if (posB >= ((uint64_t)1 << 32) - ((uint64_t)1 << 20)) return 1;
        }
    }
    // upload: one table buffer
    const uint32_t nwg = (uint32_t)(wg.size() / 4), heavy1 = maxfd > 0.25 ? 1u : 0u;
    std::vector<uint32_t> tab;
    auto put = [&](const std::vector<uint32_t> &v) { while (tab.size() & 3) tab.push_back(0); const size_t at = tab.size(); tab.insert(tab.end(), v.begin(), v.end()); return at; };
    const size_t o_vb1 = put(vbase1), o_vc1 = put(vcap1), o_lt1 = put(lt1), o_ow1 = put(own1), o_h1 = put(std::vector<uint32_t>{heavy1});
    const size_t o_cb2 = put(cbase2), o_c2 = put(cap2), o_lt2 = put(lt2), o_ow2 = put(own2), o_h2 = put(heavy2), o_wg = put(wg), o_rp = put(rpart), o_r0 = put(pr0), o_nr = put(pnr);
    RET(ensure(c, sp.tab, tab.size() * 4));
    HIPCHK(c, hipMemcpyAsync(sp.tab.p, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipStreamSynchronize(st)); // tab is pageable
    const uint32_t *T = (const uint32_t *)sp.tab.p;
    sp.vbase1 = T + o_vb1; sp.vcap1 = T + o_vc1; sp.lt1 = T + o_lt1; sp.own1 = T + o_ow1; sp.heavy1_d = T + o_h1;
    sp.cbase2 = T + o_cb2; sp.cap2 = T + o_c2; sp.lt2 = T + o_lt2; sp.own2 = T + o_ow2; sp.heavy2 = T + o_h2; sp.wg2 = T + o_wg; sp.rpart = T + o_rp; sp.pr0 = T + o_r0; sp.pnr = T + o_nr;
    sp.n = R.n; sp.b1 = b1; sp.b2 = b2; sp.span = (uint32_t)span; sp.nspans = (uint32_t)nspans; sp.nwg2 = nwg; sp.nranges = nwg * P2;
    sp.sizeA = posA; sp.sizeB = posB; sp.heavy1 = heavy1 != 0; sp.any_heavy2 = any_heavy; sp.any_light2 = any_light; sp.sample_size = ns;
    sp.valid = true;
    return 0;
}

int ensure_part(hj_ctx *c, Buf &b, size_t bytes);

// ---- a look before the first optimistic attempt (round 6; VERDICT r5 item 4's first-call cost) ----
// The histogram-free passes are optimistic: on a heavily skewed relation the first call on a binding used to pay a whole failed attempt (both
// passes over the relation: 8-9 ms at 2^31 tuples), partition buffers sized for it (then re-allocated larger for the sampled path: device
// allocations are the bulk of a first call) and only then the sample.  For a large relation nothing is known about, 2^16 of its keys are
// counted by pass-1 and by pass-2 digit first (k_skew_probe + 4 KiB read back: ~50 us): a digit at more than twice its share (+6 sigma) means
// slots WILL overflow, and the relation goes to the sampled path at once.  Milder skew is not seen by so small a sample and is found by the attempt,
// as before.  Once per binding; never inside a captured step (the eager first call of a binding comes first).
int skew_probe(hj_ctx *c, Rel &R) {
    if (R.probed) return 0;
    R.probed = true;
    if (!c->skew_probe_log2 || R.n < ((uint64_t)1 << c->skew_probe_log2) || R.prefer_exact || !c->bits2 || !c->fast_path || c->cfg.exact_only ||
        c->bits1 + c->bits2 > 18 || c->bits1 > 9 || c->bits2 > 9) return 0;
    const double t0 = now_ms(), a0 = c->prof.alloc_ms;
    const uint32_t nsamp = 1u << 16, P1 = 1u << c->bits1, P2 = 1u << c->bits2;
    if (R.n < (uint64_t)nsamp * 16) return 0;
    RET(ensure(c, c->probe_hist, 1024 * 4));
    uint32_t h[1024];
    HIPCHK(c, hipMemsetAsync(c->probe_hist.p, 0, 1024 * 4, c->stream));
    { Timed t(c, "k_skew_probe"); HIPCHK(c, launch_skew_probe(c->stream, R.in_k, R.n, nsamp, c->bits1, c->bits2, (uint32_t *)c->probe_hist.p)); }
    HIPCHK(c, hipMemcpyAsync(h, c->probe_hist.p, (size_t)(P1 + P2) * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    auto over = [&](const uint32_t *v, uint32_t cnt) {
        const double mean = (double)nsamp / cnt;
        uint32_t mx = 0;
        for (uint32_t i = 0; i < cnt; i++) mx = std::max(mx, v[i]);
        return (double)mx > 2.0 * mean + 6.0 * std::sqrt(mean);
    };
    const bool skew1 = over(h, P1), skew2 = over(h + P1, P2);
    if (skew1 || skew2) R.prefer_exact = true; // what a failed attempt would have taught
    if (c->debug) fprintf(stderr, "[hj] skew probe of %llu tuples: pass-1 digits %s, pass-2 digits %s\n", (unsigned long long)R.n, skew1 ? "skewed" : "flat", skew2 ? "skewed" : "flat");
    c->prof.plan_ms += (now_ms() - t0) - (c->prof.alloc_ms - a0);
    return 0;
}

// the two launches; *done = false when the relation cannot take this path (the caller goes on to the exact passes)
int partition_sampled(hj_ctx *c, int r, uint32_t b1, uint32_t b2, uint32_t *flag, bool *done, bool want_hot) {
    Rel &R = c->rel[r];
    const Rel &O = c->rel[1 - r];
    *done = false;
    // the heavy-hitter bypass: wanted by the running entry point, this is the probe side, and the look at the keys said it pays
    if (want_hot && !R.sph.hot_ready) {
        const int hr = hot_prepare(c, R, O, R.sph);
        if (hr < 0) return hr;
        if (hr > 0) { R.hot_useless = true; want_hot = false; }
    }
    if (want_hot && (!R.sph.valid || R.sph.n != R.n || R.sph.b1 != b1 || R.sph.b2 != b2)) {
        R.sph.valid = false;
        const int rc = plan_sampled(c, R, b1, b2, R.sph);
        if (rc < 0) return rc;
        if (rc > 0 || !R.sph.valid) { R.sph.valid = false; R.sph.hot_ready = false; R.hot_useless = true; want_hot = false; } // the plain plan below
    }
    if (!want_hot && b1 + b2 > 17) return 0; // (18 radix bits only with the bypass: *done stays false, the exact passes take the relation)
    Rel::Sampled &sp = want_hot ? R.sph : R.sp;
    if (!sp.valid || sp.n != R.n || sp.b1 != b1 || sp.b2 != b2 || c->replan) {
        sp.valid = false;
        const int rc = plan_sampled(c, R, b1, b2, sp);
        if (rc < 0) return rc;
        if (rc > 0 || !sp.valid) { R.sampled_failed = true; return 0; }
    }
    const uint32_t P1 = 1u << b1, P2 = 1u << b2;
    hipStream_t st = c->stream;
    RET(ensure_part(c, R.a_k, (size_t)(sp.sizeA + PAD) * 4)); RET(ensure_part(c, R.a_p, (size_t)(sp.sizeA + PAD) * 4));
    RET(ensure_part(c, R.b_k, (size_t)(sp.sizeB + PAD) * 4)); RET(ensure_part(c, R.b_p, (size_t)(sp.sizeB + PAD) * 4));
    RET(ensure(c, R.s1beg, (size_t)P1 * sp.nspans * 8)); RET(ensure(c, R.s1end, (size_t)P1 * sp.nspans * 8));
    RET(ensure(c, sp.rbeg, (size_t)sp.nranges * 8)); RET(ensure(c, sp.rend, (size_t)sp.nranges * 8));
    FastArgs fa{};
    fa.keys = R.in_k; fa.pays = R.in_p; fa.n = R.n; fa.span = sp.span; fa.nspans = sp.nspans;
    fa.shift = b2; fa.P = P1;
    fa.out_keys = (int32_t *)R.a_k.p; fa.out_pays = (int32_t *)R.a_p.p;
    fa.obeg = (uint64_t *)R.s1beg.p; fa.oend = (uint64_t *)R.s1end.p; fa.ovf = flag;
    VarArgs va{};
    va.vbase = sp.vbase1; va.vcap = sp.vcap1; va.lt = sp.lt1; va.own = sp.own1; va.heavy = sp.heavy1_d; va.wg = nullptr;
    if (want_hot) { // what the other relation holds for the candidates NOW (4 bytes per tuple of it), then pass 1 joins their tuples itself
        uint64_t *sc = (uint64_t *)c->scalars.p;
        uint32_t *d_cand = (uint32_t *)sp.hot_tab.p, *d_cnt = d_cand + HOT_SLOTS;
        int32_t *d_pay = (int32_t *)(d_cnt + HOT_SLOTS);
        HIPCHK(c, hipMemsetAsync(d_cnt, 0, (size_t)HOT_SLOTS * 4, st));
        { Timed t(c, "k_hot_build"); HIPCHK(c, launch_hot_build(st, O.in_k, O.in_p, O.n, d_cand, d_cnt, d_pay, reinterpret_cast<unsigned long long *>(sc + 13))); }
        va.hot.mode = (uint32_t)c->hot_request; va.hot.cand = d_cand; va.hot.cnt = d_cnt; va.hot.pay = d_pay;
        va.hot.acc = reinterpret_cast<unsigned long long *>(sc + 13);
        va.hot.out_key = c->hot_out[0];
        va.hot.out_tab = r == HJ_REL_S ? c->hot_out[1] : c->hot_out[2]; // the other relation's payload column
        va.hot.out_str = r == HJ_REL_S ? c->hot_out[2] : c->hot_out[1];
        va.hot.out_cap = c->hot_cap;
        va.hot.cursor = reinterpret_cast<unsigned long long *>(sc + SC_CURSOR);
        va.hot.stamps = c->stamps_part2; // (experiment builds: per-workgroup phase times of the bypassing pass 1, 8 words each)
    }
    { Timed t(c, "k_part1_var"); HIPCHK(c, launch_part1_var(st, fa, va, sp.heavy1)); }
    FastArgs fb{};
    fb.keys = (const int32_t *)R.a_k.p; fb.pays = (const int32_t *)R.a_p.p;
    fb.sbeg = (const uint64_t *)R.s1beg.p; fb.send = (const uint64_t *)R.s1end.p; fb.nparents = P1; fb.spp = sp.nspans;
    fb.shift = 0; fb.P = P2;
    fb.out_keys = (int32_t *)R.b_k.p; fb.out_pays = (int32_t *)R.b_p.p;
    fb.obeg = (uint64_t *)sp.rbeg.p; fb.oend = (uint64_t *)sp.rend.p; fb.ovf = flag;
    VarArgs vb{};
    vb.vbase = sp.cbase2; vb.vcap = sp.cap2; vb.lt = sp.lt2; vb.own = sp.own2; vb.heavy = sp.heavy2; vb.wg = reinterpret_cast<const uint4 *>(sp.wg2);
    { Timed t(c, "k_part2_var"); HIPCHK(c, launch_part2_var(st, fb, vb, sp.nwg2, sp.any_heavy2, sp.any_light2)); }
    R.nparts = P1 * P2;
    R.nranges = sp.nranges; R.rpart = sp.rpart; R.pr0 = sp.pr0; R.pnr = sp.pnr; R.rstride = P2;
    R.part_k = (const int32_t *)R.b_k.p; R.part_p = (const int32_t *)R.b_p.p;
    R.part_beg = (const uint64_t *)sp.rbeg.p; R.part_end = (const uint64_t *)sp.rend.p;
    R.part_off = nullptr;
    R.n_alloc = sp.sizeB;
    R.pb1 = b1; R.pb2 = b2;
    R.partitioned = true; R.fast_tried = true; R.sampled = true;
    R.flag_unread = true;
    R.hot_mode = want_hot ? c->hot_request : 0;
    c->join_planned = false;
    *done = true;
    return 0;
}

// Partition buffers are allocated with 10 % of headroom: a relation whose slots overflow (skew) moves to the sampled geometry, whose
// buffers are 5-7 % larger than the uniform ones; without the headroom that switch frees and re-allocates every buffer of the
// relation — and device allocation is what a cold first call spends its time on when the driver has pages to scrub (19 ms ... 2.7 s
// for the 70 GiB of config 4, profiles/r4_first_call.txt).
int ensure_part(hj_ctx *c, Buf &b, size_t bytes) {
    if (bytes <= b.cap && b.p) return 0;
    return ensure(c, b, bytes + bytes / 10);
}

int partition_rel(hj_ctx *c, int r, FastPair *defer, bool assume_clean) {
    Rel &R = c->rel[r];
    if (defer) defer->used = false;
    if (!R.bound) return fail(c, HJ_EINVAL, "relation %d not loaded", r);
    // Positions inside the partition kernels are 32-bit LINE numbers (2^37 tuples): what bounds a relation is the card's memory,
    // checked here before anything is (re)allocated — pass-1 and final buffers of both columns, ~1.14 x 16 bytes per tuple on top
    // of the input (the reference's CLI accepts up to ULONG_MAX/4 tuples, main.cu:491-514).
    if (R.n > ((uint64_t)1 << 34)) return fail(c, HJ_EINVAL, "relation too large (n <= 2^34 tuples: pass-2 parents are addressed in 32-bit units)");
    if (c->force_sampled & (1 << r)) R.prefer_exact = true; // experiment knob: this relation takes the sampled path whatever its distribution
    else if (c->force_sampled & (4 << r)) { R.prefer_exact = false; R.sampled_failed = false; } // bits 2, 3: forget what was learned (back to the plain passes)
    choose_bits(c);
    RET(skew_probe(c, R));
    bool histogram_free = false; // the relation will take the plain or the sampled histogram-free passes (neither reads R.root)
    {
        const uint32_t P1g = 1u << c->bits1, P2g = 1u << c->bits2;
        FastPlan fg{};
        const bool fastg = c->bits2 && c->fast_path && !c->cfg.exact_only && !R.prefer_exact && plan_fast(c, R.n, P1g, P2g, fg);
        histogram_free = fastg || (R.prefer_exact && R.sp.valid && !R.sampled_failed && !R.force_exact && c->bits2 && c->fast_path && !c->cfg.exact_only);
        const uint64_t wantA = c->bits2 ? (std::max<uint64_t>(R.n, fastg ? fg.sizeA : 0) + PAD) * 4 : 0, wantB = (std::max<uint64_t>(R.n, fastg ? fg.sizeB : 0) + PAD) * 4;
        uint64_t grow = 0, give_back = 0;
        for (const Buf *b : {&R.a_k, &R.a_p}) if (wantA > b->cap) { grow += wantA + wantA / 10; give_back += b->cap; }
        for (const Buf *b : {&R.b_k, &R.b_p}) if (wantB > b->cap) { grow += wantB + wantB / 10; give_back += b->cap; }
        size_t free_b = 0, total_b = 0;
        if (grow && hipMemGetInfo(&free_b, &total_b) == hipSuccess && grow > (uint64_t)free_b + give_back)
            return fail(c, HJ_ENOMEM, "partitioning %llu tuples needs %.1f GiB of partition buffers (%.1f GiB still to allocate), %.1f of %.1f GiB are free on device %d",
                        (unsigned long long)R.n, (2.0 * wantA + 2.0 * wantB) / 1073741824.0, (grow - give_back) / 1073741824.0, free_b / 1073741824.0,
                        total_b / 1073741824.0, c->device);
    }
    hipStream_t st = c->stream;
    RET(ensure(c, R.root, 2 * 8));
    uint32_t *const flag = reinterpret_cast<uint32_t *>((uint64_t *)c->scalars.p + 8 + r); // travels with the result block
    // root offsets for the exact passes + the flag reset — not needed in front of histogram-free passes whose flag is known to be 0
    // (steady-state steps of hj_join: two launches fewer per step)
    // histogram_free is a PREDICTION of the path taken below (the sampled path can still decline): every branch that reads R.root asks
    // for it again, so a skipped launch is made up for in front of the exact passes
    bool root_done = false;
    auto set_root = [&]() -> int {
        if (root_done) return 0;
        Timed t(c, "k_set_root");
        HIPCHK(c, launch_set_root(st, (uint64_t *)R.root.p, R.n, flag));
        R.flag_unread = false; R.flag_maybe_set = false; // reset in stream order
        root_done = true;
        return 0;
    };
    if (!(assume_clean && histogram_free && !R.flag_unread && !R.flag_maybe_set)) RET(set_root());
    uint32_t b1 = c->bits1, b2 = c->bits2;
    // A relation known to be skewed (its histogram-free attempt overflowed) is split as evenly as possible between
    // the two exact passes: fewer than 512 digits per pass leave LDS lines to deal to the heavy digits (k_scatter_wc).
    // The final partition id is the same low b1+b2 key bits whatever the split, so the other relation is unaffected.
    if (R.prefer_exact && b2 && !c->cfg.force_bits && !c->cfg.bits1 && b1 + b2 <= 16) { const uint32_t t = b1 + b2; b1 = (t + 1) / 2; b2 = t - b1; }
    R.fast_tried = false;
    R.flag_known_good = false;
    R.part_off = nullptr;
    R.n_bound = 0;
    R.sampled = false; R.rpart = nullptr; R.pr0 = R.pnr = nullptr;
    R.hot_mode = 0;
    // known to be skewed: the sampled path — on either side of the join since round 4 (a build partition that is a list of ranges
    // is built into the LDS table piece by piece: general items, plan_join).  Up to 17 radix bits: at 16 (2^28 x 2^31 Zipf) it takes
    // 18.9 ms where the exact passes take 24.6, at 17 bits 23.6 against 26.9; at 18 bits both passes are 512-way, a heavy digit has
    // ONE LDS line and most of its tuples bypass it: no faster than the exact passes, the materialising join slower (profiles/
    // r4_sampled_16_17_bits.txt)
    // the heavy-hitter bypass: asked for by the running entry point (hj_join: count; hj_join_and_materialize: write), for the PROBE side
    // of the join, whose other side is there to be looked at
    const bool want_hot = c->hot_request && c->hot_enable && r == 1 - c->build && !R.hot_useless && c->rel[1 - r].bound && c->rel[1 - r].n &&
                          !c->rel[c->build].sampled && !c->rel[c->build].prefer_exact;
    // (round 6: with the bypass the sampled path also takes 18 radix bits — what made a 512-way pass under skew no faster than the exact passes
    // was the hot digits' tuples leaving tuple by tuple, and those are the tuples the bypass takes out; without it the cap stays at 17)
    if (R.prefer_exact && !R.sampled_failed && !R.force_exact && b2 && b1 + b2 <= (want_hot ? 18u : 17u) && c->fast_path && !c->cfg.exact_only &&
        R.n >= ((uint64_t)1 << 20)) {
        bool done = false;
        RET(partition_sampled(c, r, b1, b2, flag, &done, want_hot));
        if (done) return 0;
    }
    if (b1 == 0) { // nothing to partition: one partition = the input itself
        RET(set_root());
        R.part_k = R.in_k; R.part_p = R.in_p; R.nparts = 1; R.nranges = 1; R.n_alloc = R.n;
        R.part_off = (const uint64_t *)R.root.p;
        R.part_beg = R.part_off; R.part_end = R.part_off + 1;
        R.pb1 = R.pb2 = 0;
        R.partitioned = true;
        c->join_planned = false;
        return 0;
    }
    const uint32_t P1 = 1u << b1, P2 = 1u << b2;
    const uint32_t nparts = b2 ? P1 * P2 : P1;
    FastPlan f{};
    const bool fast = b2 && c->fast_path && !c->cfg.exact_only && !R.prefer_exact && plan_fast(c, R.n, P1, P2, f);
    const uint64_t elemsA = std::max<uint64_t>(R.n, fast ? f.sizeA : 0), elemsB = std::max<uint64_t>(R.n, fast ? f.sizeB : 0);
    RET(ensure_part(c, R.b_k, (size_t)(elemsB + PAD) * 4));
    RET(ensure_part(c, R.b_p, (size_t)(elemsB + PAD) * 4));
    RET(ensure(c, R.beg, (size_t)nparts * 8));
    RET(ensure(c, R.end, (size_t)nparts * 8));
    uint64_t *beg = (uint64_t *)R.beg.p, *end = (uint64_t *)R.end.p;
    if (b2 == 0) {
        RET(ensure(c, R.off2, (size_t)(P1 + 1) * 8));
        RET(set_root());
        RET(run_pass(c, r, 0, R.in_k, R.in_p, R.n, (const uint64_t *)R.root.p, 1, 0, P1, P1 - 1, (int32_t *)R.b_k.p,
                     (int32_t *)R.b_p.p, (uint64_t *)R.off2.p, beg, end));
    } else {
        RET(ensure_part(c, R.a_k, (size_t)(elemsA + PAD) * 4));
        RET(ensure_part(c, R.a_p, (size_t)(elemsA + PAD) * 4));
        if (fast) {
            // ---- histogram-free passes.  Optimistic: if a slot overflows, the kernels raise the relation's flag, the
            //      join's planning kernel then produces no work, and the host — which reads the flag with the next result
            //      block — redoes this relation with the exact passes (retry_overflowed) ----
            RET(ensure(c, R.s1beg, (size_t)P1 * f.nspans * 8));
            RET(ensure(c, R.s1end, (size_t)P1 * f.nspans * 8));
            uint32_t *ovf = flag;
            FastArgs fa{};
            fa.keys = R.in_k; fa.pays = R.in_p; fa.n = R.n; fa.span = f.span; fa.nspans = f.nspans;
            fa.shift = b2; fa.P = P1; fa.cap = f.cap1;
            fa.out_keys = (int32_t *)R.a_k.p; fa.out_pays = (int32_t *)R.a_p.p;
            fa.obeg = (uint64_t *)R.s1beg.p; fa.oend = (uint64_t *)R.s1end.p; fa.ovf = ovf;
            FastArgs fb{};
            fb.keys = (const int32_t *)R.a_k.p; fb.pays = (const int32_t *)R.a_p.p;
            fb.sbeg = (const uint64_t *)R.s1beg.p; fb.send = (const uint64_t *)R.s1end.p; fb.nparents = P1; fb.spp = f.nspans;
            fb.shift = 0; fb.P = P2; fb.cap = f.cap2;
            fb.out_keys = (int32_t *)R.b_k.p; fb.out_pays = (int32_t *)R.b_p.p;
            fb.obeg = beg; fb.oend = end; fb.ovf = ovf;
            fb.stamps = c->stamps_part2;
            if (defer) { defer->fa = fa; defer->fb = fb; defer->used = true; } // the caller launches (merged with the other relation's)
            else {
                { Timed t(c, "k_part1_fast"); HIPCHK(c, launch_part1_fast(st, fa)); }
                { Timed t(c, "k_part2_fast"); HIPCHK(c, launch_part2_fast(st, fb)); }
            }
            R.fast_tried = true;
            R.flag_unread = true;
        } else {
            RET(ensure(c, R.off1, (size_t)(P1 + 1) * 8));
            RET(ensure(c, R.off2, ((size_t)P1 * P2 + 1) * 8));
            RET(set_root());
            // pass 1 on key bits [b2, b2+b1), pass 2 on bits [0, b2): final partition id = low b1+b2 key
            // bits, pass-1 digit major — the order of jp.cu:402 ((pid << log_parts2) + j)
            RET(run_pass(c, r, 0, R.in_k, R.in_p, R.n, (const uint64_t *)R.root.p, 1, b2, P1, P1 - 1, (int32_t *)R.a_k.p,
                         (int32_t *)R.a_p.p, (uint64_t *)R.off1.p));
            RET(run_pass(c, r, 0, (const int32_t *)R.a_k.p, (const int32_t *)R.a_p.p, R.n, (const uint64_t *)R.off1.p, P1, 0, P2,
                         P2 - 1, (int32_t *)R.b_k.p, (int32_t *)R.b_p.p, (uint64_t *)R.off2.p, beg, end));
        }
    }
    R.nparts = nparts;
    R.nranges = nparts;
    R.part_k = (const int32_t *)R.b_k.p;
    R.part_p = (const int32_t *)R.b_p.p;
    R.part_beg = beg; R.part_end = end;
    R.part_off = R.fast_tried ? nullptr : (const uint64_t *)R.off2.p;
    R.n_alloc = elemsB;
    R.pb1 = b1; R.pb2 = b2;
    R.partitioned = true;
    c->join_planned = false;
    return 0;
}

// A relation whose histogram-free passes were queued is only known good once its flag has been read.  [sync]
// If the flag is up the relation is re-partitioned with the exact passes (asynchronously, on the context stream).
int resolve_layout(hj_ctx *c, Rel &R) {
    if (!R.fast_tried) return 0;
    uint32_t ovf = 0;
    HIPCHK(c, hipMemcpyAsync(&ovf, (uint64_t *)c->scalars.p + 8 + (&R - c->rel), 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    R.flag_unread = false; R.flag_maybe_set = ovf != 0;
    if (ovf) {
        if (R.sampled) R.sampled_failed = true;
        R.prefer_exact = true;
        drop_graph(c);
        RET(partition_rel(c, (int)(&R - c->rel)));
        if (R.fast_tried) return resolve_layout(c, R); // fast -> sampled -> exact: at most twice
    } else {
        R.flag_known_good = true;
    }
    return 0;
}

// introspection wants one gap-free range per partition: a sampled layout is redone with the exact passes
int exact_for_introspection(hj_ctx *c, Rel &R) {
    drop_graph(c); // resolve_layout / the exact redo rewrite the relation's layout state under a captured step
    RET(whole_partitions(c, 0));
    RET(resolve_layout(c, R));
    if (!R.sampled) return 0;
    R.force_exact = true;
    const int rc = partition_rel(c, (int)(&R - c->rel));
    R.force_exact = false;
    return rc;
}

// work-item list of the current partitions: k_join_plan + scan + k_join_expand (decompose_chains, jp.cu:843-874)
// gen_ok: the kernel that follows takes general items (the count kernel and the one-probe materialiser do; the late-materialising
// kernel does not: a sampled build side is redone with the exact passes for it)
// keep_cursor: the planning kernel leaves the materialiser's output cursor alone (a second probe-side group appends to the first's output)
int plan_join(hj_ctx *c, JoinArgs &a_out, bool &tag16, bool gen_ok, bool keep_cursor) {
    c->join_planned = false;
    Rel &B = c->rel[c->build], &Pb = c->rel[1 - c->build];
    if (!B.partitioned || !Pb.partitioned) return fail(c, HJ_EINVAL, "both relations must be partitioned before the join");
    if (B.sampled && !gen_ok) {
        B.force_exact = true;
        const int rc = partition_rel(c, c->build);
        B.force_exact = false;
        RET(rc);
    }
    if (B.nparts != Pb.nparts || B.pb1 + B.pb2 != Pb.pb1 + Pb.pb2)
        return fail(c, HJ_EINVAL, "relations were partitioned with different radix bits (%u+%u vs %u+%u): partition both after loading both",
                    B.pb1, B.pb2, Pb.pb1, Pb.pb2);
    hipStream_t st = c->stream;
    // A build relation known to be skewed (its slots overflowed once): GENERAL items.  Its partitions may be lists of ranges (sampled
    // path) that are built into the LDS table piece by piece, and a build partition that does not fit the table while the other
    // side's is smaller is joined with the roles flipped (jp.cu:929-1003) — one item per chunk of the big side instead of one
    // workgroup looping over hundreds of table chunks.
    // The same items serve a designated build relation that is the LARGER one (hj_config.build_side): the radix bits follow the smaller
    // relation (choose_bits), so its partitions would not fit the table either.
    const bool general = gen_ok && (B.sampled || B.prefer_exact || (B.n > Pb.n && !c->force_build_r)); // (streaming probe: R builds by design, bits follow |R|)
    // sampled probe side: several ranges per partition.  Whole ranges are packed into list items (one table build for all of them)
    const bool lists = Pb.sampled && Pb.pr0;
    const uint32_t nparts = (lists || general) ? Pb.nparts : Pb.nranges; // planning threads: partitions, or probe RANGES (== partitions unless sampled)
    const uint32_t rbits = B.pb1 + B.pb2;
    // The tag shortcut of jp.cu:1029 is exact with >= 16 radix bits (D2): what is left of a key fits the 16 stored bits.  Below that
    // the table stores full keys.  (Round 3 also built 16-bit tags at 14 / 15 radix bits, the extra key bits folded into the bucket
    // index: parity-green, measured no faster than full keys, removed — profiles/r3_tag_extra_ab.txt.)
    // Round 6: exact whenever rbits + log2(heads) >= 16 — the key bits [rbits, 16) the tag drops are part of the bucket index, and the tag
    // (key >> max(rbits, 16)) is only ever compared inside one bucket's chain.  Config 2 (2^27, 15 radix bits) and everything smaller now
    // run the 8-byte-per-hop tag kernels; full keys are left for tables of very few heads.
    tag16 = c->nh >= 16 && (c->tags_legacy ? rbits : rbits + ceil_log2(c->nh)) >= 16;
    const uint64_t held_b = B.n_bound ? B.n_bound : B.n, held_p = Pb.n_bound ? Pb.n_bound : Pb.n; // (multi-GPU: what this rank can hold, not the nominal size)
    const uint64_t max_items64 = (uint64_t)Pb.nranges + held_p / c->chunk + 1 + (general ? (uint64_t)B.nranges + held_b / c->chunk + 1 : 0); // flipped partitions are cut on the build relation
    if (max_items64 > 0x7FFFFFFFull) return fail(c, HJ_EINVAL, "too many work items");
    c->max_items = (uint32_t)max_items64;
    RET(ensure(c, c->items_cnt, (size_t)nparts * 4));
    RET(ensure(c, c->items, (size_t)c->max_items * sizeof(JoinItem)));
    const uint64_t nwave = (uint64_t)c->max_items * JOIN_WAVES;
    RET(ensure(c, c->wave_counts, (size_t)nwave * 8));
    RET(ensure(c, c->wave_agg, (size_t)nwave * 8));
    uint64_t nch = (nwave + SCAN_CHUNK - 1) / SCAN_CHUNK + 2;
    uint64_t nch2 = ((uint64_t)nparts + SCAN_CHUNK - 1) / SCAN_CHUNK + 2;
    if (nch2 > nch) nch = nch2;
    RET(ensure(c, c->jchunk_sums, (size_t)nch * 8));
    RET(ensure(c, c->jchunk_prefix, (size_t)nch * 8));
    uint64_t *sc = (uint64_t *)c->scalars.p;
    const size_t lds = join_lds_bytes(c->nh, c->cap, tag16);
    if (lds > 160 * 1024) return fail(c, HJ_EINVAL, "LDS hash table of %zu bytes exceeds 160 KiB", lds);
    HIPCHK(c, join_set_lds_limit(c->device, lds)); // per device, only ever raised (contexts share the functions)
    JoinArgs &a = a_out;
    a = JoinArgs{};
    a.bk = B.part_k; a.bp = B.part_p; a.bbeg = B.part_beg; a.bend = B.part_end; a.b_nalloc = B.n_alloc;
    a.pk = Pb.part_k; a.pp = Pb.part_p; a.pbeg = Pb.part_beg; a.pend = Pb.part_end; a.p_nalloc = Pb.n_alloc;
    a.rpart = Pb.sampled && !lists ? Pb.rpart : nullptr;
    if (lists) { a.pr0 = Pb.pr0; a.pnr = Pb.pnr; a.rstride = Pb.rstride; }
    if (B.sampled) { a.br0 = B.pr0; a.bnr = B.pnr; a.bstride = B.rstride; }
    a.general = general ? 1u : 0u;
    a.items = (const JoinItem *)c->items.p;
    a.n_items = sc + 0;
    a.radix_bits = rbits; a.cap = c->cap; a.nh = c->nh; a.chunk = c->chunk;
    a.bflag = (B.fast_tried && !B.flag_known_good) ? reinterpret_cast<const uint32_t *>(sc + 8 + c->build) : nullptr;
    a.pflag = (Pb.fast_tried && !Pb.flag_known_good) ? reinterpret_cast<const uint32_t *>(sc + 8 + (1 - c->build)) : nullptr;
    a.out_cursor = reinterpret_cast<unsigned long long *>(sc + SC_CURSOR);
    a.stamps = c->stamps_join;
    uint64_t *const zero_cursor = keep_cursor ? sc + 12 : sc + SC_CURSOR; // (sc[12]: a word nobody reads)
    const bool atomic_plan = c->plan_atomic;
    if (nparts <= 1024 && !Pb.sampled && !general) { // one workgroup's worth of partitions: plan + scan + expand in one single-workgroup launch
        Timed t(c, "k_join_plan");
        HIPCHK(c, launch_join_plan_fused(st, a, nparts, (JoinItem *)c->items.p, sc + 1, zero_cursor, sc + 0));
    } else if (atomic_plan && !Pb.sampled && !general) { // any partition count in ONE launch: item slots reserved on the counter pass 2 zeroed
        if (!c->items_zeroed) HIPCHK(c, hipMemsetAsync(sc + 0, 0, 8, st));
        Timed t(c, "k_join_plan");
        HIPCHK(c, launch_join_plan_atomic(st, a, nparts, (JoinItem *)c->items.p, sc + 1, zero_cursor, sc + 0));
    } else if (nparts <= 16384 && !Pb.sampled && !general) {
        Timed t(c, "k_join_plan");
        HIPCHK(c, launch_join_plan_fused(st, a, nparts, (JoinItem *)c->items.p, sc + 1, zero_cursor, sc + 0));
    } else {
    { Timed t(c, "k_join_plan"); HIPCHK(c, launch_join_plan(st, a, nparts, (uint32_t *)c->items_cnt.p, sc + 1, zero_cursor)); }
    { Timed t(c, "k_scan"); HIPCHK(c, launch_scan_u32(st, (uint32_t *)c->items_cnt.p, nullptr, nparts, nparts, (uint64_t *)c->jchunk_sums.p,
                                                      (uint64_t *)c->jchunk_prefix.p, sc + 0)); }
    { Timed t(c, "k_join_expand"); HIPCHK(c, launch_join_expand(st, a, nparts, (const uint32_t *)c->items_cnt.p,
                                                                (const uint64_t *)c->jchunk_prefix.p, (JoinItem *)c->items.p)); }
    }
    c->items_zeroed = false; // the counter holds this plan's item count now
    a.wave_counts = (uint64_t *)c->wave_counts.p;
    a.wave_agg = (uint64_t *)c->wave_agg.p;
    return 0;
}

// work-item list + per-wave counts + their sums
int run_count(hj_ctx *c, JoinArgs &a_out, bool &tag16, const JoinArgs *late = nullptr) {
    // general items are for the count kernel and the one-probe materialiser; the late-materialising kernel wants plain items
    RET(plan_join(c, a_out, tag16, !late));
    JoinArgs &a = a_out;
    hipStream_t st = c->stream;
    uint64_t *sc = (uint64_t *)c->scalars.p;
    if (late) {
        a.Db = late->Db; a.Dp = late->Dp; a.ncb = late->ncb; a.ncp = late->ncp; a.sb = late->sb; a.sp = late->sp;
        Timed t(c, "k_join_late_mat");
        HIPCHK(c, launch_join(st, a, c->max_items, tag16, 2));
    } else {
        Timed t(c, "k_join_count");
        HIPCHK(c, launch_join(st, a, c->max_items, tag16, 0));
    }
    // n_items is a uint64 on the device; the reductions take its low word as their length (little endian)
    const uint32_t *len = reinterpret_cast<const uint32_t *>(sc + 0);
    // matches and aggregate in one launch into sc[1], sc[2] (zeroed by k_join_plan)
    // ... plus what pass 1 of the probe side counted itself (the heavy-hitter bypass)
    const Rel &Pb = c->rel[1 - c->build];
    { Timed t(c, "k_sum2"); HIPCHK(c, launch_sum2(st, a.wave_counts, a.wave_agg, len, JOIN_WAVES, sc + 1, (!late && Pb.hot_mode == 1) ? sc + 13 : nullptr)); }
    return 0;
}

int fetch_scalars(hj_ctx *c) {
    // one copy: the results, the overflow flags of the two relations' histogram-free passes, what the heavy-hitter bypass counted, the
    // output cursor (a 128-byte line of its own, SC_CURSOR)
    HIPCHK(c, hipMemcpyAsync(c->h_scalars, c->scalars.p, (SC_CURSOR + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->redo_mask = 0;
    for (int r = 0; r < 2; r++) {
        Rel &R = c->rel[r];
        if (!R.fast_tried || R.flag_known_good) continue;
        R.flag_unread = false; R.flag_maybe_set = (uint32_t)c->h_scalars[8 + r] != 0;
        if ((uint32_t)c->h_scalars[8 + r]) { // slots overflowed: ranges invalid
            if (c->debug) fprintf(stderr, "[hj] rel %d overflow flag 0x%x (sampled %d) nwg2 %u nspans %u\n", r, (uint32_t)c->h_scalars[8 + r], (int)R.sampled, R.sp.nwg2, R.sp.nspans);
            // even the sampled capacities: the exact passes are what is left — unless the plan was the bypass's (the other relation may
            // have stopped holding a candidate exactly once: its tuples came back into slots sized without them): the plain sampled path then
            if (R.sampled && R.hot_mode) { R.hot_useless = true; R.sph.valid = false; R.sph.hot_ready = false; }
            else if (R.sampled) R.sampled_failed = true;
            R.prefer_exact = true; c->redo_mask |= 1u << r;
        }
        else R.flag_known_good = true;
    }
    resolve_completed(c); // every [sync] entry point folds finished stamps: the event backlog stays bounded
    return 0;
}

// run_count + result read-back; relations whose histogram-free partitions turn out to have overflowed (skew) are
// re-partitioned with the exact passes and the join runs again — once: the exact passes cannot overflow.
int count_and_fetch(hj_ctx *c, JoinArgs &a, bool &tag16, const JoinArgs *late = nullptr) {
    RET(run_count(c, a, tag16, late));
    RET(fetch_scalars(c));
    if (c->redo_mask && c->prof.t0 > 0 && c->prof.attempt_ms == 0) // the optimistic attempt of this call came back flagged: what it cost
        c->prof.attempt_ms = (now_ms() - c->prof.t0) - c->prof.alloc_ms;
    for (int attempt = 0; c->redo_mask; attempt++) { // histogram-free -> (sampled capacities with the heavy-hitter bypass ->) sampled capacities -> exact passes: at most three redos
        if (attempt == 3) return fail(c, HJ_EHIP, "exact passes reported an overflow");
        const uint32_t m = c->redo_mask;
        for (int r = 0; r < 2; r++)
            if (m & (1u << r)) RET(partition_rel(c, r));
        RET(run_count(c, a, tag16, late));
        RET(fetch_scalars(c));
    }
    return 0;
}

// hj_dist.hip: the count without the local redo — a raised overflow flag must reach every rank before anybody re-partitions.
// The flags are in c->h_scalars[8], [9] afterwards (and c->redo_mask); with a flag up the kernels did nothing: matches = 0.
int hj_join_count_noretry(hj_ctx *c, uint64_t *matches, uint64_t *agg) {
    JoinArgs a;
    bool tag16;
    RET(run_count(c, a, tag16));
    RET(fetch_scalars(c));
    if (matches) *matches = c->h_scalars[1];
    if (agg) *agg = c->h_scalars[2];
    return 0;
}

// ... and without any host read: the count kernels are enqueued, matches / aggregate end up in scalars[1], [2]
int hj_join_count_enqueue(hj_ctx *c) {
    JoinArgs a;
    bool tag16;
    return run_count(c, a, tag16);
}

int hj_join_materialize_enqueue(hj_ctx *c, int32_t *d_key, int32_t *d_payR, int32_t *d_payS, uint64_t cap, bool keep_cursor) {
    JoinArgs a;
    bool tag16;
    RET(whole_partitions(c, c->hot_request == 2 ? 2 : 0)); // (hj_join_and_materialize: pass 1 wrote the hot tuples to these very columns)
    RET(plan_join(c, a, tag16, true, keep_cursor));
    a.out_key = d_key;
    a.out_bpay = c->build == HJ_REL_R ? d_payR : d_payS;
    a.out_ppay = c->build == HJ_REL_R ? d_payS : d_payR;
    a.out_cap = cap;
    const size_t lds = join_mat_lds_bytes(a.nh, a.cap, tag16);
    if (lds > 160 * 1024) return fail(c, HJ_EINVAL, "LDS hash table of %zu bytes exceeds 160 KiB", lds);
    { Timed t(c, "k_join_materialize"); HIPCHK(c, launch_join_mat_reg(c->stream, a, c->max_items, tag16)); }
    c->join_planned = false;
    return 0;
}

// Partitions made under the heavy-hitter bypass lack the tuples pass 1 joined itself.  An entry point that needs them all (a
// materialising probe after hj_join, introspection, ...) gets the relation partitioned again without the bypass (hot_request is 0
// outside hj_join / hj_join_and_materialize).  ok_mode: a bypass mode the caller can work with (1: the counts are in scalars[13], [14]).
int whole_partitions(hj_ctx *c, int ok_mode) {
    for (int r = 0; r < 2; r++) {
        Rel &R = c->rel[r];
        if (R.partitioned && R.hot_mode && R.hot_mode != ok_mode) {
            drop_graph(c);
            const int saved = c->hot_request;
            c->hot_request = 0;
            const int rc = partition_rel(c, r);
            c->hot_request = saved;
            RET(rc);
        }
    }
    return 0;
}

void drop_graph(hj_ctx *c) {
    if (c->graph_exec) (void)hipGraphExecDestroy(c->graph_exec);
    c->graph_exec = nullptr;
    c->graph_warm = false;
}

void hj_invalidate_all(hj_ctx *c) {
    invalidate(c);
    for (int r = 0; r < 2; r++) { c->rel[r].fast_tried = false; c->rel[r].flag_known_good = false; c->rel[r].bound = false; c->rel[r].flag_unread = true; }
}

} // namespace hjx

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

const char *hj_version(void) { return "hj-mi355x 0.6 (gfx950)"; }

// every experiment / test knob the single-GPU path reads (DESIGN.md §9 lists them): once per context
static void read_knobs(hj_ctx *c) {
    if (const char *fp = getenv("HJ_FAST_PATH")) c->fast_path = atoi(fp); // 0: exact (histogram) passes only
    if (const char *ts = getenv("HJ_TARGET_SPANS")) c->target_spans = (uint32_t)atoi(ts);
    if (const char *fl = getenv("HJ_FORK_LOG2")) c->fork_log2 = (uint32_t)std::max(0, std::min(40, atoi(fl)));
    if (const char *ml = getenv("HJ_MERGE_LOG2")) c->merge_log2 = (uint32_t)std::max(0, std::min(40, atoi(ml)));
    if (const char *pa = getenv("HJ_PLAN_ATOMIC")) c->plan_atomic = atoi(pa) != 0;
    if (const char *vg = getenv("HJ_VAR_GUIDE")) c->var_guide = atof(vg);
    if (const char *fs = getenv("HJ_FORCE_SAMPLED")) c->force_sampled = atoi(fs); // bit 0: R, bit 1: S; bits 2, 3: forget what was learned
    c->replan = getenv("HJ_REPLAN") != nullptr;
    if (const char *tl = getenv("HJ_TAGS_LEGACY")) c->tags_legacy = atoi(tl) != 0;
    if (const char *ho = getenv("HJ_HOT")) c->hot_enable = atoi(ho);
    if (const char *hm = getenv("HJ_HOT_MIN_SHARE")) c->hot_min_share = atof(hm);
    if (const char *bs = getenv("HJ_BITS1_SIMILAR")) c->bits1_similar = (uint32_t)std::max(0, std::min(9, atoi(bs))); // 0 or 9: the first pass always takes 9 bits (rounds 2-5)
    if (const char *sk = getenv("HJ_SKEW_PROBE")) c->skew_probe_log2 = (uint32_t)std::max(0, std::min(63, atoi(sk))); // 0: no look before the first attempt
    c->debug = getenv("HJ_DEBUG") != nullptr;
}

int hj_create(hj_ctx **out, int device) {
    if (!out) return HJ_EINVAL;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return HJ_EHIP; // no GPU: fail loudly, there is no CPU fallback
    if (device < 0 || device >= ndev) return HJ_EINVAL;
    if (hipSetDevice(device) != hipSuccess) return HJ_EHIP;
    hj_ctx *c = new hj_ctx();
    c->device = device;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) { delete c; return HJ_EHIP; }
    c->stream = c->own_stream;
    if (hipMalloc(&c->scalars.p, SC_BYTES) != hipSuccess) { delete c; return HJ_ENOMEM; }
    c->scalars.cap = SC_BYTES;
    (void)hipMemset(c->scalars.p, 0, SC_BYTES);
    if (hipHostMalloc((void **)&c->h_scalars, SC_BYTES, hipHostMallocDefault) != hipSuccess) { delete c; return HJ_ENOMEM; }
    memset(c->h_scalars, 0, SC_BYTES);
    if (const char *ev = getenv("HJ_KERNEL_EVENTS")) c->events = !strcmp(ev, "all") ? 2 : (!strcmp(ev, "none") ? 0 : 1);
    read_knobs(c);
    (void)hipDeviceGetAttribute(&c->ncu, hipDeviceAttributeMultiprocessorCount, device);
    *out = c;
    return HJ_OK;
}

/* experiments only (tools/experiments/: same-context A/Bs switch a knob between two calls): read the environment knobs again.  The
 * library itself reads them ONCE, in hj_create — a variable set later by the host application changes nothing, and no entry point of the
 * path calls getenv. */
int hj_reload_knobs(hj_ctx *c) {
    if (!c) return HJ_EINVAL;
    read_knobs(c);
    return HJ_OK;
}

int hj_destroy(hj_ctx *c) {
    if (!c) return HJ_EINVAL;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    drop_graph(c);
    for (auto &s : c->stamps) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
    for (auto &e : c->pool) (void)hipEventDestroy(e);
    for (int r = 0; r < 2; r++) {
        Rel &R = c->rel[r];
        release(R.own_k); release(R.own_p); release(R.a_k); release(R.a_p); release(R.b_k); release(R.b_p);
        release(R.off1); release(R.off2); release(R.root);
        release(R.beg); release(R.end); release(R.s1beg); release(R.s1end);
        release(R.comp_k); release(R.comp_p); release(R.comp_off);
        release(R.sp.tab); release(R.sp.rbeg); release(R.sp.rend);
        release(R.sph.tab); release(R.sph.rbeg); release(R.sph.rend); release(R.sph.hot_tab);
    }
    for (int i = 0; i < 2; i++) { release(c->ws[i].span_start); release(c->ws[i].hist); release(c->ws[i].chunk_sums); release(c->ws[i].chunk_prefix); }
    for (int i = 0; i < 2; i++) { release(c->seg_k[i]); release(c->seg_p[i]); release(c->cop_k[i]); release(c->cop_p[i]); if (c->seg_ready[i]) (void)hipEventDestroy(c->seg_ready[i]); if (c->seg_joined[i]) (void)hipEventDestroy(c->seg_joined[i]); }
    release(c->seg_res);
    if (c->copy) (void)hipStreamDestroy(c->copy);
    if (c->aux) { (void)hipStreamSynchronize(c->aux); (void)hipStreamDestroy(c->aux); }
    if (c->aux_hi) { (void)hipStreamSynchronize(c->aux_hi); (void)hipStreamDestroy(c->aux_hi); }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->d2h) (void)hipStreamDestroy(c->d2h);
    for (int i = 0; i < 2; i++) {
        release(c->out_k[i]); release(c->out_p1[i]); release(c->out_p2[i]);
        if (c->out_ready[i]) (void)hipEventDestroy(c->out_ready[i]);
        if (c->out_free[i]) (void)hipEventDestroy(c->out_free[i]);
        if (c->host_k[i]) (void)hipHostFree(c->host_k[i]);
        if (c->host_p[i]) (void)hipHostFree(c->host_p[i]);
    }
    release(c->shard_root); release(c->shard_off);
    if (c->h_shard_off) (void)hipHostFree(c->h_shard_off);
    release(c->items_cnt); release(c->items); release(c->wave_counts); release(c->wave_agg);
    release(c->jchunk_sums); release(c->jchunk_prefix); release(c->scalars); release(c->probe_hist);
    if (c->h_scalars) (void)hipHostFree(c->h_scalars);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return HJ_OK;
}

const char *hj_error(const hj_ctx *c) { return c ? c->err.c_str() : "null context"; }

int hj_set_stream(hj_ctx *c, void *s) {
    if (!c) return HJ_EINVAL;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    resolve_stamps(c);
    c->stream = (s == HJ_OWN_STREAM) ? c->own_stream : (hipStream_t)s;
    drop_graph(c);
    return HJ_OK;
}

int hj_configure(hj_ctx *c, const hj_config *cfg) {
    if (!c || !cfg) return HJ_EINVAL;
    if (cfg->bits1 > 9 || cfg->bits2 > 9) return fail(c, HJ_EINVAL, "at most 9 radix bits per pass");
    if (cfg->lds_capacity > 65535) return fail(c, HJ_EINVAL, "lds_capacity must be <= 65535 (16-bit chain links)");
    if (cfg->lds_heads & (cfg->lds_heads - 1)) return fail(c, HJ_EINVAL, "lds_heads must be a power of two");
    if (cfg->reserved0 || cfg->reserved1) return fail(c, HJ_EINVAL, "hj_config.lds_stage / materialize_two_pass were removed (one-probe materialisation is the only form): the fields must be 0");
    c->cfg = *cfg;
    invalidate(c);
    drop_graph(c);
    return HJ_OK;
}

int hj_get_config(const hj_ctx *c, hj_config *cfg) {
    if (!c || !cfg) return HJ_EINVAL;
    hj_ctx *m = const_cast<hj_ctx *>(c);
    choose_bits(m);
    memset(cfg, 0, sizeof *cfg);
    cfg->bits1 = c->bits1; cfg->bits2 = c->bits2; cfg->force_bits = c->cfg.force_bits;
    cfg->build_side = c->build == HJ_REL_R ? 1 : 2;
    cfg->lds_capacity = c->cap; cfg->lds_heads = c->nh; cfg->probe_chunk = c->chunk;
    cfg->exact_only = c->cfg.exact_only || !c->fast_path;
    cfg->graph = c->cfg.graph;
    return HJ_OK;
}

int hj_sync(hj_ctx *c) {
    if (!c) return HJ_EINVAL;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return HJ_OK;
}

int hj_load_host(hj_ctx *c, int rel, const int32_t *keys, const int32_t *pays, uint64_t n, int mode) {
    RET(check_rel(c, rel));
    if (n && !keys) return fail(c, HJ_EINVAL, "keys == NULL");
    if (mode == HJ_PAYLOAD_GIVEN && n && !pays) return fail(c, HJ_EINVAL, "payload_mode GIVEN needs a payload column");
    if (mode < HJ_PAYLOAD_ONES || mode > HJ_PAYLOAD_GIVEN) return fail(c, HJ_EINVAL, "bad payload_mode");
    HIPCHK(c, hipSetDevice(c->device));
    Rel &R = c->rel[rel];
    RET(ensure(c, R.own_k, (size_t)(n + PAD) * 4));
    RET(ensure(c, R.own_p, (size_t)(n + PAD) * 4));
    if (n) HIPCHK(c, hipMemcpyAsync(R.own_k.p, keys, n * 4, hipMemcpyHostToDevice, c->stream));
    if (mode == HJ_PAYLOAD_GIVEN) {
        if (n) HIPCHK(c, hipMemcpyAsync(R.own_p.p, pays, n * 4, hipMemcpyHostToDevice, c->stream));
    } else {
        HIPCHK(c, launch_fill(c->stream, (int32_t *)R.own_p.p, n, mode, 0));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    R.in_k = (const int32_t *)R.own_k.p;
    R.in_p = (const int32_t *)R.own_p.p;
    R.n = n;
    R.bound = true;
    R.prefer_exact = false; // new data: the histogram-free passes get their chance again
    R.probed = false;
    R.sampled_failed = false; R.sp.valid = false; R.sph.valid = false; R.sph.hot_ready = false; R.hot_useless = false;
    { Rel &O = c->rel[1 - rel]; O.sph.valid = false; O.sph.hot_ready = false; O.hot_useless = false; } // its bypass plan looked at THIS relation's keys
    invalidate(c, rel);
    return HJ_OK;
}

int hj_bind_device(hj_ctx *c, int rel, const int32_t *d_keys, const int32_t *d_pays, uint64_t n) {
    RET(check_rel(c, rel));
    if (n && (!d_keys || !d_pays)) return fail(c, HJ_EINVAL, "device columns == NULL");
    if (((uintptr_t)d_keys | (uintptr_t)d_pays) & 15) return fail(c, HJ_EINVAL, "device columns must be 16-byte aligned");
    Rel &R = c->rel[rel];
    // re-binding the same columns keeps what the last run learned about them (skewed keys: exact passes at once)
    if (R.in_k != d_keys || R.in_p != d_pays || R.n != n) {
        R.prefer_exact = false; R.probed = false; R.sampled_failed = false; R.sp.valid = false; R.sph.valid = false; R.sph.hot_ready = false; R.hot_useless = false;
        Rel &O = c->rel[1 - rel]; // its bypass plan looked at THIS relation's keys
        O.sph.valid = false; O.sph.hot_ready = false; O.hot_useless = false;
    }
    R.in_k = d_keys; R.in_p = d_pays; R.n = n; R.bound = true;
    invalidate(c, rel);
    return HJ_OK;
}

int hj_partition(hj_ctx *c, int rel) {
    RET(check_rel(c, rel));
    HIPCHK(c, hipSetDevice(c->device));
    drop_graph(c); // a captured step describes the layouts ITS passes produce: re-partitioning outside hj_join voids it
    return partition_rel(c, rel);
}

int hj_join_count(hj_ctx *c, uint64_t *matches, uint64_t *agg) {
    if (!c) return HJ_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    RET(whole_partitions(c, 1)); // (partitions whose hot tuples were COUNTED by pass 1 are fine: their count is added in)
    JoinArgs a;
    bool tag16;
    RET(count_and_fetch(c, a, tag16));
    if (matches) *matches = c->h_scalars[1];
    if (agg) *agg = c->h_scalars[2];
    // a materialising call on the same partitions can skip the count (item list + per-wave counts are in HBM)
    c->last_args = a; c->last_tag16 = tag16; c->last_matches = c->h_scalars[1]; c->last_agg = c->h_scalars[2];
    c->join_planned = true;
    return HJ_OK;
}

} // extern "C"

namespace hjx {

// ONE probe: plan the work items, k_join_mat_reg finds, reserves and writes; the cursor comes back with the result block
int materialize_local(hj_ctx *c, int32_t *d_key, int32_t *d_payR, int32_t *d_payS, uint64_t cap, uint64_t *n_out) {
    RET(whole_partitions(c, 0));
    for (int attempt = 0; attempt < 4; attempt++) {
        JoinArgs a;
        bool tag16;
        if (c->join_planned) { // the item list of these partitions is on the device (a count ran): only the cursor is reset
            a = c->last_args; tag16 = c->last_tag16;
            HIPCHK(c, hipMemsetAsync((uint64_t *)c->scalars.p + SC_CURSOR, 0, 8, c->stream));
        } else {
            RET(plan_join(c, a, tag16));
        }
        a.out_key = d_key;
        a.out_bpay = c->build == HJ_REL_R ? d_payR : d_payS;
        a.out_ppay = c->build == HJ_REL_R ? d_payS : d_payR;
        a.out_cap = cap;
        const size_t lds = join_mat_lds_bytes(a.nh, a.cap, tag16);
        if (lds > 160 * 1024) return fail(c, HJ_EINVAL, "LDS hash table of %zu bytes exceeds 160 KiB", lds);
        { Timed t(c, "k_join_materialize"); HIPCHK(c, launch_join_mat_reg(c->stream, a, c->max_items, tag16)); }
        c->join_planned = false;
        RET(fetch_scalars(c)); // [sync]
        if (!c->redo_mask) break;
        if (attempt == 3) return fail(c, HJ_EHIP, "exact passes reported an overflow");
        const uint32_t m = c->redo_mask; // slots overflowed (skew): the kernel did nothing; sampled capacities / exact passes, then again
        for (int r = 0; r < 2; r++)
            if (m & (1u << r)) RET(partition_rel(c, r));
    }
    if (n_out) *n_out = c->h_scalars[SC_CURSOR];
    return 0;
}

} // namespace hjx

extern "C" {

int hj_join_materialize(hj_ctx *c, int32_t *d_key, int32_t *d_payR, int32_t *d_payS, uint64_t cap, uint64_t *n_out) {
    if (!c) return HJ_EINVAL;
    if (cap && (!d_key || !d_payR || !d_payS)) return fail(c, HJ_EINVAL, "output columns == NULL");
    HIPCHK(c, hipSetDevice(c->device));
    uint64_t n = 0;
    RET(materialize_local(c, d_key, d_payR, d_payS, cap, &n));
    if (n_out) *n_out = n;
    if (n > cap) return fail(c, HJ_ECAPACITY, "join produced %llu tuples, capacity %llu", (unsigned long long)n, (unsigned long long)cap);
    return HJ_OK;
}

} // extern "C"

namespace {

// hj_config.graph: the whole step (partition R, partition S, plan, build+probe, result copy: ~14-22 launches) captured once
// into a hipGraph and replayed with ONE host call per step — below ~2^24 tuples a step is bound by the host's launch rate, not
// by the kernels.  The graph is tied to everything its launches captured: the bound columns and sizes, the configuration, the
// stream, kernel events off, and the partition path each relation takes.  The first call on a binding runs eagerly (it
// allocates and learns whether a relation is skewed), the second captures, later ones replay.
// Both relations' partition passes.  S's passes go to a second stream and run beside R's; the join waits for both.  At small
// sizes the two chains of short dependent kernels overlap (2^20: 0.125 -> 0.103 ms); at large sizes every pass kernel is one or
// two waves of 1-per-CU workgroups, and the other relation's kernel fills the CUs that one kernel's tail leaves idle (same-box
// A/B, profiles/r3_fork_ab.txt: 2^26 -2 %, 2^27 -2.5...-5 %, 2^28 -1.2 %, 2^30 -1...-3.6 %).  HJ_FORK_LOG2 sets the largest
// |R|+|S| (log2) that forks.  Not with kernel events on (the instrumented steps time serial kernels).  Works the same under
// stream capture (fork / join by events).
int partition_both(hj_ctx *c) {
    const bool fork = c->rel[0].n + c->rel[1].n <= ((uint64_t)1 << c->fork_log2) && c->events == 0;
    // Small and medium inputs: ONE launch per pass for both relations (k_part1_fast2 / k_part2_fast2), one stream, no event fork and
    // join, no k_set_root in front of a relation whose flag is known to be 0, the join's item counter zeroed by pass 2: a steady-state
    // step is pass 1, pass 2, plan, build+probe, sum, result copy.  With kernel events on as well (round 6: the instrumented steps time the
    // kernels a step really launches — with a 7-bit first pass one relation's pass 2 alone is 128 workgroups on 256 CUs, which no step
    // ever runs).  Not for relations of very different sizes (their passes are arranged by priority below).
    if (c->merge_log2 && c->rel[0].n + c->rel[1].n <= ((uint64_t)1 << c->merge_log2) && c->rel[0].n && c->rel[1].n &&
        std::max(c->rel[0].n, c->rel[1].n) < 4 * std::min(c->rel[0].n, c->rel[1].n)) {
        FastPair pr[2];
        for (int r = 0; r < 2; r++) {
            const int rc = partition_rel(c, r, &pr[r], true);
            if (rc) { // a relation prepared earlier in this loop was marked partitioned, but its launches were deferred and never made
                invalidate(c);
                for (int q = 0; q < 2; q++) { c->rel[q].fast_tried = false; c->rel[q].flag_known_good = false; c->rel[q].flag_unread = true; }
                c->items_zeroed = false;
                return rc;
            }
        }
        if (pr[0].used && pr[1].used) {
            pr[0].fb.zero_items = (uint64_t *)c->scalars.p + 0;
            { Timed t(c, "k_part1_fast2"); HIPCHK(c, launch_part1_fast2(c->stream, pr[0].fa, pr[1].fa)); }
            { Timed t(c, "k_part2_fast2"); HIPCHK(c, launch_part2_fast2(c->stream, pr[0].fb, pr[1].fb)); }
            c->items_zeroed = true;
        } else {
            for (int r = 0; r < 2; r++)
                if (pr[r].used) {
                    { Timed t(c, "k_part1_fast"); HIPCHK(c, launch_part1_fast(c->stream, pr[r].fa)); }
                    { Timed t(c, "k_part2_fast"); HIPCHK(c, launch_part2_fast(c->stream, pr[r].fb)); }
                }
        }
        return 0;
    }
    if (!fork) {
        RET(partition_rel(c, HJ_REL_R));
        return partition_rel(c, HJ_REL_S);
    }
    // Relations of very different sizes (PK-FK 2^27 x 2^31): the LARGER one's passes go to a HIGH-priority second stream, so that the
    // small relation's workgroups are dispatched only where the big kernels have none left to dispatch — in their tails — instead of
    // taking CUs between the big kernel's first and second wave of workgroups (which cost more than running the two one after the
    // other: profiles/r4_asymmetric_streams.txt, 19.1 -> 17.5 ms per step on one box, 18.3 -> 18.0 on another).  Relations of
    // similar size share the chip at equal priority (a high-priority stream costs 2-3 % there).
    const int larger = c->rel[1].n >= c->rel[0].n ? HJ_REL_S : HJ_REL_R;
    const bool asym = std::max(c->rel[0].n, c->rel[1].n) >= 4 * std::min(c->rel[0].n, c->rel[1].n);
    hipStream_t &aux = asym ? c->aux_hi : c->aux;
    if (!aux) {
        int lo = 0, hi = 0; // (numerically lower = higher priority)
        if (asym && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi < lo) HIPCHK(c, hipStreamCreateWithPriority(&aux, hipStreamNonBlocking, hi));
        else HIPCHK(c, hipStreamCreateWithFlags(&aux, hipStreamNonBlocking));
    }
    if (!c->ev_fork) HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    if (!c->ev_join) HIPCHK(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    hipStream_t main = c->stream;
    HIPCHK(c, hipEventRecord(c->ev_fork, main));
    HIPCHK(c, hipStreamWaitEvent(aux, c->ev_fork, 0));
    const int auxrel = asym ? larger : HJ_REL_S;
    c->stream = aux;
    int rc = partition_rel(c, auxrel);
    c->stream = main;
    if (!rc) rc = partition_rel(c, 1 - auxrel);
    HIPCHK(c, hipEventRecord(c->ev_join, aux));
    HIPCHK(c, hipStreamWaitEvent(main, c->ev_join, 0));
    return rc;
}

struct GraphKey { const int32_t *k[2], *p[2]; uint64_t n[2]; hipStream_t st; bool pe[2]; };
bool same_key(const hj_ctx *c) {
    for (int r = 0; r < 2; r++)
        if (c->gkey_k[r] != c->rel[r].in_k || c->gkey_p[r] != c->rel[r].in_p || c->gkey_n[r] != c->rel[r].n || c->gkey_pe[r] != c->rel[r].prefer_exact) return false;
    return c->gkey_st == c->stream;
}
void set_key(hj_ctx *c) {
    for (int r = 0; r < 2; r++) { c->gkey_k[r] = c->rel[r].in_k; c->gkey_p[r] = c->rel[r].in_p; c->gkey_n[r] = c->rel[r].n; c->gkey_pe[r] = c->rel[r].prefer_exact; }
    c->gkey_st = c->stream;
}

int join_graph(hj_ctx *c, uint64_t *matches, uint64_t *agg, bool *done) {
    *done = false;
    if (!c->cfg.graph || c->events != 0 || !c->rel[0].bound || !c->rel[1].bound) return 0;
    if (!same_key(c)) { drop_graph(c); return 0; } // new binding: an eager call first
    if (!c->graph_warm) return 0;
    if (!c->graph_exec) {
        // capture: every buffer has its size from the eager call (ensure() is a no-op), no host read happens in between
        hipGraph_t gr = nullptr;
        if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { // e.g. HIP's legacy default stream
            (void)hipGetLastError();
            drop_graph(c);
            return 0;
        }
        int rc = partition_both(c);
        JoinArgs a;
        bool tag16 = false;
        if (!rc) rc = run_count(c, a, tag16);
        hipError_t e = rc ? hipSuccess : hipMemcpyAsync(c->h_scalars, c->scalars.p, (SC_CURSOR + 1) * 8, hipMemcpyDeviceToHost, c->stream);
        const hipError_t e2 = hipStreamEndCapture(c->stream, &gr);
        if (rc || e != hipSuccess || e2 != hipSuccess || !gr) { // not capturable: the eager path answers this call (and reports its own errors)
            if (gr) (void)hipGraphDestroy(gr);
            (void)hipGetLastError();
            drop_graph(c);
            invalidate(c); // whatever the aborted capture recorded about the partitions never ran
            c->err.clear();
            return 0;
        }
        e = hipGraphInstantiate(&c->graph_exec, gr, nullptr, nullptr, 0);
        (void)hipGraphDestroy(gr);
        if (e != hipSuccess) { c->graph_exec = nullptr; drop_graph(c); return 0; }
        c->last_args = a; c->last_tag16 = tag16;
    }
    HIPCHK(c, hipGraphLaunch(c->graph_exec, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->redo_mask = 0;
    for (int r = 0; r < 2; r++) {
        Rel &R = c->rel[r];
        if (!R.fast_tried) continue;
        R.flag_unread = false; R.flag_maybe_set = (uint32_t)c->h_scalars[8 + r] != 0;
        if ((uint32_t)c->h_scalars[8 + r]) { // the data under the binding changed: skewed now
            if (R.sampled && R.hot_mode) { R.hot_useless = true; R.sph.valid = false; R.sph.hot_ready = false; }
            else if (R.sampled) R.sampled_failed = true;
            R.prefer_exact = true; c->redo_mask |= 1u << r;
        }
        else R.flag_known_good = true;
    }
    if (c->redo_mask) { drop_graph(c); return 0; } // the eager path redoes the flagged relation
    if (matches) *matches = c->h_scalars[1];
    if (agg) *agg = c->h_scalars[2];
    c->last_matches = c->h_scalars[1]; c->last_agg = c->h_scalars[2];
    c->join_planned = true;
    *done = true;
    return 0;
}

} // namespace

extern "C" {

int hj_partition_both(hj_ctx *c) {
    if (!c) return HJ_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    drop_graph(c);
    return partition_both(c);
}

static int join_impl(hj_ctx *c, uint64_t *matches, uint64_t *agg) {
    bool done = false;
    RET(join_graph(c, matches, agg, &done));
    if (!done) {
        RET(partition_both(c));
        RET(hj_join_count(c, matches, agg));
        if (c->cfg.graph) { set_key(c); c->graph_warm = true; } // the next call on this binding may capture
    }
    return HJ_OK;
}

int hj_join(hj_ctx *c, uint64_t *matches, uint64_t *agg) {
    if (!c) return HJ_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    c->prof = hj_ctx::CallProf{};
    c->prof.t0 = now_ms();
    c->hot_request = 1; // a skewed probe side: pass 1 counts the matches of its heavy hitters itself (the result is all this call hands out)
    const int rc = join_impl(c, matches, agg);
    c->hot_request = 0;
    c->prof.total_ms = now_ms() - c->prof.t0;
    c->prof.t0 = 0;
    return rc;
}

static int join_and_materialize_impl(hj_ctx *c, int32_t *d_key, int32_t *d_payR, int32_t *d_payS, uint64_t cap, uint64_t *n_out) {
    uint64_t *sc = (uint64_t *)c->scalars.p;
    for (int attempt = 0; attempt < 4; attempt++) {
        // the output cursor starts at 0 BEFORE the passes: pass 1 of a skewed probe side appends the tuples of its heavy hitters, the probe
        // appends the rest behind them (plan_join leaves the cursor alone)
        HIPCHK(c, hipMemsetAsync(sc + SC_CURSOR, 0, 8, c->stream));
        RET(partition_both(c));
        RET(hj_join_materialize_enqueue(c, d_key, d_payR, d_payS, cap, true));
        RET(fetch_scalars(c)); // [sync]
        if (!c->redo_mask) break;
        if (attempt == 3) return fail(c, HJ_EHIP, "exact passes reported an overflow");
        // slots overflowed (skew): nothing of this attempt counts; the flagged relation takes its next path (sampled capacities, exact
        // passes), and BOTH are partitioned again — the tuples pass 1 wrote belong to a cursor that restarts
    }
    if (n_out) *n_out = c->h_scalars[SC_CURSOR];
    return 0;
}

int hj_join_and_materialize(hj_ctx *c, int32_t *d_key, int32_t *d_payR, int32_t *d_payS, uint64_t cap, uint64_t *n_out) {
    if (!c) return HJ_EINVAL;
    if (cap && (!d_key || !d_payR || !d_payS)) return fail(c, HJ_EINVAL, "output columns == NULL");
    HIPCHK(c, hipSetDevice(c->device));
    drop_graph(c);
    c->prof = hj_ctx::CallProf{};
    c->prof.t0 = now_ms();
    c->hot_request = 2; c->hot_out[0] = d_key; c->hot_out[1] = d_payR; c->hot_out[2] = d_payS; c->hot_cap = cap;
    uint64_t n = 0;
    const int rc = join_and_materialize_impl(c, d_key, d_payR, d_payS, cap, &n);
    c->hot_request = 0; c->hot_out[0] = c->hot_out[1] = c->hot_out[2] = nullptr; c->hot_cap = 0;
    c->prof.total_ms = now_ms() - c->prof.t0;
    c->prof.t0 = 0;
    RET(rc);
    if (n_out) *n_out = n;
    if (n > cap) return fail(c, HJ_ECAPACITY, "join produced %llu tuples, capacity %llu", (unsigned long long)n, (unsigned long long)cap);
    return HJ_OK;
}

/* experiments only: device buffers (4 x uint64 per work item of the join / per parent of pass 2) that a library built with -DHJ_STAMPS
 * (`make stamps`) fills with per-workgroup start / end times and hardware ids; the shipped build ignores them. */
int hj_debug_set_stamps(hj_ctx *c, void *d_join, void *d_part2) {
    if (!c) return HJ_EINVAL;
    c->stamps_join = (unsigned long long *)d_join; c->stamps_part2 = (unsigned long long *)d_part2;
    c->join_planned = false;
    drop_graph(c);
    return HJ_OK;
}

int hj_hot_stats(const hj_ctx *c, int *mode, uint32_t *keys, double *share, uint64_t *matches) {
    if (!c) return HJ_EINVAL;
    const hj_ctx::Rel *H = nullptr;
    for (int r = 0; r < 2; r++) if (c->rel[r].partitioned && c->rel[r].hot_mode) H = &c->rel[r];
    if (mode) *mode = H ? H->hot_mode : 0;
    if (keys) *keys = H ? H->sph.hot_keys : 0;
    if (share) *share = H ? H->sph.hot_share : 0.0;
    if (matches) *matches = (H && H->hot_mode == 1) ? c->h_scalars[13] : 0;
    return HJ_OK;
}

int hj_last_call_breakdown(const hj_ctx *c, double *alloc_ms, uint32_t *allocations, double *failed_attempt_ms, double *sample_plan_ms, double *total_ms) {
    if (!c) return HJ_EINVAL;
    if (alloc_ms) *alloc_ms = c->prof.alloc_ms;
    if (allocations) *allocations = c->prof.allocs;
    if (failed_attempt_ms) *failed_attempt_ms = c->prof.attempt_ms;
    if (sample_plan_ms) *sample_plan_ms = c->prof.plan_ms;
    if (total_ms) *total_ms = c->prof.total_ms;
    return HJ_OK;
}

int hj_device_malloc(hj_ctx *c, void **d_ptr, uint64_t bytes) {
    if (!c || !d_ptr) return HJ_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMalloc(d_ptr, bytes ? bytes : 256));
    return HJ_OK;
}

int hj_device_free(hj_ctx *c, void *d_ptr) {
    if (!c) return HJ_EINVAL;
    if (d_ptr) HIPCHK(c, hipFree(d_ptr));
    return HJ_OK;
}

int hj_memcpy_d2h(hj_ctx *c, void *h_dst, const void *d_src, uint64_t bytes) {
    if (!c) return HJ_EINVAL;
    if (bytes) HIPCHK(c, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return HJ_OK;
}

int hj_memcpy_h2d(hj_ctx *c, void *d_dst, const void *h_src, uint64_t bytes) {
    if (!c) return HJ_EINVAL;
    if (bytes) HIPCHK(c, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return HJ_OK;
}

int hj_join_late_materialize(hj_ctx *c, const int32_t *d_Dr, uint32_t ncolR, uint64_t strideR, const int32_t *d_Ds,
                             uint32_t ncolS, uint64_t strideS, uint64_t *matches, uint64_t *sum) {
    if (!c) return HJ_EINVAL;
    if ((ncolR && !d_Dr) || (ncolS && !d_Ds)) return fail(c, HJ_EINVAL, "extra-column table == NULL");
    if ((ncolR && strideR < c->rel[HJ_REL_R].n) || (ncolS && strideS < c->rel[HJ_REL_S].n))
        return fail(c, HJ_EINVAL, "column stride smaller than the relation");
    HIPCHK(c, hipSetDevice(c->device));
    RET(whole_partitions(c, 0));
    choose_bits(c);
    JoinArgs late{};
    const bool r_builds = c->build == HJ_REL_R;
    late.Db = r_builds ? d_Dr : d_Ds; late.ncb = r_builds ? ncolR : ncolS; late.sb = r_builds ? strideR : strideS;
    late.Dp = r_builds ? d_Ds : d_Dr; late.ncp = r_builds ? ncolS : ncolR; late.sp = r_builds ? strideS : strideR;
    JoinArgs a;
    bool tag16;
    RET(count_and_fetch(c, a, tag16, &late));
    if (matches) *matches = c->h_scalars[1];
    if (sum) *sum = c->h_scalars[2];
    return HJ_OK;
}

int hj_join_nonpartitioned(hj_ctx *c, int kind, uint64_t *matches, uint64_t *agg) {
    if (!c) return HJ_EINVAL;
    if (kind != 0 && kind != 1) return fail(c, HJ_EINVAL, "kind must be 0 (perfect array) or 1 (global chained table)");
    if (!c->rel[0].bound || !c->rel[1].bound) return fail(c, HJ_EINVAL, "load or bind both relations first");
    HIPCHK(c, hipSetDevice(c->device));
    choose_bits(c);
    const Rel &B = c->rel[c->build], &Pb = c->rel[1 - c->build];
    uint64_t *sc = (uint64_t *)c->scalars.p;
    HIPCHK(c, hipMemsetAsync(sc + 5, 0, 3 * 8, c->stream)); // [5] max key, [6] matches, [7] agg
    Buf t1, t2;
    int rc = 0;
    if (kind == 0) {
        // direct-address table over [0, max build key]: the best case of a non-partitioned join; needs
        // unique, non-negative build keys (jp.cu:620-627 "perfect hashing")
        hipError_t e = launch_np_max(c->stream, B.in_k, B.n, reinterpret_cast<uint32_t *>(sc + 5));
        if (e != hipSuccess) return fail(c, HJ_EHIP, "k_np_max: %s", hipGetErrorString(e));
        RET(fetch_scalars(c));
        const uint64_t range = (uint32_t)c->h_scalars[5] + (uint64_t)1;
        if (range > ((uint64_t)1 << 31)) return fail(c, HJ_EINVAL, "perfect array needs non-negative build keys");
        rc = ensure(c, t1, (size_t)range * 4);
        if (!rc) {
            hipError_t e2 = hipMemsetAsync(t1.p, 0, (size_t)range * 4, c->stream);
            if (e2 == hipSuccess) { Timed t(c, "k_np_perfect"); e2 = launch_np_perfect(c->stream, B.in_k, B.n, B.in_p, Pb.in_k, Pb.in_p, Pb.n, (int32_t *)t1.p, range, sc + 6); }
            if (e2 != hipSuccess) rc = fail(c, HJ_EHIP, "perfect array: %s", hipGetErrorString(e2));
        }
    } else {
        uint32_t lg = B.n > 1 ? ceil_log2(B.n) : 1; // one slot per build tuple (the reference passes p = 27 for 2^27)
        if (lg > 31) lg = 31;
        rc = ensure(c, t1, ((size_t)1 << lg) * 4);
        if (!rc) rc = ensure(c, t2, (size_t)(B.n + 1) * 4);
        if (!rc) {
            hipError_t e2 = hipMemsetAsync(t1.p, 0, ((size_t)1 << lg) * 4, c->stream);
            if (e2 == hipSuccess) { Timed t(c, "k_np_chained"); e2 = launch_np_chained(c->stream, B.in_k, B.in_p, B.n, Pb.in_k, Pb.in_p, Pb.n, lg, (int32_t *)t1.p, (int32_t *)t2.p, sc + 6); }
            if (e2 != hipSuccess) rc = fail(c, HJ_EHIP, "chained table: %s", hipGetErrorString(e2));
        }
    }
    if (!rc) rc = fetch_scalars(c);
    release(t1);
    release(t2);
    if (rc) return rc;
    if (matches) *matches = c->h_scalars[6];
    if (agg) *agg = c->h_scalars[7];
    return HJ_OK;
}

} // extern "C"

extern "C" {

int hj_get_partitions(hj_ctx *c, int rel, const int32_t **d_keys, const int32_t **d_pays, const uint64_t **d_offsets,
                      uint64_t *nparts) {
    RET(check_rel(c, rel));
    Rel &R = c->rel[rel];
    if (!R.partitioned) return fail(c, HJ_EINVAL, "relation %d not partitioned", rel);
    HIPCHK(c, hipSetDevice(c->device));
    RET(exact_for_introspection(c, R));
    if (R.fast_tried) {
        // slotted layout (histogram-free passes): hand out a gap-free copy with contiguous offsets
        const uint32_t np = R.nparts;
        std::vector<uint64_t> hb(np), he(np), off((size_t)np + 1);
        HIPCHK(c, hipMemcpyAsync(hb.data(), R.part_beg, (size_t)np * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(he.data(), R.part_end, (size_t)np * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        off[0] = 0;
        for (uint32_t i = 0; i < np; i++) off[i + 1] = off[i] + (he[i] - hb[i]);
        if (off[np] != R.n) return fail(c, HJ_EHIP, "partition sizes add up to %llu, relation has %llu tuples",
                                        (unsigned long long)off[np], (unsigned long long)R.n);
        RET(ensure(c, R.comp_k, (size_t)(R.n + PAD) * 4));
        RET(ensure(c, R.comp_p, (size_t)(R.n + PAD) * 4));
        RET(ensure(c, R.comp_off, ((size_t)np + 1) * 8));
        HIPCHK(c, hipMemcpyAsync(R.comp_off.p, off.data(), ((size_t)np + 1) * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, launch_compact(c->stream, R.part_k, R.part_p, R.part_beg, R.part_end, np, (const uint64_t *)R.comp_off.p,
                                 (int32_t *)R.comp_k.p, (int32_t *)R.comp_p.p));
        HIPCHK(c, hipStreamSynchronize(c->stream)); // `off` is pageable host memory
        if (d_keys) *d_keys = (const int32_t *)R.comp_k.p;
        if (d_pays) *d_pays = (const int32_t *)R.comp_p.p;
        if (d_offsets) *d_offsets = (const uint64_t *)R.comp_off.p;
        if (nparts) *nparts = np;
        return HJ_OK;
    }
    if (d_keys) *d_keys = R.part_k;
    if (d_pays) *d_pays = R.part_p;
    if (d_offsets) *d_offsets = R.part_off;
    if (nparts) *nparts = R.nparts;
    return HJ_OK;
}

int hj_partition_layout(hj_ctx *c, int rel, int *slotted) {
    RET(check_rel(c, rel));
    Rel &R = c->rel[rel];
    if (!R.partitioned) return fail(c, HJ_EINVAL, "relation %d not partitioned", rel);
    HIPCHK(c, hipSetDevice(c->device));
    RET(resolve_layout(c, R));
    if (slotted) *slotted = R.sampled ? 2 : (R.fast_tried ? 1 : 0);
    return HJ_OK;
}

int hj_enable_timings(hj_ctx *c, int level) {
    if (!c || level < 0 || level > 2) return HJ_EINVAL;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    resolve_stamps(c);
    c->events = level;
    drop_graph(c);
    return HJ_OK;
}

int hj_timings_reset(hj_ctx *c) {
    if (!c) return HJ_EINVAL;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    resolve_stamps(c);
    for (auto &k : c->kstats) { k.launches = 0; k.total_ms = 0; k.last_ms = 0; }
    return HJ_OK;
}

int hj_timings(hj_ctx *c, hj_kernel_time *out, uint32_t cap, uint32_t *n) {
    if (!c) return HJ_EINVAL;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    resolve_stamps(c);
    uint32_t k = 0;
    for (auto &s : c->kstats) {
        if (out && k < cap) {
            memset(&out[k], 0, sizeof out[k]);
            strncpy(out[k].name, s.name.c_str(), sizeof(out[k].name) - 1);
            out[k].launches = s.launches; out[k].total_ms = s.total_ms; out[k].last_ms = s.last_ms;
        }
        k++;
    }
    if (n) *n = k;
    return HJ_OK;
}

int hj_shard_split_ordered(hj_ctx *c, const int32_t *d_keys, const int32_t *d_pays, uint64_t n, uint32_t nshards,
                           const uint32_t *h_position, int32_t *d_out_keys, int32_t *d_out_pays, uint64_t *h_counts) {
    if (!c) return HJ_EINVAL;
    if (nshards == 0 || nshards > (uint32_t)MAX_PARTS) return fail(c, HJ_EINVAL, "nshards out of range");
    if (n && (!d_keys || !d_pays || !d_out_keys || !d_out_pays)) return fail(c, HJ_EINVAL, "null column");
    if (((uintptr_t)d_keys | (uintptr_t)d_pays) & 15) return fail(c, HJ_EINVAL, "device columns must be 16-byte aligned");
    HIPCHK(c, hipSetDevice(c->device));
    // persistent small buffers: no hipMalloc/hipFree in the steady state (hipFree synchronises the whole
    // device and would stall an all-to-all that is in flight on another stream)
    RET(ensure(c, c->shard_root, 16));
    RET(ensure(c, c->shard_off, (size_t)(MAX_PARTS + 1) * 8 + (size_t)MAX_PARTS * 4));
    if (!c->h_shard_off) HIPCHK(c, hipHostMalloc((void **)&c->h_shard_off, (size_t)(MAX_PARTS + 1) * 8 + (size_t)MAX_PARTS * 4, hipHostMallocDefault));
    uint32_t *d_remap = nullptr;
    if (h_position) { // shard v goes to output position h_position[v] (a permutation of 0..nshards-1)
        std::vector<uint8_t> seen(nshards, 0);
        for (uint32_t i = 0; i < nshards; i++) {
            if (h_position[i] >= nshards || seen[h_position[i]]) return fail(c, HJ_EINVAL, "position table is not a permutation");
            seen[h_position[i]] = 1;
        }
        uint32_t *h_remap = reinterpret_cast<uint32_t *>(c->h_shard_off + MAX_PARTS + 1);
        memcpy(h_remap, h_position, (size_t)nshards * 4);
        d_remap = reinterpret_cast<uint32_t *>((uint64_t *)c->shard_off.p + MAX_PARTS + 1);
        HIPCHK(c, hipMemcpyAsync(d_remap, h_remap, (size_t)nshards * 4, hipMemcpyHostToDevice, c->stream));
    }
    HIPCHK(c, launch_set_root(c->stream, (uint64_t *)c->shard_root.p, n));
    RET(run_pass(c, 0, 1, d_keys, d_pays, n, (const uint64_t *)c->shard_root.p, 1, 0, nshards, nshards, d_out_keys, d_out_pays,
                 (uint64_t *)c->shard_off.p, nullptr, nullptr, d_remap));
    HIPCHK(c, hipMemcpyAsync(c->h_shard_off, c->shard_off.p, (size_t)(nshards + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (h_counts) for (uint32_t i = 0; i < nshards; i++) h_counts[i] = c->h_shard_off[i + 1] - c->h_shard_off[i];
    return HJ_OK;
}

int hj_shard_split(hj_ctx *c, const int32_t *d_keys, const int32_t *d_pays, uint64_t n, uint32_t nshards,
                   int32_t *d_out_keys, int32_t *d_out_pays, uint64_t *h_counts) {
    return hj_shard_split_ordered(c, d_keys, d_pays, n, nshards, nullptr, d_out_keys, d_out_pays, h_counts);
}

int hj_shard_count(hj_ctx *c, const int32_t *d_keys, uint64_t n, uint32_t nshards, uint64_t *h_counts) {
    if (!c || !h_counts) return HJ_EINVAL;
    if (nshards == 0 || nshards > (uint32_t)MAX_PARTS) return fail(c, HJ_EINVAL, "nshards out of range");
    if (n && !d_keys) return fail(c, HJ_EINVAL, "null column");
    HIPCHK(c, hipSetDevice(c->device));
    RET(ensure(c, c->shard_off, (size_t)(MAX_PARTS + 1) * 8 + (size_t)MAX_PARTS * 4));
    if (!c->h_shard_off) HIPCHK(c, hipHostMalloc((void **)&c->h_shard_off, (size_t)(MAX_PARTS + 1) * 8 + (size_t)MAX_PARTS * 4, hipHostMallocDefault));
    HIPCHK(c, hipMemsetAsync(c->shard_off.p, 0, (size_t)nshards * 8, c->stream));
    if (n) HIPCHK(c, launch_shard_count(c->stream, d_keys, n, nshards, (uint64_t *)c->shard_off.p));
    HIPCHK(c, hipMemcpyAsync(c->h_shard_off, c->shard_off.p, (size_t)nshards * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (uint32_t i = 0; i < nshards; i++) h_counts[i] = c->h_shard_off[i];
    return HJ_OK;
}

int hj_gen_unique(hj_ctx *c, int32_t *d_keys, uint64_t n, uint64_t first, uint64_t domain, uint64_t seed) {
    if (!c) return HJ_EINVAL;
    if (domain == 0 || domain > ((uint64_t)1 << 32)) return fail(c, HJ_EINVAL, "domain must be in [1, 2^32]");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, launch_gen_unique(c->stream, d_keys, n, first, domain, seed));
    return HJ_OK;
}

int hj_gen_zipf(hj_ctx *c, int32_t *d_keys, uint64_t n, uint64_t first, uint64_t alphabet, double theta, uint64_t seed) {
    if (!c) return HJ_EINVAL;
    if (alphabet == 0 || alphabet >= ((uint64_t)1 << 32) - 1) return fail(c, HJ_EINVAL, "alphabet must be in [1, 2^32-2]");
    if (!(theta >= 0.0) || theta > 8.0) return fail(c, HJ_EINVAL, "theta out of range");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, launch_gen_zipf(c->stream, d_keys, n, first, alphabet, theta, seed));
    return HJ_OK;
}

int hj_fill_payload(hj_ctx *c, int32_t *d_pays, uint64_t n, int mode, uint64_t first_rowid) {
    if (!c) return HJ_EINVAL;
    if (mode != HJ_PAYLOAD_ONES && mode != HJ_PAYLOAD_ROWID) return fail(c, HJ_EINVAL, "bad payload_mode");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, launch_fill(c->stream, d_pays, n, mode, first_rowid));
    return HJ_OK;
}

int hj_ubench(hj_ctx *c, int kind, const int32_t *d_in_k, const int32_t *d_in_p, int32_t *d_out_k, int32_t *d_out_p, uint64_t n,
              uint32_t reps, double *avg_ms, uint64_t *bytes_per_launch) {
    if (!c || kind < 0 || kind > 7 || !reps) return HJ_EINVAL;
    if (n < 32 || !d_in_k || !d_in_p || !d_out_k || !d_out_p) return fail(c, HJ_EINVAL, "hj_ubench needs four columns of >= 32 tuples");
    if (kind >= 4 && (d_in_p != d_in_k + n || d_out_p != d_out_k + n || (n & 31)))
        return fail(c, HJ_EINVAL, "hj_ubench kinds 4-7 move ONE array per side: the payload column must follow the key column (p == k + n, n a multiple of 32)");
    if (((uintptr_t)d_in_k | (uintptr_t)d_in_p | (uintptr_t)d_out_k | (uintptr_t)d_out_p) & 15) return fail(c, HJ_EINVAL, "columns must be 16-byte aligned");
    HIPCHK(c, hipSetDevice(c->device));
    hipEvent_t a = get_event(c), b = get_event(c);
    if (!a || !b) return fail(c, HJ_EHIP, "no HIP events");
    HIPCHK(c, launch_ubench(c->stream, kind, d_in_k, d_in_p, d_out_k, d_out_p, n)); // warm-up
    HIPCHK(c, hipEventRecord(a, c->stream));
    for (uint32_t i = 0; i < reps; i++) HIPCHK(c, launch_ubench(c->stream, kind, d_in_k, d_in_p, d_out_k, d_out_p, n));
    HIPCHK(c, hipEventRecord(b, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, a, b));
    c->pool.push_back(a); c->pool.push_back(b);
    uint64_t lines = n / 32, pow2 = 1;
    while (pow2 * 2 <= lines) pow2 *= 2;
    if (avg_ms) *avg_ms = (double)ms / reps;
    if (kind >= 4) { // line PAIRS (32 tuples, 256 bytes): the scattering kinds cover a power-of-two number of them
        uint64_t p2 = 1;
        while (p2 * 2 <= lines) p2 *= 2;
        if (bytes_per_launch) *bytes_per_launch = ((kind == 6 || kind == 7) ? p2 : lines) * 32 * 16;
        return HJ_OK;
    }
    if (bytes_per_launch) *bytes_per_launch = (kind == 1 ? pow2 * 32 : (n / 4) * 4) * (kind >= 2 ? 8 : 16); // 8 B read + 8 B written per tuple (one-way kinds: one of them)
    return HJ_OK;
}

static int digest_common(hj_ctx *c, const int32_t *a, const int32_t *b, const int32_t *d, uint64_t n, uint64_t *out) {
    if (!c || !out) return HJ_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    uint64_t *sc = (uint64_t *)c->scalars.p;
    HIPCHK(c, hipMemsetAsync(sc + 3, 0, 8, c->stream));
    HIPCHK(c, launch_digest(c->stream, a, b, d, n, sc + 3));
    RET(fetch_scalars(c));
    *out = c->h_scalars[3];
    return HJ_OK;
}

int hj_digest_pairs(hj_ctx *c, const int32_t *d_keys, const int32_t *d_pays, uint64_t n, uint64_t *digest) {
    return digest_common(c, d_keys, d_pays, nullptr, n, digest);
}

int hj_digest_triples(hj_ctx *c, const int32_t *d_key, const int32_t *d_payR, const int32_t *d_payS, uint64_t n,
                      uint64_t *digest) {
    return digest_common(c, d_key, d_payR, d_payS, n, digest);
}

int hj_verify_partitions(hj_ctx *c, int rel, uint64_t *misplaced, uint64_t *d_digests) {
    RET(check_rel(c, rel));
    Rel &R = c->rel[rel];
    if (!R.partitioned) return fail(c, HJ_EINVAL, "relation %d not partitioned", rel);
    HIPCHK(c, hipSetDevice(c->device));
    RET(exact_for_introspection(c, R));
    uint64_t *sc = (uint64_t *)c->scalars.p;
    HIPCHK(c, hipMemsetAsync(sc + 4, 0, 8, c->stream));
    HIPCHK(c, launch_verify_partitions(c->stream, R.part_k, R.part_p, R.part_beg, R.part_end, R.nparts, 0, 0, sc + 4, d_digests, nullptr));
    RET(fetch_scalars(c));
    if (misplaced) *misplaced = c->h_scalars[4];
    return HJ_OK;
}

} // extern "C"
