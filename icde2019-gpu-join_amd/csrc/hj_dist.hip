// hj_dist.hip — multi-GPU join behind the C ABI of include/hj_dist.h: level-0 shard split, all-to-all over xGMI with RCCL,
// local radix passes + build/probe, all-reduce of the count.
//
// Reference analogue: the co-processing path joins 16 host-made level-0 partitions independently
// (hash_join_clustered_probe.cu:1256-1266, 1503-1618); here the level-0 partitions are made on the GPUs and cross the links.
//
// One Rank object drives one GPU.  Two transports implement the same four collectives (Link):
//   RcclLink  ncclSend/ncclRecv groups, ncclAllGather, ncclAllReduce on the rank's communication stream;
//   CopyLink  ranks of ONE process that share a device (RCCL refuses duplicate GPUs): peers' buffers are pulled with
//             hipMemcpyAsync, ordered by HIP events and a host barrier.  It exists so that the whole pipeline — slicing,
//             fixed-size regions, segment tables, flag gathering, the exact fallback — runs at world sizes 2 and 3 on a
//             one-GPU box (tests/test_dist_c.py).
//
// Fast path (sliced, no host read of any count):
//   per relation, K slices; slice i: k_part1_fast<MODE 1> writes shard g's tuples into slots (g, span) of fixed capacity, so
//   that shard g's slots are ONE contiguous region of nsp*cap tuples -> one grouped send/recv per slice moves region g to
//   rank g (message sizes are a function of (n_max, G, K) alone) together with the slots' end positions -> the receiver
//   turns them into a segment table (k_dist_segments) and runs its local pass 1 over the segments (k_part2_fast in its
//   seg_pass1 geometry), pass 2 and the join as on one GPU.
//   Streams: compute (the context's) and comm.  compute: s0 s1 p0 s2 p1 ... ; comm: x0 x1 x2 ...  (s = split, x = exchange,
//   p = local pass 1), x_i after s_i, p_i after x_i: split(i+1) || exchange(i) || pass-1(i-1).
//   A slot that overflows anywhere raises that rank's flag; the flags are summed with the result (one all-reduce) and every
//   rank then repeats the join on the exact path.
// Exact path: exact split (histogram + scatter), counts to the host, all-gather of the counts, messages of exact size,
//   local partition + join (what dist.py did from Python in rounds 1-2).
// Materialising join (round 5; north_star "join output = match count AND materialised tuples" x SURVEY §8(e) "Output: stays
//   sharded"): the same two paths with the one-probe materialiser in place of the count kernel — every rank writes the
//   (key, payR, payS) tuples of the partitions it owns into ITS caller-provided device columns (the earlier probe-side group under the
//   exchange, the last group appending behind it on one output cursor), the per-rank output sizes are all-gathered, the global
//   count all-reduced with the flags.  Reference: join_partitioned_results per level-0 partition, hjcp.cu:1503-1618, jp.cu:1107-1416.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "hj.h"
#include "hj_ctx.h"
#include "hj_dist.h"
#include "hj_internal.h"

using namespace hj;
using namespace hjx;

namespace {

constexpr uint64_t PAD = 16;
std::atomic<int> g_debug_stall_rank{0}; // tests only (hj_dist_debug_stall_rank): rank + 1 that stalls in its next exchange; 0 = none
constexpr size_t MSG_CHUNK = (size_t)512 << 20; // RCCL 2.26 / ROCm 7 corrupted single messages of >= 2 GiB (tools/experiments/rccl_2gib_repro.py)

// librccl is bound at run time, not at link time: a process that has PyTorch loaded already carries PyTorch's own copy of
// librccl, and a second copy pulled in by libhj.so's dependency list gives two sets of RCCL globals in one process (observed:
// "double free or corruption" at interpreter exit).  The copy that is already loaded is used; otherwise ROCm's is opened.
struct RcclApi {
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;                       // optional: not part of `ok`
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t *) = nullptr; // optional
    bool ok = false;
};
const RcclApi &rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        void *h = nullptr;
        for (const char *n : {"librccl.so", "librccl.so.1"})
            if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);          // the copy this process already has (PyTorch's)
        for (const char *n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
            if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        bool all = true;
        auto sym = [&](const char *n) { void *p = dlsym(h, n); all &= p != nullptr; return p; };
        api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
        api.CommInitAll = (decltype(api.CommInitAll))sym("ncclCommInitAll");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
        api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
        api.Send = (decltype(api.Send))sym("ncclSend");
        api.Recv = (decltype(api.Recv))sym("ncclRecv");
        api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
        api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
        api.ok = all;
        api.CommAbort = (decltype(api.CommAbort))dlsym(h, "ncclCommAbort");
        api.CommGetAsyncError = (decltype(api.CommGetAsyncError))dlsym(h, "ncclCommGetAsyncError");
    });
    return api;
}

struct Msg {          // one peer's share of an exchange
    int peer;
    const void *src;  // what I send to peer
    size_t sbytes;
    void *dst;        // where peer's message to me lands
    size_t rbytes;
};

// The collectives the pipeline needs; every call is enqueued on `st` (device buffers throughout).
struct Link {
    virtual ~Link() {}
    virtual const char *name() const = 0;
    // A rank that fails must not leave its peers waiting inside a collective: abort() makes every later (and, where the transport
    // allows it, every pending) collective of this rank's group fail instead of block.  failed(): somebody aborted; why says who.
    virtual void abort(const std::string &why) = 0;
    virtual bool failed(std::string *why) = 0;
    virtual int exchange(const std::vector<Msg> &msgs, hipStream_t st, std::string &err) = 0;
    virtual int allgather(const void *src, void *dst, size_t bytes, hipStream_t st, std::string &err) = 0; // dst[world][bytes]
    virtual int allreduce_sum_u64(uint64_t *inout, size_t n, uint64_t *scratch /* [world*n] */, hipStream_t st, std::string &err) = 0;
};

// ---- RCCL over xGMI ----
struct RcclLink : Link {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    bool own = true;
    bool dead = false;
    std::string dead_why;
    ~RcclLink() override {
        if (!comm || !own) return;
        if (dead && rccl().CommAbort) (void)rccl().CommAbort(comm); // never ncclCommDestroy on a communicator with a stuck collective: it waits for it
        else if (!dead) (void)rccl().CommDestroy(comm);
    }
    const char *name() const override { return "rccl"; }
    // The communicator is unusable afterwards (the next join on it fails at once).  ncclCommAbort itself is left to the
    // destructor: aborting while this process still has kernels of the communicator queued is what RCCL documents as safe only
    // from the thread that owns it, which is the one that destroys the rank.
    void abort(const std::string &why) override { if (!dead) { dead = true; dead_why = why; } }
    bool failed(std::string *why) override {
        if (!dead && comm && rccl().CommGetAsyncError) {
            ncclResult_t ar = ncclSuccess;
            if (rccl().CommGetAsyncError(comm, &ar) == ncclSuccess && ar != ncclSuccess && ar != ncclInProgress) {
                dead = true; dead_why = std::string("RCCL asynchronous error: ") + rccl().GetErrorString(ar);
            }
        }
        if (dead && why) *why = dead_why;
        return dead;
    }
    static int chk(ncclResult_t r, const char *what, std::string &err) {
        if (r == ncclSuccess) return 0;
        err = std::string(what) + ": " + rccl().GetErrorString(r);
        return HJ_EHIP;
    }
    bool refuse(std::string &err) { if (dead) err = "communicator aborted: " + dead_why; return dead; }
    int exchange(const std::vector<Msg> &msgs, hipStream_t st, std::string &err) override {
        if (refuse(err)) return HJ_EHIP;
        // ONE group = one all-to-all-v: every ordered pair has its own xGMI link, nothing is relayed
        int rc = chk(rccl().GroupStart(), "ncclGroupStart", err);
        for (const Msg &m : msgs) {
            for (size_t o = 0; o < m.sbytes && !rc; o += MSG_CHUNK)
                rc = chk(rccl().Send((const char *)m.src + o, std::min(MSG_CHUNK, m.sbytes - o), ncclInt8, m.peer, comm, st), "ncclSend", err);
            for (size_t o = 0; o < m.rbytes && !rc; o += MSG_CHUNK)
                rc = chk(rccl().Recv((char *)m.dst + o, std::min(MSG_CHUNK, m.rbytes - o), ncclInt8, m.peer, comm, st), "ncclRecv", err);
        }
        const int rc2 = chk(rccl().GroupEnd(), "ncclGroupEnd", err);
        return rc ? rc : rc2;
    }
    int allgather(const void *src, void *dst, size_t bytes, hipStream_t st, std::string &err) override {
        if (refuse(err)) return HJ_EHIP;
        return chk(rccl().AllGather(src, dst, bytes, ncclInt8, comm, st), "ncclAllGather", err);
    }
    int allreduce_sum_u64(uint64_t *inout, size_t n, uint64_t *, hipStream_t st, std::string &err) override {
        if (refuse(err)) return HJ_EHIP;
        return chk(rccl().AllReduce(inout, inout, n, ncclUint64, ncclSum, comm, st), "ncclAllReduce", err); // wraps mod 2^64
    }
};

// ---- ranks of one process: pull the peers' buffers with device copies (no RCCL kernel, no CU, no LDS: the copy engines) ----
// Ranks that share a device (one-GPU test boxes: RCCL refuses duplicate GPUs) or sit on distinct devices with peer access enabled
// (hipMemcpyPeerAsync over xGMI: the A/B partner of the RCCL transport on a multi-GPU node, hj_dist_create_transport).
struct CopyGroup {
    int world;
    std::mutex mu;
    std::condition_variable cv;
    int waiting = 0;
    uint64_t generation = 0;
    bool aborted = false;                    // under mu: a rank failed — every barrier returns at once from now on
    std::string why;
    double timeout_s = 120.0;
    std::vector<char> arrived;               // per rank: at the current barrier (diagnostics of a deadline)
    std::vector<int> dev;                    // per rank: HIP device
    std::vector<std::vector<Msg>> posted;    // per rank: its messages of the current exchange
    std::vector<const void *> gsrc;          // per rank: source of the current all-gather
    std::vector<hipEvent_t> ready;           // per rank: "my buffers of the current collective are written"
    explicit CopyGroup(int w) : world(w), arrived(w, 0), dev(w, 0), posted(w), gsrc(w, nullptr), ready(w, nullptr) {}
    // false: the group was aborted, or the deadline passed (then this call aborts it and says who was missing)
    bool barrier(int rank, const char *what, std::string &err) {
        std::unique_lock<std::mutex> lk(mu);
        if (aborted) { err = "device-copy transport: " + why; return false; }
        const uint64_t gen = generation;
        arrived[rank] = 1;
        if (++waiting == world) { waiting = 0; generation++; std::fill(arrived.begin(), arrived.end(), 0); cv.notify_all(); return true; }
        const bool ok = cv.wait_for(lk, std::chrono::duration<double>(timeout_s), [&] { return generation != gen || aborted; });
        if (aborted) { err = "device-copy transport: " + why; return false; }
        if (ok) return true;
        std::string missing;
        for (int q = 0; q < world; q++) if (!arrived[q]) missing += (missing.empty() ? "" : ", ") + std::to_string(q);
        why = "deadline of " + std::to_string(timeout_s) + " s passed at the " + what + " barrier: rank " + std::to_string(rank) + " waited for rank(s) " + missing;
        aborted = true;
        err = "device-copy transport: " + why;
        cv.notify_all();
        return false;
    }
    void abort(const std::string &w) {
        std::lock_guard<std::mutex> lk(mu);
        if (!aborted) { aborted = true; why = w; }
        cv.notify_all();
    }
    bool failed(std::string *w) {
        std::lock_guard<std::mutex> lk(mu);
        if (aborted && w) *w = why;
        return aborted;
    }
};

__global__ void k_sum_ranks(const uint64_t *__restrict__ gathered, uint32_t world, uint32_t n, uint64_t *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t s = 0;
    for (uint32_t q = 0; q < world; q++) s += gathered[(uint64_t)q * n + i];
    out[i] = s;
}

struct CopyLink : Link {
    CopyGroup *g = nullptr;
    int rank = 0;
    const char *name() const override { return "device-copy"; }
    void abort(const std::string &why) override { g->abort(why); }
    bool failed(std::string *why) override { return g->failed(why); }
    static int chk(hipError_t e, const char *what, std::string &err) {
        if (e == hipSuccess) return 0;
        err = std::string(what) + ": " + hipGetErrorString(e);
        return HJ_EHIP;
    }
    // dst on my device, src on rank q's: a plain device-to-device copy when the ranks share a device, a peer copy otherwise
    int pull(void *dst, const void *src, size_t bytes, int q, hipStream_t st, std::string &err) {
        if (!bytes) return 0;
        if (g->dev[q] == g->dev[rank]) return chk(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st), "hipMemcpyAsync", err);
        return chk(hipMemcpyPeerAsync(dst, g->dev[rank], src, g->dev[q], bytes, st), "hipMemcpyPeerAsync", err);
    }
    int exchange(const std::vector<Msg> &msgs, hipStream_t st, std::string &err) override {
        g->posted[rank] = msgs;
        int rc = chk(hipEventRecord(g->ready[rank], st), "hipEventRecord", err);
        if (!g->barrier(rank, "exchange (post)", err)) return HJ_EHIP; // every rank has posted and recorded
        for (size_t j = 0; j < msgs.size(); j++) {
            if (rc) break;
            const Msg &m = msgs[j];
            // my k-th message with peer p pairs with p's k-th message with me (both sides list their columns in the same order)
            size_t k = 0;
            for (size_t i = 0; i < j; i++) k += msgs[i].peer == m.peer;
            const Msg *theirs = nullptr;
            for (const Msg &pm : g->posted[m.peer])
                if (pm.peer == rank) { if (k == 0) { theirs = &pm; break; } k--; }
            if (!theirs || theirs->sbytes != m.rbytes) {
                err = "device-copy exchange: rank " + std::to_string(rank) + " expects " + std::to_string(m.rbytes) + " bytes from rank " +
                      std::to_string(m.peer) + ", which sends " + (theirs ? std::to_string(theirs->sbytes) : std::string("nothing")) +
                      " (message " + std::to_string(j) + " of " + std::to_string(msgs.size()) + ", peer posted " + std::to_string(g->posted[m.peer].size()) + ")";
                rc = HJ_EHIP;
                break;
            }
            rc = chk(hipStreamWaitEvent(st, g->ready[m.peer], 0), "hipStreamWaitEvent", err);
            if (!rc) rc = pull(m.dst, theirs->src, m.rbytes, m.peer, st, err);
        }
        if (rc) { g->abort("rank " + std::to_string(rank) + ": " + err); return rc; } // the peers leave their barrier instead of waiting for me
        if (!g->barrier(rank, "exchange (done)", err)) return HJ_EHIP; // the posted lists may be overwritten
        return 0;
    }
    int allgather(const void *src, void *dst, size_t bytes, hipStream_t st, std::string &err) override {
        g->gsrc[rank] = src;
        int rc = chk(hipEventRecord(g->ready[rank], st), "hipEventRecord", err);
        if (!g->barrier(rank, "all-gather (post)", err)) return HJ_EHIP;
        for (int q = 0; q < g->world && !rc; q++) {
            rc = chk(hipStreamWaitEvent(st, g->ready[q], 0), "hipStreamWaitEvent", err);
            if (!rc) rc = pull((char *)dst + (size_t)q * bytes, g->gsrc[q], bytes, q, st, err);
        }
        if (rc) { g->abort("rank " + std::to_string(rank) + ": " + err); return rc; }
        if (!g->barrier(rank, "all-gather (done)", err)) return HJ_EHIP;
        return 0;
    }
    int allreduce_sum_u64(uint64_t *inout, size_t n, uint64_t *scratch, hipStream_t st, std::string &err) override {
        int rc = allgather(inout, scratch, n * 8, st, err);
        if (rc) return rc;
        // every rank has copied every contribution before anybody overwrites its own: the stream order of each rank alone
        // does not give that, so the ranks meet once more behind their copies
        rc = chk(hipEventRecord(g->ready[rank], st), "hipEventRecord", err);
        if (!g->barrier(rank, "all-reduce (copied)", err)) return HJ_EHIP;
        for (int q = 0; q < g->world && !rc; q++) rc = chk(hipStreamWaitEvent(st, g->ready[q], 0), "hipStreamWaitEvent", err);
        if (rc) { g->abort("rank " + std::to_string(rank) + ": " + err); return rc; }
        if (!g->barrier(rank, "all-reduce (waited)", err)) return HJ_EHIP;
        hipLaunchKernelGGL(k_sum_ranks, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, scratch, (uint32_t)g->world, (uint32_t)n, inout);
        return chk(hipGetLastError(), "k_sum_ranks", err);
    }
};


} // namespace

// The materialising join's output: this rank's caller-provided device columns (the tuples of the partitions it owns) — SURVEY
// §8(e) "Output: stays sharded"; join_partitioned_results writes (payR, payS) per level-0 partition in the co-processing analogue,
// hjcp.cu:1503-1618, jp.cu:1107-1416.
struct MatOut {
    int32_t *key = nullptr, *payR = nullptr, *payS = nullptr;
    uint64_t cap = 0;
    bool want_agg = false;
    uint64_t n_out = 0; // result: tuples this rank produced (may exceed cap: nothing beyond cap was written)
};

// ================================================================================================
// one rank
// ================================================================================================
struct hj_dist_rank {
    hj_ctx *c = nullptr;
    bool own_ctx = false;
    int rank = 0, world = 1;
    std::unique_ptr<Link> link;
    hj_dist_config cfg{};
    std::string err;
    hipStream_t comm = nullptr;
    hj_dist_stats st{};
    double timeout_s = 120.0;  // deadline of every wait on a collective (hj_dist_config.timeout_ms, HJ_DIST_TIMEOUT_S)
    const char *stage = "idle"; // what the host thread last enqueued / waits for (diagnostics of a deadline)
    int stage_rel = -1, stage_slice = -1;
    uint32_t cur_maxK = 0, cur_K[2] = {0, 0}; // the slices of the running sliced join (for the event census of a deadline)
    std::vector<uint8_t> enq;  // [2*maxK] split events, [2*maxK] exchange events recorded by the running join
    bool prefer_exact = false; // the last fast attempt overflowed somewhere: exact path until new columns are bound
    bool args_rejected = false; // the last join returned HJ_EINVAL because some rank's arguments were bad (learned from the first all-gather): the link is fine
    const int32_t *last_cols[4] = {nullptr, nullptr, nullptr, nullptr};
    uint64_t last_n[2] = {0, 0};
    // buffers (grow-only)
    hj_ctx::Buf send_k[2], send_p[2], recv_k[2], recv_p[2];
    hj_ctx::Buf s_beg[2], s_end[2];      // slot ranges written by the split [K][G*nsp]
    hj_ctx::Buf r_end[2];                // received end positions [K][G*nsp]
    hj_ctx::Buf seg_beg[2], seg_end[2];  // segment tables of the local pass 1 [K][nsp*G]
    hj_ctx::Buf small;                   // device scratch: sizes, flags, results (u64 words)
    uint64_t *h_small = nullptr;         // pinned mirror
    std::vector<hipEvent_t> ev_split, ev_xchg, ev_t; // per (relation, slice); timing events
    hipEvent_t ev_misc[4] = {}, ev_coll[2] = {};
    hj_ctx::Buf x_send_k[2], x_send_p[2], x_recv_k[2], x_recv_p[2], x_counts, x_bal; // exact path
    // the geometry the ranks last AGREED they could allocate for (grow-only buffers: the same geometry needs no second agreement)
    uint64_t agreed[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<uint64_t> n_out_all; // materialising join: output tuples of every rank (all-gathered)

    int fail(int code, const char *fmt, ...) {
        char buf[600];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
        return code;
    }
};

namespace {

#define DCHK(r, call)                                                                                       \
    do {                                                                                                    \
        hipError_t e__ = (call);                                                                            \
        if (e__ != hipSuccess) return (r)->fail(e__ == hipErrorOutOfMemory ? HJ_ENOMEM : HJ_EHIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e__)); \
    } while (0)
#define DRET(r, x)                                                                                          \
    do {                                                                                                    \
        int r__ = (x);                                                                                      \
        if (r__) { if ((r)->err.empty()) (r)->err = hj_error((r)->c); return r__; }                         \
    } while (0)
#define LRET(r, x)                                                                                          \
    do {                                                                                                    \
        int r__ = (x);                                                                                      \
        if (r__) return r__;                                                                                \
    } while (0)

int dist_ensure(hj_dist_rank *r, hj_ctx::Buf &b, size_t bytes) {
    int rc = ensure(r->c, b, bytes);
    if (rc) r->err = hj_error(r->c);
    return rc;
}

double env_timeout_s() {
    static double v = [] { const char *e = getenv("HJ_DIST_TIMEOUT_S"); const double x = e ? atof(e) : 0.0; return x > 0 ? x : 120.0; }();
    return v;
}

// Which slice the links (or a peer) still owe: an event census of the running sliced join, for the message of a deadline.
std::string slice_census(hj_dist_rank *r) {
    std::string s;
    for (int x = 0; x < 2 && r->cur_maxK; x++)
        for (uint32_t i = 0; i < r->cur_K[x]; i++) {
            const size_t e = (size_t)x * r->cur_maxK + i;
            if (e >= r->ev_xchg.size()) continue;
            const size_t half = 2 * (size_t)r->cur_maxK;
            const bool spq = e < r->enq.size() && r->enq[e], xcq = half + e < r->enq.size() && r->enq[half + e];
            const bool sp = spq && hipEventQuery(r->ev_split[e]) == hipSuccess, xc = xcq && hipEventQuery(r->ev_xchg[e]) == hipSuccess;
            if (sp && xc) continue;
            s += std::string(s.empty() ? "" : "; ") + (x ? "S" : "R") + " slice " + std::to_string(i) + ": split " + (sp ? "done" : spq ? "NOT done" : "not enqueued") +
                 ", exchange " + (xc ? "done" : xcq ? "NOT complete" : "not enqueued");
        }
    (void)hipGetLastError(); // hipErrorNotReady from the queries is not an error
    return s.empty() ? std::string("every slice's exchange has completed") : s;
}

// Wait for a stream with a deadline: no wait of the multi-GPU path blocks for ever.  On expiry the message names the rank, the
// stage, the slices whose exchange has not completed and the peers a message is owed by (every other rank: one grouped
// send/recv per slice), the rank's group is aborted, and the caller returns HJ_EHIP.
int wait_stream(hj_dist_rank *r, hipStream_t s, const char *what) {
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0;; spins++) {
        const hipError_t e = hipStreamQuery(s);
        if (e == hipSuccess) return 0;
        if (e != hipErrorNotReady) return r->fail(HJ_EHIP, "rank %d of %d: %s: %s", r->rank, r->world, what, hipGetErrorString(e));
        (void)hipGetLastError();
        if ((spins & 63u) == 63u) {
            std::string why;
            if (r->link && r->link->failed(&why)) return r->fail(HJ_EHIP, "rank %d of %d: %s: given up, the group was aborted: %s", r->rank, r->world, what, why.c_str());
            const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (el > r->timeout_s) {
                const std::string census = slice_census(r);
                r->fail(HJ_EHIP, "rank %d of %d (%s transport): deadline of %.1f s passed waiting for %s [stage: %s, relation %d, slice %d]; %s; "
                                 "peers owed/owing a message: every rank but %d.  A peer has failed or stalled, or a link is down "
                                 "(HJ_DIST_TIMEOUT_S / hj_dist_config.timeout_ms set the deadline)",
                        r->rank, r->world, r->link ? r->link->name() : "?", r->timeout_s, what, r->stage, r->stage_rel, r->stage_slice, census.c_str(), r->rank);
                fprintf(stderr, "[hj_dist] %s\n", r->err.c_str());
                if (r->link) r->link->abort(r->err);
                return HJ_EHIP;
            }
        }
        if (spins < 4096) std::this_thread::yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
}

int rank_init(hj_dist_rank *r) {
    r->timeout_s = env_timeout_s();
    DCHK(r, hipSetDevice(r->c->device));
    DCHK(r, hipStreamCreateWithFlags(&r->comm, hipStreamNonBlocking));
    LRET(r, dist_ensure(r, r->small, 16384));
    DCHK(r, hipMemset(r->small.p, 0, 16384));
    DCHK(r, hipHostMalloc((void **)&r->h_small, 16384, hipHostMallocDefault));
    for (auto &e : r->ev_misc) DCHK(r, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto &e : r->ev_coll) DCHK(r, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return 0;
}

void rank_free(hj_dist_rank *r) {
    if (!r) return;
    if (r->c) (void)hipSetDevice(r->c->device);
    // A group that failed (deadline, peer gone) may still have a collective kernel stuck on the communication stream: hipFree
    // synchronises the whole device and hipStreamDestroy the stream, so both would block behind it for ever.  The link goes first —
    // its destructor is what reaches ncclCommAbort, which takes the stuck kernel off the device — and only then the buffers, events
    // and streams.  A healthy group drains its stream and frees in the usual order (the communicator is destroyed behind it).
    const bool failed = r->link && r->link->failed(nullptr);
    if (failed) r->link.reset();
    else if (r->comm) (void)hipStreamSynchronize(r->comm);
    for (int x = 0; x < 2; x++) {
        release(r->send_k[x]); release(r->send_p[x]); release(r->recv_k[x]); release(r->recv_p[x]);
        release(r->s_beg[x]); release(r->s_end[x]); release(r->r_end[x]); release(r->seg_beg[x]); release(r->seg_end[x]);
        release(r->x_send_k[x]); release(r->x_send_p[x]); release(r->x_recv_k[x]); release(r->x_recv_p[x]);
    }
    release(r->x_counts);
    release(r->x_bal);
    release(r->small);
    if (r->h_small) (void)hipHostFree(r->h_small);
    for (auto e : r->ev_split) if (e) (void)hipEventDestroy(e);
    for (auto e : r->ev_xchg) if (e) (void)hipEventDestroy(e);
    for (auto e : r->ev_t) if (e) (void)hipEventDestroy(e);
    for (auto e : r->ev_misc) if (e) (void)hipEventDestroy(e);
    for (auto e : r->ev_coll) if (e) (void)hipEventDestroy(e);
    r->link.reset();
    if (r->comm) (void)hipStreamDestroy(r->comm);
    if (r->own_ctx && r->c) hj_destroy(r->c);
    delete r;
}

// One communicator, one stream: EVERY collective of a rank is issued on its communication stream (RCCL serialises a
// communicator's operations; issuing them from two streams leaves their relative order to the library).  Small collectives that
// belong to the compute stream's program order are bracketed by events: comm waits for compute, compute waits for comm.
int coll_begin(hj_dist_rank *r) {
    DCHK(r, hipEventRecord(r->ev_coll[0], r->c->stream));
    DCHK(r, hipStreamWaitEvent(r->comm, r->ev_coll[0], 0));
    return 0;
}
int coll_end(hj_dist_rank *r) {
    DCHK(r, hipEventRecord(r->ev_coll[1], r->comm));
    DCHK(r, hipStreamWaitEvent(r->c->stream, r->ev_coll[1], 0));
    return 0;
}

// Every rank reports whether it is still good (local_rc == 0) and every rank learns whether all are: a rank that failed while
// planning or allocating must not leave the others alone inside the exchange.  One 8-byte all-gather on the communication stream,
// read by the host; the caller returns when anybody failed.  [sync, with the deadline]
int agree(hj_dist_rank *r, int local_rc, const char *phase) {
    hj_ctx *c = r->c;
    uint64_t *small = (uint64_t *)r->small.p; // [1024] mine, [1032 ..] everybody's (world <= 64)
    r->h_small[1024] = local_rc ? (uint64_t)(uint32_t)(-local_rc) : 0;
    r->stage = phase; r->stage_rel = -1; r->stage_slice = -1;
    const std::string mine = r->err;
    DCHK(r, hipMemcpyAsync(small + 1024, r->h_small + 1024, 8, hipMemcpyHostToDevice, c->stream));
    LRET(r, coll_begin(r));
    LRET(r, r->link->allgather(small + 1024, small + 1032, 8, r->comm, r->err));
    LRET(r, coll_end(r));
    DCHK(r, hipMemcpyAsync(r->h_small + 1032, small + 1032, (size_t)r->world * 8, hipMemcpyDeviceToHost, c->stream));
    LRET(r, wait_stream(r, c->stream, phase));
    if (local_rc) { r->err = mine; return local_rc; }
    for (int q = 0; q < r->world; q++)
        if (r->h_small[1032 + q]) return r->fail(HJ_EHIP, "rank %d: rank %d failed %s (code -%llu): every rank leaves the join", r->rank, q, phase, (unsigned long long)r->h_small[1032 + q]);
    return 0;
}

// Materialising join: every rank learns every rank's output size (and whether it fitted its owner's columns): one all-gather of
// {n_out, cap} per rank on the communication stream, read by the host.  [sync, with the deadline]
int gather_outputs(hj_dist_rank *r, MatOut *mat) {
    hj_ctx *c = r->c;
    uint64_t *small = (uint64_t *)r->small.p; // [24..25] mine, [512 .. 512 + 2 * world) everybody's
    r->h_small[24] = mat->n_out; r->h_small[25] = mat->cap;
    DCHK(r, hipMemcpyAsync(small + 24, r->h_small + 24, 16, hipMemcpyHostToDevice, c->stream));
    LRET(r, coll_begin(r));
    LRET(r, r->link->allgather(small + 24, small + 512, 16, r->comm, r->err));
    LRET(r, coll_end(r));
    DCHK(r, hipMemcpyAsync(r->h_small + 512, small + 512, (size_t)r->world * 16, hipMemcpyDeviceToHost, c->stream));
    r->stage = "all-gather of the output sizes"; r->stage_rel = -1; r->stage_slice = -1;
    LRET(r, wait_stream(r, c->stream, "the all-gather of the ranks' output sizes (materialising join)"));
    LRET(r, wait_stream(r, r->comm, "the communication stream after the all-gather of the output sizes"));
    r->n_out_all.assign((size_t)r->world, 0);
    std::string over;
    for (int q = 0; q < r->world; q++) {
        r->n_out_all[q] = r->h_small[512 + 2 * q];
        if (r->h_small[512 + 2 * q] > r->h_small[512 + 2 * q + 1])
            over += (over.empty() ? "rank " : ", rank ") + std::to_string(q) + ": " + std::to_string(r->h_small[512 + 2 * q]) + " tuples for a capacity of " + std::to_string(r->h_small[512 + 2 * q + 1]);
    }
    r->st.materialized = mat->n_out;
    if (!over.empty()) return r->fail(HJ_ECAPACITY, "materialised output does not fit its owner's columns (%s): nothing beyond a capacity was written", over.c_str());
    return 0;
}

// Geometry of the sliced exchange of one relation: a function of (n_max over the ranks, G, K) and the radix bits only —
// identical on every rank, which is what makes the message sizes known without asking anybody.
struct SliceGeom {
    uint32_t K = 0, nsp = 0, span = 0, cap0 = 0, cap1 = 0, cap2 = 0, NS = 0;
    uint64_t L = 0, region = 0, sizeA = 0, sizeB = 0;
};
bool plan_slices(uint64_t nmax, uint32_t G, uint32_t Kwant, uint32_t P1, uint32_t P2, SliceGeom &g) {
    if (nmax == 0) nmax = 1;
    // Pass 2 reads at most 1024 segments per parent = local pass-1 spans over all slices: K slices of <= 1024/K spans.  The
    // default K = 4 keeps every split / pass-1 launch 256 workgroups wide (one per CU); K = 8 halves the exposed first split and
    // last pass 1 but runs those kernels 128 wide.
    uint32_t K = Kwant ? Kwant : 4;
    while (K > 1 && nmax / K < ((uint64_t)1 << 16)) K--;
    uint64_t L = (nmax + K - 1) / K;
    L = ((L + TILE - 1) / TILE) * TILE;
    K = (uint32_t)((nmax + L - 1) / L);
    const uint32_t want = std::min<uint32_t>(256u, 1024u / K);
    uint64_t span = (L + want - 1) / want;
    span = ((span + TILE - 1) / TILE) * TILE;
    const uint32_t nsp = (uint32_t)((L + span - 1) / span);
    if ((uint64_t)K * nsp > 1024) return false;
    g.K = K; g.L = L; g.span = (uint32_t)span; g.nsp = nsp; g.NS = K * nsp;
    g.cap0 = fast_slot_cap((span + G - 1) / G, G);
    g.region = (uint64_t)nsp * g.cap0;
    // the local pass-1 workgroup reads the G slots (*, s): span tuples in expectation
    uint64_t sd = 1;
    while (sd * sd < span) sd++;
    g.cap1 = fast_slot_cap((span + 8 * sd + P1 - 1) / P1, P1);
    g.cap2 = fast_slot_cap((nmax + (uint64_t)P1 * P2 - 1) / ((uint64_t)P1 * P2), P2);
    g.sizeA = (uint64_t)P1 * g.NS * g.cap1;
    g.sizeB = (uint64_t)P1 * P2 * g.cap2;
    const uint64_t lim = ((uint64_t)1 << 32) - ((uint64_t)1 << 20);
    return g.sizeA < lim && g.sizeB < lim && (uint64_t)K * G * g.region < lim;
}

float ev_ms(hipEvent_t a, hipEvent_t b) { float ms = 0; return hipEventElapsedTime(&ms, a, b) == hipSuccess ? ms : 0.f; }

// ---- the sliced fixed-size pipeline.  Returns 0 with *flagged set when some slot overflowed somewhere (result void). ----
// mat != nullptr: the materialising join — every probe-side group is joined by the one-probe materialiser into this rank's output
// columns (the earlier group under the exchange, as it is counted otherwise); out[0] = global matches, out[1] = global aggregate (0
// unless mat->want_agg), the per-rank output sizes are all-gathered into r->n_out_all.
int join_fast(hj_dist_rank *r, const int32_t *const cols[4], const uint64_t n[2], const uint64_t nmax[2], uint64_t out[2], bool *flagged,
              bool *applicable, MatOut *mat) {
    hj_ctx *c = r->c;
    const bool phantom = r->world == 1 && r->cfg.phantom_world > 1; // one-GPU measurement mode: the shape of a G-GPU job
    const uint32_t G = phantom ? r->cfg.phantom_world : (uint32_t)r->world, me = (uint32_t)r->rank;
    *applicable = false;
    // radix bits from the nominal sizes (every rank computes the same)
    for (int x = 0; x < 2; x++) { c->rel[x].n = nmax[x]; c->rel[x].bound = true; }
    { const bool kn = c->keep_nine; c->keep_nine = true; choose_bits(c); c->keep_nine = kn; } // (slices run side by side here, not relations: 9 bits first)
    const uint32_t b1 = c->bits1, b2 = c->bits2;
    if (!b2 || !c->fast_path || c->cfg.exact_only) return 0; // single-pass sizes: exact path
    const uint32_t P1 = 1u << b1, P2 = 1u << b2;
    SliceGeom g[2];
    for (int x = 0; x < 2; x++)
        if (!plan_slices(nmax[x], G, r->cfg.slices, P1, P2, g[x])) return 0;
    *applicable = true;
    hipStream_t cs = c->stream, ms = r->comm;
    uint64_t *sc = (uint64_t *)c->scalars.p;
    uint64_t *small = (uint64_t *)r->small.p; // [0..3] result block, [4..5] received, [8..] gathered flags, [64..] all-reduce scratch
    const size_t nev = 0;
    (void)nev;
    // The relation that builds goes first.  The other one (the probe side) is joined in up to two GROUPS of slices — all but
    // the last, and the last — each with its own pass 2 and its own build+probe against the finished build side: pass 2 and
    // the join of the first group run while the last slice is still on the links, and what is left when the last byte has
    // arrived is pass 1 + pass 2 + join of ONE slice (the build tables are rebuilt once more: hidden under the exchange).
    const int first = c->build, second = 1 - c->build;
    struct Grp { uint32_t s0, s1, NS, cap2; uint64_t sizeA, sizeB, offA, offB; size_t s1off, boff; };
    std::vector<Grp> grp[2];
    for (int x = 0; x < 2; x++) {
        const SliceGeom &q = g[x];
        std::vector<std::pair<uint32_t, uint32_t>> cuts;
        if (x == second && q.K >= 2 && !r->cfg.single_group) { cuts.push_back({0, q.K - 1}); cuts.push_back({q.K - 1, q.K}); }
        else cuts.push_back({0, q.K});
        uint64_t offA = 0, offB = 0;
        size_t s1off = 0, boff = 0;
        for (auto &cu : cuts) {
            Grp gr{};
            gr.s0 = cu.first; gr.s1 = cu.second; gr.NS = (gr.s1 - gr.s0) * q.nsp;
            const uint64_t share = (nmax[x] * (gr.s1 - gr.s0) + q.K - 1) / q.K;
            gr.cap2 = fast_slot_cap((share + (uint64_t)P1 * P2 - 1) / ((uint64_t)P1 * P2), P2);
            gr.sizeA = (uint64_t)P1 * gr.NS * q.cap1; gr.sizeB = (uint64_t)P1 * P2 * gr.cap2;
            gr.offA = offA; gr.offB = offB; gr.s1off = s1off; gr.boff = boff;
            offA += gr.sizeA + PAD; offB += gr.sizeB + PAD; s1off += (size_t)P1 * gr.NS; boff += (size_t)P1 * P2;
            offA = (offA + 3) & ~(uint64_t)3; offB = (offB + 3) & ~(uint64_t)3;
            const uint64_t lim = ((uint64_t)1 << 32) - ((uint64_t)1 << 20);
            if (gr.sizeA >= lim || gr.sizeB >= lim) { *applicable = false; return 0; }
            grp[x].push_back(gr);
        }
    }
    // buffers + events.  The memory budget is checked before anything is (re)allocated — at 2^30 tuples per relation and G = 8
    // a rank holds 2 x 2 x 9.0 GiB of send/receive regions and 2 x (2 x 9.2 + 2 x 9.2) GiB of partition buffers beside its 16 GiB
    // of input — and the ranks AGREE on the outcome before the first message leaves: a rank that cannot allocate takes everybody
    // out of the join with a message, instead of leaving its peers inside a collective.
    uint32_t maxK = std::max(g[0].K, g[1].K);
    auto prepare = [&]() -> int {
        struct Want { hj_ctx::Buf *b; size_t bytes; };
        std::vector<Want> wants;
        for (int x = 0; x < 2; x++) {
            const SliceGeom &q = g[x];
            const size_t el = (size_t)q.K * G * q.region + PAD, slots = (size_t)q.K * G * q.nsp;
            hj_ctx::Rel &R = c->rel[x];
            const Grp &lastg = grp[x].back();
            const uint64_t totA = lastg.offA + lastg.sizeA + PAD, totB = lastg.offB + lastg.sizeB + PAD;
            for (hj_ctx::Buf *b : {&r->send_k[x], &r->send_p[x], &r->recv_k[x], &r->recv_p[x]}) wants.push_back({b, el * 4});
            for (hj_ctx::Buf *b : {&r->s_beg[x], &r->s_end[x], &r->r_end[x], &r->seg_beg[x], &r->seg_end[x]}) wants.push_back({b, slots * 8});
            wants.push_back({&R.a_k, (size_t)totA * 4}); wants.push_back({&R.a_p, (size_t)totA * 4});
            wants.push_back({&R.b_k, (size_t)totB * 4}); wants.push_back({&R.b_p, (size_t)totB * 4});
            wants.push_back({&R.s1beg, (size_t)P1 * q.NS * 8}); wants.push_back({&R.s1end, (size_t)P1 * q.NS * 8});
            wants.push_back({&R.beg, (size_t)P1 * P2 * 8 * grp[x].size()}); wants.push_back({&R.end, (size_t)P1 * P2 * 8 * grp[x].size()});
            wants.push_back({&R.root, 16});
        }
        size_t grow = 0, give_back = 0, total = 0;
        for (const Want &w : wants) { total += w.bytes; if (!(w.bytes <= w.b->cap && w.b->p)) { grow += w.bytes; give_back += w.b->cap; } }
        size_t free_b = 0, total_b = 0;
        DCHK(r, hipMemGetInfo(&free_b, &total_b));
        if (grow > free_b + give_back)
            return r->fail(HJ_ENOMEM, "rank %d of %d: the sliced exchange of 2 x %llu tuples per rank at G = %u needs %.2f GiB of buffers (%.2f GiB still to allocate), "
                                      "%.2f GiB of %.2f GiB are free on device %d", r->rank, r->world, (unsigned long long)std::max(nmax[0], nmax[1]), G,
                           total / 1073741824.0, (grow - give_back) / 1073741824.0, free_b / 1073741824.0, total_b / 1073741824.0, c->device);
        while (r->ev_split.size() < 2 * (size_t)maxK) {
            hipEvent_t a, b, t0, t1, t2, t3;
            DCHK(r, hipEventCreateWithFlags(&a, hipEventDisableTiming));
            DCHK(r, hipEventCreateWithFlags(&b, hipEventDisableTiming));
            DCHK(r, hipEventCreate(&t0)); DCHK(r, hipEventCreate(&t1)); DCHK(r, hipEventCreate(&t2)); DCHK(r, hipEventCreate(&t3));
            r->ev_split.push_back(a); r->ev_xchg.push_back(b);
            r->ev_t.push_back(t0); r->ev_t.push_back(t1); r->ev_t.push_back(t2); r->ev_t.push_back(t3);
        }
        while (r->ev_t.size() < 4 * 2 * (size_t)maxK + 8) { hipEvent_t t; DCHK(r, hipEventCreate(&t)); r->ev_t.push_back(t); }
        for (const Want &w : wants) LRET(r, dist_ensure(r, *w.b, w.bytes));
        return 0;
    };
    {   // The agreement is a host-synchronous round trip in front of the first split; buffers only grow, so a geometry the ranks have
        // agreed on once needs no second agreement (every rank sees the same sequence of geometries: they derive from all-gathered sizes).
        const uint64_t key[8] = {nmax[0], nmax[1], G, g[0].K, g[1].K, ((uint64_t)b1 << 32) | b2, (uint64_t)grp[0].size() | ((uint64_t)grp[1].size() << 8), 1};
        if (memcmp(key, r->agreed, sizeof key)) {
            memset(r->agreed, 0, sizeof r->agreed);
            LRET(r, agree(r, prepare(), "preparing the sliced exchange (memory budget, buffers)"));
            memcpy(r->agreed, key, sizeof key);
        } else {
            LRET(r, prepare());
        }
    }
    r->cur_maxK = maxK; r->cur_K[0] = g[0].K; r->cur_K[1] = g[1].K;
    r->enq.assign(4 * (size_t)maxK, 0); // which of this join's split / exchange events have been recorded (an older record reads "done")
    // flags of both relations down, received counters and the per-group results zero
    for (int x = 0; x < 2; x++) { Timed t(c, "k_set_root"); DCHK(r, launch_set_root(cs, (uint64_t *)c->rel[x].root.p, nmax[x], reinterpret_cast<uint32_t *>(sc + 8 + x))); }
    DCHK(r, hipMemsetAsync(small + 4, 0, 32, cs)); // [4],[5] tuples received per relation, [6],[7] of which this rank's own
    DCHK(r, hipMemsetAsync(small + 20, 0, 32, cs));
    r->st.link_bytes = 0; r->st.payload_bytes = 0;

    auto group_of = [&](int x, uint32_t i) -> const Grp & {
        for (const Grp &gr : grp[x]) if (i >= gr.s0 && i < gr.s1) return gr;
        return grp[x].back();
    };
    bool xchg_started = false;
    auto split = [&](int x, uint32_t i) -> int {
        const SliceGeom &q = g[x];
        r->stage = "split"; r->stage_rel = x; r->stage_slice = (int)i;
        const uint64_t lo = std::min<uint64_t>((uint64_t)i * q.L, n[x]), hi = std::min<uint64_t>(lo + q.L, n[x]);
        FastArgs fa{};
        fa.keys = cols[2 * x] + lo; fa.pays = cols[2 * x + 1] + lo; fa.n = hi - lo; fa.span = q.span; fa.nspans = q.nsp;
        fa.shift = 0; fa.P = G; fa.cap = q.cap0; fa.mode = 1;
        const uint64_t base = (uint64_t)i * G * q.region; // slice i's G regions; positions inside the kernel are relative to it
        fa.out_keys = (int32_t *)r->send_k[x].p + base; fa.out_pays = (int32_t *)r->send_p[x].p + base;
        fa.obeg = (uint64_t *)r->s_beg[x].p + (size_t)i * G * q.nsp; fa.oend = (uint64_t *)r->s_end[x].p + (size_t)i * G * q.nsp;
        fa.ovf = reinterpret_cast<uint32_t *>(sc + 8 + x);
        DCHK(r, hipEventRecord(r->ev_t[4 * (x * maxK + i) + 0], cs));
        { Timed t(c, "k_split_fast"); DCHK(r, launch_part1_fast(cs, fa)); }
        DCHK(r, hipEventRecord(r->ev_t[4 * (x * maxK + i) + 1], cs));
        DCHK(r, hipEventRecord(r->ev_split[x * maxK + i], cs));
        r->enq[x * maxK + i] = 1;
        return 0;
    };
    auto exchange = [&](int x, uint32_t i) -> int {
        const SliceGeom &q = g[x];
        r->stage = "exchange"; r->stage_rel = x; r->stage_slice = (int)i;
        // test hook (hj_dist_debug_stall_rank; was read from the environment inside every exchange until round 5): this rank stops taking
        // part for longer than the deadline
        if (x == second && i == 0 && g_debug_stall_rank.load(std::memory_order_relaxed) == r->rank + 1)
            std::this_thread::sleep_for(std::chrono::duration<double>(r->timeout_s * 2.5 + 0.2));
        DCHK(r, hipStreamWaitEvent(ms, r->ev_split[x * maxK + i], 0));
        if (!xchg_started) { DCHK(r, hipEventRecord(r->ev_t[4 * 2 * maxK + 6], ms)); xchg_started = true; } // the links are busy from here ...
        const uint64_t base = (uint64_t)i * G * q.region;
        const size_t eb = (size_t)i * G * q.nsp;
        std::vector<Msg> mk, mp, me_;
        for (uint32_t step = 0; step < G; step++) {
            const uint32_t p = (me + step) % G; // rank-staggered peer order
            const int32_t *sk = (const int32_t *)r->send_k[x].p + base + (uint64_t)p * q.region, *sp = (const int32_t *)r->send_p[x].p + base + (uint64_t)p * q.region;
            int32_t *dk = (int32_t *)r->recv_k[x].p + base + (uint64_t)p * q.region, *dp = (int32_t *)r->recv_p[x].p + base + (uint64_t)p * q.region;
            const uint64_t *se = (const uint64_t *)r->s_end[x].p + eb + (size_t)p * q.nsp;
            uint64_t *de = (uint64_t *)r->r_end[x].p + eb + (size_t)p * q.nsp;
            if (p != me) r->st.link_bytes += (uint64_t)q.region * 8 + (uint64_t)q.nsp * 8; // phantom world: what WOULD cross a link
            if (phantom || (p == me && !r->cfg.self_via_link)) { // own share (phantom world: every share): device copies, no link involved
                DCHK(r, hipMemcpyAsync(dk, sk, q.region * 4, hipMemcpyDeviceToDevice, ms));
                DCHK(r, hipMemcpyAsync(dp, sp, q.region * 4, hipMemcpyDeviceToDevice, ms));
                DCHK(r, hipMemcpyAsync(de, se, (size_t)q.nsp * 8, hipMemcpyDeviceToDevice, ms));
                continue;
            }
            mk.push_back(Msg{(int)p, sk, (size_t)q.region * 4, dk, (size_t)q.region * 4});
            mp.push_back(Msg{(int)p, sp, (size_t)q.region * 4, dp, (size_t)q.region * 4});
            me_.push_back(Msg{(int)p, se, (size_t)q.nsp * 8, de, (size_t)q.nsp * 8});
        }
        std::vector<Msg> all(mk);
        all.insert(all.end(), mp.begin(), mp.end());
        all.insert(all.end(), me_.begin(), me_.end());
        if (!phantom && (!all.empty() || G > 1)) {
            int rc = r->link->exchange(all, ms, r->err);
            if (rc) return rc;
        }
        DCHK(r, hipEventRecord(r->ev_xchg[x * maxK + i], ms));
        DCHK(r, hipEventRecord(r->ev_t[4 * 2 * maxK + 7], ms)); // ... to the last record of this event
        r->enq[2 * (size_t)maxK + x * maxK + i] = 1;
        return 0;
    };
    auto pass1 = [&](int x, uint32_t i) -> int {
        const SliceGeom &q = g[x];
        hj_ctx::Rel &R = c->rel[x];
        const Grp &gr = group_of(x, i);
        DCHK(r, hipStreamWaitEvent(cs, r->ev_xchg[x * maxK + i], 0));
        const size_t eb = (size_t)i * G * q.nsp;
        uint64_t *sb = (uint64_t *)r->seg_beg[x].p + eb, *se = (uint64_t *)r->seg_end[x].p + eb;
        DCHK(r, hipEventRecord(r->ev_t[4 * (x * maxK + i) + 2], cs));
        DCHK(r, launch_dist_segments(cs, (const uint64_t *)r->r_end[x].p + eb, G, q.nsp, q.cap0, phantom ? 0xFFFFFFFFu : me, (uint64_t)i * G * q.region, sb, se,
                                     reinterpret_cast<uint32_t *>(sc + 8 + x), small + 4 + x, small + 6 + x));
        FastArgs fb{};
        fb.keys = (const int32_t *)r->recv_k[x].p; fb.pays = (const int32_t *)r->recv_p[x].p;
        fb.sbeg = sb; fb.send = se; fb.nparents = q.nsp; fb.spp = G;
        fb.shift = b2; fb.P = P1; fb.cap = q.cap1; fb.seg_pass1 = 1; fb.span0 = (i - gr.s0) * q.nsp; fb.nspans = gr.NS;
        fb.out_keys = (int32_t *)R.a_k.p + gr.offA; fb.out_pays = (int32_t *)R.a_p.p + gr.offA;
        fb.obeg = (uint64_t *)R.s1beg.p + gr.s1off; fb.oend = (uint64_t *)R.s1end.p + gr.s1off;
        fb.ovf = reinterpret_cast<uint32_t *>(sc + 8 + x);
        { Timed t(c, "k_part1_fast"); DCHK(r, launch_part2_fast(cs, fb)); }
        DCHK(r, hipEventRecord(r->ev_t[4 * (x * maxK + i) + 3], cs));
        return 0;
    };
    // pass 2 of one group of slices; the relation's Rel then describes that group's partitions
    auto pass2 = [&](int x, const Grp &gr) -> int {
        const SliceGeom &q = g[x];
        hj_ctx::Rel &R = c->rel[x];
        FastArgs fb{};
        fb.keys = (const int32_t *)R.a_k.p + gr.offA; fb.pays = (const int32_t *)R.a_p.p + gr.offA;
        fb.sbeg = (const uint64_t *)R.s1beg.p + gr.s1off; fb.send = (const uint64_t *)R.s1end.p + gr.s1off; fb.nparents = P1; fb.spp = gr.NS;
        fb.shift = 0; fb.P = P2; fb.cap = gr.cap2;
        fb.out_keys = (int32_t *)R.b_k.p + gr.offB; fb.out_pays = (int32_t *)R.b_p.p + gr.offB;
        fb.obeg = (uint64_t *)R.beg.p + gr.boff; fb.oend = (uint64_t *)R.end.p + gr.boff;
        fb.ovf = reinterpret_cast<uint32_t *>(sc + 8 + x);
        { Timed t(c, "k_part2_fast"); DCHK(r, launch_part2_fast(cs, fb)); }
        R.nparts = P1 * P2; R.nranges = P1 * P2; R.sampled = false; R.rpart = nullptr;
        R.part_k = (const int32_t *)R.b_k.p + gr.offB; R.part_p = (const int32_t *)R.b_p.p + gr.offB;
        R.part_beg = (const uint64_t *)R.beg.p + gr.boff; R.part_end = (const uint64_t *)R.end.p + gr.boff;
        R.part_off = nullptr;
        R.n_alloc = gr.sizeB;
        R.pb1 = b1; R.pb2 = b2;
        R.partitioned = true; R.fast_tried = true; R.flag_known_good = false; R.flag_unread = true;
        R.n_bound = (uint64_t)(gr.s1 - gr.s0) * G * q.region; // upper bound of what this rank can hold here (sizes the work-item list; R.n stays nominal)
        return 0;
    };
    hipEvent_t t_tail0 = r->ev_t[4 * 2 * maxK + 0], t_tail1 = r->ev_t[4 * 2 * maxK + 1];
    uint32_t joined = 0, early_mask = 0;
    // build + probe of one probe-side group against the build side; the result is parked on the device (no host read here)
    auto join_group = [&]() -> int {
        if (mat) { // ONE probe that writes: the second group appends behind the first (the cursor is zeroed by the first group's plan only)
            int rc = hj_join_materialize_enqueue(c, mat->key, mat->payR, mat->payS, mat->cap, joined > 0);
            if (rc) { r->err = hj_error(c); return rc; }
            joined++;
            return 0;
        }
        int rc = hj_join_count_enqueue(c);
        if (rc) { r->err = hj_error(c); return rc; }
        DCHK(r, hipMemcpyAsync(small + 20 + 2 * joined, sc + 1, 16, hipMemcpyDeviceToDevice, cs));
        joined++;
        return 0;
    };
    // after the pass 1 of slice (x, i): a finished group gets its pass 2, a finished probe-side group its join
    auto after_pass1 = [&](int x, uint32_t i) -> int {
        const Grp &gr = group_of(x, i);
        if (i + 1 != gr.s1) return 0;
        const bool last = x == second && gr.s1 == g[x].K;
        // the last group's pass 2 + join are the tail (nothing is on the links any more); earlier ones run under the exchange
        hipEvent_t e0 = last ? t_tail0 : r->ev_t[4 * 2 * maxK + 2 + 2 * (x == second)], e1 = last ? t_tail1 : r->ev_t[4 * 2 * maxK + 3 + 2 * (x == second)];
        DCHK(r, hipEventRecord(e0, cs));
        LRET(r, pass2(x, gr));
        if (x == second) LRET(r, join_group());
        DCHK(r, hipEventRecord(e1, cs));
        if (!last) early_mask |= 1u << (x == second);
        return 0;
    };
    // ---- enqueue: the schedule of the header comment, the build side first ----
    struct Step { int x; uint32_t i; };
    std::vector<Step> order;
    for (int x : {first, second}) for (uint32_t i = 0; i < g[x].K; i++) order.push_back(Step{x, i});
    c->join_planned = false;
    for (size_t j = 0; j < order.size(); j++) {
        LRET(r, split(order[j].x, order[j].i));
        LRET(r, exchange(order[j].x, order[j].i));
        if (j >= 1) { LRET(r, pass1(order[j - 1].x, order[j - 1].i)); LRET(r, after_pass1(order[j - 1].x, order[j - 1].i)); }
    }
    LRET(r, pass1(order.back().x, order.back().i));
    LRET(r, after_pass1(order.back().x, order.back().i));
    // the per-group results and, with the result block, the flags of this rank's kernels; [sync]
    if (mat && mat->want_agg) { // sum payR * payS over what was written (the cursor is on the device: k_dot reads it there)
        DCHK(r, hipMemsetAsync(sc + 3, 0, 8, cs));
        DCHK(r, launch_dot(cs, mat->payR, mat->payS, sc + SC_CURSOR, mat->cap, sc + 3));
    }
    DCHK(r, hipMemcpyAsync(r->h_small + 16, small + 20, 32, hipMemcpyDeviceToHost, cs));
    r->stage = "pipeline drained"; r->stage_rel = -1; r->stage_slice = -1;
    LRET(r, wait_stream(r, cs, "the sliced pipeline (splits, exchanges, local passes, joins)")); // every exchange is behind this: the deadline, not a blocking sync
    if (fetch_scalars(c)) { r->err = hj_error(c); return HJ_EHIP; }
    uint64_t m = r->h_small[16] + r->h_small[18], a = r->h_small[17] + r->h_small[19];
    if (mat) { m = c->h_scalars[SC_CURSOR]; a = mat->want_agg ? c->h_scalars[3] : 0; mat->n_out = m; } // the output cursor = this rank's matches
    // one all-reduce: matches, aggregate, the two flags (a rank whose local slots overflowed must take everybody along)
    r->h_small[0] = m; r->h_small[1] = a; r->h_small[2] = c->h_scalars[8] & 0xFFFFFFFFu; r->h_small[3] = c->h_scalars[9] & 0xFFFFFFFFu;
    DCHK(r, hipMemcpyAsync(small, r->h_small, 32, hipMemcpyHostToDevice, cs));
    LRET(r, coll_begin(r));
    LRET(r, r->link->allreduce_sum_u64(small, 4, small + 64, ms, r->err));
    LRET(r, coll_end(r));
    DCHK(r, hipMemcpyAsync(r->h_small + 8, small, 64, hipMemcpyDeviceToHost, cs));
    r->stage = "all-reduce of the result";
    LRET(r, wait_stream(r, cs, "the all-reduce of the result (matches, aggregate, overflow flags)"));
    LRET(r, wait_stream(r, ms, "the communication stream after the all-reduce"));
    out[0] = r->h_small[8]; out[1] = r->h_small[9];
    *flagged = (r->h_small[10] | r->h_small[11]) != 0;
    if (mat && !*flagged) LRET(r, gather_outputs(r, mat));
    r->st.received[0] = r->h_small[12]; r->st.received[1] = r->h_small[13];
    // tuple bytes among the link bytes: what this rank sent to others = its local tuples minus the ones it kept (phantom world: the
    // shards other than shard 0 of what it holds)
    r->st.payload_bytes = 0;
    for (int x = 0; x < 2; x++) { const uint64_t all = phantom ? r->h_small[12 + x] : n[x], self = r->h_small[14 + x]; r->st.payload_bytes += 8 * (all > self ? all - self : 0); }
    // stage times
    r->st.path = 0; r->st.slices = maxK; r->st.spans_per_slice = g[0].nsp; r->st.slot_capacity[0] = g[0].cap0; r->st.slot_capacity[1] = g[1].cap0;
    for (int x = 0; x < 2; x++) {
        r->st.split_ms[x] = 0; r->st.pass1_ms[x] = 0;
        for (uint32_t i = 0; i < g[x].K; i++) {
            r->st.split_ms[x] += ev_ms(r->ev_t[4 * (x * maxK + i) + 0], r->ev_t[4 * (x * maxK + i) + 1]);
            r->st.pass1_ms[x] += ev_ms(r->ev_t[4 * (x * maxK + i) + 2], r->ev_t[4 * (x * maxK + i) + 3]);
        }
    }
    r->st.first_split_ms = ev_ms(r->ev_t[4 * (first * maxK) + 0], r->ev_t[4 * (first * maxK) + 1]);
    r->st.last_pass1_ms = ev_ms(r->ev_t[4 * (second * maxK + g[second].K - 1) + 2], r->ev_t[4 * (second * maxK + g[second].K - 1) + 3]);
    r->st.pass2_join_ms = ev_ms(t_tail0, t_tail1);
    r->st.early_pass2_join_ms = 0;
    for (int k = 0; k < 2; k++)
        if (early_mask & (1u << k)) r->st.early_pass2_join_ms += ev_ms(r->ev_t[4 * 2 * maxK + 2 + 2 * k], r->ev_t[4 * 2 * maxK + 3 + 2 * k]);
    r->st.probe_groups = (uint32_t)grp[second].size();
    r->st.exchange_ms = xchg_started ? ev_ms(r->ev_t[4 * 2 * maxK + 6], r->ev_t[4 * 2 * maxK + 7]) : 0.f;
    return 0;
}

// ---- exact-count exchange: exact split, counts read by the host, messages of exact size, standard local path ----
int join_exact(hj_dist_rank *r, const int32_t *const cols[4], const uint64_t n[2], uint64_t out[2], bool balance, MatOut *mat) {
    hj_ctx *c = r->c;
    const uint32_t G = (uint32_t)r->world, me = (uint32_t)r->rank;
    hipStream_t cs = c->stream, ms = r->comm;
    uint64_t *small = (uint64_t *)r->small.p;
    LRET(r, dist_ensure(r, r->x_counts, (size_t)G * G * 8 + (size_t)G * 8));
    uint64_t recv_tot[2] = {0, 0};
    r->st.link_bytes = 0;
    // Skew across GPUs: a heavy hitter cannot be split, but its shard need not share a GPU with an average load.  8 virtual
    // shards per GPU; both relations are counted per virtual shard (keys only), the counts summed over the ranks, the shards
    // dealt to GPUs longest-first by |R|+|S| (the reference's size-aware placement idea, partition-primitives.cu:307-468), and the
    // split writes them in owner order so that every peer still gets ONE contiguous run per column.  Deterministic: every rank
    // computes the same assignment from the same summed counts.
    const uint32_t V = 8, ns = balance ? G * V : G;
    std::vector<uint32_t> owner(ns), position(ns);
    for (uint32_t v = 0; v < ns; v++) { owner[v] = v % G; position[v] = v; }
    if (balance) {
        if (ns > 512) return r->fail(HJ_EINVAL, "size-aware assignment needs world <= 64");
        std::vector<uint64_t> cr(ns), csz(ns), tot(ns);
        DRET(r, hj_shard_count(c, cols[0], n[0], ns, cr.data()));
        DRET(r, hj_shard_count(c, cols[2], n[1], ns, csz.data()));
        for (uint32_t v = 0; v < ns; v++) r->h_small[256 + v] = cr[v] + csz[v];
        LRET(r, dist_ensure(r, r->x_bal, (size_t)ns * 8 * (G + 1)));
        uint64_t *d_cnt = (uint64_t *)r->x_bal.p;
        DCHK(r, hipMemcpyAsync(d_cnt, r->h_small + 256, (size_t)ns * 8, hipMemcpyHostToDevice, cs));
        LRET(r, coll_begin(r));
        LRET(r, r->link->allreduce_sum_u64(d_cnt, ns, d_cnt + ns, ms, r->err));
        LRET(r, coll_end(r));
        DCHK(r, hipMemcpyAsync(tot.data(), d_cnt, (size_t)ns * 8, hipMemcpyDeviceToHost, cs));
        r->stage = "all-reduce of the shard sizes";
        LRET(r, wait_stream(r, cs, "the all-reduce of the virtual-shard sizes"));
        std::vector<uint32_t> by(ns);
        for (uint32_t v = 0; v < ns; v++) by[v] = v;
        std::stable_sort(by.begin(), by.end(), [&](uint32_t a, uint32_t b) { return tot[a] > tot[b]; });
        std::vector<uint64_t> load(G, 0);
        for (uint32_t v : by) {
            uint32_t g = 0;
            for (uint32_t q = 1; q < G; q++) if (load[q] < load[g]) g = q;
            owner[v] = g; load[g] += tot[v];
        }
        std::vector<uint32_t> order(ns);
        for (uint32_t v = 0; v < ns; v++) order[v] = v;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return owner[a] < owner[b]; });
        for (uint32_t pos = 0; pos < ns; pos++) position[order[pos]] = pos;
    }
    for (int x = 0; x < 2; x++) {
        {
            int arc = dist_ensure(r, r->x_send_k[x], (size_t)(n[x] + PAD) * 4);
            if (!arc) arc = dist_ensure(r, r->x_send_p[x], (size_t)(n[x] + PAD) * 4);
            LRET(r, agree(r, arc, "allocating the send side of the exact exchange"));
        }
        std::vector<uint64_t> cnt(G, 0);
        // level-0 split, one contiguous run per owner; [sync]: the run lengths come back to the host
        if (!balance) {
            DRET(r, hj_shard_split(c, cols[2 * x], cols[2 * x + 1], n[x], G, (int32_t *)r->x_send_k[x].p, (int32_t *)r->x_send_p[x].p, cnt.data()));
        } else {
            std::vector<uint64_t> per_pos(ns, 0);
            DRET(r, hj_shard_split_ordered(c, cols[2 * x], cols[2 * x + 1], n[x], ns, position.data(), (int32_t *)r->x_send_k[x].p,
                                           (int32_t *)r->x_send_p[x].p, per_pos.data()));
            for (uint32_t v = 0; v < ns; v++) cnt[owner[v]] += per_pos[position[v]];
        }
        // every rank's counts to every rank
        uint64_t *d_mine = (uint64_t *)r->x_counts.p + (size_t)G * G, *d_all = (uint64_t *)r->x_counts.p;
        DCHK(r, hipMemcpyAsync(d_mine, cnt.data(), (size_t)G * 8, hipMemcpyHostToDevice, cs));
        DCHK(r, hipStreamSynchronize(cs)); // cnt is pageable
        LRET(r, coll_begin(r));
        LRET(r, r->link->allgather(d_mine, d_all, (size_t)G * 8, ms, r->err));
        LRET(r, coll_end(r));
        std::vector<uint64_t> all((size_t)G * G);
        DCHK(r, hipMemcpyAsync(all.data(), d_all, (size_t)G * G * 8, hipMemcpyDeviceToHost, cs));
        r->stage = "all-gather of the shard counts"; r->stage_rel = x;
        LRET(r, wait_stream(r, cs, "the all-gather of the exact shard counts"));
        std::vector<uint64_t> soff(G + 1, 0), roff(G + 1, 0);
        for (uint32_t q = 0; q < G; q++) { soff[q + 1] = soff[q] + cnt[q]; roff[q + 1] = roff[q] + all[(size_t)q * G + me]; }
        recv_tot[x] = roff[G];
        {   // the receive side is sized from the counts: the ranks agree that everybody could allocate before anything is sent
            int arc = dist_ensure(r, r->x_recv_k[x], (size_t)(roff[G] + PAD) * 4);
            if (!arc) arc = dist_ensure(r, r->x_recv_p[x], (size_t)(roff[G] + PAD) * 4);
            LRET(r, agree(r, arc, "allocating the receive side of the exact exchange"));
        }
        // the split ran on the compute stream; the exchange goes to the communication stream
        DCHK(r, hipEventRecord(r->ev_misc[x], cs));
        DCHK(r, hipStreamWaitEvent(ms, r->ev_misc[x], 0));
        std::vector<Msg> msgs;
        for (int col = 0; col < 2; col++) {
            const int32_t *sb = (const int32_t *)(col ? r->x_send_p[x].p : r->x_send_k[x].p);
            int32_t *rb = (int32_t *)(col ? r->x_recv_p[x].p : r->x_recv_k[x].p);
            for (uint32_t step = 0; step < G; step++) {
                const uint32_t p = (me + step) % G;
                if (p == me && !r->cfg.self_via_link) {
                    if (cnt[p]) DCHK(r, hipMemcpyAsync(rb + roff[p], sb + soff[p], cnt[p] * 4, hipMemcpyDeviceToDevice, ms));
                    continue;
                }
                msgs.push_back(Msg{(int)p, sb + soff[p], (size_t)cnt[p] * 4, rb + roff[p], (size_t)all[(size_t)p * G + me] * 4});
                if (p != me) r->st.link_bytes += cnt[p] * 4;
            }
        }
        LRET(r, r->link->exchange(msgs, ms, r->err));
        DCHK(r, hipEventRecord(r->ev_misc[2 + x], ms));
    }
    // local path on what arrived: R's partition passes run while S is still on the links
    DRET(r, hj_bind_device(c, HJ_REL_R, (const int32_t *)r->x_recv_k[0].p, (const int32_t *)r->x_recv_p[0].p, recv_tot[0]));
    DRET(r, hj_bind_device(c, HJ_REL_S, (const int32_t *)r->x_recv_k[1].p, (const int32_t *)r->x_recv_p[1].p, recv_tot[1]));
    DCHK(r, hipStreamWaitEvent(cs, r->ev_misc[2], 0));
    DRET(r, hj_partition(c, HJ_REL_R));
    DCHK(r, hipStreamWaitEvent(cs, r->ev_misc[3], 0));
    DRET(r, hj_partition(c, HJ_REL_S));
    r->stage = "exact exchange"; r->stage_rel = -1;
    LRET(r, wait_stream(r, ms, "the exact-size exchange (grouped send/recv of both relations)")); // hj_join_count synchronises: the deadline first
    LRET(r, wait_stream(r, cs, "the local partition passes behind the exact exchange"));
    uint64_t m = 0, a = 0;
    if (mat) { // what arrived is local from here on: the one-probe materialiser with its own redo of a skewed relation
        DRET(r, materialize_local(c, mat->key, mat->payR, mat->payS, mat->cap, &m));
        mat->n_out = m;
        if (mat->want_agg) {
            uint64_t *sc = (uint64_t *)c->scalars.p;
            DCHK(r, hipMemsetAsync(sc + 3, 0, 8, cs));
            DCHK(r, launch_dot(cs, mat->payR, mat->payS, sc + SC_CURSOR, mat->cap, sc + 3));
            if (fetch_scalars(c)) { r->err = hj_error(c); return HJ_EHIP; }
            a = c->h_scalars[3];
        }
    } else {
        DRET(r, hj_join_count(c, &m, &a));
    }
    r->h_small[0] = m; r->h_small[1] = a;
    DCHK(r, hipMemcpyAsync(small, r->h_small, 16, hipMemcpyHostToDevice, cs));
    LRET(r, coll_begin(r));
    LRET(r, r->link->allreduce_sum_u64(small, 2, small + 64, ms, r->err));
    LRET(r, coll_end(r));
    DCHK(r, hipMemcpyAsync(r->h_small + 8, small, 16, hipMemcpyDeviceToHost, cs));
    r->stage = "all-reduce of the result";
    LRET(r, wait_stream(r, cs, "the all-reduce of the result"));
    LRET(r, wait_stream(r, ms, "the communication stream after the all-reduce"));
    out[0] = r->h_small[8]; out[1] = r->h_small[9];
    if (mat) LRET(r, gather_outputs(r, mat));
    r->st.received[0] = recv_tot[0]; r->st.received[1] = recv_tot[1];
    r->st.path = 1; r->st.slices = 1; r->st.balanced = balance ? 1 : 0; r->st.exchange_ms = 0;
    r->st.payload_bytes = r->st.link_bytes;
    return 0;
}

int rank_join_inner(hj_dist_rank *r, const int32_t *Rk, const int32_t *Rp, uint64_t nR, const int32_t *Sk, const int32_t *Sp, uint64_t nS,
                    uint64_t *matches, uint64_t *agg, MatOut *mat);

// A rank that fails for any reason aborts its group on the way out: peers inside (or on their way into) a collective then leave
// with an error that names this rank, instead of waiting for the deadline — or, in the reference's terms, instead of the
// print-and-exit of CHK_ERROR (common.h:132-141) taking one process down while the others hang.
int rank_join(hj_dist_rank *r, const int32_t *Rk, const int32_t *Rp, uint64_t nR, const int32_t *Sk, const int32_t *Sp, uint64_t nS,
              uint64_t *matches, uint64_t *agg, MatOut *mat = nullptr) {
    r->err.clear();
    r->cur_maxK = 0;
    r->timeout_s = r->cfg.timeout_ms ? r->cfg.timeout_ms * 1e-3 : env_timeout_s();
    {
        std::string why;
        if (r->link->failed(&why)) return r->fail(HJ_EHIP, "rank %d: the group was aborted by an earlier failure (%s): create a new one", r->rank, why.c_str());
    }
    r->st.materializing = mat ? 1u : 0u; r->st.materialized = 0;
    const int rc = rank_join_inner(r, Rk, Rp, nR, Sk, Sp, nS, matches, agg, mat);
    if (rc == HJ_ECAPACITY && mat) { r->stage = "idle"; return rc; } // every rank saw the same all-gathered sizes: an answer, not a failure of the group
    if (rc == HJ_EINVAL && r->args_rejected) { r->stage = "idle"; return rc; } // ... and so is a rejected argument: all ranks learned it from the first all-gather
    if (rc) {
        if (r->err.empty()) r->err = hj_error(r->c);
        r->link->abort("rank " + std::to_string(r->rank) + " failed: " + r->err);
        hj_invalidate_all(r->c);
    }
    r->stage = "idle";
    return rc;
}

int rank_join_inner(hj_dist_rank *r, const int32_t *Rk, const int32_t *Rp, uint64_t nR, const int32_t *Sk, const int32_t *Sp, uint64_t nS,
                    uint64_t *matches, uint64_t *agg, MatOut *mat) {
    hj_ctx *c = r->c;
    // A caller's mistake on ONE rank: the verdict travels with the first all-gather, every rank returns HJ_EINVAL naming the
    // rank and the reason, nothing has been exchanged and the group stays usable — the link is aborted for failures, not for arguments.
    static const char *const kBadArgs[] = {"", "null column", "output columns == NULL", "device columns must be 16-byte aligned"};
    uint64_t bad = 0;
    if ((nR && (!Rk || !Rp)) || (nS && (!Sk || !Sp))) bad = 1;
    else if (mat && mat->cap && (!mat->key || !mat->payR || !mat->payS)) bad = 2;
    else if ((((uintptr_t)Rk | (uintptr_t)Rp | (uintptr_t)Sk | (uintptr_t)Sp) & 15)) bad = 3;
    r->args_rejected = false;
    DCHK(r, hipSetDevice(c->device));
    const auto t0 = std::chrono::steady_clock::now();
    const int32_t *cols[4] = {Rk, Rp, Sk, Sp};
    const uint64_t n[2] = {nR, nS};
    if (memcmp(cols, r->last_cols, sizeof cols) || n[0] != r->last_n[0] || n[1] != r->last_n[1]) r->prefer_exact = false; // new data
    memcpy(r->last_cols, cols, sizeof cols); r->last_n[0] = n[0]; r->last_n[1] = n[1];
    // nominal sizes = the largest local slice of each relation over the ranks, and whether anybody wants the exact path:
    // one small all-gather, read by the host before anything is planned (sizes decide the geometry on every rank)
    uint64_t *small = (uint64_t *)r->small.p;
    r->h_small[0] = nR; r->h_small[1] = nS; r->h_small[2] = ((r->prefer_exact || r->cfg.exact_only) ? 1 : 0) | (bad << 1);
    r->h_small[3] = (r->prefer_exact || r->cfg.balance_size) ? 1 : 0; // skew was seen on these columns (or the caller asks): size-aware shards
    DCHK(r, hipMemcpyAsync(small + 16, r->h_small, 32, hipMemcpyHostToDevice, c->stream));
    LRET(r, coll_begin(r));
    LRET(r, r->link->allgather(small + 16, small + 128, 32, r->comm, r->err));
    LRET(r, coll_end(r));
    DCHK(r, hipMemcpyAsync(r->h_small + 32, small + 128, (size_t)r->world * 32, hipMemcpyDeviceToHost, c->stream));
    r->stage = "all-gather of the sizes";
    LRET(r, wait_stream(r, c->stream, "the all-gather of the local sizes (first collective of the join)"));
    uint64_t nmax[2] = {0, 0};
    bool exact = false, balance = false;
    for (int q = 0; q < r->world; q++) {
        if (const uint64_t why = r->h_small[32 + 4 * q + 2] >> 1) { // (the lowest such rank: every rank reports the same one)
            r->args_rejected = true;
            return r->fail(HJ_EINVAL, "rank %d of %d rejected its arguments: %s (nothing was exchanged; the group stays usable)", q, r->world, kBadArgs[why < 4 ? why : 0]);
        }
        nmax[0] = std::max(nmax[0], r->h_small[32 + 4 * q]); nmax[1] = std::max(nmax[1], r->h_small[32 + 4 * q + 1]);
        exact |= (r->h_small[32 + 4 * q + 2] & 1) != 0;
        balance |= r->h_small[32 + 4 * q + 3] != 0;
    }
    uint64_t out[2] = {0, 0};
    int rc = 0;
    bool done = false;
    if (!exact) {
        bool flagged = false, applicable = false;
        rc = join_fast(r, cols, n, nmax, out, &flagged, &applicable, mat);
        if (rc == HJ_ECAPACITY && mat) { if (matches) *matches = out[0]; if (agg) *agg = out[1]; }
        if (rc) return rc;
        if (applicable && !flagged) done = true;
        if (flagged) { r->prefer_exact = true; balance = true; } // every rank saw the same summed flags: everybody goes exact together
    }
    if (!done) {
        hj_invalidate_all(c);
        rc = join_exact(r, cols, n, out, balance && r->world > 1, mat);
        if (rc == HJ_ECAPACITY && mat) { if (matches) *matches = out[0]; if (agg) *agg = out[1]; }
        if (rc) return rc;
    }
    r->st.wall_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (matches) *matches = out[0];
    if (agg) *agg = out[1];
    return HJ_OK;
}

} // namespace

// ================================================================================================
// one process, G ranks
// ================================================================================================
struct hj_dist {
    int world = 0;
    std::vector<hj_dist_rank *> ranks;
    std::unique_ptr<CopyGroup> copy;
    std::vector<int> dev;                    // HIP device of every rank
    std::string err, transport;
    struct Bound { const int32_t *k = nullptr, *p = nullptr; uint64_t n = 0; } bound[64][2];
    MatOut outb[64];
};

extern "C" {

// The links of a group: made at creation and again by hj_dist_set_transport (the ranks, their contexts and every buffer stay).
// want: "" / "auto" ($HJ_DIST_TRANSPORT, else RCCL for distinct devices and device copies for shared ones), "rccl", "device-copy".
static int make_links(hj_dist *d, std::string want) {
    const int nranks = d->world;
    const std::vector<int> &dev = d->dev;
    bool distinct = true;
    for (int r = 0; r < nranks; r++) for (int q = 0; q < r; q++) distinct &= dev[q] != dev[r];
    if (want.empty() || want == "auto") { const char *e = getenv("HJ_DIST_TRANSPORT"); want = e ? e : ""; }
    if (want == "copy") want = "device-copy";
    if (!want.empty() && want != "auto" && want != "rccl" && want != "device-copy") { d->err = "unknown transport '" + want + "' (rccl, device-copy = copy, auto)"; return HJ_EINVAL; }
    if (want == "rccl" && !distinct) { d->err = "RCCL refuses ranks that share a device: device-copy is the transport of such a group"; return HJ_EINVAL; }
    const bool use_rccl = want == "rccl" || ((want.empty() || want == "auto") && distinct);
    std::vector<ncclComm_t> comms(nranks, nullptr);
    std::unique_ptr<CopyGroup> copy;
    if (use_rccl) {
        if (!rccl().ok || rccl().CommInitAll(comms.data(), nranks, dev.data()) != ncclSuccess) { d->err = "ncclCommInitAll failed"; return HJ_EHIP; }
    } else {
        copy.reset(new CopyGroup(nranks));
        copy->dev = dev;
        copy->timeout_s = (!d->ranks.empty() && d->ranks[0]->cfg.timeout_ms) ? d->ranks[0]->cfg.timeout_ms * 1e-3 : env_timeout_s();
        // distinct devices: every rank reads its peers' regions directly (xGMI peer access)
        for (int r = 0; r < nranks && distinct; r++) {
            if (hipSetDevice(dev[r]) != hipSuccess) return HJ_EHIP;
            for (int q = 0; q < nranks; q++) {
                if (q == r) continue;
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, dev[r], dev[q]) != hipSuccess || !can) { d->err = "no peer access between the devices of ranks " + std::to_string(r) + " and " + std::to_string(q); return HJ_EHIP; }
                const hipError_t e = hipDeviceEnablePeerAccess(dev[q], 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return HJ_EHIP;
                (void)hipGetLastError();
            }
        }
        for (int r = 0; r < nranks; r++)
            if (hipSetDevice(dev[r]) != hipSuccess || hipEventCreateWithFlags(&copy->ready[r], hipEventDisableTiming) != hipSuccess) {
                for (auto e : copy->ready) if (e) (void)hipEventDestroy(e);
                return HJ_EHIP;
            }
    }
    // the old links go (a healthy group: drained first), the new ones come
    for (int r = 0; r < nranks; r++) {
        hj_dist_rank *k = d->ranks[r];
        if (k->link) { (void)hipSetDevice(dev[r]); if (!k->link->failed(nullptr) && k->comm) (void)hipStreamSynchronize(k->comm); k->link.reset(); }
    }
    if (d->copy) for (auto e : d->copy->ready) if (e) (void)hipEventDestroy(e);
    d->copy = std::move(copy);
    for (int r = 0; r < nranks; r++) {
        hj_dist_rank *k = d->ranks[r];
        if (use_rccl) { RcclLink *l = new RcclLink(); l->comm = comms[r]; l->rank = r; l->world = nranks; k->link.reset(l); }
        else { CopyLink *l = new CopyLink(); l->g = d->copy.get(); l->rank = r; k->link.reset(l); }
        memset(k->agreed, 0, sizeof k->agreed);
    }
    d->transport = use_rccl ? "rccl" : "device-copy";
    return HJ_OK;
}

int hj_dist_create_transport(hj_dist **out, int nranks, const int *devices, const char *transport) {
    if (!out) return HJ_EINVAL;
    *out = nullptr;
    if (nranks < 1 || nranks > 64) return HJ_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return HJ_EHIP; // no GPU: fail loudly
    hj_dist *d = new hj_dist();
    d->world = nranks;
    d->dev.resize(nranks);
    for (int r = 0; r < nranks; r++) {
        d->dev[r] = devices ? devices[r] : r;
        if (d->dev[r] < 0 || d->dev[r] >= ndev) { delete d; return HJ_EINVAL; } // fewer GPUs visible than ranks asked for
    }
    int rc = 0;
    for (int r = 0; r < nranks && !rc; r++) {
        hj_dist_rank *k = new hj_dist_rank();
        d->ranks.push_back(k);
        k->rank = r; k->world = nranks; k->own_ctx = true;
        rc = hj_create(&k->c, d->dev[r]);
        if (!rc) rc = rank_init(k);
    }
    if (!rc) rc = make_links(d, transport ? transport : "");
    if (rc) { hj_dist_destroy(d); return rc; }
    *out = d;
    return HJ_OK;
}

// Another transport for the SAME group (contexts, bound columns, every buffer stay): what a multi-GPU node uses to time one workload
// over RCCL's kernels and over the copy engines back to back (bench.py dist.alt_transport).  On failure the group keeps its links.
int hj_dist_set_transport(hj_dist *d, const char *transport) {
    if (!d) return HJ_EINVAL;
    for (auto *k : d->ranks) { std::string why; if (k->link && k->link->failed(&why)) { d->err = "the group was aborted (" + why + "): create a new one"; return HJ_EHIP; } }
    return make_links(d, transport ? transport : "");
}

int hj_dist_create(hj_dist **out, int nranks, const int *devices) { return hj_dist_create_transport(out, nranks, devices, nullptr); }

int hj_dist_destroy(hj_dist *d) {
    if (!d) return HJ_EINVAL;
    for (auto *k : d->ranks) rank_free(k);
    if (d->copy) for (auto e : d->copy->ready) if (e) (void)hipEventDestroy(e);
    delete d;
    return HJ_OK;
}

const char *hj_dist_error(const hj_dist *d) { return d ? d->err.c_str() : "null hj_dist"; }
int hj_dist_world(const hj_dist *d) { return d ? d->world : 0; }
const char *hj_dist_transport(const hj_dist *d) { return d ? d->transport.c_str() : ""; }
hj_ctx *hj_dist_context(hj_dist *d, int rank) { return (d && rank >= 0 && rank < d->world) ? d->ranks[rank]->c : nullptr; }

int hj_dist_configure(hj_dist *d, const hj_dist_config *cfg) {
    if (!d || !cfg) return HJ_EINVAL;
    if (cfg->slices > 64) { d->err = "at most 64 slices"; return HJ_EINVAL; }
    if (cfg->phantom_world > 512 || (cfg->phantom_world > 1 && d->world != 1)) { d->err = "phantom_world needs world size 1 and <= 512 shards"; return HJ_EINVAL; }
    for (auto *k : d->ranks) k->cfg = *cfg;
    if (d->copy) { std::lock_guard<std::mutex> lk(d->copy->mu); d->copy->timeout_s = cfg->timeout_ms ? cfg->timeout_ms * 1e-3 : env_timeout_s(); }
    return HJ_OK;
}

int hj_dist_bind(hj_dist *d, int rank, int rel, const int32_t *d_keys, const int32_t *d_pays, uint64_t n) {
    if (!d || rank < 0 || rank >= d->world || (rel != HJ_REL_R && rel != HJ_REL_S)) return HJ_EINVAL;
    d->bound[rank][rel].k = d_keys; d->bound[rank][rel].p = d_pays; d->bound[rank][rel].n = n;
    return HJ_OK;
}

int hj_dist_join(hj_dist *d, uint64_t *matches, uint64_t *agg) {
    if (!d) return HJ_EINVAL;
    std::vector<int> rc(d->world, 0);
    std::vector<uint64_t> m(d->world, 0), a(d->world, 0);
    std::vector<std::thread> th;
    // one host thread per GPU: each enqueues its rank's pipeline; the collectives meet on the links
    for (int r = 0; r < d->world; r++)
        th.emplace_back([&, r] {
            const auto &b = d->bound[r];
            rc[r] = rank_join(d->ranks[r], b[0].k, b[0].p, b[0].n, b[1].k, b[1].p, b[1].n, &m[r], &a[r]);
        });
    for (auto &t : th) t.join();
    for (int r = 0; r < d->world; r++)
        if (rc[r]) { d->err = "rank " + std::to_string(r) + ": " + d->ranks[r]->err; return rc[r]; }
    for (int r = 1; r < d->world; r++)
        if (m[r] != m[0] || a[r] != a[0]) { d->err = "ranks disagree on the all-reduced result"; return HJ_EHIP; }
    if (matches) *matches = m[0];
    if (agg) *agg = a[0];
    return HJ_OK;
}

int hj_dist_bind_output(hj_dist *d, int rank, int32_t *d_key, int32_t *d_payR, int32_t *d_payS, uint64_t cap) {
    if (!d || rank < 0 || rank >= d->world) return HJ_EINVAL;
    if (cap && (!d_key || !d_payR || !d_payS)) { d->err = "output columns == NULL"; return HJ_EINVAL; }
    MatOut &o = d->outb[rank];
    o.key = d_key; o.payR = d_payR; o.payS = d_payS; o.cap = cap;
    return HJ_OK;
}

int hj_dist_join_materialize(hj_dist *d, uint64_t *matches, uint64_t *agg, uint64_t *n_out) {
    if (!d) return HJ_EINVAL;
    std::vector<int> rc(d->world, 0);
    std::vector<uint64_t> m(d->world, 0), a(d->world, 0);
    std::vector<std::thread> th;
    for (int r = 0; r < d->world; r++)
        th.emplace_back([&, r] {
            const auto &b = d->bound[r];
            d->outb[r].want_agg = agg != nullptr;
            rc[r] = rank_join(d->ranks[r], b[0].k, b[0].p, b[0].n, b[1].k, b[1].p, b[1].n, &m[r], &a[r], &d->outb[r]);
        });
    for (auto &t : th) t.join();
    int hard = 0, soft = 0;
    for (int r = 0; r < d->world; r++) {
        if (rc[r] && rc[r] != HJ_ECAPACITY && !hard) { d->err = "rank " + std::to_string(r) + ": " + d->ranks[r]->err; hard = rc[r]; }
        if (rc[r] == HJ_ECAPACITY && !soft) { soft = rc[r]; if (!hard) d->err = d->ranks[r]->err; }
    }
    if (hard) return hard;
    for (int r = 1; r < d->world; r++)
        if (m[r] != m[0] || a[r] != a[0]) { d->err = "ranks disagree on the all-reduced result"; return HJ_EHIP; }
    if (matches) *matches = m[0];
    if (agg) *agg = a[0];
    if (n_out) for (int r = 0; r < d->world; r++) n_out[r] = r < (int)d->ranks[0]->n_out_all.size() ? d->ranks[0]->n_out_all[r] : 0;
    return soft;
}

/* tests only: rank `rank` (>= 0) of every group of this process stops taking part in its next probe-side exchange for 2.5 deadlines — a
 * stalled peer; -1 = nobody (include/hj_dist.h, "tests only"). */
int hj_dist_debug_stall_rank(int rank) { g_debug_stall_rank.store(rank + 1); return HJ_OK; }

int hj_dist_get_stats(hj_dist *d, int rank, hj_dist_stats *out) {
    if (!d || !out || rank < 0 || rank >= d->world) return HJ_EINVAL;
    *out = d->ranks[rank]->st;
    return HJ_OK;
}

// ---- one process per GPU ----
int hj_dist_unique_id(void *id128) {
    if (!id128) return HJ_EINVAL;
    static_assert(sizeof(ncclUniqueId) == HJ_DIST_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    if (!rccl().ok || rccl().GetUniqueId(&id) != ncclSuccess) return HJ_EHIP;
    memcpy(id128, &id, sizeof id);
    return HJ_OK;
}

int hj_dist_rank_create(hj_dist_rank **out, hj_ctx *ctx, int rank, int world, const void *id128) {
    if (!out) return HJ_EINVAL;
    *out = nullptr;
    // the pipeline's fixed-size scratch (sizes, flags, all-reduce staging) and the split's LDS lines are laid out for <= 64 ranks
    if (!ctx || !id128 || world < 1 || world > 64 || rank < 0 || rank >= world) return HJ_EINVAL;
    if (hipSetDevice(ctx->device) != hipSuccess) return HJ_EHIP;
    if (!rccl().ok) return HJ_EHIP;
    // ncclCommInitRank is collective and blocks until every rank has called it: it runs on a helper thread so that a rank that
    // never arrives costs a deadline and a message, not a hang.  On expiry the helper is left behind (it cannot be cancelled).
    struct Init { std::mutex mu; std::condition_variable cv; bool done = false; ncclResult_t res = ncclSuccess; ncclComm_t comm = nullptr; };
    auto st = std::make_shared<Init>();
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    const int device = ctx->device;
    std::thread([st, id, rank, world, device] {
        (void)hipSetDevice(device);
        ncclComm_t comm = nullptr;
        const ncclResult_t res = rccl().CommInitRank(&comm, world, id, rank);
        std::lock_guard<std::mutex> lk(st->mu);
        st->res = res; st->comm = comm; st->done = true;
        st->cv.notify_all();
    }).detach();
    {
        std::unique_lock<std::mutex> lk(st->mu);
        if (!st->cv.wait_for(lk, std::chrono::duration<double>(env_timeout_s()), [&] { return st->done; })) {
            fprintf(stderr, "[hj_dist] rank %d of %d: deadline of %.1f s passed inside ncclCommInitRank: some rank never called hj_dist_rank_create "
                            "(or the id did not reach it)\n", rank, world, env_timeout_s());
            return HJ_EHIP;
        }
        if (st->res != ncclSuccess) return HJ_EHIP;
    }
    RcclLink *l = new RcclLink();
    l->rank = rank; l->world = world; l->comm = st->comm;
    hj_dist_rank *k = new hj_dist_rank();
    k->c = ctx; k->rank = rank; k->world = world; k->link.reset(l);
    int rc = rank_init(k);
    if (rc) { rank_free(k); return rc; }
    *out = k;
    return HJ_OK;
}

int hj_dist_rank_destroy(hj_dist_rank *r) {
    if (!r) return HJ_EINVAL;
    rank_free(r);
    return HJ_OK;
}

const char *hj_dist_rank_error(const hj_dist_rank *r) { return r ? r->err.c_str() : "null hj_dist_rank"; }

int hj_dist_rank_configure(hj_dist_rank *r, const hj_dist_config *cfg) {
    if (!r || !cfg) return HJ_EINVAL;
    if (cfg->slices > 64) return r->fail(HJ_EINVAL, "at most 64 slices");
    if (cfg->phantom_world > 512 || (cfg->phantom_world > 1 && r->world != 1)) return r->fail(HJ_EINVAL, "phantom_world needs world size 1 and <= 512 shards");
    r->cfg = *cfg;
    return HJ_OK;
}

int hj_dist_rank_join(hj_dist_rank *r, const int32_t *d_Rk, const int32_t *d_Rp, uint64_t nR, const int32_t *d_Sk, const int32_t *d_Sp,
                      uint64_t nS, uint64_t *matches, uint64_t *agg) {
    if (!r) return HJ_EINVAL;
    return rank_join(r, d_Rk, d_Rp, nR, d_Sk, d_Sp, nS, matches, agg);
}

int hj_dist_rank_join_materialize(hj_dist_rank *r, const int32_t *d_Rk, const int32_t *d_Rp, uint64_t nR, const int32_t *d_Sk, const int32_t *d_Sp,
                                  uint64_t nS, int32_t *d_out_key, int32_t *d_out_payR, int32_t *d_out_payS, uint64_t cap, uint64_t *n_out,
                                  uint64_t *n_out_all, uint64_t *matches, uint64_t *agg) {
    if (!r) return HJ_EINVAL;
    MatOut o;
    o.key = d_out_key; o.payR = d_out_payR; o.payS = d_out_payS; o.cap = cap; o.want_agg = agg != nullptr;
    const int rc = rank_join(r, d_Rk, d_Rp, nR, d_Sk, d_Sp, nS, matches, agg, &o);
    if (rc && rc != HJ_ECAPACITY) return rc;
    if (n_out) *n_out = o.n_out;
    if (n_out_all) for (int q = 0; q < r->world; q++) n_out_all[q] = q < (int)r->n_out_all.size() ? r->n_out_all[q] : 0;
    return rc;
}

int hj_dist_rank_get_stats(hj_dist_rank *r, hj_dist_stats *out) {
    if (!r || !out) return HJ_EINVAL;
    *out = r->st;
    return HJ_OK;
}

} // extern "C"
