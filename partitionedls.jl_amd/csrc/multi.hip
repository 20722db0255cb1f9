// multi.hip — fit(Opt) and fit(BnB) over several GPUs of one node behind the C ABI (include/partls.h: partls_fit_opt_multi,
// partls_fit_bnb_multi).
//
// Opt: the loop of the reference, Opt.jl:85-94, has no loop-carried state: the 2^K' sign patterns are sharded over the devices by
// Gray-index range, exactly as partitionedls.jl_amd/dist.py does across processes — here inside one process, one host thread
// and one partls_ctx per device, so that a Julia `fit(Opt, X, y, P)` uses the whole node without any glue on its side.
// The winner is the lexicographic minimum (objective, reference pattern index) — argmin's first-index rule, Opt.jl:96 — taken
// with two RCCL all-reduces (ncclMin on the objective, ncclMin on the index masked to the minimisers): 24 + 8 bytes over xGMI,
// latency-bound; the shards' near ties (<= 4 candidates of 16 bytes per rank) are handed to rank 0 so that the finish re-ranks the
// set a single context would.
// BnB: the subtrees below two open nodes are independent given the incumbent (BnB.jl:94-132); every rank thread runs the same
// native frontier (frontier.h), bounds its share of every round on its own GPU (warm-started from the parent's tableau snapshot,
// which lives on the rank that bounded the parent), and ONE all-gather of (bound, branch, slot) per round — ncclAllGather over
// xGMI — carries the incumbent.
// RCCL is loaded on first use (573 MB on disk: a single-GPU fit never pays for it).
//
// PROTOCOL.  A fit is a fixed sequence of phases; every phase ends in a rendezvous (HostBarrier) at which the ranks publish their
// status, and after which ALL of them either go on or leave — a rank that failed joins the rendezvous it owes from its error path,
// so nobody is ever left waiting, and no collective is entered unless every rank has just agreed to enter it.  Between such an
// agreement and the enqueue of the collective there is no fallible host allocation (buffers are sized before the threads start).
// A rank that cannot keep the protocol (an exception on its thread, a rendezvous that times out: PARTLS_MULTI_TIMEOUT_S, default
// 3600 s) POISONS the barrier: every other rank's next rendezvous fails at once and the fit returns PARTLS_ERR_STATE instead of
// hanging.  A failed RCCL enqueue aborts every communicator of the handle (ncclCommAbort releases the ranks already inside the
// collective); the handle then reduces through host memory for the rest of its life.
#include "ctx.h"
#include "frontier.h"
#include <new>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace partls;

namespace {

// ---- RCCL, bound at run time --------------------------------------------------------------------------------------------
struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {getenv("PARTLS_RCCL_LIB"), "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
        // a copy the process already holds (PyTorch brings its own) wins: two RCCLs in one process would each initialise the fabric
        r.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        for (const char *n : names) {
            if (r.handle) break;
            if (n && *n) r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        }
        if (!r.handle) { r.error = std::string("librccl.so.1 could not be loaded: ") + (dlerror() ? dlerror() : "not found"); return; }
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(r.handle, "ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.handle, "ncclCommDestroy"));
        r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(r.handle, "ncclCommAbort"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.handle, "ncclAllReduce"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.handle, "ncclAllGather"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.handle, "ncclGetErrorString"));
        if (!r.CommInitAll || !r.CommDestroy || !r.AllReduce || !r.AllGather || !r.GetErrorString) r.error = "librccl.so.1 lacks ncclCommInitAll / ncclAllReduce / ncclAllGather";
    });
    return r;
}

// A barrier all ranks pass together; reusable (generation counter).  wait() returns false when the barrier is BROKEN: a rank did not
// arrive within the timeout, or a rank that could not keep the protocol poisoned it.  A broken barrier stays broken until reset()
// (start of the next fit, when no rank thread exists): every later wait() fails at once.
struct HostBarrier {
    std::mutex m;
    std::condition_variable cv;
    int n = 1, waiting = 0;
    unsigned gen = 0;
    bool broken = false;
    double timeout_s = 3600.0;
    bool wait()
    {
        std::unique_lock<std::mutex> lk(m);
        if (broken) return false;
        const unsigned g = gen;
        if (++waiting == n) { waiting = 0; ++gen; cv.notify_all(); return true; }
        const bool woke = cv.wait_for(lk, std::chrono::duration<double>(timeout_s), [&] { return gen != g || broken; });
        if (gen != g) return true;                           // the barrier completed (a poison that came later concerns the next one)
        if (!woke) { broken = true; cv.notify_all(); }       // timeout: nobody else shall wait for the missing rank either
        return false;
    }
    void poison() { std::lock_guard<std::mutex> lk(m); broken = true; cv.notify_all(); }
    bool is_broken() { std::lock_guard<std::mutex> lk(m); return broken; }
    void reset() { std::lock_guard<std::mutex> lk(m); waiting = 0; broken = false; }
};

constexpr int64_t NO_CANDIDATE = (int64_t)1 << 62;

// CRC-32 of the group -> Gray-bit assignment: exact in a double (it rides in the objective all-reduce)
double order_key(const int64_t *gbit, int kb)
{
    uint32_t crc = 0xFFFFFFFFu;
    for (int k = 0; k < kb; ++k)
        for (int b = 0; b < 8; ++b) {
            crc ^= (uint32_t)((uint64_t)gbit[k] >> (8 * b)) & 0xFFu;
            for (int i = 0; i < 8; ++i) crc = (crc >> 1) ^ (0xEDB88320u & (0u - (crc & 1u)));
        }
    return (double)(crc ^ 0xFFFFFFFFu);
}

struct Cand { int n = 0; double obj[4] = {0, 0, 0, 0}; int64_t pat[4] = {-1, -1, -1, -1}; };

}  // namespace

struct partls_multi {
    int ndev = 0;
    std::vector<int> devices;
    std::vector<partls_ctx *> ctx;
    bool use_rccl = false;
    std::mutex comm_mutex;                   // abort_rccl
    std::vector<ncclComm_t> comms;
    std::vector<DevBuf> red;                 // per rank: [objective, key, -key | index] on its device
    HostBarrier bar;
    // per-call exchange between the rank threads
    std::vector<partls_status> st;
    std::vector<std::string> msg;
    std::vector<char> secondary;             // the rank's status only says "another rank failed": report that rank's instead
    std::vector<double> obj, key;
    std::vector<int64_t> pat, unconv;
    std::vector<Cand> cand;                  // every rank's winner + near ties of its shard (partls_opt_candidates)
    std::vector<std::vector<double>> all_opt;
    double win_obj = 0.0;
    int64_t win_pat = -1;
    std::vector<std::vector<double>> t_ms;
    // row-sharded fits: did rank r get as far as the Gram rendezvous (a rank that failed earlier joins it from its error path), the
    // status of its part of the exchange, and the host images of the partial Gram products of the host-reduced mode (sized before the
    // rank threads start: nothing is allocated between two rendezvous)
    std::vector<char> at_gram, xst;
    std::vector<std::vector<double>> gram_img, gram_sum;
    bool shard_rows = false;
    bool replicate = false;                  // PARTLS_MULTI_REPLICATE (read at create): every rank uploads all of X (A/B tests)
    // BnB: the per-round exchange of (bound, branch, slot).  Host mode: xbuf[parity][rank * stride ...] (written before the round's
    // rendezvous, read after it; two parities so that a fast rank's next round cannot overwrite what a slow one still reads).
    // RCCL mode: per-rank device buffers [send | recv] and pinned host images of both.
    std::vector<double> xbuf[2];
    size_t xstride = 0;
    std::vector<DevBuf> xdev;
    std::vector<PinnedDoubles> xpin;
    int64_t bnb_nodes = 0;
    // fault injection (tests): PARTLS_MULTI_FAULT="rank:stage[:vanish]", read at create.  Stage 1: before the upload, 2: inside the Gram
    // exchange, 3: after the sweep / before the search, 4: in the reduction / the second search round, 5: rank 0's finish.  Default: the
    // rank FAILS there (error status, protocol kept); "vanish" (stages 1, 3, 4): it leaves its thread without a word — the bounded
    // rendezvous must catch it.
    int fault_rank = -1, fault_stage = 0;
    bool fault_vanish = false;
};

namespace {

struct FitArgs {
    int kind;                                                // 0: fit(Opt), 1: fit(BnB)
    const double *X; int64_t N, M, ldX; const double *y; const int64_t *P; int64_t K, ldP; double eta; uint32_t flags;
    double *alpha, *beta, *t, *opt; int64_t *best_index; double *all_opt; int64_t *nopen;
};

void fail(partls_multi *mc, int r, partls_status st, bool secondary = false)
{
    if (mc->st[(size_t)r] != PARTLS_OK && !mc->secondary[(size_t)r]) return;     // keep the first primary failure of the rank
    mc->st[(size_t)r] = st;
    mc->secondary[(size_t)r] = secondary ? 1 : 0;
    try { mc->msg[(size_t)r] = partls_last_error(); } catch (...) { }
}

// a status returned by a call on this rank: recorded unless the rank already has one (the first failure is the one that explains the rest;
// a "another rank failed" status must not be promoted to a failure of this rank)
void note(partls_multi *mc, int r, partls_status st)
{
    if (st != PARTLS_OK && mc->st[(size_t)r] == PARTLS_OK) fail(mc, r, st);
}

partls_status lost(partls_multi *mc, int r)
{
    set_error("rank %d: a rendezvous of the rank threads failed (a rank did not arrive within %.0f s, or left the protocol)", r, mc->bar.timeout_s);
    fail(mc, r, PARTLS_ERR_STATE, true);
    return PARTLS_ERR_STATE;
}

bool any_failed(const partls_multi *mc)
{
    for (int q = 0; q < mc->ndev; ++q) if (mc->st[(size_t)q] != PARTLS_OK) return true;
    return false;
}

// 0: no fault here; 1: fail; 2: vanish
int fault_at(const partls_multi *mc, int r, int stage)
{
    if (mc->fault_rank != r || mc->fault_stage != stage) return 0;
    return mc->fault_vanish ? 2 : 1;
}
struct Vanish {};                                            // thrown by an injected "vanish": unwinds to the thread's entry, which returns silently

// A collective could not be enqueued on some rank: ranks that did enqueue theirs are stuck on the device until the communicators are
// aborted.  Once per handle; afterwards the handle reduces through host memory.
void abort_rccl(partls_multi *mc)
{
    std::lock_guard<std::mutex> lk(mc->comm_mutex);
    if (!mc->use_rccl) return;
    Rccl &R = rccl();
    for (ncclComm_t cm : mc->comms) if (cm) (void)(R.CommAbort ? R.CommAbort(cm) : R.CommDestroy(cm));
    mc->comms.clear();
    mc->use_rccl = false;
}

// Rows of X sharded over the ranks: every rank has built the Gram products of ITS row block; their sum is the problem's.  Called by
// ctx_prepare between the Gram build and the tableau preparation (ctx.h: gram_hook), on the rank's thread and stream.
//   rendezvous 1: is every rank here?  (a rank that failed earlier joins from its error path, prepare_rank)
//   exchange:     G <- sum_r G_r : ncclAllReduce(ncclSum) over xGMI on (M + 2)^2 doubles padded to ldg^2 (0.8 MB at C3) — every rank
//                 receives the same bits, so all of them prepare the same tableau and derive the same visiting order — or, without a
//                 communicator, device -> host images
//   rendezvous 2: did every rank's part of the exchange work?
//   (host mode)   the same sum in rank order; rendezvous 3: everybody has read the images
partls_status gram_rendezvous(partls_multi *mc, int r, partls_ctx *c)
{
    const int R = mc->ndev;
    mc->at_gram[(size_t)r] = 1;
    if (!mc->bar.wait()) return lost(mc, r);
    for (int q = 0; q < R; ++q)
        if (mc->at_gram[(size_t)q] != 1) { set_error("partls_fit_*_multi: another rank failed before the Gram products could be combined"); fail(mc, r, PARTLS_ERR_STATE, true); return PARTLS_ERR_STATE; }
    const size_t count = (size_t)c->ldg * c->ldg;
    const bool with_rccl = mc->use_rccl;                     // (cannot change here: abort_rccl runs only after rendezvous 2)
    partls_status mine = PARTLS_OK;
    if (fault_at(mc, r, 2)) { set_error("injected fault (stage 2) on rank %d", r); mine = PARTLS_ERR_HIP; }
    if (with_rccl) {
        Rccl &Rc = rccl();
        // enqueued whatever `mine` says: the other ranks are about to enqueue theirs
        const ncclResult_t e = Rc.AllReduce(c->G.p, c->G.p, count, ncclDouble, ncclSum, mc->comms[(size_t)r], c->stream);
        if (e != ncclSuccess) { set_error("ncclAllReduce(sum of the Gram products) failed on rank %d: %s", r, Rc.GetErrorString(e)); mine = PARTLS_ERR_HIP; mc->xst[(size_t)r] = 2; }
    } else if (mine == PARTLS_OK) {
        std::vector<double> &img = mc->gram_img[(size_t)r];
        if (img.size() < count || mc->gram_sum[(size_t)r].size() < count) { set_error("internal: Gram image larger than planned"); mine = PARTLS_ERR_STATE; }
        else if (hipMemcpyAsync(img.data(), c->G.p, count * sizeof(double), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                 hipStreamSynchronize(c->stream) != hipSuccess) { set_error("rank %d: copy of the Gram block to the host failed", r); mine = PARTLS_ERR_HIP; }
    }
    if (mine != PARTLS_OK) { fail(mc, r, mine); if (mc->xst[(size_t)r] == 0) mc->xst[(size_t)r] = 1; }
    if (!mc->bar.wait()) return lost(mc, r);
    bool ok = true, enqueue_failed = false;
    for (int q = 0; q < R; ++q) { ok = ok && mc->xst[(size_t)q] == 0; enqueue_failed = enqueue_failed || mc->xst[(size_t)q] == 2; }
    if (!ok) {
        if (enqueue_failed) abort_rccl(mc);                  // releases the ranks whose collective is waiting for the failed one
        if (mine != PARTLS_OK) return mine;
        set_error("partls_fit_*_multi: another rank failed while the Gram products were combined");
        fail(mc, r, PARTLS_ERR_STATE, true);
        return PARTLS_ERR_STATE;
    }
    if (with_rccl) return PARTLS_OK;
    std::vector<double> &sum = mc->gram_sum[(size_t)r];
    std::memcpy(sum.data(), mc->gram_img[0].data(), count * sizeof(double));
    for (int q = 1; q < R; ++q) {
        const double *src = mc->gram_img[(size_t)q].data();
        for (size_t i = 0; i < count; ++i) sum[i] += src[i];
    }
    if (!mc->bar.wait()) return lost(mc, r);                 // every rank has read the images before anybody's next fit rewrites them
    PARTLS_HIP_CHECK(hipMemcpyAsync(c->G.p, sum.data(), count * sizeof(double), hipMemcpyHostToDevice, c->stream));
    PARTLS_HIP_CHECK(hipStreamSynchronize(c->stream));
    return PARTLS_OK;
}

// upload of this rank's row block, Gram products, their sum, tableau: phase 1 of both fits
partls_status prepare_rank(partls_multi *mc, int r, const FitArgs &a, uint32_t flags)
{
    partls_ctx *c = mc->ctx[(size_t)r];
    const int R = mc->ndev;
    // rank r uploads and owns rows [r0, r1) of X and y (column-major: a row block is a strided view, uploaded with one 2-D copy): 1 / R of
    // the PCIe traffic of a replicated upload; every later pass over the data runs on all blocks (ctx.h: peers)
    const int64_t r0 = mc->shard_rows ? (int64_t)(((__int128)r * a.N) / R) : 0, r1 = mc->shard_rows ? (int64_t)(((__int128)(r + 1) * a.N) / R) : a.N;
    c->gram_hook = nullptr;
    // (the hook runs inside partls_opt_prepare, whose C-ABI guard would turn an exception into a status and let this rank walk on out of
    // step with the others' rendezvous: anything thrown in there poisons the rendezvous instead)
    if (mc->shard_rows) c->gram_hook = [mc, r](partls_ctx *cc) -> partls_status {
        try { return gram_rendezvous(mc, r, cc); }
        catch (...) { set_error("rank %d: an exception in the Gram exchange", r); fail(mc, r, PARTLS_ERR_BAD_ARG); mc->bar.poison(); return PARTLS_ERR_BAD_ARG; }
    };
    partls_status st;
    const int flt = fault_at(mc, r, 1);
    if (flt == 2) { c->gram_hook = nullptr; throw Vanish(); }
    if (flt == 1) { set_error("injected fault (stage 1) on rank %d", r); st = PARTLS_ERR_HIP; }
    else {
        try { st = partls_opt_prepare(c, a.X ? a.X + r0 : nullptr, r1 - r0, a.M, a.ldX, a.y ? a.y + r0 : nullptr, 0, a.P, a.K, a.ldP, a.eta, flags); }
        catch (...) { c->gram_hook = nullptr; throw; }
    }
    c->gram_hook = nullptr;
    note(mc, r, st);                                         // (the hook records its own failures, primary or secondary)
    if (mc->shard_rows && mc->at_gram[(size_t)r] == 0) {     // failed before the rendezvous: join it, so that the others can leave it
        mc->at_gram[(size_t)r] = 2;
        if (!mc->bar.wait()) return lost(mc, r);
    }
    return st;
}

// lexicographic minimum over the ranks through RCCL: every rank ends with the same (objective, pattern).  Both collectives are
// enqueued whatever happens in between (the other ranks enqueue theirs); errors are reported afterwards.
partls_status reduce_rccl(partls_multi *mc, int r, double obj, int64_t pat, double key, double *gobj, int64_t *gpat, bool *enqueue_failed)
{
    Rccl &R = rccl();
    partls_ctx *c = mc->ctx[(size_t)r];
    double *d = mc->red[(size_t)r].as<double>();
    const double h[3] = {pat >= 0 ? obj : INFINITY, key, -key};
    double g[3] = {INFINITY, 0.0, 0.0};
    partls_status st = PARTLS_OK;
    auto hipok = [&](hipError_t e, const char *what) { if (e != hipSuccess && st == PARTLS_OK) { set_error("%s failed on rank %d: %s", what, r, hipGetErrorString(e)); st = PARTLS_ERR_HIP; } };
    hipok(hipMemcpyAsync(d, h, sizeof(h), hipMemcpyHostToDevice, c->stream), "hipMemcpyAsync");
    ncclResult_t e = R.AllReduce(d, d, 3, ncclDouble, ncclMin, mc->comms[(size_t)r], c->stream);
    if (e != ncclSuccess) { if (st == PARTLS_OK) { set_error("ncclAllReduce(min objective) failed on rank %d: %s", r, R.GetErrorString(e)); st = PARTLS_ERR_HIP; } *enqueue_failed = true; return st; }
    hipok(hipMemcpyAsync(g, d, sizeof(g), hipMemcpyDeviceToHost, c->stream), "hipMemcpyAsync");
    hipok(hipStreamSynchronize(c->stream), "hipStreamSynchronize");
    int64_t idx = (st == PARTLS_OK && pat >= 0 && obj == g[0]) ? pat : NO_CANDIDATE, gidx = NO_CANDIDATE;
    int64_t *di = reinterpret_cast<int64_t *>(d + 4);
    hipok(hipMemcpyAsync(di, &idx, sizeof(idx), hipMemcpyHostToDevice, c->stream), "hipMemcpyAsync");
    e = R.AllReduce(di, di, 1, ncclInt64, ncclMin, mc->comms[(size_t)r], c->stream);
    if (e != ncclSuccess) { if (st == PARTLS_OK) { set_error("ncclAllReduce(min index) failed on rank %d: %s", r, R.GetErrorString(e)); st = PARTLS_ERR_HIP; } *enqueue_failed = true; return st; }
    hipok(hipMemcpyAsync(&gidx, di, sizeof(gidx), hipMemcpyDeviceToHost, c->stream), "hipMemcpyAsync");
    hipok(hipStreamSynchronize(c->stream), "hipStreamSynchronize");
    if (st != PARTLS_OK) return st;
    if (g[1] != -g[2]) {
        set_error("partls_fit_opt_multi: the ranks visit the patterns in different orders (bit-order keys differ), their Gray-index "
                  "ranges do not partition the pattern space");
        return PARTLS_ERR_STATE;
    }
    *gobj = g[0];
    *gpat = gidx == NO_CANDIDATE ? -1 : gidx;
    return PARTLS_OK;
}

// the same reduction from the host slots every rank filled before the agreement rendezvous (device list with duplicates, or a handle
// whose communicators were aborted): no further rendezvous needed — the slots are rewritten by the next fit only
partls_status reduce_host(partls_multi *mc, double *gobj, int64_t *gpat)
{
    double bo = INFINITY;
    int64_t bp = NO_CANDIDATE;
    bool same = true;
    for (int q = 0; q < mc->ndev; ++q) {
        same = same && mc->key[(size_t)q] == mc->key[0];
        if (mc->pat[(size_t)q] < 0) continue;
        if (mc->obj[(size_t)q] < bo || (mc->obj[(size_t)q] == bo && mc->pat[(size_t)q] < bp)) { bo = mc->obj[(size_t)q]; bp = mc->pat[(size_t)q]; }
    }
    if (!same) { set_error("partls_fit_opt_multi: the ranks visit the patterns in different orders (bit-order keys differ)"); return PARTLS_ERR_STATE; }
    *gobj = bo;
    *gpat = bp == NO_CANDIDATE ? -1 : bp;
    return PARTLS_OK;
}

void rank_opt(partls_multi *mc, int r, const FitArgs &a)
{
    partls_ctx *c = mc->ctx[(size_t)r];
    const int R = mc->ndev;
    // ---- phase 1: upload, Gram products and their sum, tableau ------------------------------------------------------------------
    partls_status st = prepare_rank(mc, r, a, a.flags);
    if (mc->bar.is_broken()) return;
    // ---- phase 2: visiting order and the sweep of this rank's shard ---------------------------------------------------------------
    double bobj = INFINITY, key = 0.0;
    int64_t bpat = -1, unconv = 0;
    if (st == PARTLS_OK) {
        const int flt = fault_at(mc, r, 3);
        if (flt == 2) throw Vanish();
        if (flt == 1) { set_error("injected fault (stage 3) on rank %d", r); st = PARTLS_ERR_HIP; }
    }
    if (st == PARTLS_OK) {
        const int64_t npat = partls_opt_num_patterns(c);
        int64_t gbit[40];
        st = partls_opt_bit_order(c, gbit, nullptr);
        if (st == PARTLS_OK) {
            key = order_key(gbit, c->kbits);
            const int64_t g0 = (int64_t)(((__int128)r * npat) / R), g1 = (int64_t)(((__int128)(r + 1) * npat) / R);
            double *ao = a.all_opt ? mc->all_opt[(size_t)r].data() : nullptr;      // this rank's image of all_opt (NaN outside its shard), merged by the caller
            st = partls_opt_sweep(c, g0, g1, &bobj, &bpat, ao, &unconv);
        }
    }
    note(mc, r, st);
    mc->unconv[(size_t)r] = unconv;
    mc->obj[(size_t)r] = bpat >= 0 ? bobj : INFINITY;
    mc->pat[(size_t)r] = bpat;
    mc->key[(size_t)r] = key;
    Cand &cd = mc->cand[(size_t)r];
    cd.n = 0;
    if (st == PARTLS_OK) {
        int64_t n = 0;
        if (partls_opt_candidates(c, 4, cd.obj, cd.pat, &n) == PARTLS_OK) cd.n = (int)n;
    }
    for (int w = 0; w < PARTLS_T_COUNT; ++w) mc->t_ms[(size_t)r][(size_t)w] = c->ms[w];
    // agree on the outcome so far BEFORE any collective: a rank that failed must not leave the others waiting inside RCCL
    if (!mc->bar.wait()) { lost(mc, r); return; }
    if (any_failed(mc)) return;
    // ---- phase 3: the global lexicographic minimum ----------------------------------------------------------------------------------
    double gobj = INFINITY;
    int64_t gpat = -1;
    const bool with_rccl = mc->use_rccl;
    bool enqueue_failed = false;
    const int flt4 = fault_at(mc, r, 4);
    if (flt4 == 2) throw Vanish();
    st = with_rccl ? reduce_rccl(mc, r, bobj, bpat, key, &gobj, &gpat, &enqueue_failed) : reduce_host(mc, &gobj, &gpat);
    if (flt4 == 1 && st == PARTLS_OK) { set_error("injected fault (stage 4) on rank %d", r); st = PARTLS_ERR_HIP; }
    if (enqueue_failed) abort_rccl(mc);                      // the ranks inside the collective are released; all of them fail below
    note(mc, r, st);
    if (with_rccl) {                                         // did the collectives work everywhere?
        if (!mc->bar.wait()) { lost(mc, r); return; }
        if (any_failed(mc)) return;
    } else if (st != PARTLS_OK) return;                      // (host mode: every rank computed the same thing from the same slots)
    if (r != 0) return;
    // ---- phase 4: the winner, on the first device ------------------------------------------------------------------------------------
    mc->win_obj = gobj;
    mc->win_pat = gpat;
    if (gpat < 0) { set_error("sweep produced no candidate"); fail(mc, 0, PARTLS_ERR_NOT_CONVERGED); return; }
    // near ties of EVERY shard: rank 0 re-ranks the set a single context would have had (Opt.jl:90,96)
    double co[4 * 64]; int64_t cp[4 * 64];
    int64_t nc = 0;
    for (int q = 0; q < R; ++q) for (int i = 0; i < mc->cand[(size_t)q].n; ++i) { co[nc] = mc->cand[(size_t)q].obj[i]; cp[nc] = mc->cand[(size_t)q].pat[i]; ++nc; }
    double mo = INFINITY; int64_t mp = -1;
    st = partls_opt_merge_candidates(c, nc, co, cp, &mo, &mp);
    if (st == PARTLS_OK && mp != gpat) { set_error("internal: the merged candidate lists name pattern %lld, the reduction %lld", (long long)mp, (long long)gpat); st = PARTLS_ERR_STATE; }
    if (st != PARTLS_OK) { fail(mc, 0, st); return; }
    // the winner is re-solved on the first device; its passes over the data (refinement, objective, KKT check) cover every rank's
    // row block: rank 0's thread drives the other devices' streams as well (every rank has finished its sweep: rendezvous above)
    if (mc->shard_rows) c->peers.assign(mc->ctx.begin() + 1, mc->ctx.end());
    if (fault_at(mc, 0, 5) == 1) { set_error("injected fault (stage 5) on rank 0"); fail(mc, 0, PARTLS_ERR_HIP); return; }
    st = partls_opt_finish(c, gpat, a.alpha, a.beta, a.t, a.opt, a.best_index);
    mc->t_ms[0][PARTLS_T_FINISH] = c->ms[PARTLS_T_FINISH];
    if (st != PARTLS_OK) fail(mc, 0, st);
}

// fit(BnB) over the ranks: every rank thread runs the same frontier; per round it bounds its share on its own GPU (warm-started from
// the parent's snapshot, which it holds), one all-gather of (bound, branch, slot) shares the results — and with them the incumbent —
// and every rank prunes, branches and counts snapshot references identically (frontier.h).  BnB.jl:94-132.
void rank_bnb(partls_multi *mc, int r, const FitArgs &a)
{
    partls_ctx *c = mc->ctx[(size_t)r];
    const int R = mc->ndev;
    partls_status st = prepare_rank(mc, r, a, PARTLS_OPT_FAITHFUL_INTERCEPT);
    if (mc->bar.is_broken()) return;
    const int Kp = (int)a.K + 1;
    const int64_t batch = std::max(1, c->knobs.bnb_batch);
    partls_frontier f;
    std::vector<uint64_t> bp, bf;
    std::vector<int32_t> src, dst, br, per_rank, gbr, gdst;
    std::vector<double> lb, glb;
    if (st == PARTLS_OK) {
        const int flt = fault_at(mc, r, 3);
        if (flt == 2) throw Vanish();
        if (flt == 1) { set_error("injected fault (stage 3) on rank %d", r); st = PARTLS_ERR_HIP; }
    }
    if (st == PARTLS_OK) st = partls_bnb_snap_begin(c);
    if (st == PARTLS_OK) {
        try {
            f.rank = r; f.world = R; f.batch = batch;
            f.refs.resize((size_t)R);
            f.best_free = ((uint64_t)1 << Kp) - 1;
            f.heap.push_node({0.0, 0ULL, ((uint64_t)1 << Kp) - 1, f.seq++, -1, -1});     // root: everything free (Σ = [], BnB.jl:33)
            bp.resize((size_t)batch); bf.resize((size_t)batch); src.resize((size_t)batch); dst.resize((size_t)batch); br.resize((size_t)batch);
            lb.resize((size_t)batch); per_rank.resize((size_t)R);
            glb.resize((size_t)batch * R); gbr.resize((size_t)batch * R); gdst.resize((size_t)batch * R);
        } catch (const std::bad_alloc &) { set_error("out of host memory"); st = PARTLS_ERR_BAD_ARG; }
    }
    note(mc, r, st);
    for (int w = 0; w < PARTLS_T_COUNT; ++w) mc->t_ms[(size_t)r][(size_t)w] = c->ms[w];
    if (!mc->bar.wait()) { lost(mc, r); return; }             // every rank is ready to search (or nobody searches)
    if (any_failed(mc)) return;
    // ---- rounds ---------------------------------------------------------------------------------------------------------------------
    const size_t stride = mc->xstride;                          // doubles per rank and parity: [status, total, count, - | lb | branch | slot]
    for (unsigned round = 0;; ++round) {
        int64_t mine = 0;
        const int64_t total = f.next(&mine, bp.data(), bf.data(), src.data(), per_rank.data());
        st = PARTLS_OK;
        const int flt = fault_at(mc, r, 4);
        if (flt == 2 && round == 1) throw Vanish();
        if (flt == 1 && round == 1) { set_error("injected fault (stage 4) on rank %d", r); st = PARTLS_ERR_HIP; }
        if (st == PARTLS_OK && mine > 0) st = partls_bnb_bound_snap(c, mine, bp.data(), bf.data(), src.data(), dst.data(), lb.data(), br.data());
        note(mc, r, st);
        // publish: status and results in this round's parity slots (host mode: that IS the exchange)
        double *slot = mc->xbuf[round & 1].data() + (size_t)r * stride;
        slot[0] = st == PARTLS_OK ? 0.0 : 1.0; slot[1] = (double)total; slot[2] = (double)mine;
        for (int64_t i = 0; i < mine; ++i) { slot[4 + i] = lb[(size_t)i]; slot[4 + batch + i] = (double)br[(size_t)i]; slot[4 + 2 * batch + i] = (double)dst[(size_t)i]; }
        f.prefetch();                                           // the next round's pops, while the slower ranks finish this one
        if (!mc->bar.wait()) { lost(mc, r); return; }
        bool ok = true, same = true;
        for (int q = 0; q < R; ++q) {
            const double *sq = mc->xbuf[round & 1].data() + (size_t)q * stride;
            ok = ok && sq[0] == 0.0;
            same = same && sq[1] == (double)total && sq[2] == (double)per_rank[(size_t)q];
        }
        if (!ok) { if (st == PARTLS_OK) { set_error("partls_fit_bnb_multi: another rank failed in round %u of the search", round); fail(mc, r, PARTLS_ERR_STATE, true); } return; }
        if (!same) { set_error("partls_fit_bnb_multi: the ranks' frontiers disagree in round %u (internal error)", round); fail(mc, r, PARTLS_ERR_STATE); return; }
        if (total == 0) break;                                  // the same decision on every rank
        const double *xsrc = mc->xbuf[round & 1].data();
        if (mc->use_rccl) {
            // the exchange proper: ncclAllGather over xGMI of [lb | branch | slot] (3 x per doubles per rank, per = the largest share)
            Rccl &Rc = rccl();
            int64_t per = 0;
            for (int q = 0; q < R; ++q) per = std::max<int64_t>(per, per_rank[(size_t)q]);
            const size_t cnt = (size_t)3 * per;
            double *hs = mc->xpin[(size_t)r].data(), *hr = hs + (size_t)3 * batch;
            for (int64_t i = 0; i < mine; ++i) { hs[i] = lb[(size_t)i]; hs[per + i] = (double)br[(size_t)i]; hs[2 * per + i] = (double)dst[(size_t)i]; }
            double *ds = mc->xdev[(size_t)r].as<double>(), *dr = ds + (size_t)3 * batch;
            partls_status xs = PARTLS_OK;
            bool enqueue_failed = false;
            if (hipMemcpyAsync(ds, hs, cnt * sizeof(double), hipMemcpyHostToDevice, c->stream) != hipSuccess) xs = PARTLS_ERR_HIP;
            const ncclResult_t e = Rc.AllGather(ds, dr, cnt, ncclDouble, mc->comms[(size_t)r], c->stream);
            if (e != ncclSuccess) { set_error("ncclAllGather(bounds of round %u) failed on rank %d: %s", round, r, Rc.GetErrorString(e)); xs = PARTLS_ERR_HIP; enqueue_failed = true; }
            else if (hipMemcpyAsync(hr, dr, cnt * R * sizeof(double), hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
                if (xs == PARTLS_OK) set_error("rank %d: copy of the gathered bounds failed", r);
                xs = PARTLS_ERR_HIP;
            }
            if (enqueue_failed) abort_rccl(mc);
            note(mc, r, xs);
            if (!mc->bar.wait()) { lost(mc, r); return; }
            if (any_failed(mc)) return;
            int64_t j = 0;
            for (int q = 0; q < R; ++q) {
                const double *g = hr + (size_t)q * cnt;
                for (int64_t i = 0; i < per_rank[(size_t)q]; ++i, ++j) { glb[(size_t)j] = g[i]; gbr[(size_t)j] = (int32_t)g[per + i]; gdst[(size_t)j] = (int32_t)g[2 * per + i]; }
            }
        } else {
            int64_t j = 0;
            for (int q = 0; q < R; ++q) {
                const double *g = xsrc + (size_t)q * stride + 4;
                for (int64_t i = 0; i < per_rank[(size_t)q]; ++i, ++j) { glb[(size_t)j] = g[i]; gbr[(size_t)j] = (int32_t)g[batch + i]; gdst[(size_t)j] = (int32_t)g[2 * batch + i]; }
            }
        }
        f.ingest(glb.data(), gbr.data(), gdst.data());
        if (!f.dead.empty()) {
            st = partls_bnb_snap_release(c, (int64_t)f.dead.size(), f.dead.data());
            f.dead.clear();
            note(mc, r, st);               // reported at the next round's rendezvous
        }
    }
    if (r != 0) return;
    mc->bnb_nodes = f.bounded;
    if (!(f.mu < INFINITY)) { set_error("partls_fit_bnb_multi: no feasible leaf found"); fail(mc, 0, PARTLS_ERR_NOT_CONVERGED); return; }
    if (a.nopen) *a.nopen = f.bounded;
    if (mc->shard_rows) c->peers.assign(mc->ctx.begin() + 1, mc->ctx.end());
    if (fault_at(mc, 0, 5) == 1) { set_error("injected fault (stage 5) on rank 0"); fail(mc, 0, PARTLS_ERR_HIP); return; }
    st = partls_bnb_leaf(c, f.best_pat, f.best_free, a.alpha, a.beta, a.t, a.opt);
    if (st != PARTLS_OK) fail(mc, 0, st);
}

// entry of a rank thread: nothing may leave it — an exception poisons the rendezvous, so that every other rank fails its next one
void rank_main(partls_multi *mc, int r, const FitArgs &a)
{
    try {
        if (a.kind == 0) rank_opt(mc, r, a); else rank_bnb(mc, r, a);
        return;
    }
    catch (const Vanish &) { mc->ctx[(size_t)r]->gram_hook = nullptr; return; }                 // injected: the rank is simply gone (the others time out)
    catch (const std::bad_alloc &) { set_error("rank %d: out of host memory", r); }
    catch (...) { set_error("rank %d: internal error: an exception reached the rank thread's entry", r); }
    mc->ctx[(size_t)r]->gram_hook = nullptr;
    fail(mc, r, PARTLS_ERR_BAD_ARG);
    mc->bar.poison();
}

partls_status run_fit(partls_multi *mc, const FitArgs &a)
{
    const int R = mc->ndev;
    for (int r = 0; r < R; ++r) {
        mc->st[(size_t)r] = PARTLS_OK; mc->msg[(size_t)r].clear(); mc->secondary[(size_t)r] = 0; mc->unconv[(size_t)r] = 0;
        mc->at_gram[(size_t)r] = 0; mc->xst[(size_t)r] = 0; mc->cand[(size_t)r].n = 0;
    }
    mc->bar.reset();
    // rows sharded over the ranks whenever every rank gets a reasonable block (tiny problems: every rank takes all rows — nothing to save)
    // (one rank: the block is all of X and the sum has one term — the same code path, which is what a one-GPU box can test of it)
    mc->shard_rows = a.N >= (int64_t)64 * R && !mc->replicate;
    // everything the rank threads exchange is sized HERE, before they start: nothing is allocated between two rendezvous
    if (mc->shard_rows && !mc->use_rccl && a.M >= 1 && a.M <= 1022) {
        int chunks = 0, ldg = 0;
        (void)gram_slab_doubles(a.N / R + 1, a.M, mc->ctx[0]->knobs.gram_S, mc->ctx[0]->knobs.gram_cr, &chunks, &ldg);
        for (int r = 0; r < R; ++r) { mc->gram_img[(size_t)r].resize((size_t)ldg * ldg); mc->gram_sum[(size_t)r].resize((size_t)ldg * ldg); }
    }
    if (a.kind == 0 && a.all_opt) for (int r = 0; r < R; ++r) mc->all_opt[(size_t)r].resize((size_t)1 << (a.K + 1));
    if (a.kind == 1) {
        const size_t batch = (size_t)std::max(1, mc->ctx[0]->knobs.bnb_batch);
        mc->xstride = 4 + 3 * batch;
        for (int p = 0; p < 2; ++p) mc->xbuf[p].assign(mc->xstride * (size_t)R, 0.0);
        if (mc->use_rccl)
            for (int r = 0; r < R; ++r) {
                PARTLS_HIP_CHECK(hipSetDevice(mc->devices[(size_t)r]));
                PARTLS_HIP_CHECK(mc->xdev[(size_t)r].ensure((size_t)3 * batch * (R + 1) * sizeof(double)));
                PARTLS_HIP_CHECK(mc->xpin[(size_t)r].resize((size_t)3 * batch * (R + 1)));
            }
    }
    std::vector<std::thread> th;
    th.reserve((size_t)R);
    bool spawn_failed = false;
    for (int r = 1; r < R && !spawn_failed; ++r) {
        try { th.emplace_back(rank_main, mc, r, std::cref(a)); }
        catch (...) { spawn_failed = true; }                    // std::system_error: out of threads
    }
    if (spawn_failed) mc->bar.poison();                         // the ranks that did start fail their first rendezvous and return
    else rank_main(mc, 0, a);                                   // rank 0 on the caller's thread
    for (std::thread &w : th) w.join();
    if (spawn_failed) { set_error("partls_fit_*_multi: a rank thread could not be started"); return PARTLS_ERR_BAD_ARG; }
    // the first PRIMARY failure is the one to report ("another rank failed" statuses only point at it)
    int bad = -1;
    for (int r = 0; r < R && bad < 0; ++r) if (mc->st[(size_t)r] != PARTLS_OK && !mc->secondary[(size_t)r]) bad = r;
    for (int r = 0; r < R && bad < 0; ++r) if (mc->st[(size_t)r] != PARTLS_OK) bad = r;
    if (bad < 0 && mc->bar.is_broken()) { set_error("partls_fit_*_multi: the rendezvous of the rank threads broke down"); return PARTLS_ERR_STATE; }
    if (bad >= 0) {
        set_error("rank %d (device %d): %s", bad, mc->devices[(size_t)bad], mc->msg[(size_t)bad].c_str());
        return mc->st[(size_t)bad];
    }
    return PARTLS_OK;
}

}  // namespace

extern "C" {

partls_status partls_multi_create(const int *devices, int ndev, partls_multi **out)
try {
    if (!out) { set_error("partls_multi_create: out is NULL"); return PARTLS_ERR_BAD_ARG; }
    *out = nullptr;
    const int visible = partls_device_count();
    if (visible <= 0) { set_error("partls_multi_create: no HIP device; this library has no CPU fallback"); return PARTLS_ERR_NO_DEVICE; }
    if (ndev < 0 || ndev > 64) { set_error("partls_multi_create: ndev = %d", ndev); return PARTLS_ERR_BAD_ARG; }
    if (ndev == 0) { ndev = visible; devices = nullptr; }
    partls_multi *mc = new (std::nothrow) partls_multi();
    if (!mc) { set_error("out of host memory"); return PARTLS_ERR_BAD_ARG; }
    mc->ndev = ndev;
    bool distinct = true;
    for (int r = 0; r < ndev; ++r) {
        const int d = devices ? devices[r] : r;
        if (d < 0 || d >= visible) { set_error("partls_multi_create: device %d of %d visible", d, visible); delete mc; return PARTLS_ERR_NO_DEVICE; }
        for (int q : mc->devices) distinct = distinct && q != d;
        mc->devices.push_back(d);
    }
    mc->st.assign((size_t)ndev, PARTLS_OK); mc->msg.assign((size_t)ndev, std::string());
    for (std::string &m : mc->msg) m.reserve(600);               // set_error's buffer is 512 bytes: fail() never allocates
    mc->secondary.assign((size_t)ndev, 0);
    mc->obj.assign((size_t)ndev, 0.0); mc->key.assign((size_t)ndev, 0.0);
    mc->pat.assign((size_t)ndev, -1); mc->unconv.assign((size_t)ndev, 0);
    mc->cand.assign((size_t)ndev, Cand());
    mc->all_opt.resize((size_t)ndev);
    mc->replicate = getenv("PARTLS_MULTI_REPLICATE") != nullptr;
    if (const char *e = getenv("PARTLS_MULTI_TIMEOUT_S")) { const double v = atof(e); if (v > 0.0) mc->bar.timeout_s = v; }
    if (const char *e = getenv("PARTLS_MULTI_FAULT")) {
        int fr = -1, fs = 0; char kind[16] = "";
        const int got = sscanf(e, "%d:%d:%15s", &fr, &fs, kind);
        if (got >= 2) { mc->fault_rank = fr; mc->fault_stage = fs; mc->fault_vanish = got == 3 && !strcmp(kind, "vanish"); }
    }
    mc->at_gram.assign((size_t)ndev, 0); mc->xst.assign((size_t)ndev, 0);
    mc->gram_img.resize((size_t)ndev); mc->gram_sum.resize((size_t)ndev);
    mc->t_ms.assign((size_t)ndev, std::vector<double>((size_t)PARTLS_T_COUNT, 0.0));
    mc->bar.n = ndev;
    mc->red.resize((size_t)ndev);
    mc->xdev.resize((size_t)ndev);
    mc->xpin.resize((size_t)ndev);
    for (int r = 0; r < ndev; ++r) {
        partls_ctx *c = nullptr;
        partls_status st = partls_create(mc->devices[(size_t)r], &c);
        if (st != PARTLS_OK) { partls_multi_destroy(mc); return st; }
        mc->ctx.push_back(c);
        if (hipSetDevice(mc->devices[(size_t)r]) != hipSuccess || mc->red[(size_t)r].ensure(8 * sizeof(double)) != hipSuccess) {
            set_error("partls_multi_create: device buffer allocation failed on device %d", mc->devices[(size_t)r]);
            partls_multi_destroy(mc);
            return PARTLS_ERR_HIP;
        }
    }
    if (distinct) {
        Rccl &R = rccl();
        if (!R.error.empty()) { set_error("partls_multi_create: %s", R.error.c_str()); partls_multi_destroy(mc); return PARTLS_ERR_UNSUPPORTED; }
        mc->comms.assign((size_t)ndev, nullptr);
        const ncclResult_t e = R.CommInitAll(mc->comms.data(), ndev, mc->devices.data());
        if (e != ncclSuccess) {
            mc->comms.clear();
            set_error("ncclCommInitAll over %d devices failed: %s", ndev, R.GetErrorString(e));
            partls_multi_destroy(mc);
            return PARTLS_ERR_HIP;
        }
        mc->use_rccl = true;
    }
    *out = mc;
    return PARTLS_OK;
}
catch (const std::bad_alloc &) { partls::set_error("out of host memory"); return PARTLS_ERR_BAD_ARG; }
catch (...) { partls::set_error("internal error: an exception reached the C ABI"); return PARTLS_ERR_BAD_ARG; }

void partls_multi_destroy(partls_multi *mc)
{
    if (!mc) return;
    int ndev = 0;
    const bool alive = hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0;
    if (alive) {
        for (size_t r = 0; r < mc->ctx.size(); ++r)
            if (mc->ctx[r] && mc->ctx[r]->stream && hipSetDevice(mc->devices[r]) == hipSuccess) (void)hipStreamSynchronize(mc->ctx[r]->stream);
        for (ncclComm_t cm : mc->comms) if (cm) (void)rccl().CommDestroy(cm);
        for (size_t r = 0; r < mc->red.size(); ++r)
            if (hipSetDevice(mc->devices[r]) == hipSuccess) { mc->red[r].release(); if (r < mc->xdev.size()) mc->xdev[r].release(); if (r < mc->xpin.size()) mc->xpin[r].release(); }
    }
    for (partls_ctx *c : mc->ctx) partls_destroy(c);
    delete mc;
}

int partls_multi_size(const partls_multi *mc) { return mc ? mc->ndev : 0; }
int partls_multi_uses_rccl(const partls_multi *mc) { return (mc && mc->use_rccl) ? 1 : 0; }
partls_ctx *partls_multi_context(partls_multi *mc, int rank) { return (mc && rank >= 0 && rank < mc->ndev) ? mc->ctx[(size_t)rank] : nullptr; }

partls_status partls_multi_get_timing(const partls_multi *mc, int rank, int which, double *ms)
try {
    if (!mc || !ms || rank < 0 || rank >= mc->ndev || which < 0 || which >= PARTLS_T_COUNT) { set_error("partls_multi_get_timing: bad argument"); return PARTLS_ERR_BAD_ARG; }
    *ms = mc->t_ms[(size_t)rank][(size_t)which];
    return PARTLS_OK;
}
catch (const std::bad_alloc &) { partls::set_error("out of host memory"); return PARTLS_ERR_BAD_ARG; }
catch (...) { partls::set_error("internal error: an exception reached the C ABI"); return PARTLS_ERR_BAD_ARG; }

partls_status partls_fit_opt_multi(partls_multi *mc, const double *X, int64_t N, int64_t M, int64_t ldX, const double *y,
                                   const int64_t *P, int64_t K, int64_t ldP, double eta, uint32_t flags,
                                   double *alpha, double *beta, double *t, double *opt, int64_t *best_index, double *all_opt)
try {
    if (!mc) { set_error("partls_fit_opt_multi: handle is NULL"); return PARTLS_ERR_BAD_ARG; }
    if (!alpha || !beta || !t || !opt) { set_error("partls_fit_opt_multi: NULL output"); return PARTLS_ERR_BAD_ARG; }
    if (all_opt && (K < 1 || K > 39)) { set_error("partls_fit_opt_multi: all_opt needs 1 <= K <= 39"); return PARTLS_ERR_UNSUPPORTED; }
    if (all_opt) flags |= PARTLS_OPT_FAITHFUL_INTERCEPT;
    const FitArgs a{0, X, N, M, ldX, y, P, K, ldP, eta, flags, alpha, beta, t, opt, best_index, all_opt, nullptr};
    const int R = mc->ndev;
    partls_status st = run_fit(mc, a);
    if (st != PARTLS_OK) {
        for (auto &v : mc->all_opt) std::vector<double>().swap(v);
        return st;
    }
    if (all_opt) {                                           // every pattern belongs to exactly one shard: the others hold NaN there
        const size_t np = (size_t)1 << (K + 1);
        std::memcpy(all_opt, mc->all_opt[0].data(), np * sizeof(double));
        for (int r = 1; r < R; ++r) {
            const double *src = mc->all_opt[(size_t)r].data();
            for (size_t i = 0; i < np; ++i) if (src[i] == src[i]) all_opt[i] = src[i];
        }
        for (auto &v : mc->all_opt) std::vector<double>().swap(v);
    }
    int64_t unconv = 0;
    for (int r = 0; r < R; ++r) unconv += mc->unconv[(size_t)r];
    if (unconv) { set_error("%lld subproblems hit the pivot cap", (long long)unconv); return PARTLS_ERR_NOT_CONVERGED; }
    return PARTLS_OK;
}
catch (const std::bad_alloc &) { partls::set_error("out of host memory"); return PARTLS_ERR_BAD_ARG; }
catch (...) { partls::set_error("internal error: an exception reached the C ABI"); return PARTLS_ERR_BAD_ARG; }

partls_status partls_fit_bnb_multi(partls_multi *mc, const double *X, int64_t N, int64_t M, int64_t ldX, const double *y,
                                   const int64_t *P, int64_t K, int64_t ldP, double eta,
                                   double *alpha, double *beta, double *t, double *opt, int64_t *nopen)
try {
    if (!mc) { set_error("partls_fit_bnb_multi: handle is NULL"); return PARTLS_ERR_BAD_ARG; }
    if (!alpha || !beta || !t || !opt) { set_error("partls_fit_bnb_multi: NULL output"); return PARTLS_ERR_BAD_ARG; }
    const FitArgs a{1, X, N, M, ldX, y, P, K, ldP, eta, PARTLS_OPT_FAITHFUL_INTERCEPT, alpha, beta, t, opt, nullptr, nullptr, nopen};
    return run_fit(mc, a);
}
catch (const std::bad_alloc &) { partls::set_error("out of host memory"); return PARTLS_ERR_BAD_ARG; }
catch (...) { partls::set_error("internal error: an exception reached the C ABI"); return PARTLS_ERR_BAD_ARG; }

}  // extern "C"
