// multi.hip — fit(Opt) over several GPUs of one node behind the C ABI (include/partls.h: partls_fit_opt_multi).
//
// The loop of the reference, Opt.jl:85-94, has no loop-carried state: the 2^K' sign patterns are sharded over the devices by
// Gray-index range, exactly as partitionedls.jl_amd/dist.py does across processes — here inside one process, one host thread
// and one partls_ctx per device, so that a Julia `fit(Opt, X, y, P)` uses the whole node without any glue on its side.
// The winner is the lexicographic minimum (objective, reference pattern index) — argmin's first-index rule, Opt.jl:96 — taken
// with two RCCL all-reduces (ncclMin on the objective, ncclMin on the index masked to the minimisers): 24 + 8 bytes over xGMI,
// latency-bound.  RCCL is loaded on first use (573 MB on disk: a single-GPU fit never pays for it).
#include "ctx.h"
#include <new>
#include <rccl/rccl.h>
#include <dlfcn.h>
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
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
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
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.handle, "ncclAllReduce"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.handle, "ncclGetErrorString"));
        if (!r.CommInitAll || !r.CommDestroy || !r.AllReduce || !r.GetErrorString) r.error = "librccl.so.1 lacks ncclCommInitAll / ncclAllReduce";
    });
    return r;
}

// a barrier all ranks pass together; reusable (generation counter)
struct HostBarrier {
    std::mutex m;
    std::condition_variable cv;
    int n = 1, waiting = 0;
    unsigned gen = 0;
    void wait()
    {
        std::unique_lock<std::mutex> lk(m);
        const unsigned g = gen;
        if (++waiting == n) { waiting = 0; ++gen; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g; });
    }
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

}  // namespace

struct partls_multi {
    int ndev = 0;
    std::vector<int> devices;
    std::vector<partls_ctx *> ctx;
    bool use_rccl = false;
    std::vector<ncclComm_t> comms;
    std::vector<DevBuf> red;                 // per rank: [objective, key, -key | index] on its device
    HostBarrier bar;
    // per-call exchange between the rank threads
    std::vector<partls_status> st;
    std::vector<std::string> msg;
    std::vector<double> obj, key;
    std::vector<int64_t> pat, unconv;
    std::vector<std::vector<double>> all_opt;
    double win_obj = 0.0;
    int64_t win_pat = -1;
    std::vector<std::vector<double>> t_ms;
    // row-sharded fits: did rank r get as far as the Gram rendezvous (a rank that failed earlier joins it from its error path), and
    // the host images of the partial Gram products of the host-reduced (rehearsal) mode
    std::vector<char> at_gram;
    std::vector<std::vector<double>> gram_img;
    bool shard_rows = false;
    bool replicate = false;                  // PARTLS_MULTI_REPLICATE (read at create): every rank uploads all of X (A/B tests)
};

namespace {

struct FitArgs {
    const double *X; int64_t N, M, ldX; const double *y; const int64_t *P; int64_t K, ldP; double eta; uint32_t flags;
    double *alpha, *beta, *t, *opt; int64_t *best_index; double *all_opt;
};

void fail(partls_multi *mc, int r, partls_status st)
{
    mc->st[(size_t)r] = st;
    mc->msg[(size_t)r] = partls_last_error();
}

// lexicographic minimum over the ranks through RCCL: every rank ends with the same (objective, pattern)
partls_status reduce_rccl(partls_multi *mc, int r, double obj, int64_t pat, double key, double *gobj, int64_t *gpat)
{
    Rccl &R = rccl();
    partls_ctx *c = mc->ctx[(size_t)r];
    double *d = mc->red[(size_t)r].as<double>();
    const double h[3] = {pat >= 0 ? obj : INFINITY, key, -key};
    double g[3];
    PARTLS_HIP_CHECK(hipMemcpyAsync(d, h, sizeof(h), hipMemcpyHostToDevice, c->stream));
    ncclResult_t e = R.AllReduce(d, d, 3, ncclDouble, ncclMin, mc->comms[(size_t)r], c->stream);
    if (e != ncclSuccess) { set_error("ncclAllReduce(min objective) failed on rank %d: %s", r, R.GetErrorString(e)); return PARTLS_ERR_HIP; }
    PARTLS_HIP_CHECK(hipMemcpyAsync(g, d, sizeof(g), hipMemcpyDeviceToHost, c->stream));
    PARTLS_HIP_CHECK(hipStreamSynchronize(c->stream));
    int64_t idx = (pat >= 0 && obj == g[0]) ? pat : NO_CANDIDATE, gidx = NO_CANDIDATE;
    int64_t *di = reinterpret_cast<int64_t *>(d + 4);
    PARTLS_HIP_CHECK(hipMemcpyAsync(di, &idx, sizeof(idx), hipMemcpyHostToDevice, c->stream));
    e = R.AllReduce(di, di, 1, ncclInt64, ncclMin, mc->comms[(size_t)r], c->stream);
    if (e != ncclSuccess) { set_error("ncclAllReduce(min index) failed on rank %d: %s", r, R.GetErrorString(e)); return PARTLS_ERR_HIP; }
    PARTLS_HIP_CHECK(hipMemcpyAsync(&gidx, di, sizeof(gidx), hipMemcpyDeviceToHost, c->stream));
    PARTLS_HIP_CHECK(hipStreamSynchronize(c->stream));
    // both collectives have completed on every rank before anybody can return an error: no rank is left waiting inside RCCL
    if (g[1] != -g[2]) {
        set_error("partls_fit_opt_multi: the ranks visit the patterns in different orders (bit-order keys differ), their Gray-index "
                  "ranges do not partition the pattern space");
        return PARTLS_ERR_STATE;
    }
    *gobj = g[0];
    *gpat = gidx == NO_CANDIDATE ? -1 : gidx;
    return PARTLS_OK;
}

// the same reduction through host memory (device list with duplicates: no communicator can be formed)
partls_status reduce_host(partls_multi *mc, int r, double obj, int64_t pat, double key, double *gobj, int64_t *gpat)
{
    mc->obj[(size_t)r] = pat >= 0 ? obj : INFINITY;
    mc->pat[(size_t)r] = pat;
    mc->key[(size_t)r] = key;
    mc->bar.wait();
    double bo = INFINITY;
    int64_t bp = NO_CANDIDATE;
    bool same = true;
    for (int q = 0; q < mc->ndev; ++q) {
        same = same && mc->key[(size_t)q] == mc->key[0];
        if (mc->pat[(size_t)q] < 0) continue;
        if (mc->obj[(size_t)q] < bo || (mc->obj[(size_t)q] == bo && mc->pat[(size_t)q] < bp)) { bo = mc->obj[(size_t)q]; bp = mc->pat[(size_t)q]; }
    }
    mc->bar.wait();                                        // nobody rewrites the slots while another rank still reads them
    if (!same) { set_error("partls_fit_opt_multi: the ranks visit the patterns in different orders (bit-order keys differ)"); return PARTLS_ERR_STATE; }
    *gobj = bo;
    *gpat = bp == NO_CANDIDATE ? -1 : bp;
    return PARTLS_OK;
}

// Rows of X sharded over the ranks: every rank has built the Gram products of ITS row block; their sum is the problem's.  Called by
// ctx_prepare between the Gram build and the tableau preparation (ctx.h: gram_hook), on the rank's thread and stream.  The ranks first
// agree on the host that ALL of them got here (a rank that failed earlier joins the rendezvous from its error path): nobody may be
// left waiting inside a collective.  Then  G <- sum_r G_r : ncclAllReduce(ncclSum) over xGMI on (M + 2)^2 doubles padded to ldg^2
// (0.8 MB at C3) — every rank receives the same bits, so all of them prepare the same tableau and derive the same visiting order —
// or, in the rehearsal mode without a communicator, the same sum in rank order through host memory.
partls_status gram_rendezvous(partls_multi *mc, int r, partls_ctx *c)
{
    const int R = mc->ndev;
    mc->at_gram[(size_t)r] = 1;
    mc->bar.wait();
    bool all_here = true;
    for (int q = 0; q < R; ++q) all_here = all_here && mc->at_gram[(size_t)q] == 1;
    if (!all_here) { set_error("partls_fit_opt_multi: another rank failed before the Gram products could be combined"); return PARTLS_ERR_STATE; }
    const size_t count = (size_t)c->ldg * c->ldg;
    if (mc->use_rccl) {
        Rccl &Rc = rccl();
        const ncclResult_t e = Rc.AllReduce(c->G.p, c->G.p, count, ncclDouble, ncclSum, mc->comms[(size_t)r], c->stream);
        if (e != ncclSuccess) { set_error("ncclAllReduce(sum of the Gram products) failed on rank %d: %s", r, Rc.GetErrorString(e)); return PARTLS_ERR_HIP; }
        return PARTLS_OK;
    }
    std::vector<double> &img = mc->gram_img[(size_t)r];
    img.resize(count);
    PARTLS_HIP_CHECK(hipMemcpyAsync(img.data(), c->G.p, count * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    PARTLS_HIP_CHECK(hipStreamSynchronize(c->stream));
    mc->bar.wait();
    std::vector<double> sum(mc->gram_img[0]);
    for (int q = 1; q < R; ++q) {
        const double *src = mc->gram_img[(size_t)q].data();
        for (size_t i = 0; i < count; ++i) sum[i] += src[i];
    }
    mc->bar.wait();                                        // every rank has read the images before anybody's next fit rewrites them
    PARTLS_HIP_CHECK(hipMemcpyAsync(c->G.p, sum.data(), count * sizeof(double), hipMemcpyHostToDevice, c->stream));
    PARTLS_HIP_CHECK(hipStreamSynchronize(c->stream));     // `sum` is pageable and goes out of scope
    return PARTLS_OK;
}

void rank_main(partls_multi *mc, int r, const FitArgs &a)
{
    partls_ctx *c = mc->ctx[(size_t)r];
    const int R = mc->ndev;
    // rank r uploads and owns rows [r0, r1) of X and y (column-major: a row block is a strided view, uploaded with one 2-D copy): 1 / R of
    // the PCIe traffic of a replicated upload; every later pass over the data runs on all blocks (ctx.h: peers)
    const int64_t r0 = mc->shard_rows ? (int64_t)(((__int128)r * a.N) / R) : 0, r1 = mc->shard_rows ? (int64_t)(((__int128)(r + 1) * a.N) / R) : a.N;
    mc->at_gram[(size_t)r] = 0;
    c->gram_hook = nullptr;
    if (mc->shard_rows) c->gram_hook = [mc, r](partls_ctx *cc) { return gram_rendezvous(mc, r, cc); };
    partls_status st = partls_opt_prepare(c, a.X ? a.X + r0 : nullptr, r1 - r0, a.M, a.ldX, a.y ? a.y + r0 : nullptr, 0, a.P, a.K, a.ldP, a.eta, a.flags);
    c->gram_hook = nullptr;
    if (mc->shard_rows && mc->at_gram[(size_t)r] == 0) {  // failed before the rendezvous: join it, so that the others can leave it
        mc->at_gram[(size_t)r] = 2;
        mc->bar.wait();
    }
    double bobj = INFINITY, key = 0.0;
    int64_t bpat = -1, unconv = 0;
    if (st == PARTLS_OK) {
        const int64_t npat = partls_opt_num_patterns(c);
        int64_t gbit[40];
        st = partls_opt_bit_order(c, gbit, nullptr);
        if (st == PARTLS_OK) {
            key = order_key(gbit, c->kbits);
            const int64_t g0 = (int64_t)(((__int128)r * npat) / R), g1 = (int64_t)(((__int128)(r + 1) * npat) / R);
            double *ao = nullptr;
            bool have_buf = true;
            if (a.all_opt) {                                   // this rank's image of all_opt (NaN outside its shard), merged by the caller
                try { mc->all_opt[(size_t)r].resize((size_t)((int64_t)1 << (a.K + 1))); ao = mc->all_opt[(size_t)r].data(); }
                catch (...) { have_buf = false; }               // no exception may leave a rank's thread, let alone the C ABI
            }
            if (have_buf) st = partls_opt_sweep(c, g0, g1, &bobj, &bpat, ao, &unconv);
            else { set_error("out of host memory for the per-rank image of all_opt (2^(K+1) doubles per rank)"); st = PARTLS_ERR_BAD_ARG; }
        }
    }
    if (st != PARTLS_OK) fail(mc, r, st);
    mc->unconv[(size_t)r] = unconv;
    for (int w = 0; w < PARTLS_T_COUNT; ++w) mc->t_ms[(size_t)r][(size_t)w] = c->ms[w];
    // agree on the outcome so far BEFORE any collective: a rank that failed must not leave the others waiting inside RCCL
    mc->bar.wait();
    bool all_ok = true;
    for (int q = 0; q < R; ++q) all_ok = all_ok && mc->st[(size_t)q] == PARTLS_OK;
    if (!all_ok) return;
    double gobj = INFINITY;
    int64_t gpat = -1;
    st = mc->use_rccl ? reduce_rccl(mc, r, bobj, bpat, key, &gobj, &gpat) : reduce_host(mc, r, bobj, bpat, key, &gobj, &gpat);
    if (st != PARTLS_OK) { fail(mc, r, st); return; }
    if (r != 0) return;
    mc->win_obj = gobj;
    mc->win_pat = gpat;
    if (gpat < 0) { set_error("sweep produced no candidate"); fail(mc, 0, PARTLS_ERR_NOT_CONVERGED); return; }
    // the winner is re-solved on the first device; its passes over the data (refinement, objective, KKT check) cover every rank's
    // row block: rank 0's thread drives the other devices' streams as well (every rank has finished its sweep: reduce above)
    if (mc->shard_rows) c->peers.assign(mc->ctx.begin() + 1, mc->ctx.end());
    st = partls_opt_finish(c, gpat, a.alpha, a.beta, a.t, a.opt, a.best_index);
    mc->t_ms[0][PARTLS_T_FINISH] = c->ms[PARTLS_T_FINISH];
    if (st != PARTLS_OK) fail(mc, 0, st);
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
    mc->obj.assign((size_t)ndev, 0.0); mc->key.assign((size_t)ndev, 0.0);
    mc->pat.assign((size_t)ndev, -1); mc->unconv.assign((size_t)ndev, 0);
    mc->all_opt.resize((size_t)ndev);
    mc->replicate = getenv("PARTLS_MULTI_REPLICATE") != nullptr;
    mc->at_gram.assign((size_t)ndev, 0);
    mc->gram_img.resize((size_t)ndev);
    mc->t_ms.assign((size_t)ndev, std::vector<double>((size_t)PARTLS_T_COUNT, 0.0));
    mc->bar.n = ndev;
    mc->red.resize((size_t)ndev);
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
            if (hipSetDevice(mc->devices[r]) == hipSuccess) mc->red[r].release();
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
    if (all_opt) flags |= PARTLS_OPT_FAITHFUL_INTERCEPT;
    const FitArgs a{X, N, M, ldX, y, P, K, ldP, eta, flags, alpha, beta, t, opt, best_index, all_opt};
    const int R = mc->ndev;
    for (int r = 0; r < R; ++r) { mc->st[(size_t)r] = PARTLS_OK; mc->msg[(size_t)r].clear(); mc->unconv[(size_t)r] = 0; }
    // rows sharded over the ranks whenever every rank gets a reasonable block (tiny problems: every rank takes all rows — nothing to save)
    // (one rank: the block is all of X and the sum has one term — the same code path, which is what a one-GPU box can test of it)
    mc->shard_rows = N >= (int64_t)64 * R && !mc->replicate;
    std::vector<std::thread> th;
    for (int r = 1; r < R; ++r) th.emplace_back(rank_main, mc, r, std::cref(a));
    rank_main(mc, 0, a);                                     // rank 0 on the caller's thread
    for (std::thread &w : th) w.join();
    for (int r = 0; r < R; ++r)
        if (mc->st[(size_t)r] != PARTLS_OK) {
            set_error("rank %d (device %d): %s", r, mc->devices[(size_t)r], mc->msg[(size_t)r].c_str());
            for (auto &v : mc->all_opt) std::vector<double>().swap(v);
            return mc->st[(size_t)r];
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

}  // extern "C"
