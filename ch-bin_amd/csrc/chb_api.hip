// Host side of libchbin_hip.so: the C ABI declared in include/chbin_hip.h.
//
// The fit loop reproduces algorithm.py:37-76 exactly -- including the in-place (Gauss-Seidel)
// label updates -- while evaluating K contigs at a time:
//
//   batch = K consecutive positions of the sweep's permutation.
//   T0(j,c): the m nearest members of bin c among samples OUTSIDE the batch (labels frozen at batch
//            start).  One heavy launch per batch (topm_base).
//   round r: every position j takes, for each bin c, T0(j,c) merged with the batch members that
//            belong to c when j is visited: EARLIER positions under their label of round r-1,
//            LATER positions under their pre-batch label (they have not been visited yet).
//            Then hull distances, strict-'>' argmin (algorithm.py:57), giving label_r.
//   Let f = first position with label_r != label_{r-1}.  Positions <= f were computed from labels
//   that can no longer change, hence are final; the next round only re-evaluates positions > f.
//   No change (f = K) means label_r is the unique fixed point = the sequential result.
// In sweep 1 the pre-batch labels of the batch are all -1; in later sweeps they are last sweep's
// labels and a batch typically converges in one round.
#include "chb_internal.h"
#include "../../include/chbin_hip.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <unordered_set>
#include <vector>

using namespace chb;

namespace {

thread_local std::string g_err;

// a batch's verdict slot: [0] first changed position of the round, [1] tiles of the largest bin, [2] tiles of all bins,
// [3..5] wave-tiles skipped / seen / never loaded (tile skipping), [6] fill mark of the persistent pack's arena,
// [7..8] candidates admitted / pairs (threshold pools), sampled by the batch's base shortlist launch; the rest spare
constexpr int kSlotInts = 16;

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

// RCCL is resolved at run time (dlopen) so that the library neither forces a second copy of RCCL
// into a process that already has one (PyTorch bundles its own) nor needs it for single-GPU use.
struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

RcclApi *rccl()
{
    static RcclApi api;
    static bool tried = false;
    if (tried) return api.ok ? &api : nullptr;
    tried = true;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);   // a copy already in the process
        if (api.lib) break;
    }
    if (!api.lib)
        for (const char *n : names) {
            api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (api.lib) break;
        }
    if (!api.lib) return nullptr;
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.lib, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.lib, "ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
    api.AllGather = (decltype(api.AllGather))dlsym(api.lib, "ncclAllGather");
    api.Broadcast = (decltype(api.Broadcast))dlsym(api.lib, "ncclBroadcast");
    api.CommCount = (decltype(api.CommCount))dlsym(api.lib, "ncclCommCount");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
    api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllGather && api.GetErrorString &&
             api.Broadcast && api.CommCount;
    return api.ok ? &api : nullptr;
}

#define NCCLCHK(expr)                                                                      \
    do {                                                                                   \
        ncclResult_t r_ = (expr);                                                          \
        if (r_ != ncclSuccess)                                                             \
            return fail(CHB_EHIP, std::string(#expr) + ": " + rccl()->GetErrorString(r_)); \
    } while (0)

#define HIPCHK(expr)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(CHB_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));      \
    } while (0)

struct ProfEntry {
    double ms = 0.0;
    int64_t launches = 0;
    double work = 0.0;
};

struct Pending {
    hipEvent_t a, b;
    std::string name;
    double work;
};

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t n)
    {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        hipError_t e = hipMalloc((void **)&p, std::max<size_t>(n, 1) * sizeof(T));
        if (e == hipSuccess) cap = n;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
};

// pinned host staging (labels, permutations): asynchronous copies at DMA speed instead of the driver's
// bounce-buffer path for pageable memory
template <typename T>
struct PinBuf {
    T *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t n)
    {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        hipError_t e = hipHostMalloc((void **)&p, std::max<size_t>(n, 1) * sizeof(T), hipHostMallocDefault);
        if (e == hipSuccess) cap = n;
        return e;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
    PinBuf() = default;
    PinBuf(const PinBuf &) = delete;
    PinBuf &operator=(const PinBuf &) = delete;
    ~PinBuf() { release(); }
};

}  // namespace

// device storage behind a MemberPack
struct PackBufs {
    DevBuf<unsigned short> Z;
    DevBuf<float> bias, sn, cs, cb, bb, tsn;   // (bias / sn / cs / cb: the batch-entry pack's per-row columns)
    DevBuf<int> pad_ptr;
    hipError_t ensure(size_t rows, size_t B, size_t Dz)
    {
        hipError_t e;
        if ((e = Z.ensure((rows + 64) * Dz)) != hipSuccess) return e;   // + slack: whole 32-row tiles are read
        DevBuf<float> *f[] = {&bias, &sn, &cs, &cb};
        for (auto *b : f)
            if ((e = b->ensure(rows + 64)) != hipSuccess) return e;
        if ((e = bb.ensure(4 * B)) != hipSuccess) return e;
        if ((e = tsn.ensure(rows / 32 + B + 68)) != hipSuccess) return e;   // (+ 64: the tile-skipping kernel reads 64 at a time)
        return pad_ptr.ensure(B + 1);
    }
    void release()
    {
        Z.release(); bias.release(); sn.release(); cs.release(); cb.release();
        bb.release(); tsn.release(); pad_ptr.release();
    }
    chb::MemberPack view()
    {
        return chb::MemberPack{Z.p, bias.p, sn.p, cs.p, cb.p, tsn.p, pad_ptr.p, reinterpret_cast<float4 *>(bb.p)};
    }
};

struct chb_ctx {
    int dev = 0;
    hipStream_t stream = nullptr;
    // samples
    DevBuf<double> X;
    int64_t N = 0;
    int D = 0, Dp = 0;
    // fit state
    int B = 0, m = 0;
    int metric = 0;   // CHB_METRIC_CONVEX / CHB_METRIC_AFFINE
    bool fit_open = false;
    DevBuf<int> labels, inb;
    // batch state
    int K = 0, Kcap = 0, q_lo = 0, q_hi = 0;
    bool batch_open = false;
    DevBuf<int> bq, lab_old, lab_prev, lab_new, first_change;
    int *bq_cur = nullptr;      // the open batch's sample indices: bq.p, or a window of perm (no copy)
    int *fc_host = nullptr;     // pinned landing places of the two verdict slots (kSlotInts ints each, as on the device)
    int *fc_cur = nullptr;      // slot of the open batch (first_change.p + 0 / kSlotInts): {first changed position, tiles of the
                                // batch's largest bin, tiles of all bins, wave-tiles skipped / seen by the first
                                // workgroups of its shortlist launch}: bin sizes and skip statistics ride home with the
                                // verdict in one 20-byte copy
    hipEvent_t fc_event[2] = {nullptr, nullptr};
    bool speculate = true;      // CHB_SPECULATE=0: never enqueue the next batch ahead of the convergence test
    bool argmin_in_place = false;   // chb_fit_cluster without exchange: argmin also stores the label to lab_prev
    DevBuf<double> mind, mind2, dist;   // winning hull distance, runner-up (margin report), all distances
    bool want_margin = false;
    DevBuf<double> l0d, l1d, l2d;
    DevBuf<int> l0i, l1i, l0c, l1c, l2i, l2c;
    int round_in_batch = 0;   // rounds alternate between the list sets 1 and 2 (the other = previous)
    DevBuf<int> cnt, bin_ptr, cursor, memb_id;
    DevBuf<int> cnt2, bin_ptr2, cursor2, memb2_id, memb2_code;
    DevBuf<int> perm;
    PinBuf<int> pin_a, pin_b, pin_c;   // host staging: labels in, labels out, permutation
    // two-stage selection: fp16 shadow copies + shortlists (prefilter_kernels.hip)
    DevBuf<unsigned short> Gs, Zs;     // per sample: query-side row (global centre), member-side row (own bin)
    DevBuf<float> gq, ms;              // per sample: float2 {||qh||^2, rho}, float4 {bias, rho, ||zh||^2, amax}
    DevBuf<double> mu_g, colsum_part;  // global mean, scratch of its two-pass sum
    DevBuf<unsigned int> rmax;
    double shadow_scale = 1.0;         // S
    PackBufs pk, pk2;                  // padded member packs: base members, the batch's own entries
    DevBuf<float> qn;                  // [N][B] float2 exact sample-to-centre norms (per fit)
    DevBuf<double> centers;
    int Dz = 0;
    bool shadow_ok = false, use_prefilter = true, overflow_total_valid = false;
    bool pf_base = true, pf_update = true;   // developer switches (CHB_PF_BASE / CHB_PF_UPDATE; -DCHB_DEV_KNOBS builds only)
    DevBuf<int> cand, cand_cnt, flags64, flaglist, nflag, overflow;
    DevBuf<int> flaglist2, nflag2;   // what the second-chance launch of a pool batch leaves for the brute-force kernel
    DevBuf<int> active, n_active, act_blk;
    // fused selection + hull distance (m <= 16): batch-entry candidates of this / the previous round,
    // the base stage's tau (bound of the m-th nearest distance), the exact path's work list
    bool fused = false, allow_fused = true;
    bool pf_fit = false;        // this fit uses the shortlist stage (use_prefilter, D <= 160, m <= 16)
    bool lists_valid = false;   // the open batch was started with need_lists (chb_topm_per_bin)
    DevBuf<int> candu[2], candu_cnt[2], slow, n_slow;
    // the shortlist stage's contract as checked by the fused kernels (FusedArgs::short_cnt): pairs of this fit whose base
    // shortlist held fewer than min(m, bin size) candidates or a wild index -- any is an internal error of the fit
    DevBuf<int> short_cnt;
    DevBuf<int> agree;   // chb_bcast_samples: {status, N, D, root} of every rank; chb_fit_cluster: the fit's agreement table
    // framed exchange of the sharded loop (aux_kernels.hip: xchg_pack / xchg_unpack): every rank's {header, label slice},
    // the number of the fit's next exchange (part of the tag every frame carries), and the device's "a rank was out of
    // step" record {flag, my tag, its tag, rank}
    DevBuf<int> xg, xerr;
    int xseq = 0;
    int dev_inject_batches = 0;   // developer builds: batch starts of this context so far (CHB_SL_INJECT_SHORT)
#ifdef CHB_DEV_KNOBS
    chb::ShortlistArgs dev_pa{}; bool dev_pa_valid = false;   // the open batch's base shortlist launch (CHB_DEV_OVERLAP)
#endif
    // the persistent base pack (prefilter_kernels.hip): the member pack kept across the batches of a fit
    DevBuf<int> pp_start, pp_cap, pp_fill, pp_live, pp_nt, pp_memb, pp_row, pp_ctl, pp_ovf, pp_dest;
    int pp_arena_rows = 0;
    int64_t pp_mark = 0;       // rows handed out from which on the host asks for a rebuild (pack_state_build)
    bool pp_allowed = true;    // CHB_PACK_INCR=0: every batch start rebuilds CSR and pack (the form up to round 3; A/B tests)
    bool pp_fit = false;       // inside chb_fit_cluster (the stepwise entry points and chb_topm_per_bin always rebuild)
    bool pp_valid = false;     // the pack on the device matches the labels
    bool pp_batch = false;     // the open batch was started on it
    bool pp_rebuild = false;   // much of the arena is used up: the next batch start outside a look-ahead window rebuilds
    int64_t stats_pp_batches = 0, stats_pp_builds = 0;
    chb::PackState pack_state()
    {
        return chb::PackState{pp_start.p, pp_cap.p, pp_fill.p, pp_live.p, pp_nt.p, pp_memb.p, pp_row.p, pp_ctl.p, pp_ovf.p,
                              pp_dest.p, pp_arena_rows};
    }
    // threshold pools of the shortlist stage (prefilter_kernels.hip, "threshold pools"): for every (bin, home bin) the 32
    // base members of the bin nearest to the home bin's centre; built at a fit's start, maintained by every commit
    DevBuf<unsigned short> pool_Z;
    DevBuf<int> pool_id, pool_hole, pool_ok, pool_stat;
    DevBuf<float> pool_key, pool_sn, pool_tsn;
    bool pool_force = false;    // CHB_POOL_TAU=2: pools whatever the size of the fit (A/B tests)
    bool pool_allowed = true;   // CHB_POOL_TAU=0: the base shortlist launch always streams a bin twice (the form up to round 4; A/B tests)
    bool pool_fit = false;      // inside chb_fit_cluster (the stepwise entry points and chb_topm_per_bin never use pools)
    bool pool_valid = false;    // the pools on the device match the labels
    bool pool_holes = true;     // the open batch may hold labelled samples (their pool slots are holes until the commit)
    int pool_state = 0;         // this fit: 0 undecided = on, 1 kept on, -1 off (its shortlists came out long: overlapping bins)
    int pool_batches = 0;
    long long pool_cand = 0, pool_pairs = 0;
    long long pool_off_key = -1;   // (bins, neighbours, metric) of the fit that turned them off on these samples
    int64_t stats_pool_batches = 0;
    chb::PoolState pool_view()
    {
        return chb::PoolState{pool_Z.p, pool_id.p, pool_key.p, pool_sn.p, pool_hole.p, pool_tsn.p, pool_ok.p};
    }
    int64_t short_seen = 0;
    // bins far larger than the rest are cut into segments for the shortlist stage (SegPlan, prefilter_kernels.hip): plan
    // buffers, and the bin sizes last seen by the host (they come home with the rounds' verdicts)
    DevBuf<int> seg_nseg, seg_gflag;
    DevBuf<int4> seg_items;
    DevBuf<float> seg_lists;
    int seg_gcap = 0;
    int hint_max_tiles = 0, hint_total_tiles = 0;
    bool allow_segments = true;   // CHB_SEGMENTS=0: never (A/B tests)
    // shells: the CSR of the base members is keyed (bin, shell of the member's distance from the bin's centre), outermost
    // shell first, so that the rows of a 32-row tile have similar norms (tile skipping in the shortlist kernel)
    DevBuf<float> shell_inv;
    int nsh = 1;
    // tile skipping in the base shortlist launch (needs the shells above and queries seated by nearest bin centre):
    // nearest-centre keys, the seating order, and the fit's verdict on whether it pays (0 undecided = on, 1 on, -1 off)
    DevBuf<unsigned long long> ckey;
    DevBuf<int> qord, home;
    // (the fit loop orders the positions of ALL batches of a sweep in one launch; the open batch's part: qord_cur / home_cur)
    DevBuf<int> qord_all, home_all;
    DevBuf<int4> geo_all;
    PinBuf<int4> pin_geo;
    int *qord_cur = nullptr, *home_cur = nullptr;
    bool allow_skip = true;       // CHB_TILE_SKIP=0: never (A/B tests)
    int skip_state = 0, skip_batches = 0;
    long long skip_off_key = -1;   // (bins, neighbours, metric) of the fit that found nothing to skip on these samples
    long long last_batch = 0;
    long long skip_skipped = 0, skip_seen = 0, skip_unloaded = 0;
    DevBuf<float> tau;
    // scratch for the indexed / explicit-point entry points
    DevBuf<int> xq, xhull, xcnt;
    DevBuf<double> xdist, xalpha, xpts;
    // profiling
    int prof = 0;   // 0 off, 1 every kernel, 2 only the two dominant ones (cheap enough for a timed region)
    std::map<std::string, ProfEntry> prof_acc;
    std::vector<Pending> pending;
    int64_t stats[4] = {0, 0, 0, 0};
    int64_t stats_seg_batches = 0;   // batches of the last fit that ran the segment launches
    int64_t stats_lookahead = 0;     // batches of the last fit whose successor was enqueued ahead of their verdict and kept
    int64_t stats_lookahead_failed = 0;   // ... and discarded (the batch needed further rounds)
    // multi-GPU: one context per process per GPU, RCCL communicator over all ranks
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    // host-staged exchange (chb_comm_init_hook): the same sharded loop with the all-gathers done by a
    // caller-supplied function on host buffers -- MPI, gloo, pipes; also how two ranks can share one GPU
    chb_allgather_fn hook = nullptr;
    void *hook_user = nullptr;
    std::vector<char> hook_send, hook_recv;
    bool force_gather = false;   // CHB_FORCE_GATHER=1: run the exchange path even with one rank (tests)
    // work-unit hints for the profile (pairs = queries x members streamed)
    double hint_base_members = 0.0, hint_batch_entries = 0.0;

    Lists L0() { return Lists{l0d.p, l0i.p, l0c.p}; }
    Lists L1() { return Lists{l1d.p, l1i.p, l1c.p}; }
    Lists L2() { return Lists{l2d.p, l2i.p, l2c.p}; }
    Lists Lcur() { return (round_in_batch & 1) ? L2() : L1(); }
    Lists Lprev() { return (round_in_batch & 1) ? L1() : L2(); }
};

// In-place all-gather of `count` elements per rank inside the device buffer `buf` (rank r's slice sits at
// buf + r * count): RCCL on the context's stream, or the host-staged hook.
static int exchange_all_gather(chb_ctx *h, void *buf, size_t count, size_t elem, ncclDataType_t dt)
{
    char *b = static_cast<char *>(buf);
    if (h->hook != nullptr) {
        const size_t bytes = count * elem;
        h->hook_send.resize(bytes);
        h->hook_recv.resize(bytes * (size_t)h->world);
        HIPCHK(hipMemcpyAsync(h->hook_send.data(), b + (size_t)h->rank * bytes, bytes, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        if (h->hook(h->hook_user, h->hook_send.data(), h->hook_recv.data(), bytes) != 0)
            return fail(CHB_EHIP, "the exchange hook reported a failure");
        HIPCHK(hipMemcpyAsync(b, h->hook_recv.data(), bytes * (size_t)h->world, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));   // (hook_recv is reused by the next exchange)
        return CHB_OK;
    }
    NCCLCHK(rccl()->AllGather(b + (size_t)h->rank * count * elem, b, count, dt, h->comm, h->stream));
    return CHB_OK;
}

namespace {

struct Timed {
    chb_ctx *h;
    Pending p;
    bool on;
    Timed(chb_ctx *h_, const char *name, double work)
        : h(h_), on(h_->prof == 1 || (h_->prof == 2 && (!strcmp(name, "prefilter") || !strcmp(name, "hull_qp"))))
    {
        if (!on) return;
        p.name = name; p.work = work;
        (void)hipEventCreate(&p.a);
        (void)hipEventCreate(&p.b);
        (void)hipEventRecord(p.a, h->stream);
    }
    ~Timed()
    {
        if (!on) return;
        (void)hipEventRecord(p.b, h->stream);
        h->pending.push_back(p);
    }
};

void drain_profile(chb_ctx *h)
{
    for (auto &p : h->pending) {
        (void)hipEventSynchronize(p.b);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, p.a, p.b);
        auto &e = h->prof_acc[p.name];
        e.ms += ms; e.launches += 1; e.work += p.work;
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    h->pending.clear();
}

long long skip_key(const chb_ctx *h) { return (long long)h->B | ((long long)h->m << 32) | ((long long)h->metric << 40); }

int ensure_batch_buffers(chb_ctx *h, int Kcap)
{
    const size_t B = h->B, m = h->m, K = Kcap;
    const size_t Kpad = K + (size_t)h->world;   // allgather of ceil(K/world)-sized slices
    HIPCHK(h->bq.ensure(K));
    HIPCHK(h->lab_old.ensure(K));
    HIPCHK(h->lab_prev.ensure(Kpad));
    HIPCHK(h->lab_new.ensure(Kpad));
    HIPCHK(h->first_change.ensure(2 * kSlotInts));
    h->fc_cur = h->first_change.p;
    HIPCHK(h->xg.ensure(Kpad + (size_t)(kXchgHdr + 1) * (size_t)h->world));
    { const size_t had = h->xerr.cap; HIPCHK(h->xerr.ensure(4)); if (!had) HIPCHK(hipMemsetAsync(h->xerr.p, 0, 4 * sizeof(int), h->stream)); }
    HIPCHK(h->mind.ensure(Kpad));
    HIPCHK(h->mind2.ensure(Kpad));
    HIPCHK(h->dist.ensure(K * B));
    HIPCHK(h->l0d.ensure(K * B * m));
    HIPCHK(h->l1d.ensure(K * B * m));
    HIPCHK(h->l2d.ensure(K * B * m));
    HIPCHK(h->l2i.ensure(K * B * m));
    HIPCHK(h->l2c.ensure(K * B));
    HIPCHK(h->l0i.ensure(K * B * m));
    HIPCHK(h->l1i.ensure(K * B * m));
    HIPCHK(h->l0c.ensure(K * B));
    HIPCHK(h->l1c.ensure(K * B));
    { const size_t had = h->cnt.cap; HIPCHK(h->cnt.ensure(B * (size_t)kShells)); if (h->cnt.cap != had) HIPCHK(hipMemsetAsync(h->cnt.p, 0, sizeof(int) * h->cnt.cap, h->stream)); }
    HIPCHK(h->bin_ptr.ensure(B + 1));
    HIPCHK(h->cursor.ensure(B * (size_t)kShells));
    HIPCHK(h->memb_id.ensure((size_t)h->N));
    HIPCHK(h->cnt2.ensure(B));
    HIPCHK(h->bin_ptr2.ensure(B + 1));
    HIPCHK(h->cursor2.ensure(B));
    HIPCHK(h->memb2_id.ensure(2 * K));
    HIPCHK(h->memb2_code.ensure(2 * K));
    if (h->pf_fit) {
        HIPCHK(h->cand.ensure(K * B * (size_t)kCandCap));
        HIPCHK(h->cand_cnt.ensure(K * B));
        HIPCHK(h->active.ensure(K * B));
        HIPCHK(h->n_active.ensure(1));
        HIPCHK(h->act_blk.ensure((K * B + 4095) / 4096 + 1));
        HIPCHK(h->flags64.ensure(B * ((K + kQTile - 1) / kQTile)));
        HIPCHK(h->flaglist.ensure(B * ((K + kQTile - 1) / kQTile)));
        HIPCHK(h->nflag.ensure(1));
        HIPCHK(h->flaglist2.ensure(B * ((K + kQTile - 1) / kQTile)));
        HIPCHK(h->nflag2.ensure(1));
        launch_fill_i32(h->flags64.p, 0, (int)(B * ((K + kQTile - 1) / kQTile)), h->stream);   // kept zero by its consumer
        HIPCHK(h->overflow.ensure(1));
        HIPCHK(h->pk.ensure((size_t)h->N + 32 * B, B, (size_t)h->Dz));
        HIPCHK(h->pk2.ensure(2 * K + 32 * B, B, (size_t)h->Dz));
        h->seg_gcap = (int)std::min<size_t>(64, B / 4 + 1);
        HIPCHK(h->qord.ensure(K));
        HIPCHK(h->home.ensure(B));
        HIPCHK(h->seg_nseg.ensure(1));
        HIPCHK(h->seg_gflag.ensure(B));
        HIPCHK(h->seg_items.ensure(16 * (size_t)h->seg_gcap));
        // (seg_lists -- giant slots x 16 segments x K x list length floats, 0.27 / 0.86 GB at 1M x 200 bins for m = 5 / 15 --
        //  is allocated by the first batch that really runs the segment launches: batch_begin_dev)
        if (h->fused) {
            for (int i = 0; i < 2; ++i) {
                HIPCHK(h->candu[i].ensure(K * B * (size_t)kCandCapU));
                HIPCHK(h->candu_cnt[i].ensure(K * B));
            }
            HIPCHK(h->slow.ensure(K * B));
            HIPCHK(h->n_slow.ensure(1));
            { const size_t had = h->short_cnt.cap; HIPCHK(h->short_cnt.ensure(1)); if (!had) HIPCHK(hipMemsetAsync(h->short_cnt.p, 0, sizeof(int), h->stream)); }
            HIPCHK(h->tau.ensure(K * B));
        }
    }
    h->Kcap = Kcap;
    return CHB_OK;
}

// sync = false (chb_fit_cluster): the uploads and kernels of the fit's start are only enqueued -- the caller goes on with
// its own host work (permutation check / conversion) while they run, and the stream orders everything behind them
int fit_begin_impl(chb_ctx *h, int64_t B, const int64_t *initial, int m, bool sync = true)
{
    if (!h->X.p) return fail(CHB_ESTATE, "chb_set_samples has not been called");
    if (B <= 0) return fail(CHB_EINVAL, "num_clusters must be positive");
    if (B > 8192) return fail(CHB_EUNSUPPORTED, "more than 8192 bins (per-block LDS histograms of the CSR build)");
    if (m < 1 || m > CHB_MAX_NEIGHBORS)
        return fail(CHB_EUNSUPPORTED, "num_neighbors must be in [1, 64]");
    if (m > kMaxM && !hull_generic_supported())
        return fail(CHB_EUNSUPPORTED, "num_neighbors > 16 needs 68 KB of LDS per workgroup, which this device does not grant");
    h->B = (int)B; h->m = m;
    // the fp16 shortlist stage and the tuned kernels hold lists of up to 16 entries; beyond that the plain
    // one-wavefront-per-problem kernels run (brute-force selection, LDS-resident solver)
    h->pf_fit = h->use_prefilter && h->shadow_ok && m <= kMaxM;
    // (a fit that found nothing to skip settles it for later fits over the same samples with the same bin count,
    //  neighbour count and metric -- the verdict depends on all three)
    // (wide rows, Dz > 160: the plain two-sweep builds only -- no tile skipping, pools or segments)
    h->skip_state = (h->skip_off_key == skip_key(h) || h->Dz > 160) ? -1 : 0;
    h->skip_batches = 0; h->skip_skipped = 0; h->skip_seen = 0; h->skip_unloaded = 0;
    h->fused = h->allow_fused && h->pf_fit && h->pf_base && h->pf_update && fused_supported(m, h->Dp);
    HIPCHK(h->pin_a.ensure((size_t)h->N));
    int *lab = h->pin_a.p;
    std::vector<int64_t> bin_size((size_t)B, 0);
    for (int64_t i = 0; i < h->N; ++i) {
        const int64_t v = initial[i];
        if (v >= B) return fail(CHB_EINVAL, "initial_bins contains a label >= num_clusters");
        lab[(size_t)i] = v < 0 ? -1 : (int)v;
        if (v >= 0) ++bin_size[(size_t)v];
    }
    {   // bin sizes as the first batch will see them (later ones come home with the rounds' verdicts)
        int64_t mx = 0, tot = 0;
        for (int64_t c = 0; c < B; ++c) { const int64_t t = (bin_size[(size_t)c] + 31) / 32; mx = std::max(mx, t); tot += t; }
        h->hint_max_tiles = (int)mx; h->hint_total_tiles = (int)std::min<int64_t>(tot, 0x7fffffff);
    }
    HIPCHK(h->labels.ensure((size_t)h->N));
    HIPCHK(h->inb.ensure((size_t)h->N));
    HIPCHK(hipMemcpyAsync(h->labels.p, lab, sizeof(int) * h->N, hipMemcpyHostToDevice, h->stream));
    launch_fill_i32(h->inb.p, -1, (int)h->N, h->stream);
    HIPCHK(h->cnt.ensure((size_t)B * kShells));
    HIPCHK(hipMemsetAsync(h->cnt.p, 0, sizeof(int) * h->cnt.cap, h->stream));   // (kept zero by scan_kernel from here on)
    HIPCHK(h->bin_ptr.ensure((size_t)B + 1));
    HIPCHK(h->cursor.ensure((size_t)B * kShells));
    h->nsh = 1;
    HIPCHK(h->memb_id.ensure((size_t)h->N));
    if (h->pf_fit) {
        // Bin centres for the shortlist stage: the mean of each bin's initially labelled members
        // (the seeds), fixed for the whole fit -- any fixed point keeps the bounds valid, one near
        // the bin keeps them tight.  Then every labelled sample's shadow row against its own bin.
        HIPCHK(h->centers.ensure((size_t)B * h->Dp));
        {
            Timed t(h, "fit_start", (double)h->N);
            launch_bucket_base(h->labels.p, h->inb.p, (int)h->N, h->B, h->cnt.p, h->bin_ptr.p, h->cursor.p,
                               h->memb_id.p, nullptr, nullptr, h->stream);
            launch_bin_centers(h->X.p, h->D, h->Dp, h->memb_id.p, h->bin_ptr.p, h->B, h->centers.p, h->stream);
            launch_sample_shadow(h->X.p, h->D, h->Dp, nullptr, (int)h->N, h->labels.p, h->B, h->centers.p,
                                 h->mu_g.p, h->shadow_scale, h->Zs.p, h->Dz, h->ms.p, nullptr, nullptr, h->stream);
        }
        // every sample's exact norm against every (fixed) centre, and its nearest centre: once per fit, not per batch.
        // N x B x 8 bytes (51 MB at 100k x 64, 1.6 GB at 1M x 200, 65 GB at 1M x 8192): a table that does not fit the
        // device sends the fit to the brute-force selection (as a feature width beyond the shortlist stage's does) instead
        // of failing it
        if (h->qn.ensure((size_t)h->N * (size_t)B * 2) != hipSuccess || h->ckey.ensure((size_t)h->N) != hipSuccess) {
            (void)hipGetLastError();
            h->qn.release(); h->ckey.release();
            h->pf_fit = false; h->fused = false;
        }
    }
    if (h->pf_fit) {
        {
            Timed t(h, "query_norms", (double)h->N * (double)B);
            launch_query_norms(h->X.p, h->D, h->Dp, (int)h->N, h->B, h->centers.p, h->shadow_scale, h->qn.p, h->ckey.p, h->stream);
        }
        // the unit of the CSR's shell key per bin (from the initially labelled members; fixed for the fit)
        int nsh = kShells;
        while (nsh > 1 && (int64_t)B * nsh > kMaxKeys) nsh >>= 1;
        HIPCHK(h->shell_inv.ensure((size_t)B));
        launch_shell_scale(h->ms.p, h->memb_id.p, h->bin_ptr.p, h->B, nsh, h->shell_inv.p, h->stream);
        h->nsh = nsh;
        HIPCHK(hipGetLastError());
    }
    if (sync) HIPCHK(hipStreamSynchronize(h->stream));
    h->fit_open = true; h->batch_open = false; h->Kcap = 0;
    h->overflow_total_valid = false;
    h->short_seen = 0;
    if (h->short_cnt.p) HIPCHK(hipMemsetAsync(h->short_cnt.p, 0, sizeof(int), h->stream));
    h->pp_valid = false; h->pp_batch = false; h->pp_rebuild = false;
    if (h->pp_ctl.p) HIPCHK(hipMemsetAsync(h->pp_ctl.p, 0, 4 * sizeof(int), h->stream));
    return CHB_OK;
}

// The persistent base pack from the labels as they stand (no batch open): compact CSR of all labelled samples, then
// regions with room to grow.  Allocates on first use (12 N + 1024 B rows: the layout takes at most 3.5 N + 128 B, a commit's
// moves at most 3 (N + K) + 96 B, the host rebuilds from 5 N + 512 B on) -- hence only ever called outside a look-ahead window.
int pack_state_build(chb_ctx *h)
{
    const size_t B = h->B;
    // (layout: at most 2 N + 1.5 N + 128 B rows; 2 * row + 1 must fit an int)
    const int arena = (int)std::min<int64_t>(12 * h->N + 1024 * (int64_t)B, 0x3fffff00);
    // The host asks for a rebuild (compaction) once `mark` rows are handed out; the request is honoured up to three commits
    // later (it rides home with a verdict, lags a batch under the look-ahead and waits for a batch start outside a window).
    // What three commits can take: ONE mass move -- every bin's region tripling at once, 3 (N + K) + 96 B rows, after which
    // the regions hold three times their members and cannot move again at once -- plus the appends of the other two (K rows
    // each): 3 N + 5 K + 128 B.  The mark leaves that much room; it always lies above a fresh layout (3.5 N + 128 B) since a
    // batch never holds more than N samples.
    {
        const int64_t K = std::max(h->Kcap, 1);
        h->pp_mark = std::min<int64_t>(5 * h->N + 512 * (int64_t)B, (int64_t)arena - (3 * h->N + 5 * K + 128 * (int64_t)B));
        if (h->pp_mark < 7 * h->N / 2 + 128 * (int64_t)B) { h->pp_valid = false; h->pp_fit = false; return CHB_OK; }   // (capped arena)
    }
    if (h->pp_arena_rows < arena || !h->pp_memb.p) {
        if (h->pk.ensure((size_t)arena, B, (size_t)h->Dz) != hipSuccess || h->pp_memb.ensure((size_t)arena + 64) != hipSuccess) {
            (void)hipGetLastError();   // (no room for the arena: this fit rebuilds its pack per batch)
            h->pp_memb.release();
            HIPCHK(h->pk.ensure((size_t)h->N + 32 * B, B, (size_t)h->Dz));
            h->pp_arena_rows = 0; h->pp_valid = false; h->pp_fit = false;
            return CHB_OK;
        }
        h->pp_arena_rows = arena;
    }
    HIPCHK(h->pp_row.ensure((size_t)h->N));
    DevBuf<int> *pb[] = {&h->pp_start, &h->pp_cap, &h->pp_fill, &h->pp_live, &h->pp_nt};
    for (auto *b : pb) HIPCHK(b->ensure(B + 1));
    { const size_t had = h->pp_ctl.cap; HIPCHK(h->pp_ctl.ensure(4)); if (!had) HIPCHK(hipMemsetAsync(h->pp_ctl.p, 0, 4 * sizeof(int), h->stream)); }
    HIPCHK(h->pp_ovf.ensure((size_t)std::max(h->Kcap, 1)));
    HIPCHK(h->pp_dest.ensure((size_t)std::max(h->Kcap, 1)));
    // (room to grow: were all N samples labelled and spread evenly, a bin would hold N / B rows -- half as much again)
    const int grow = (int)std::min<int64_t>((3 * h->N / 2) / std::max<int64_t>(h->B, 1) + 64, 0x3fffffff);
    {
        Timed t(h, "bucket", (double)h->N);
        launch_bucket_base(h->labels.p, h->inb.p, (int)h->N, h->B, h->cnt.p, h->bin_ptr.p, h->cursor.p, h->memb_id.p, nullptr,
                           nullptr, h->stream);
        launch_pack_state_build(h->pack_state(), h->pk.view(), h->Zs.p, h->ms.p, h->D, h->Dz, h->memb_id.p, h->bin_ptr.p, h->B,
                                (int)h->N, grow, h->stream);
    }
    HIPCHK(hipGetLastError());
    h->pp_valid = true; h->pp_rebuild = false;
    h->stats_pp_builds += 1;
    return CHB_OK;
}

// The threshold pools from the labels as they stand at a fit's start (the CSR fit_begin_impl has just made): allocated on
// first use, B x B tiles of 32 shadow rows -- 38 MB at 100k x 136 x 64, 410 MB at 1M x 146 x 200; a fit whose pools would
// take more than kPoolMaxBytes keeps the two-sweep shortlist launch.
constexpr size_t kPoolMaxBytes = (size_t)1 << 30;
int pool_build(chb_ctx *h)
{
    h->pool_valid = false;
    const size_t B = h->B, slots = B * B * (size_t)kPoolRows;
    // (m > 8: the 16-lane hull kernel pays for every candidate beyond 16 with extra rows and second tiles -- with the pools'
    //  17.4 instead of 16.8 candidates per pair at m = 15 it ran 34.4 instead of 28.8 ms per sweep, more than the shortlist
    //  kernel saved (5.1 instead of 5.9): those fits keep the two sweeps)
    if (!h->pool_allowed || !h->fused || !h->pf_fit || !h->pf_base || h->ckey.p == nullptr || B < 2 || h->m > 8 || h->Dz > 160 ||
        slots * (size_t)h->Dz * sizeof(unsigned short) > kPoolMaxBytes)
        return CHB_OK;
    // (small fits: a bin of a few tiles has no threshold sweep worth replacing, while the pools' build and upkeep are per
    //  fit and per batch -- BASELINE configs[1], 10k x 32 = 10 tiles per bin, went from 1.45 to 1.85 ms per sweep with them.
    //  From 16 tiles per bin on average; CHB_POOL_TAU=2 keeps them whatever the size)
    if (!h->pool_force && (size_t)h->N < 512 * B) return CHB_OK;
    if (h->pool_Z.ensure(slots * (size_t)h->Dz) != hipSuccess || h->pool_id.ensure(slots) != hipSuccess ||
        h->pool_hole.ensure(slots) != hipSuccess || h->pool_key.ensure(slots) != hipSuccess || h->pool_sn.ensure(slots) != hipSuccess ||
        h->pool_tsn.ensure(B * B + 64) != hipSuccess || h->pool_ok.ensure(B * B) != hipSuccess) {
        (void)hipGetLastError();   // (no room: the fit keeps the two-sweep launch)
        h->pool_Z.release(); h->pool_id.release(); h->pool_hole.release(); h->pool_key.release(); h->pool_sn.release();
        return CHB_OK;
    }
    {
        Timed t(h, "pool", (double)h->N);
        launch_pool_build(h->pool_view(), h->Zs.p, h->ms.p, h->qn.p, h->D, h->Dz, h->memb_id.p, h->bin_ptr.p, h->B, h->stream);
    }
    HIPCHK(hipGetLastError());
    h->pool_valid = true;
    return CHB_OK;
}

// bq already holds the K sample indices (device).  need_lists: the caller wants the exact base lists
// L0 (chb_topm_per_bin); the fit loop of the fused path (m <= 16) works on the shortlists directly.
int batch_begin_dev(chb_ctx *h, int K, int q_lo, int q_hi, bool need_lists)
{
    const bool fusedp = h->fused && !need_lists;
    h->lists_valid = !fusedp;
    h->K = K; h->q_lo = q_lo; h->q_hi = q_hi;
    h->round_in_batch = 0;
    hipStream_t s = h->stream;
    // segmented bins: the plan is made by the CSR scan on the device, but only if the host will also enqueue the two
    // segment launches -- which it does when the bin sizes it saw last (one or two batches old) say that a bin may
    // have more than kSegMinTiles tiles and four times the average
    SegPlan sp{};
    const bool pf_base_path = h->pf_fit && h->pf_base && h->cand.p;
    // tile skipping: on until the fit's first batches have shown that it skips (next to) nothing
    const bool skip_on = pf_base_path && h->allow_skip && h->nsh > 1 && h->skip_state >= 0 && h->ckey.p != nullptr;
    if (pf_base_path && h->seg_gflag.p) {
        sp.nseg = h->seg_nseg.p; sp.items = h->seg_items.p; sp.gflag = h->seg_gflag.p; sp.lists = h->seg_lists.p;
        sp.cap = 16 * h->seg_gcap; sp.gcap = h->seg_gcap;
        const long long est = (long long)h->hint_max_tiles * 3 / 2 + 8;
        sp.launch = h->allow_segments && h->Dz <= 160 && est > kSegMinTiles && est * h->B > 3LL * std::max(1, h->hint_total_tiles);
        if (sp.launch) {
            HIPCHK(h->seg_lists.ensure((size_t)h->seg_gcap * 16 * (size_t)h->Kcap * (size_t)shortlist_list_len(h->m)));
            sp.lists = h->seg_lists.p;
        }
    }
    // the persistent base pack serves the fit loop's batches whenever the shortlist launch does not skip tiles (whose
    // shell order needs the rebuild); built / rebuilt only outside a look-ahead window
    bool pp_now = false;
    if (h->pp_fit && h->pp_allowed && fusedp && pf_base_path && !skip_on) {
        if ((!h->pp_valid || h->pp_rebuild) && g_gate.flag == nullptr) { const int r_ = pack_state_build(h); if (r_) return r_; }
        pp_now = h->pp_valid;
    }
    if (!pp_now) h->pp_valid = false;   // (this batch's commit will not maintain the pack)
    h->pp_batch = pp_now;
    // threshold pools: the base shortlist launch streams a bin once where a pool tile gives the threshold
    if (!(h->pool_fit && fusedp && pf_base_path)) h->pool_valid = false;   // (this batch will not maintain them)
    if (h->pool_state < 0) h->pool_valid = false;                          // (turned off for this fit: no upkeep either)
    const bool pool_on = h->pool_valid && h->pool_state >= 0;
    if (pp_now) {
        // the batch is opened (its members' rows become holes) and tiles per bin / statistics / segment plan written:
        // one launch instead of count + scan + fill + gather
        Timed t(h, "bucket", (double)K);
        launch_pack_state_start(h->pack_state(), h->pk.view(), h->D, h->Dz, h->labels.p, h->inb.p, h->bq_cur, K, h->lab_old.p,
                                h->B, sp.gflag ? &sp : nullptr, h->fc_cur + 1, h->nflag.p, s);
        h->stats_pp_batches += 1;
    } else {
        // (the batch is opened -- labels remembered, members marked -- inside the CSR count's launch)
        Timed t(h, "bucket", (double)h->N);
        launch_bucket_base(h->labels.p, h->inb.p, (int)h->N, h->B, h->cnt.p, h->bin_ptr.p,
                           h->cursor.p, h->memb_id.p, h->pk.pad_ptr.p, h->nflag.p, s, h->bq_cur, K, h->lab_old.p,
                           sp.gflag ? &sp : nullptr, h->fc_cur + 1, pf_base_path ? h->ms.p : nullptr,
                           skip_on ? h->shell_inv.p : nullptr, skip_on ? h->nsh : 1);
    }
    if (h->pool_valid) {
        // (the batch's samples are marked: their slots in the pools are holes while it is open)
        Timed t(h, "pool", (double)K);
        launch_pool_open(h->pool_view(), h->inb.p, h->D, h->Dz, h->B, h->nflag2.p, s);
    }
    h->pool_holes = true;
    TopmArgs a{};
    a.X = h->X.p; a.Dp = h->Dp; a.bq = h->bq_cur; a.pos_begin = q_lo; a.pos_end = q_hi;
    a.bin_ptr = h->bin_ptr.p; a.memb_id = h->memb_id.p; a.memb_code = nullptr;
    if (pp_now) { a.bin_ptr = h->pp_start.p; a.bin_cnt = h->pp_fill.p; a.memb_id = h->pp_memb.p; }
    a.B = h->B; a.m = h->m; a.Kcap = h->Kcap;
    a.in = Lists{nullptr, nullptr, nullptr};
    a.out = h->L0();
    if (h->pf_fit && h->pf_base && h->cand.p) {
        // two-stage exact selection: fp16 matrix-core shortlist, exact fp64 on the shortlist,
        // brute force only for (query tile, bin) pairs whose shortlist overflowed
        const int nq64 = (q_hi - q_lo + kQTile - 1) / kQTile;
        (void)nq64;   // (flags64 is all zero here: launch_topm_flagged clears what it serves)
        if (!h->overflow_total_valid) { launch_fill_i32(h->overflow.p, 0, 1, s); h->overflow_total_valid = true; }
        if (!pp_now) {
            // the members' shadow rows (relative to their bin's centre) gathered into padded CSR order, and the per-bin
            // bounds: one launch
            Timed t(h, "bucket", 0.0);
            launch_pack_build(h->Zs.p, h->ms.p, h->D, h->Dz, h->memb_id.p, h->bin_ptr.p, h->B, (int)h->N, h->pk.view(), skip_on, s);
        }
        int *qord_p = h->qord.p, *home_p = h->home.p;
        if ((skip_on || pool_on) && h->qord_cur != nullptr) { qord_p = h->qord_cur; home_p = h->home_cur; }   // (done for the whole sweep)
        else if (skip_on || pool_on) {   // (the queries are seated in the order of their nearest bin centre)
            Timed t(h, "bucket", 0.0);
            launch_query_order(h->ckey.p, h->bq_cur, q_lo, q_hi, h->B, h->qord.p, h->home.p, s);
        }
        ShortlistArgs pa{};
        pa.Gs = h->Gs.p; pa.gq = reinterpret_cast<const float2 *>(h->gq.p);
        pa.qn = reinterpret_cast<const float2 *>(h->qn.p);
        pa.P = h->pk.view(); pa.Dz = h->Dz; pa.S = h->shadow_scale;
        pa.bq = h->bq_cur; pa.pos_begin = q_lo; pa.pos_end = q_hi;
        pa.bin_ptr = h->bin_ptr.p; pa.memb_id = h->memb_id.p; pa.update = false;
        if (pp_now) {   // (a bin = its region: first row, tiles in use; a row's sample, -1 for a hole)
            pa.P.pad_ptr = h->pp_start.p; pa.P.nt = h->pp_nt.p;
            pa.bin_ptr = h->pp_start.p; pa.memb_id = h->pp_memb.p;
        }
        pa.B = h->B; pa.m = h->m; pa.Kcap = h->Kcap;
        pa.cand = h->cand.p; pa.cand_cnt = h->cand_cnt.p; pa.cand_cap = kCandCap; pa.overflow = h->overflow.p;
        if (fusedp) pa.tau_out = h->tau.p;
        pa.seg = sp;
        if (sp.launch) h->stats_seg_batches += 1;
        if (skip_on) { pa.qord = qord_p; pa.home = home_p; pa.skip = 1; pa.skip_stat = h->fc_cur + 3; }
        if (pool_on) {
            pa.qord = qord_p; pa.home = home_p; pa.ckey = h->ckey.p; pa.pool = h->pool_view(); pa.pool_stat = h->fc_cur + 7;
            h->stats_pool_batches += 1;
        }
#ifdef CHB_DEV_KNOBS
        if (skip_on) { if (const char *ev = getenv("CHB_SKIP_NEVER")) if (atoi(ev)) pa.skip = 1 | 2 * atoi(ev); }
#endif
#ifdef CHB_DEV_KNOBS
        // CHB_SL_DBG=<file>: per-wavefront timeline of the 10th base shortlist launch of the process (tools/sl_timeline.py)
        static int dbg_launch = 0;
        static unsigned long long *dbg_dev = nullptr;
        const char *dbg_path = getenv("CHB_SL_DBG");
        const size_t dbg_words = (size_t)65536 * 16;
        if (dbg_path != nullptr && ++dbg_launch == 10) {
            HIPCHK(hipMalloc(&dbg_dev, dbg_words * 8));
            HIPCHK(hipMemsetAsync(dbg_dev, 0, dbg_words * 8, s));
            pa.dbg = dbg_dev;
        }
#endif
#ifdef CHB_DEV_KNOBS
        // CHB_SL_BOUNDS=1: the base shortlist launch checks its tile DMA sources and member reads against the buffers' extents,
        // skips an access that is out of range and reports the first one (instead of a GPU memory fault)
        static int *viol_dev = nullptr;
        if (getenv("CHB_SL_BOUNDS") != nullptr) {
            if (viol_dev == nullptr) HIPCHK(hipMalloc(&viol_dev, 8 * sizeof(int)));
            HIPCHK(hipMemsetAsync(viol_dev, 0, 8 * sizeof(int), s));
            pa.viol = viol_dev;
            pa.viol_rows = pp_now ? (long long)h->pp_arena_rows : (long long)h->N + 32LL * h->B + 64;
            pa.viol_pool_rows = (long long)h->B * h->B * kPoolRows;
            pa.viol_members = pp_now ? (long long)h->pp_arena_rows : (long long)h->N;
        }
#endif
        {
            Timed t(h, "prefilter", (double)(q_hi - q_lo) * h->hint_base_members);
            pa.flaglist = h->flaglist.p; pa.nflag = h->nflag.p;   // (counter reset by the CSR scan / the batch CSR kernel)
            launch_shortlist(pa, h->flags64.p, s);
        }
        const int *fb_list = h->flaglist.p, *fb_n = h->nflag.p;   // the brute-force kernel's work list
        if (pool_on) {
            // second chance for the pairs whose pool threshold was too loose for a 128-entry shortlist: the exact two-sweep
            // selection on the overflow list's work items; only what overflows again goes to the brute-force kernel
            Timed t(h, "prefilter_retry", 0.0);
            ShortlistArgs pb = pa;
            pb.worklist = h->flaglist.p; pb.nwork = h->nflag.p; pb.flaglist = h->flaglist2.p; pb.nflag = h->nflag2.p;
            launch_shortlist_worklist(pb, h->flags64.p, s);
            fb_list = h->flaglist2.p; fb_n = h->nflag2.p;
        }
#ifdef CHB_DEV_KNOBS
        if (pa.viol != nullptr) {
            int hv[8];
            HIPCHK(hipStreamSynchronize(s));
            HIPCHK(hipMemcpy(hv, viol_dev, sizeof(hv), hipMemcpyDeviceToHost));
            if (hv[0] != 0) {
                fprintf(stderr, "[chb bounds] code %d: %d %d %d %d %d %d (workgroup %d); skip %d pool %d pp %d K %d q %d..%d\n", hv[0], hv[1], hv[2], hv[3],
                        hv[4], hv[5], hv[6], hv[7], (int)skip_on, (int)pool_on, (int)pp_now, K, q_lo, q_hi);
                return fail(CHB_ESTATE, "shortlist bounds check failed");
            }
        }
        h->dev_pa = pa; h->dev_pa_valid = true;   // (CHB_DEV_OVERLAP)
#endif
#ifdef CHB_DEV_KNOBS
        if (pa.dbg != nullptr) {
            std::vector<unsigned long long> hostd(dbg_words);
            HIPCHK(hipStreamSynchronize(s));
            HIPCHK(hipMemcpy(hostd.data(), dbg_dev, dbg_words * 8, hipMemcpyDeviceToHost));
            if (FILE *fp = fopen(dbg_path, "wb")) { fwrite(hostd.data(), 8, dbg_words, fp); fclose(fp); }
            HIPCHK(hipFree(dbg_dev)); dbg_dev = nullptr;
        }
#endif
        if (!fusedp) {
            RescoreArgs ra{};
            ra.X = h->X.p; ra.Dp = h->Dp; ra.bq = h->bq_cur; ra.pos_begin = q_lo; ra.pos_end = q_hi;
            ra.B = h->B; ra.m = h->m; ra.Kcap = h->Kcap;
            ra.cand = h->cand.p; ra.cand_cnt = h->cand_cnt.p; ra.cand_cap = kCandCap; ra.out = h->L0();
            Timed t(h, "rescore", (double)(q_hi - q_lo) * h->B);
            launch_rescore(ra, s);
        } else {
            // overflowed (query tile, bin) pairs: the brute-force kernel's exact top-m becomes the shortlist
            a.out = Lists{nullptr, nullptr, nullptr};
            a.cand_out = h->cand.p; a.cand_cnt_out = h->cand_cnt.p; a.cand_cap = kCandCap;
            a.tau_out = h->tau.p; a.S = h->shadow_scale;
        }
        {
            Timed t(h, "topm_fallback", 0.0);
            launch_topm_flagged(a, h->flags64.p, fb_list, fb_n, s);
        }
#ifdef CHB_DEV_KNOBS
        // CHB_SL_INJECT_SHORT=<n>: the n-th batch start of a context hands the hull kernels one truncated shortlist
        // (tests: the product build's check must turn it into an error)
        if (fusedp && !pp_now) if (const char *ev = getenv("CHB_SL_INJECT_SHORT")) {
            if (++h->dev_inject_batches == atoi(ev)) launch_inject_short(h->cand_cnt.p, h->B, h->Kcap, q_lo, h->bin_ptr.p, h->m, s);
        }
#endif
#ifdef CHB_DEV_KNOBS
        if (fusedp && !pp_now && getenv("CHB_SL_VALIDATE") != nullptr) {   // (the validation kernel reads the rebuilt CSR)
            static int *verr = nullptr;
            if (verr == nullptr) HIPCHK(hipMalloc(&verr, 16 + 4 * 65536));
            HIPCHK(hipMemsetAsync(verr, 0, 16 + 4 * 65536, s));
            launch_validate_batch(h->cand.p, h->cand_cnt.p, h->B, h->Kcap, q_lo, q_hi, kCandCap, (int)h->N, h->bin_ptr.p,
                                  h->memb_id.p, skip_on ? qord_p : nullptr, h->m, verr, s);
            int herr[4];
            HIPCHK(hipStreamSynchronize(s));
            HIPCHK(hipMemcpy(herr, verr, 16, hipMemcpyDeviceToHost));
            if (herr[0] != 0) {
                fprintf(stderr, "[chb validate] code %d bin %d position %d value %d (batch positions %d..%d, skip %d)\n", herr[0],
                        herr[1], herr[2], herr[3], q_lo, q_hi, (int)skip_on);
                if (herr[0] >= 5 && skip_on && q_hi - q_lo <= 256) {
                    const int nq = q_hi - q_lo;
                    std::vector<int> qo(nq);
                    HIPCHK(hipMemcpy(qo.data(), h->qord.p, 4 * (size_t)nq, hipMemcpyDeviceToHost));
                    for (int i = 0; i < nq; ++i) fprintf(stderr, "  i %d qord %d\n", i, qo[i]);
                }
                return fail(CHB_ESTATE, "shortlist validation failed");
            }
        }
#endif
    } else {
        Timed t(h, "topm_base", (double)(q_hi - q_lo) * h->hint_base_members);
        if (h->m > kMaxM) launch_topm_generic(a, s); else launch_topm(a, s);
    }
    HIPCHK(hipGetLastError());
    h->batch_open = true;
    return CHB_OK;
}

// lab_prev (device) holds the labels of the previous round.  Evaluates [max(active,q_lo), q_hi).
int batch_round_dev(chb_ctx *h, int active)
{
    hipStream_t s = h->stream;
    const int lo = std::max(active, h->q_lo), hi = h->q_hi;
    if (hi <= lo) launch_fill_i32(h->fc_cur, h->K, 1, s);
    const bool fusedp = h->fused && h->lists_valid == false;
    if (hi > lo) {
        {
            Timed t(h, "bucket", (double)h->K);
            launch_bucket_batch(h->lab_prev.p, h->lab_old.p, h->bq_cur, h->K, h->B, h->cnt2.p,
                                h->bin_ptr2.p, h->cursor2.p, h->memb2_id.p, h->memb2_code.p, h->pk2.pad_ptr.p,
                                h->fc_cur, fusedp ? h->n_slow.p : nullptr, h->nflag.p, s,
                                (h->pf_fit && h->pk2.bb.p) ? h->pk2.bb.p : nullptr);
        }
        TopmArgs a{};
        a.X = h->X.p; a.Dp = h->Dp; a.bq = h->bq_cur; a.pos_begin = lo; a.pos_end = hi;
        a.bin_ptr = h->bin_ptr2.p; a.memb_id = h->memb2_id.p; a.memb_code = h->memb2_code.p;
        a.B = h->B; a.m = h->m; a.Kcap = h->Kcap;
        a.in = h->L0(); a.out = h->Lcur();
        if (fusedp) {
            // ---- fused path: shortlist of the batch's own entries against the base stage's tau, then
            // selection + hull distance straight from the two shortlists
            const int cur = h->round_in_batch & 1;
            launch_pack_centered(h->X.p, h->D, h->Dp, h->memb2_id.p, h->memb2_code.p, h->bin_ptr2.p, h->B, 2 * h->K,
                                 h->centers.p, h->mu_g.p, h->shadow_scale, h->Dz, h->pk2.view(), s);
            ShortlistArgs pa{};
            pa.Gs = h->Gs.p; pa.gq = reinterpret_cast<const float2 *>(h->gq.p);
            pa.qn = reinterpret_cast<const float2 *>(h->qn.p);   // (the fit's table)
            pa.P = h->pk2.view(); pa.Dz = h->Dz; pa.S = h->shadow_scale;
            pa.bq = h->bq_cur; pa.pos_begin = lo; pa.pos_end = hi;
            pa.bin_ptr = h->bin_ptr2.p; pa.memb_id = h->memb2_id.p; pa.update = true;
            pa.tau_in = h->tau.p;
            pa.B = h->B; pa.m = h->m; pa.Kcap = h->Kcap;
            pa.cand = h->candu[cur].p; pa.cand_cnt = h->candu_cnt[cur].p; pa.cand_cap = kCandCapU;
            pa.overflow = h->overflow.p;
            {
                Timed t(h, "prefilter_update", (double)(hi - lo) * h->hint_batch_entries);
                pa.flaglist = h->flaglist.p; pa.nflag = h->nflag.p;   // (counter reset by the CSR scan / the batch CSR kernel)
            launch_shortlist(pa, h->flags64.p, s);
            }
            {
                // overflowed pairs: exact top-m among the (eligible) batch entries as their shortlist
                a.in = Lists{nullptr, nullptr, nullptr};
                a.out = Lists{nullptr, nullptr, nullptr};
                a.cand_out = h->candu[cur].p; a.cand_cnt_out = h->candu_cnt[cur].p; a.cand_cap = kCandCapU;
                Timed t(h, "topm_fallback", 0.0);
                launch_topm_flagged(a, h->flags64.p, h->flaglist.p, h->nflag.p, s);
            }
            FusedArgs f{};
            f.X = h->X.p; f.n_samples = h->N; f.D = h->D; f.Dp = h->Dp; f.bq = h->bq_cur; f.pos_begin = lo; f.pos_end = hi;
            f.B = h->B; f.m = h->m; f.Kcap = h->Kcap;
            f.cand = h->cand.p; f.cand_cnt = h->cand_cnt.p;
            f.candu = h->candu[cur].p; f.candu_cnt = h->candu_cnt[cur].p;
            if (h->round_in_batch > 0) { f.candp = h->candu[cur ^ 1].p; f.candp_cnt = h->candu_cnt[cur ^ 1].p; }
            f.dist = h->dist.p; f.metric = h->metric; f.slow = h->slow.p; f.n_slow = h->n_slow.p;
            f.bin_ptr = h->bin_ptr.p; f.short_cnt = h->short_cnt.p;
            if (h->pp_batch) f.bin_size = h->pp_live.p;
#ifdef CHB_DEV_KNOBS
            // CHB_DEV_OVERLAP=1: how much would the base shortlist launch of the NEXT batch gain from running beside this
            // batch's hull kernel?  The batch's own base shortlist launch is repeated into scratch buffers on a second
            // stream while the hull kernel runs (2: the same repeat in line on the main stream -- the additive baseline).
            static const int dev_overlap = getenv("CHB_DEV_OVERLAP") ? atoi(getenv("CHB_DEV_OVERLAP")) : 0;
            static hipStream_t dev_s2 = nullptr; static hipEvent_t dev_e1 = nullptr, dev_e2 = nullptr;
            static int *dev_scratch = nullptr; static size_t dev_scratch_n = 0;
            bool dev_side = false;
            if (dev_overlap && h->round_in_batch == 0 && h->dev_pa_valid) {
                const size_t KB = (size_t)h->Kcap * h->B;
                const size_t need = KB * kCandCap + 4 * KB + 64;
                if (dev_scratch_n < need) { if (dev_scratch) (void)hipFree(dev_scratch); HIPCHK(hipMalloc(&dev_scratch, need * sizeof(int))); dev_scratch_n = need; HIPCHK(hipMemset(dev_scratch, 0, need * sizeof(int))); }
                if (!dev_s2) { HIPCHK(hipStreamCreateWithFlags(&dev_s2, hipStreamNonBlocking)); HIPCHK(hipEventCreateWithFlags(&dev_e1, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&dev_e2, hipEventDisableTiming)); }
                ShortlistArgs pb = h->dev_pa;
                pb.cand = dev_scratch; pb.cand_cnt = dev_scratch + KB * kCandCap; pb.tau_out = reinterpret_cast<float *>(dev_scratch + KB * kCandCap + KB);
                pb.overflow = dev_scratch + KB * kCandCap + 2 * KB; pb.nflag = pb.overflow + 1; pb.flaglist = pb.overflow + 8;
                pb.skip_stat = nullptr;
                int *fl = dev_scratch + KB * kCandCap + 3 * KB;
                if (dev_overlap == 1) {
                    HIPCHK(hipEventRecord(dev_e1, s)); HIPCHK(hipStreamWaitEvent(dev_s2, dev_e1, 0));
                    launch_shortlist(pb, fl, dev_s2);
                    HIPCHK(hipEventRecord(dev_e2, dev_s2));
                    dev_side = true;
                } else {
                    launch_shortlist(pb, fl, s);
                }
            }
#endif
            {
                Timed t(h, "hull_qp", (double)(hi - lo) * h->B);
                launch_hull_select_qp(f, s);
            }
#ifdef CHB_DEV_KNOBS
            if (dev_side) HIPCHK(hipStreamWaitEvent(s, dev_e2, 0));
#endif
            {
                // the exact path for what the fused kernel left: cdist-rounded distances on both shortlists,
                // (distance, index) order, then the list-based hull kernel
                Timed t(h, "slow_path", 0.0);
                RescoreArgs ra{};
                ra.X = h->X.p; ra.Dp = h->Dp; ra.bq = h->bq_cur; ra.pos_begin = lo; ra.pos_end = hi;
                ra.B = h->B; ra.m = h->m; ra.Kcap = h->Kcap;
                ra.cand = h->cand.p; ra.cand_cnt = h->cand_cnt.p; ra.cand_cap = kCandCap;
                ra.cand2 = h->candu[cur].p; ra.cand2_cnt = h->candu_cnt[cur].p; ra.cand2_cap = kCandCapU;
                ra.active = h->slow.p; ra.n_active = h->n_slow.p;
                ra.out = h->L1();
                launch_rescore(ra, s);
                QpArgs q{};
                q.X = h->X.p; q.D = h->D; q.Dp = h->Dp; q.bq = h->bq_cur; q.pos_begin = lo; q.pos_end = hi;
                q.B = h->B; q.m = h->m; q.Kcap = h->Kcap; q.lists = h->L1(); q.dist = h->dist.p;
                q.prev = Lists{nullptr, nullptr, nullptr};
                q.metric = h->metric;
                q.active = h->slow.p; q.n_active = h->n_slow.p;
                launch_hull_qp(q, s);
            }
        } else {
        if (h->pf_fit && h->pf_update && h->cand.p) {
            // batch members that can displace an entry of the base list: fp16 shortlist against
            // the exact m-th distance, exact rescoring seeded with the base list
            // (fit rounds only produce the "earlier" / "later" eligibility codes, which have the affine
            // form the shortlist kernel evaluates; chb_topm_per_bin's "not equal" code stays on launch_topm)
            launch_pack_centered(h->X.p, h->D, h->Dp, h->memb2_id.p, h->memb2_code.p, h->bin_ptr2.p, h->B, 2 * h->K,
                                 h->centers.p, h->mu_g.p, h->shadow_scale, h->Dz, h->pk2.view(), s);
            ShortlistArgs pa{};
            pa.Gs = h->Gs.p; pa.gq = reinterpret_cast<const float2 *>(h->gq.p);
            pa.qn = reinterpret_cast<const float2 *>(h->qn.p);   // (the fit's table)
            pa.P = h->pk2.view(); pa.Dz = h->Dz; pa.S = h->shadow_scale;
            pa.bq = h->bq_cur; pa.pos_begin = lo; pa.pos_end = hi;
            pa.bin_ptr = h->bin_ptr2.p; pa.memb_id = h->memb2_id.p; pa.update = true;
            pa.seed = h->L0();
            pa.B = h->B; pa.m = h->m; pa.Kcap = h->Kcap;
            pa.cand = h->cand.p; pa.cand_cnt = h->cand_cnt.p; pa.cand_cap = kCandCap; pa.overflow = h->overflow.p;
            {
                Timed t(h, "prefilter_update", (double)(hi - lo) * h->hint_batch_entries);
                pa.flaglist = h->flaglist.p; pa.nflag = h->nflag.p;   // (counter reset by the CSR scan / the batch CSR kernel)
            launch_shortlist(pa, h->flags64.p, s);
                // the (position, bin) pairs with a non-empty shortlist, for rescore_kernel
                launch_compact_active(h->cand_cnt.p, lo, hi, h->B, h->Kcap, h->act_blk.p, h->active.p,
                                      h->n_active.p, s);
            }
            RescoreArgs ra{};
            ra.X = h->X.p; ra.Dp = h->Dp; ra.bq = h->bq_cur; ra.pos_begin = lo; ra.pos_end = hi;
            ra.B = h->B; ra.m = h->m; ra.Kcap = h->Kcap;
            ra.cand = h->cand.p; ra.cand_cnt = h->cand_cnt.p; ra.cand_cap = kCandCap;
            ra.active = h->active.p; ra.n_active = h->n_active.p;
            ra.in = h->L0(); ra.out = h->Lcur();
            // pairs without any candidate keep the base list
            {
                const size_t nl = (size_t)h->Kcap * h->B;
                const Lists dst = h->Lcur();
                HIPCHK(hipMemcpyAsync(dst.d, h->l0d.p, sizeof(double) * nl * h->m, hipMemcpyDeviceToDevice, s));
                HIPCHK(hipMemcpyAsync(dst.idx, h->l0i.p, sizeof(int) * nl * h->m, hipMemcpyDeviceToDevice, s));
                HIPCHK(hipMemcpyAsync(dst.cnt, h->l0c.p, sizeof(int) * nl, hipMemcpyDeviceToDevice, s));
            }
            {
                Timed t(h, "rescore_update", (double)(hi - lo) * h->B);
                launch_rescore(ra, s);
            }
            {
                Timed t(h, "topm_fallback", 0.0);
                launch_topm_flagged(a, h->flags64.p, h->flaglist.p, h->nflag.p, s);
            }
        } else {
            Timed t(h, "topm_update", (double)(hi - lo) * h->hint_batch_entries);
            if (h->m > kMaxM) launch_topm_generic(a, s); else launch_topm(a, s);
        }
        QpArgs q{};
        q.X = h->X.p; q.D = h->D; q.Dp = h->Dp; q.bq = h->bq_cur; q.pos_begin = lo; q.pos_end = hi;
        q.B = h->B; q.m = h->m; q.Kcap = h->Kcap; q.lists = h->Lcur(); q.dist = h->dist.p;
        // a (position, bin) whose vertex list is the one of the previous round keeps its distance
        q.prev = h->round_in_batch > 0 ? h->Lprev() : Lists{nullptr, nullptr, nullptr};
        q.metric = h->metric;
        {
            Timed t(h, "hull_qp", (double)(hi - lo) * h->B);
            if (h->m > kMaxM) launch_hull_generic(q, s); else launch_hull_qp(q, s);
        }
        }   // list-based paths
        {
            Timed t(h, "argmin", (double)(hi - lo));
            launch_argmin(h->dist.p, h->lab_old.p, h->lab_prev.p, lo, hi, h->B, h->lab_new.p,
                          h->mind.p, h->want_margin ? h->mind2.p : nullptr, h->fc_cur, h->argmin_in_place, s);
        }
        h->stats[2] += (int64_t)(hi - lo) * h->B;
    }
    HIPCHK(hipGetLastError());
    h->stats[1] += 1;
    h->round_in_batch += 1;
    return CHB_OK;
}

int batch_commit_dev(chb_ctx *h, const int *final_dev)
{
    hipStream_t s = h->stream;
    if (h->pp_batch)
        // ... and the rows put back into the persistent pack (in place, or appended to the new bin), then full regions moved
        launch_pack_state_commit(h->pack_state(), h->pk.view(), h->X.p, h->D, h->Dp, h->bq_cur, h->K, h->labels.p, h->B,
                                 h->centers.p, h->mu_g.p, h->shadow_scale, h->Zs.p, h->Dz, h->ms.p, final_dev, h->lab_old.p,
                                 h->inb.p, s);
    else if (h->pf_fit && h->centers.p)
        // final labels out, batch marks cleared, and the members' shadow rows recomputed against their
        // new bin's centre: one launch
        launch_sample_shadow(h->X.p, h->D, h->Dp, h->bq_cur, h->K, h->labels.p, h->B, h->centers.p, h->mu_g.p,
                             h->shadow_scale, h->Zs.p, h->Dz, h->ms.p, final_dev, h->inb.p, s);
    else
        launch_batch_close(h->labels.p, h->inb.p, h->bq_cur, final_dev, h->K, s);
    if (h->pool_valid) {
        // (labels and shadow rows are final: holes resolved, the batch's arrivals offered to their new bins' pools)
        Timed t(h, "pool", (double)h->K);
        launch_pool_commit(h->pool_view(), h->Zs.p, h->ms.p, h->qn.p, h->D, h->Dz, h->bq_cur, h->K, final_dev, h->lab_old.p,
                           h->labels.p, h->B, h->pool_holes, s);
    }
    HIPCHK(hipGetLastError());
    h->batch_open = false;
    h->pp_batch = false;
    return CHB_OK;
}

// Under an exchange the ranks run ONE fit together: the order of its collectives is a function of the arguments, of the
// context's switches and of the tile-skipping memo -- so before the first batch every rank all-gathers what it was given
// and how it is set up.  Arguments (and the formulation the switches select) must be equal: a difference fails the call on
// EVERY rank with the same message instead of leaving some of them inside a collective.  What may legitimately differ --
// the memo a context keeps from earlier fits, the A/B switches of the look-ahead, the tile skipping and the persistent
// pack -- is settled by taking the most conservative value of all ranks for this fit.
struct FitAgree {
    static constexpr int W = 24, kEq = 16;
    int v[W];
};
const char *const kAgreeNames[FitAgree::kEq] = {
    "library", "num_clusters", "num_neighbors", "n_move", "max_iter", "batch size", "N", "D", "metric", "CHB_FUSED / fused path",
    "CHB_PREFILTER / shortlist stage", "min_dist_out", "perms", "perms", "initial_bins", "initial_bins"};

uint64_t hash_i64(const int64_t *p, int64_t n)
{
    // four independent multiply-xor lanes (the order of the elements matters, which is the point)
    uint64_t a = 0x9e3779b97f4a7c15ull, b = 0xc2b2ae3d27d4eb4full, c = 0x165667b19e3779f9ull, d = 0x27d4eb2f165667c5ull;
    int64_t i = 0;
    for (; i + 4 <= n; i += 4) {
        a = (a ^ (uint64_t)p[i]) * 0x100000001b3ull;     b = (b ^ (uint64_t)p[i + 1]) * 0x100000001b3ull;
        c = (c ^ (uint64_t)p[i + 2]) * 0x100000001b3ull; d = (d ^ (uint64_t)p[i + 3]) * 0x100000001b3ull;
    }
    for (; i < n; ++i) a = (a ^ (uint64_t)p[i]) * 0x100000001b3ull;
    uint64_t x = a ^ (b * 3) ^ (c * 5) ^ (d * 7);
    x ^= x >> 29; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 32;
    return x;
}

// mine: this rank's table; on success `mine` holds the agreed table (equal entries as given, entries >= kEq the minimum
// over the ranks)
int fit_agree(chb_ctx *h, FitAgree *mine)
{
    constexpr int W = FitAgree::W;
    const int world = h->world;
    HIPCHK(h->agree.ensure((size_t)W * world));
    std::vector<int> all((size_t)W * world, 0);
    memcpy(all.data() + (size_t)W * h->rank, mine->v, sizeof(int) * W);
    HIPCHK(hipMemcpyAsync(h->agree.p + (size_t)W * h->rank, mine->v, sizeof(int) * W, hipMemcpyHostToDevice, h->stream));
    { const int r_ = exchange_all_gather(h, h->agree.p, (size_t)W, sizeof(int), ncclInt32); if (r_) return r_; }
    HIPCHK(hipMemcpyAsync(all.data(), h->agree.p, sizeof(int) * all.size(), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int r = 0; r < world; ++r)
        for (int k = 0; k < FitAgree::kEq; ++k)
            if (all[(size_t)W * r + k] != all[k])   // (against rank 0's: every rank then reports the same pair)
                return fail(CHB_EINVAL, std::string("chb_fit_cluster: rank ") + std::to_string(r) + " and rank 0 differ in `" +
                                        kAgreeNames[k] + "` -- every rank of the communicator must make the same call on the "
                                        "same data with the same switches");
    for (int k = FitAgree::kEq; k < W; ++k) {
        int mn = all[k];
        for (int r = 1; r < world; ++r) mn = std::min(mn, all[(size_t)W * r + k]);
        mine->v[k] = mn;
    }
    return CHB_OK;
}

std::vector<int> to_i32(const int64_t *p, size_t n)
{
    std::vector<int> v(n);
    for (size_t i = 0; i < n; ++i) v[i] = (int)p[i];
    return v;
}

}  // namespace

extern "C" {

const char *chb_last_error(void) { return g_err.c_str(); }
int chb_version(void) { return 1; }

int chb_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int chb_create(int device_id, chb_ctx **out)
{
    if (!out) return fail(CHB_EINVAL, "out is null");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(CHB_ENODEVICE, "no HIP device visible (this library has no CPU fallback)");
    if (device_id < 0 || device_id >= n) return fail(CHB_EINVAL, "device_id out of range");
    HIPCHK(hipSetDevice(device_id));
    chb_ctx *h = new chb_ctx();
    h->dev = device_id;
    if (const char *e = getenv("CHB_PREFILTER")) h->use_prefilter = atoi(e) != 0;
    if (const char *e = getenv("CHB_FORCE_GATHER")) h->force_gather = atoi(e) != 0;
    // CHB_FUSED=0: m <= 16 also takes the list-based path (exact rescoring of every shortlist, then the hull
    // kernel); like CHB_PREFILTER=0 a switch to the slower, independent formulation for the tests' A/B checks
    if (const char *e = getenv("CHB_FUSED")) h->allow_fused = atoi(e) != 0;
#ifdef CHB_DEV_KNOBS   // developer builds only (tools/): the product library reads no tuning knob
    if (const char *e = getenv("CHB_PF_BASE")) h->pf_base = atoi(e) != 0;
    if (const char *e = getenv("CHB_PF_UPDATE")) h->pf_update = atoi(e) != 0;
#endif
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipHostMalloc((void **)&h->fc_host, 128, hipHostMallocDefault);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&h->fc_event[i], hipEventDisableTiming);
    if (const char *ev = getenv("CHB_SPECULATE")) h->speculate = atoi(ev) != 0;
    if (const char *ev = getenv("CHB_SEGMENTS")) h->allow_segments = atoi(ev) != 0;
    if (const char *ev = getenv("CHB_TILE_SKIP")) h->allow_skip = atoi(ev) != 0;
    if (const char *ev = getenv("CHB_PACK_INCR")) h->pp_allowed = atoi(ev) != 0;
    if (const char *ev = getenv("CHB_POOL_TAU")) { h->pool_allowed = atoi(ev) != 0; h->pool_force = atoi(ev) == 2; }
    if (e != hipSuccess) { delete h; return fail(CHB_EHIP, hipGetErrorString(e)); }
    *out = h;
    return CHB_OK;
}

int chb_destroy(chb_ctx *h)
{
    if (!h) return CHB_OK;
    (void)hipSetDevice(h->dev);
    (void)hipStreamSynchronize(h->stream);
    drain_profile(h);
    if (h->comm && rccl()) { (void)rccl()->CommDestroy(h->comm); h->comm = nullptr; }
    DevBuf<int> *ib[] = {&h->labels, &h->inb, &h->bq, &h->lab_old, &h->lab_prev, &h->lab_new,
                         &h->first_change, &h->l0i, &h->l1i, &h->l0c, &h->l1c, &h->l2i, &h->l2c, &h->cnt, &h->bin_ptr,
                         &h->cursor, &h->memb_id, &h->cnt2, &h->bin_ptr2, &h->cursor2, &h->memb2_id,
                         &h->memb2_code, &h->perm, &h->xq, &h->xhull, &h->xcnt, &h->cand, &h->cand_cnt, &h->flags64, &h->flaglist, &h->nflag, &h->overflow, &h->flaglist2, &h->nflag2};
    for (auto *b : ib) b->release();
    DevBuf<double> *db[] = {&h->X, &h->mind, &h->mind2, &h->dist, &h->l0d, &h->l1d, &h->l2d, &h->xdist, &h->xalpha, &h->xpts};
    for (auto *b : db) b->release();
    h->Gs.release(); h->Zs.release(); h->gq.release(); h->ms.release(); h->mu_g.release();
    h->colsum_part.release(); h->rmax.release(); h->pk.release(); h->pk2.release(); h->qn.release();
    h->centers.release();
    h->active.release(); h->n_active.release(); h->act_blk.release();
    for (int i = 0; i < 2; ++i) { h->candu[i].release(); h->candu_cnt[i].release(); }
    h->slow.release(); h->n_slow.release(); h->tau.release(); h->short_cnt.release(); h->agree.release();
    h->xg.release(); h->xerr.release();
    h->pool_Z.release(); h->pool_id.release(); h->pool_hole.release(); h->pool_ok.release(); h->pool_stat.release();
    h->pool_key.release(); h->pool_sn.release(); h->pool_tsn.release();
    h->pp_start.release(); h->pp_cap.release(); h->pp_fill.release(); h->pp_live.release(); h->pp_nt.release();
    h->pp_memb.release(); h->pp_row.release(); h->pp_ctl.release(); h->pp_ovf.release(); h->pp_dest.release();
    h->seg_nseg.release(); h->seg_gflag.release(); h->seg_items.release(); h->seg_lists.release();
    h->shell_inv.release(); h->ckey.release(); h->qord.release(); h->home.release();
    h->qord_all.release(); h->home_all.release(); h->geo_all.release();
    (void)hipStreamDestroy(h->stream);
    if (h->fc_host) (void)hipHostFree(h->fc_host);
    for (int i = 0; i < 2; ++i) if (h->fc_event[i]) (void)hipEventDestroy(h->fc_event[i]);
    delete h;
    return CHB_OK;
}

// the resident copy X[N][Dp] (rows zero-padded to Dp): filled from `X` (host or device), or -- X == nullptr -- left to
// be filled by the caller (chb_bcast_samples on a receiving rank)
static int samples_upload(chb_ctx *h, const double *X, int64_t N, int64_t D, bool from_device)
{
    if (N <= 0 || D <= 0) return fail(CHB_EINVAL, "samples must be a non-empty N x D matrix");
    if (N >= (1LL << 31) - 64 || D > (1 << 20)) return fail(CHB_EUNSUPPORTED, "N or D too large");
    HIPCHK(hipSetDevice(h->dev));
    // rows padded to whole 128-byte lines (16 doubles): a 16-lane group of the hull kernels reads a row in 256-byte pieces
    // from its start, and a row that starts in the middle of a line makes every piece touch three lines instead of two
#ifndef CHB_ROW_PAD
#define CHB_ROW_PAD 16
#endif
    static_assert(CHB_ROW_PAD % kKChunk == 0, "the tile kernels stage kKChunk columns at a time");
    const int Dp = (int)((D + CHB_ROW_PAD - 1) / CHB_ROW_PAD) * CHB_ROW_PAD;
    HIPCHK(h->X.ensure((size_t)N * Dp));
    if (X != nullptr) {
        if (Dp != D) HIPCHK(hipMemsetAsync(h->X.p, 0, sizeof(double) * (size_t)N * Dp, h->stream));
        HIPCHK(hipMemcpy2DAsync(h->X.p, sizeof(double) * Dp, X, sizeof(double) * D, sizeof(double) * D,
                                (size_t)N, from_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                                h->stream));
    }
    h->N = N; h->D = (int)D; h->Dp = Dp;
    h->fit_open = false; h->batch_open = false;
    return CHB_OK;
}

// everything that is a function of the resident X alone (global mean, scale, query-side shadow rows)
static int samples_finish(chb_ctx *h)
{
    h->skip_off_key = -1; h->pool_off_key = -1;
    const int64_t N = h->N, D = h->D;
    const int Dp = h->Dp;
    // the shortlist stage (prefilter_kernels.hip) keeps its query fragments in registers: D <= 157 in the narrow builds,
    // D <= 573 as two to four 144-column slices in the wide ones (shadow_row_elems)
    h->shadow_ok = false;
    const int Dz = shadow_row_elems((int)D);
    if (h->use_prefilter && Dz > 0) {
        // global mean, the power-of-two scale that puts every centred feature inside +-2^11, and the
        // query-side shadow row of every sample: functions of X alone
        const int part_blocks = 1024;
        HIPCHK(h->mu_g.ensure((size_t)Dp));
        HIPCHK(h->colsum_part.ensure((size_t)part_blocks * Dp));
        HIPCHK(h->rmax.ensure(1));
        HIPCHK(h->Gs.ensure((size_t)N * Dz));
        HIPCHK(h->gq.ensure((size_t)N * 2));
        HIPCHK(h->Zs.ensure((size_t)N * Dz));
        HIPCHK(h->ms.ensure((size_t)N * 4));
        launch_global_center(h->X.p, (int)N, (int)D, Dp, h->colsum_part.p, part_blocks, h->mu_g.p, h->rmax.p,
                             h->stream);
        unsigned int rbits = 0;
        HIPCHK(hipMemcpyAsync(&rbits, h->rmax.p, sizeof(rbits), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        float R;
        memcpy(&R, &rbits, sizeof(R));
        double S = 1.0;
        if (R > 0.f && std::isfinite(R)) {
            int e = 0;
            (void)std::frexp((double)R, &e);   // R < 2^e
            S = std::ldexp(1.0, 10 - e);       // |x - mu_c| S <= 2 R S < 2^11: every member's -bias / 2 then fits the three
                                               // fp16 bias columns of its shadow row (prefilter_kernels.hip, kBiasExp)
            if (Dz > 160) S *= 0.5;            // wide rows (D <= 573): |x - mu_c| S < 2^10, -bias / 2 <= D 2^21 / 2 < 2^29.2
        }
        h->shadow_scale = S;
        launch_global_shadow(h->X.p, (int)N, (int)D, Dp, h->mu_g.p, S, h->Gs.p, Dz, h->gq.p, h->stream);
        HIPCHK(hipGetLastError());
        h->Dz = Dz;
        h->shadow_ok = true;
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    return CHB_OK;
}

static int set_samples_common(chb_ctx *h, const double *X, int64_t N, int64_t D, bool from_device)
{
    if (!h) return fail(CHB_EINVAL, "null context");
    if (!X) return fail(CHB_EINVAL, "samples must be a non-empty N x D matrix");
    const int rc = samples_upload(h, X, N, D, from_device);
    return rc ? rc : samples_finish(h);
}

int chb_set_samples(chb_ctx *h, const double *X, int64_t N, int64_t D)
{
    return set_samples_common(h, X, N, D, false);
}

int chb_set_samples_device(chb_ctx *h, const double *X, int64_t N, int64_t D)
{
    return set_samples_common(h, X, N, D, true);
}

int chb_bcast_samples(chb_ctx *h, const double *X, int64_t N, int64_t D, int root)
{
    if (!h) return fail(CHB_EINVAL, "null context");
    if (!h->comm) return fail(CHB_ESTATE, "chb_bcast_samples needs chb_comm_init (RCCL)");
    if (root < 0 || root >= h->world) return fail(CHB_EINVAL, "bad root");
    // Every rank reports {its own status, N, D, root} BEFORE the collective: a rank that returned early (no matrix on the
    // root, an allocation that failed) or ranks that disagree about the shape would otherwise leave the others blocked in
    // ncclBroadcast for good, or broadcast into buffers of the wrong size.
    int rc = (h->rank == root && !X) ? fail(CHB_EINVAL, "the root rank must pass the matrix")
                                     : samples_upload(h, h->rank == root ? X : nullptr, N, D, false);
    {
        const std::string my_err = g_err;
        const int world = h->world;
        HIPCHK(h->agree.ensure((size_t)4 * world));
        std::vector<int> all((size_t)4 * world, 0);
        int *mine = all.data() + 4 * h->rank;
        mine[0] = rc; mine[1] = (int)(N & 0x7fffffff); mine[2] = (int)(D & 0x7fffffff); mine[3] = root;
        HIPCHK(hipMemcpyAsync(h->agree.p + 4 * h->rank, mine, 4 * sizeof(int), hipMemcpyHostToDevice, h->stream));
        NCCLCHK(rccl()->AllGather(h->agree.p + 4 * h->rank, h->agree.p, 4, ncclInt32, h->comm, h->stream));
        HIPCHK(hipMemcpyAsync(all.data(), h->agree.p, sizeof(int) * all.size(), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        if (rc) return fail(rc, my_err);
        for (int r = 0; r < world; ++r) {
            const int *o = all.data() + 4 * r;
            if (o[0] != 0)
                return fail(CHB_ESTATE, "chb_bcast_samples: rank " + std::to_string(r) + " failed before the broadcast (status " +
                                        std::to_string(o[0]) + ")");
            if (o[1] != mine[1] || o[2] != mine[2] || o[3] != mine[3])
                return fail(CHB_EINVAL, "chb_bcast_samples: rank " + std::to_string(r) + " passed a different N, D or root");
        }
    }
    // the padded resident copy goes out as it lies on the root: one RCCL broadcast over xGMI
    NCCLCHK(rccl()->Broadcast(h->X.p, h->X.p, (size_t)N * (size_t)h->Dp, ncclDouble, root, h->comm, h->stream));
    return samples_finish(h);
}

int chb_comm_info(chb_ctx *h, int *rank, int *world, int *comm_ranks, int *transport)
{
    if (!h) return fail(CHB_EINVAL, "null context");
    if (rank) *rank = h->rank;
    if (world) *world = h->world;
    int cnt = 0;
    if (h->comm && rccl()) NCCLCHK(rccl()->CommCount(h->comm, &cnt));
    if (comm_ranks) *comm_ranks = cnt;
    if (transport) *transport = h->comm ? 1 : (h->hook ? 2 : 0);
    return CHB_OK;
}

int chb_pairwise_distance(chb_ctx *h, int64_t r0, int64_t r1, double *out)
{
    if (!h || !out) return fail(CHB_EINVAL, "null argument");
    if (!h->X.p) return fail(CHB_ESTATE, "chb_set_samples has not been called");
    if (r0 < 0 || r1 > h->N || r0 > r1) return fail(CHB_EINVAL, "row range out of bounds");
    HIPCHK(hipSetDevice(h->dev));
    const int64_t chunk = std::max<int64_t>(64, (int64_t)(1LL << 28) / std::max<int64_t>(h->N, 1));
    DevBuf<double> buf;
    HIPCHK(buf.ensure((size_t)std::min(chunk, r1 - r0) * h->N));
    for (int64_t a = r0; a < r1; a += chunk) {
        const int64_t b = std::min(r1, a + chunk);
        {
            Timed t(h, "pairwise", (double)(b - a) * h->N);
            launch_pairwise(h->X.p, (int)h->N, h->Dp, (int)a, (int)b, buf.p, h->stream);
        }
        hipError_t e = hipGetLastError();
        if (e == hipSuccess)
            e = hipMemcpyAsync(out + (a - r0) * h->N, buf.p, sizeof(double) * (b - a) * h->N,
                               hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) { buf.release(); return fail(CHB_EHIP, hipGetErrorString(e)); }
    }
    buf.release();
    return CHB_OK;
}

int chb_fit_begin(chb_ctx *h, int64_t B, const int64_t *initial_bins, int m)
{
    if (!h || !initial_bins) return fail(CHB_EINVAL, "null argument");
    HIPCHK(hipSetDevice(h->dev));
    const int rc = fit_begin_impl(h, B, initial_bins, m);
    // (the stepwise batches have no loop that reads the skip statistics and could turn the tile-skipping builds off
    //  where they do not pay: they run the ordinary builds)
    if (rc == CHB_OK) h->skip_state = -1;
    return rc;
}

int chb_batch_begin(chb_ctx *h, const int64_t *perm_slice, int64_t K, int64_t q_lo, int64_t q_hi)
{
    if (!h || !perm_slice) return fail(CHB_EINVAL, "null argument");
    if (!h->fit_open) return fail(CHB_ESTATE, "chb_fit_begin has not been called");
    if (h->batch_open) return fail(CHB_ESTATE, "previous batch not committed");
    if (K <= 0 || K > (1 << 24) || q_lo < 0 || q_hi > K || q_lo > q_hi)
        return fail(CHB_EINVAL, "bad batch geometry");
    HIPCHK(hipSetDevice(h->dev));
    {
        // (the batch start counts every labelled sample and subtracts the batch's own entries: a sample listed twice
        //  would be subtracted twice)
        std::vector<uint64_t> seen((size_t)(h->N + 63) / 64, 0);
        for (int64_t i = 0; i < K; ++i) {
            const int64_t v = perm_slice[i];
            if (v < 0 || v >= h->N) return fail(CHB_EINVAL, "perm entry out of range");
            uint64_t &wd = seen[(size_t)(v >> 6)];
            if (wd & (1ull << (v & 63))) return fail(CHB_EINVAL, "a batch lists a sample twice");
            wd |= 1ull << (v & 63);
        }
    }
    if ((int)K > h->Kcap) { int rc = ensure_batch_buffers(h, (int)K); if (rc) return rc; }
    std::vector<int> v = to_i32(perm_slice, (size_t)K);
    h->bq_cur = h->bq.p;
    HIPCHK(hipMemcpyAsync(h->bq_cur, v.data(), sizeof(int) * K, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return batch_begin_dev(h, (int)K, (int)q_lo, (int)q_hi, false);
}

int chb_batch_round(chb_ctx *h, const int64_t *lab_prev, int64_t active, int64_t *lab_new,
                    double *min_dist)
{
    if (!h || !lab_prev || !lab_new) return fail(CHB_EINVAL, "null argument");
    if (!h->batch_open) return fail(CHB_ESTATE, "no open batch");
    HIPCHK(hipSetDevice(h->dev));
    const int K = h->K;
    std::vector<int> v = to_i32(lab_prev, (size_t)K);
    HIPCHK(hipMemcpyAsync(h->lab_prev.p, v.data(), sizeof(int) * K, hipMemcpyHostToDevice, h->stream));
    h->argmin_in_place = false;
    int rc = batch_round_dev(h, (int)active);
    if (rc) return rc;
    const int lo = std::max((int)active, h->q_lo), hi = h->q_hi;
    if (hi > lo) {
        std::vector<int> ln((size_t)(hi - lo));
        HIPCHK(hipMemcpyAsync(ln.data(), h->lab_new.p + lo, sizeof(int) * (hi - lo), hipMemcpyDeviceToHost, h->stream));
        if (min_dist)
            HIPCHK(hipMemcpyAsync(min_dist + lo, h->mind.p + lo, sizeof(double) * (hi - lo), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        for (int i = lo; i < hi; ++i) lab_new[i] = ln[(size_t)(i - lo)];
    } else {
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return CHB_OK;
}

int chb_batch_guess(chb_ctx *h, int64_t *guess)
{
    if (!h || !guess) return fail(CHB_EINVAL, "null argument");
    if (!h->batch_open) return fail(CHB_ESTATE, "no open batch");
    HIPCHK(hipSetDevice(h->dev));
    const int lo = h->q_lo, hi = h->q_hi;
    if (hi <= lo) return CHB_OK;
    if (h->fused && !h->lists_valid) launch_guess_near(h->tau.p, h->lab_old.p, lo, hi, h->B, h->Kcap, h->lab_prev.p, h->stream);
    else launch_guess(h->l0d.p, h->l0c.p, h->lab_old.p, lo, hi, h->B, h->m, h->Kcap, h->lab_prev.p, h->stream);
    HIPCHK(hipGetLastError());
    std::vector<int> g((size_t)(hi - lo));
    HIPCHK(hipMemcpyAsync(g.data(), h->lab_prev.p + lo, sizeof(int) * (hi - lo), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int i = lo; i < hi; ++i) guess[i] = g[(size_t)(i - lo)];
    return CHB_OK;
}

int chb_batch_commit(chb_ctx *h, const int64_t *final_labels)
{
    if (!h || !final_labels) return fail(CHB_EINVAL, "null argument");
    if (!h->batch_open) return fail(CHB_ESTATE, "no open batch");
    HIPCHK(hipSetDevice(h->dev));
    std::vector<int> v = to_i32(final_labels, (size_t)h->K);
    HIPCHK(hipMemcpyAsync(h->lab_prev.p, v.data(), sizeof(int) * h->K, hipMemcpyHostToDevice, h->stream));
    int rc = batch_commit_dev(h, h->lab_prev.p);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    return CHB_OK;
}

int chb_fit_labels(chb_ctx *h, int64_t *labels_out)
{
    if (!h || !labels_out) return fail(CHB_EINVAL, "null argument");
    if (!h->fit_open) return fail(CHB_ESTATE, "chb_fit_begin has not been called");
    HIPCHK(hipSetDevice(h->dev));
    std::vector<int> v((size_t)h->N);
    HIPCHK(hipMemcpyAsync(v.data(), h->labels.p, sizeof(int) * h->N, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int64_t i = 0; i < h->N; ++i) labels_out[i] = v[(size_t)i];
    return CHB_OK;
}

int chb_fit_cluster(chb_ctx *h, int64_t B, const int64_t *initial_bins, const int64_t *perms,
                    int64_t n_move, int m, int max_iter, int batch, int64_t *labels_out,
                    int *iters_run, int64_t *changed_per_iter, double *min_dist_out)
{
    return chb_fit_cluster_ex(h, B, initial_bins, perms, n_move, m, max_iter, batch, labels_out, iters_run,
                              changed_per_iter, min_dist_out, nullptr);
}

int chb_fit_cluster_ex(chb_ctx *h, int64_t B, const int64_t *initial_bins, const int64_t *perms,
                       int64_t n_move, int m, int max_iter, int batch, int64_t *labels_out,
                       int *iters_run, int64_t *changed_per_iter, double *min_dist_out, double *margin_out)
{
    if (!h || !initial_bins || !labels_out) return fail(CHB_EINVAL, "null argument");
    if (margin_out && !min_dist_out) return fail(CHB_EINVAL, "margin_out needs min_dist_out");
    if (margin_out && h->world > 1) return fail(CHB_EUNSUPPORTED, "margin report is single-GPU");
    h->want_margin = margin_out != nullptr;
    struct MarginOff { chb_ctx *h; ~MarginOff() { h->want_margin = false; } } margin_off{h};
    if (n_move > 0 && !perms) return fail(CHB_EINVAL, "perms is null");
    if (max_iter < 0 || n_move < 0 || n_move > h->N) return fail(CHB_EINVAL, "bad n_move/max_iter");
    HIPCHK(hipSetDevice(h->dev));
    if (h->world > 1 && !h->comm && !h->hook) return fail(CHB_ESTATE, "world > 1 but chb_comm_init was not called");
    if (!h->X.p) return fail(CHB_ESTATE, "chb_set_samples has not been called");
    // every permutation entry is range-checked BEFORE anything runs (a min / max pass the compiler vectorises), so a
    // bad entry in a late sweep cannot surface after earlier sweeps have already run; duplicates inside a sweep
    // are rejected while that sweep is converted for its upload (bitmap), and any error return closes the fit
    {
        int64_t lo = 0, hi = 0;
        const int64_t tot = (int64_t)max_iter * n_move;
        for (int64_t i = 0; i < tot; ++i) { lo = std::min(lo, perms[i]); hi = std::max(hi, perms[i]); }
        if (lo < 0 || hi >= h->N) return fail(CHB_EINVAL, "perm entry out of range");
    }
    int rc = fit_begin_impl(h, B, initial_bins, m, /*sync=*/false);
    if (rc) { (void)hipStreamSynchronize(h->stream); return rc; }   // (an upload from pin_a may be in flight: the next call rewrites it)
    struct FitCloser {   // an error return must not leave an open fit / batch behind
        chb_ctx *h; bool ok = false;
        ~FitCloser() { if (!ok) { (void)hipStreamSynchronize(h->stream); h->fit_open = false; h->batch_open = false; } }
    } fit_closer{h};
    const int64_t N = h->N;
    h->pp_fit = true; h->stats_pp_batches = 0; h->stats_pp_builds = 0;
    struct PackOff {
        chb_ctx *h;
        ~PackOff() { h->pp_fit = false; h->pp_valid = false; h->pool_fit = false; h->pool_valid = false; h->qord_cur = nullptr; h->home_cur = nullptr; }
    } pack_off{h};
    // threshold pools: built from the initial labels (the CSR of the fit's start is still in place), unless an earlier fit
    // over the same samples, bins and neighbour count found that they do not pay (overlapping bins: long shortlists)
    h->pool_fit = true; h->stats_pool_batches = 0;
    h->pool_state = (h->pool_off_key == skip_key(h)) ? -1 : 0;
    h->pool_batches = 0; h->pool_cand = 0; h->pool_pairs = 0;
    if (h->pool_state >= 0) { rc = pool_build(h); if (rc) return rc; }
    std::vector<uint64_t> seen_bits;
    // default batch: 8192 positions on one GPU, growing with sqrt(world): the batch-member
    // (update) work per rank is ~K^2/world, the per-rank grids ~K/world.  Twice that from 300k contigs to move: every
    // batch rebuilds the CSR and the padded pack of ALL labelled contigs (cost ~ N per batch, ~ N^2 / K per sweep), while a
    // batch is a smaller share of the sweep and collides with itself no more often (measured: 115 against 124 ms per sweep
    // at 500k x 140 x 128, 517 against 538 at 1M x 146 x 200; 100k contigs are best served by 8192)
    // (round 5, with the pools' cheaper shortlist launches: four times from 750k contigs -- 1M x 146 x 200: 342 against 361 ms per
    //  sweep; 500k x 140 x 128 stays best at twice: 105.5 against 108.5)
    int Kmax = batch > 0 ? batch : 8192 * std::max(1, (int)std::lround(std::sqrt((double)h->world))) *
                                       ((n_move >= 750000 && h->world == 1) ? 4 : (n_move >= 300000 ? 2 : 1));   // (sharded: as before)
    if (m > kMaxM && batch <= 0) Kmax = std::min(Kmax, 512);   // the plain kernels: one wavefront per (contig, bin)
    if (Kmax > n_move) Kmax = (int)std::max<int64_t>(n_move, 1);
    h->last_batch = Kmax;
    rc = ensure_batch_buffers(h, Kmax);
    if (rc) return rc;
    hipStream_t s = h->stream;
    memset(h->stats, 0, sizeof(h->stats));
    h->stats_seg_batches = 0;
    h->stats_lookahead = 0; h->stats_lookahead_failed = 0;

    // ---- more than one rank: agree on the fit before its first collective (fit_agree above)
    const bool xchg_fit = (h->comm != nullptr || h->hook != nullptr) && (h->world > 1 || h->force_gather);
    struct SwitchRestore {   // (the agreed switches hold for this fit only)
        chb_ctx *h; bool skip, pack, spec;
        ~SwitchRestore() { h->allow_skip = skip; h->pp_allowed = pack; h->speculate = spec; }
    } switch_restore{h, h->allow_skip, h->pp_allowed, h->speculate};
    h->xseq = 0;
    if (xchg_fit) {
        FitAgree fa{};
        const uint64_t hp = hash_i64(perms, (int64_t)max_iter * n_move), hi_ = hash_i64(initial_bins, h->N);
        const int eq[FitAgree::kEq] = {0x43480005, (int)B, m, (int)(n_move & 0x7fffffff), max_iter, Kmax, (int)(h->N & 0x7fffffff),
                                       h->D, h->metric, h->fused ? 1 : 0, h->pf_fit ? 1 : 0, min_dist_out ? 1 : 0,
                                       (int)(hp & 0x7fffffff), (int)((hp >> 32) & 0x7fffffff), (int)(hi_ & 0x7fffffff),
                                       (int)((hi_ >> 32) & 0x7fffffff)};
        memcpy(fa.v, eq, sizeof(eq));
        fa.v[16] = h->skip_state; fa.v[17] = h->speculate ? 1 : 0; fa.v[18] = h->allow_skip ? 1 : 0; fa.v[19] = h->pp_allowed ? 1 : 0;
        fa.v[20] = h->pool_valid ? h->pool_state : -1;
        rc = fit_agree(h, &fa);
        if (rc) return rc;
        h->skip_state = fa.v[16]; h->speculate = fa.v[17] != 0; h->allow_skip = fa.v[18] != 0; h->pp_allowed = fa.v[19] != 0;
        if (fa.v[20] < 0) { h->pool_state = -1; h->pool_valid = false; }
    }
#ifdef CHB_DEV_KNOBS
    // developer builds, tests of the exchange schedule (tests/test_gpu_world2.py): CHB_DEV_HOOK_SPEC=1 lets the look-ahead run
    // over the host-staged hook (its exchanges synchronise the stream, so nothing is gained -- but the ORDER of the exchanges
    // is the RCCL path's); CHB_DEV_SKIP_STATS=<skipped>,<seen>,<unloaded> replaces this rank's tile-skipping statistics of
    // every batch; CHB_DEV_LOCAL_VERDICT=1 restores the behaviour before round 5 (every rank decides from its OWN statistics)
    const bool dev_hook_spec = getenv("CHB_DEV_HOOK_SPEC") != nullptr && atoi(getenv("CHB_DEV_HOOK_SPEC")) != 0;
    const bool dev_local_verdict = getenv("CHB_DEV_LOCAL_VERDICT") != nullptr && atoi(getenv("CHB_DEV_LOCAL_VERDICT")) != 0;
    int dev_stats[3] = {0, 0, 0};
    const bool dev_stats_on = getenv("CHB_DEV_SKIP_STATS") != nullptr &&
                              sscanf(getenv("CHB_DEV_SKIP_STATS"), "%d,%d,%d", &dev_stats[0], &dev_stats[1], &dev_stats[2]) == 3;
#else
    const bool dev_hook_spec = false, dev_local_verdict = false;
#endif

    // (fit_begin_impl left the converted initial labels in pin_a; their upload and the kernels of the fit's start are
    //  still running -- nothing below touches pin_a again before the sweep's final synchronisation)
    std::vector<int> prev(h->pin_a.p, h->pin_a.p + N);
    std::vector<double> mind_host, mind2_host;
    if (min_dist_out) {
        for (int64_t i = 0; i < N; ++i) min_dist_out[i] = NAN;
        mind_host.resize((size_t)Kmax);
    }
    if (margin_out) {
        for (int64_t i = 0; i < N; ++i) margin_out[i] = NAN;
        mind2_host.resize((size_t)Kmax);
    }
    int64_t assigned0 = 0;
    for (int64_t i = 0; i < N; ++i) assigned0 += prev[(size_t)i] >= 0;
    int64_t labelled = assigned0;

    int it = 0;
    bool wrote_out = false;
    bool sweep_has_labelled = true;
    for (; it < max_iter; ++it) {
        const int64_t *perm = perms + (int64_t)it * n_move;
        HIPCHK(h->perm.ensure((size_t)std::max<int64_t>(n_move, 1)));
        HIPCHK(h->pin_c.ensure((size_t)std::max<int64_t>(n_move, 1)));
        if (n_move) {
            // (pin_c, the permutations' own staging buffer: the previous sweep's upload from it completed before that
            //  sweep's final synchronisation, and the first sweep's conversion overlaps the kernels of the fit's start)
            seen_bits.assign((size_t)(N + 63) / 64, 0);
            uint64_t dup = 0;
            int any_lab = 0;   // does this sweep visit a sample that carries a label?  (sweep 1 normally does not)
            for (int64_t i = 0; i < n_move; ++i) {
                const int64_t v = perm[i];          // (in range: checked up front)
                uint64_t &wd = seen_bits[(size_t)(v >> 6)];
                const uint64_t bit = 1ull << (v & 63);
                dup |= wd & bit;
                wd |= bit;
                h->pin_c.p[i] = (int)v;
                any_lab |= prev[(size_t)v] >= 0;
            }
            sweep_has_labelled = any_lab != 0;
            if (dup) return fail(CHB_EINVAL, "a sweep's permutation lists a sample twice");
            HIPCHK(hipMemcpyAsync(h->perm.p, h->pin_c.p, sizeof(int) * n_move, hipMemcpyHostToDevice, s));
        }
        // ---- the batches of this sweep.  A batch = start (selection against the members outside it), a
        // label guess, then rounds until the first changed position is past its end.  On one GPU the
        // NEXT batch is enqueued while the first round's verdict is still on its way to the host: its
        // kernels (and this batch's commit) carry a gate on that verdict and return at once if the round
        // did not converge, in which case the remaining rounds run and the next batch is enqueued again.
        // The stream never drains while rounds converge at once -- the common case after sweep 1's start;
        // a failed guess switches the look-ahead off until a batch converges in one round again.
        const int world = h->world;
        const bool xchg = (h->comm != nullptr || h->hook != nullptr) && (world > 1 || h->force_gather);
        // (look-ahead under an exchange: the RCCL all-gather sits on the context's stream, so the first-changed position of
        //  a round is computed on the device right behind it and feeds the same gate as on one GPU; every rank sees the same
        //  labels, hence the same verdict, and the all-gathers of a gated-off batch move identical bytes between the ranks'
        //  identical buffers.  The hook transport needs the host between rounds anyway.)
        const bool can_spec = h->speculate && h->fused && (!xchg || (h->hook == nullptr && h->comm != nullptr) || dev_hook_spec) &&
                              min_dist_out == nullptr;
        // the tag of the fit's next exchange (kind 1: a batch's label guess, 2: a round's labels)
        auto next_tag = [&](int kind) { const int t = ((h->xseq & 0x7ffffff) << 4) | kind; h->xseq += 1; return t; };
        struct Geom { int64_t t0; int K, q_lo, q_hi, C; };
        auto geom_at = [&](int64_t t0) {
            // sweep 1 starts from few labelled members: do not let a batch outnumber them by much
            // (measured: a batch of up to 1.5x the labelled members costs no extra rounds and saves a batch)
            int64_t members = (it == 0) ? (assigned0 + t0) * 3 / 2 : N;
#ifdef CHB_DEV_KNOBS
            { static double r = -1.0; if (r < 0.0) { const char *e = getenv("CHB_EARLY_RATIO"); r = e ? atof(e) : 1.5; }
              if (it == 0) members = (int64_t)((assigned0 + t0) * r); }
#endif
            int K = (int)std::min<int64_t>(Kmax, n_move - t0);
            if (members < K) K = (int)std::max<int64_t>(std::min<int64_t>(64, n_move - t0), members);
            K = std::min(K, Kmax);   // (the floor of 64 above must not exceed a caller's smaller batch: buffers hold Kmax)
            // multi-GPU: rank r evaluates positions [r*C, (r+1)*C) of the batch; the label slices
            // are exchanged with one in-place RCCL all-gather per round (KB-sized)
            const int C = (K + world - 1) / world;
            const int q_lo = std::min(K, h->rank * C);
            return Geom{t0, K, q_lo, std::min(K, q_lo + C), C};
        };
        // the seating order of every batch of this sweep (tile skipping / threshold pools), in one launch: the batches are a
        // function of the sweep alone.  (Not for sweeps of thousands of tiny batches: those order theirs one by one.)
        h->qord_cur = nullptr; h->home_cur = nullptr;
        std::vector<int64_t> batch_t0;
        if (h->pf_fit && h->fused && h->ckey.p != nullptr && n_move > 0 && (h->pool_valid || (h->allow_skip && h->skip_state >= 0))) {
            std::vector<int4> geo;
            for (int64_t t = 0; t < n_move && geo.size() <= 4096;) {
                const Geom g = geom_at(t);
                geo.push_back(make_int4((int)g.t0, g.q_lo, g.q_hi, g.K));
                batch_t0.push_back(t);
                t += g.K;
            }
            if (geo.size() <= 4096) {
                HIPCHK(h->geo_all.ensure(geo.size()));
                HIPCHK(h->qord_all.ensure((size_t)n_move));
                HIPCHK(h->home_all.ensure(geo.size() * (size_t)h->B));
                // (pinned staging: the previous sweep's upload from it completed before that sweep's final synchronisation)
                HIPCHK(h->pin_geo.ensure(geo.size()));
                memcpy(h->pin_geo.p, geo.data(), sizeof(int4) * geo.size());
                HIPCHK(hipMemcpyAsync(h->geo_all.p, h->pin_geo.p, sizeof(int4) * geo.size(), hipMemcpyHostToDevice, s));
                Timed t(h, "bucket", (double)n_move);
                launch_query_order_sweep(h->ckey.p, h->perm.p, h->geo_all.p, (int)geo.size(), h->B, h->qord_all.p, h->home_all.p, s);
            } else {
                batch_t0.clear();
            }
        }
        // after a round's kernels: (multi-GPU: exchange) + first-changed position on its way to the host
        auto finish_round = [&](const Geom &g, int active, int slot) -> int {
            if (xchg) {
                // this rank's frame {tag, statistics of the batch's base shortlist launch and pack, label slice} -> all-gather
                // -> every rank's labels into lab_new / lab_prev, first changed position, statistics summed over the ranks
                // (gated kernels, not memcpys: inside a look-ahead window they must not run; single rank without exchange:
                // the argmin kernel has already written lab_prev).  What comes home in the verdict slot is then the SAME on
                // every rank -- first change, bin sizes (functions of the replicated labels), skip statistics, arena mark --
                // and with it every decision of this loop, in particular whether the next batch is enqueued ahead.
                const int tag = next_tag(2);
                const bool first = active == 0;
#ifdef CHB_DEV_KNOBS
                if (first && dev_stats_on) for (int k = 0; k < 3; ++k) launch_fill_i32(h->fc_cur + 3 + k, dev_stats[k], 1, s);
#endif
                launch_xchg_pack(h->xg.p, h->rank, g.C, h->lab_new.p, tag, h->fc_cur, first, first && h->pp_batch, true, g.K, s);
                { const int r_ = exchange_all_gather(h, h->xg.p, (size_t)(g.C + kXchgHdr), sizeof(int), ncclInt32); if (r_) return r_; }
                launch_xchg_unpack(h->xg.p, world, g.C, g.K, tag, h->lab_new.p, h->lab_prev.p, active, h->fc_cur,
                                   first && !dev_local_verdict, h->xerr.p, s);
            }
            HIPCHK(hipMemcpyAsync(h->fc_host + kSlotInts * slot, h->fc_cur, 9 * sizeof(int), hipMemcpyDeviceToHost, s));
            HIPCHK(hipEventRecord(h->fc_event[slot], s));
            return CHB_OK;
        };
        auto wait_round = [&](const Geom &g, int active, int slot, int *f) -> int {
            HIPCHK(hipEventSynchronize(h->fc_event[slot]));
            *f = h->fc_host[kSlotInts * slot];
            // (bin sizes of that batch, for the segment decision of the batches still to be enqueued; and what the tile
            //  skipping of its shortlist launch achieved: a fit whose first batches skip next to nothing turns it off)
            h->hint_max_tiles = h->fc_host[kSlotInts * slot + 1]; h->hint_total_tiles = h->fc_host[kSlotInts * slot + 2];
            // (the persistent pack's arena: rows handed out so far, as of that batch's start)
            {
                int64_t mark_at = h->pp_mark;
#ifdef CHB_DEV_KNOBS   // CHB_PACK_REBUILD_AT=<rows>: rebuild (compact) the pack from that fill mark on -- tests of the rebuild path
                { static const char *e = getenv("CHB_PACK_REBUILD_AT"); if (e) mark_at = atoll(e); }
#endif
                if (h->pp_valid && h->fc_host[kSlotInts * slot + 6] > mark_at) h->pp_rebuild = true;
            }
            // (the slot's statistics are written by the batch's one base shortlist launch: counted with the batch's first
            //  round only -- later rounds of the same batch bring the same three numbers home again)
            if (active == 0 && h->fc_host[kSlotInts * slot + 4] > 0) {
                h->skip_skipped += h->fc_host[kSlotInts * slot + 3]; h->skip_seen += h->fc_host[kSlotInts * slot + 4];
                h->skip_unloaded += h->fc_host[kSlotInts * slot + 5];
                if (h->skip_state == 0 && ++h->skip_batches >= 3)
                {
                    // (it pays from a few per cent of the wave-tiles)
                    h->skip_state = ((h->skip_skipped + h->skip_unloaded) * 50 >= h->skip_seen + h->skip_unloaded) ? 1 : -1;
                    if (h->skip_state < 0) h->skip_off_key = skip_key(h);
                    // Where tile skipping never loads a third of a bin's tiles (500k x 140 x 128: 45 %), the threshold sweep is
                    // cheap already and the pools' price -- the looser thresholds of the contigs far out in their bins: long
                    // shortlists, retries, label guesses that fail -- is higher than what they save (113 against 105 ms per
                    // sweep there; 1M x 146 x 200, 19 % never loaded: 366 against 460): such a fit drops them
                    // (checked per batch below: sweep 1's first batches stream bins of a few tiles, nothing to go by)
                }
            }
            if (active == 0 && h->skip_state == 1 && h->pool_state >= 0 && h->fc_host[kSlotInts * slot + 4] > 0) {
                const long long un = h->fc_host[kSlotInts * slot + 5], sn = h->fc_host[kSlotInts * slot + 4];
                if (un * 10 > 3 * (sn + un)) { h->pool_state = -1; h->pool_off_key = skip_key(h); }
            }
            // (threshold pools: candidates per pair of that batch's base shortlist launch, as sampled; a fit whose first
            //  batches admit far more than the exact threshold would -- overlapping bins -- goes back to the two sweeps)
            //  -- checked for EVERY batch: the pools of sweep 1's first batches hold whole bins and say nothing yet)
            if (active == 0 && h->pool_state >= 0 && h->fc_host[kSlotInts * slot + 8] > 0) {
                const long long pc = h->fc_host[kSlotInts * slot + 7], pp = h->fc_host[kSlotInts * slot + 8];
                h->pool_cand += pc; h->pool_pairs += pp;
                if (++h->pool_batches >= 3 && h->pool_state == 0) h->pool_state = 1;
                // (the benchmark configurations admit m + 0.1 .. m + 0.4 per pair; from m + 3 on the loose thresholds cost the
                //  update stage and the hull kernel more than the threshold sweep did)
                if (pc > (long long)(h->m + 3) * pp) { h->pool_state = -1; h->pool_off_key = skip_key(h); }
            }
            (void)active; (void)g;
            return CHB_OK;
        };
        // batch start + guess + round 0, nothing read back
        auto open_batch = [&](const Geom &g, int slot) -> int {
            h->bq_cur = h->perm.p + g.t0;   // the batch's sample indices: a window of the sweep's permutation
            h->fc_cur = h->first_change.p + kSlotInts * slot;
            if (!batch_t0.empty()) {
                const size_t bi = (size_t)(std::lower_bound(batch_t0.begin(), batch_t0.end(), g.t0) - batch_t0.begin());
                h->qord_cur = h->qord_all.p + g.t0 + g.q_lo; h->home_cur = h->home_all.p + bi * (size_t)h->B;
            }
            h->hint_base_members = (double)((it == 0) ? assigned0 + g.t0 : labelled - g.K);
            h->hint_batch_entries = (double)((it == 0) ? g.K : 2 * g.K);
            h->argmin_in_place = !xchg;
            int r = batch_begin_dev(h, g.K, g.q_lo, g.q_hi, false);
            if (r) return r;
            h->pool_holes = sweep_has_labelled;   // (an all-unlabelled batch leaves no holes for its commit to look for)
            // starting labels of the rounds: last sweep's label, or for still-unlabelled contigs
            // (sweep 1) the bin whose m-th nearest outside member is closest
            if (h->fused && !h->lists_valid) launch_guess_near(h->tau.p, h->lab_old.p, g.q_lo, g.q_hi, h->B, h->Kcap, h->lab_prev.p, s);
            else launch_guess(h->l0d.p, h->l0c.p, h->lab_old.p, g.q_lo, g.q_hi, h->B, h->m, h->Kcap, h->lab_prev.p, s);
            if (xchg) {
                const int tag = next_tag(1);
                launch_xchg_pack(h->xg.p, h->rank, g.C, h->lab_prev.p, tag, h->fc_cur, false, false, false, g.K, s);
                { const int r_ = exchange_all_gather(h, h->xg.p, (size_t)(g.C + kXchgHdr), sizeof(int), ncclInt32); if (r_) return r_; }
                launch_xchg_unpack(h->xg.p, world, g.C, g.K, tag, h->lab_prev.p, nullptr, 0, h->fc_cur, false, h->xerr.p, s);
            }
            r = batch_round_dev(h, 0);
            if (r) return r;
            return finish_round(g, 0, slot);
        };
        struct Snap {   // host-side batch state (the device side of a gated-off batch never changed)
            int K, q_lo, q_hi, round_in_batch; bool lists_valid, batch_open, pp_batch, pp_valid, pool_valid; int *bq_cur, *fc_cur;
            double hb, he; int64_t st[4]; size_t n_pending;
        };
        auto save = [&]() {
            Snap v{h->K, h->q_lo, h->q_hi, h->round_in_batch, h->lists_valid, h->batch_open, h->pp_batch, h->pp_valid, h->pool_valid, h->bq_cur, h->fc_cur,
                   h->hint_base_members, h->hint_batch_entries, {0, 0, 0, 0}, h->pending.size()};
            memcpy(v.st, h->stats, sizeof(v.st));
            return v;
        };
        auto restore = [&](const Snap &v) {
            h->K = v.K; h->q_lo = v.q_lo; h->q_hi = v.q_hi; h->round_in_batch = v.round_in_batch;
            h->lists_valid = v.lists_valid; h->batch_open = v.batch_open; h->bq_cur = v.bq_cur; h->fc_cur = v.fc_cur;
            h->pp_batch = v.pp_batch; h->pp_valid = v.pp_valid; h->pool_valid = v.pool_valid;
            h->hint_base_members = v.hb; h->hint_batch_entries = v.he;
            memcpy(h->stats, v.st, sizeof(v.st));
            // the launches recorded inside the window were gated off (they returned at once): they are neither
            // launches nor work of the profile
            for (size_t i = v.n_pending; i < h->pending.size(); ++i) {
                (void)hipEventDestroy(h->pending[i].a);
                (void)hipEventDestroy(h->pending[i].b);
            }
            if (h->pending.size() > v.n_pending) h->pending.resize(v.n_pending);
        };
        struct GateReset { ~GateReset() { g_gate = Gate{}; } } gate_reset;   // (error returns inside the window)

        int64_t t0 = 0;
        bool inflight = false, spec_ok = can_spec;
        int slot = 0;
        while (t0 < n_move) {
            const Geom g = geom_at(t0);
            const int K = g.K;
            if (!inflight) { rc = open_batch(g, slot); if (rc) return rc; }
            const int64_t t1 = t0 + K;
            // (a batch start that has to build or rebuild the persistent pack stays outside the look-ahead window)
            const bool skip_would = h->allow_skip && h->nsh > 1 && h->skip_state >= 0 && h->ckey.p != nullptr;
            const bool pack_sync = h->pp_fit && h->pp_allowed && h->fused && h->pf_base && h->cand.p && !skip_would &&
                                   (!h->pp_valid || h->pp_rebuild);
            const bool spec = spec_ok && t1 < n_move && !pack_sync;
            Snap snap{};
            if (spec) {
                snap = save();
                g_gate = Gate{h->first_change.p + kSlotInts * slot, K};   // "this batch's round 0 changed nothing"
                rc = batch_commit_dev(h, h->lab_prev.p);
                if (rc) return rc;
                rc = open_batch(geom_at(t1), slot ^ 1);
                if (rc) return rc;
                g_gate = Gate{};
            }
            int f = K;
            rc = wait_round(g, 0, slot, &f);
            if (rc) return rc;
            if (f < K) {
                // the guess was off at position f: everything enqueued behind the gate has skipped itself
                if (spec) { restore(snap); spec_ok = false; h->stats_lookahead_failed += 1; }
                int active = f + 1;
                while (active < K) {
                    rc = batch_round_dev(h, active);
                    if (rc) return rc;
                    rc = finish_round(g, active, slot);
                    if (rc) return rc;
                    rc = wait_round(g, active, slot, &f);
                    if (rc) return rc;
                    if (f >= K) break;
                    active = f + 1;
                }
                inflight = false;
            } else {
                inflight = spec;   // the next batch's first round is already running
                if (spec) h->stats_lookahead += 1;
                spec_ok = can_spec;
            }
            if (min_dist_out && xchg)
                { const int r_ = exchange_all_gather(h, h->mind.p, (size_t)g.C, sizeof(double), ncclFloat64); if (r_) return r_; }
            if (min_dist_out) {
                HIPCHK(hipMemcpyAsync(mind_host.data(), h->mind.p, sizeof(double) * K, hipMemcpyDeviceToHost, s));
                if (margin_out)
                    HIPCHK(hipMemcpyAsync(mind2_host.data(), h->mind2.p, sizeof(double) * K, hipMemcpyDeviceToHost, s));
                HIPCHK(hipStreamSynchronize(s));
                for (int i = 0; i < K; ++i) min_dist_out[perm[t0 + i]] = mind_host[(size_t)i];
                if (margin_out)
                    for (int i = 0; i < K; ++i) margin_out[perm[t0 + i]] = mind2_host[(size_t)i] - mind_host[(size_t)i];
            }
            if (!inflight) {   // (otherwise the commit went out with the look-ahead)
                rc = batch_commit_dev(h, h->lab_prev.p);
                if (rc) return rc;
            } else {
                slot ^= 1;
            }
#ifdef CHB_DEV_KNOBS
            { static const bool tr = getenv("CHB_DEV_TRACE_BATCHES") != nullptr;
              if (tr) fprintf(stderr, "[chb batch] sweep %d t0 %lld K %d rounds so far %lld pool %d/%d skip %d spec %d\n", it, (long long)t0, K,
                              (long long)h->stats[1], (int)h->pool_valid, h->pool_state, h->skip_state, (int)inflight); }
#endif
            h->stats[0] += 1;
            t0 = t1;
        }
        h->stats[3] += n_move * (int64_t)h->B;
        HIPCHK(h->pin_b.ensure((size_t)N));
        HIPCHK(hipMemcpyAsync(h->pin_b.p, h->labels.p, sizeof(int) * N, hipMemcpyDeviceToHost, s));
        if (h->fused && h->short_cnt.p)   // (fc_host[12]: a spare word of the first verdict slot)
            HIPCHK(hipMemcpyAsync(h->fc_host + 12, h->short_cnt.p, sizeof(int), hipMemcpyDeviceToHost, s));
        h->fc_host[13] = 0;
        if (h->pp_ctl.p)   // (fc_host[13]: another spare word -- the persistent pack's error flag)
            HIPCHK(hipMemcpyAsync(h->fc_host + 13, h->pp_ctl.p + 2, sizeof(int), hipMemcpyDeviceToHost, s));
        std::vector<int> xend;
        if (xchg) {
            // every rank's "a rank was out of step" record: all ranks then leave the sweep with the same status
            xend.assign((size_t)4 * world, 0);
            HIPCHK(hipMemcpyAsync(h->agree.p + 4 * h->rank, h->xerr.p, 4 * sizeof(int), hipMemcpyDeviceToDevice, s));
            { const int r_ = exchange_all_gather(h, h->agree.p, 4, sizeof(int), ncclInt32); if (r_) return r_; }
            HIPCHK(hipMemcpyAsync(xend.data(), h->agree.p, sizeof(int) * xend.size(), hipMemcpyDeviceToHost, s));
        }
        HIPCHK(hipStreamSynchronize(s));
        for (int r = 0; r < (int)xend.size() / 4; ++r)
            if (xend[(size_t)4 * r] != 0) {
                HIPCHK(hipMemsetAsync(h->xerr.p, 0, 4 * sizeof(int), s));
                const int *e = xend.data() + 4 * r;
                return fail(CHB_ESTATE, "internal error: the ranks' exchanges fell out of step (rank " + std::to_string(r) + " was at exchange " +
                                        std::to_string(e[1] >> 4) + " kind " + std::to_string(e[1] & 15) + " when rank " + std::to_string(e[3]) +
                                        " sent exchange " + std::to_string(e[2] >> 4) + " kind " + std::to_string(e[2] & 15) +
                                        "); labels not returned");
            }
        if (h->fc_host[13] != 0)
            return fail(CHB_ESTATE, "internal error: the persistent member pack ran out of rows; labels not returned");
        if (h->fused && h->short_cnt.p && h->fc_host[12] != 0) {
            h->short_seen = h->fc_host[12];
            return fail(CHB_ESTATE, "internal error: " + std::to_string(h->fc_host[12]) + " (position, bin) shortlists of this sweep came "
                        "out short of min(num_neighbors, bin size) candidates or held a wild index; labels not returned");
        }
        int64_t diff = 0;  // algorithm.py:63
        labelled = 0;
        {   // (one pass: change count, label count and the caller's int64 copy -- the last sweep's is what stays)
            const int *pb = h->pin_b.p;
            const int *pv = prev.data();
            for (int64_t i = 0; i < N; ++i) {
                const int v = pb[i];
                diff += pv[i] != v;
                labelled += v >= 0;
                labels_out[i] = v;
            }
        }
        wrote_out = true;
        if (changed_per_iter) changed_per_iter[it] = diff;
        if (diff == 0) { ++it; break; }  // algorithm.py:64-66
        if (it + 1 < max_iter) prev.assign(h->pin_b.p, h->pin_b.p + N);   // algorithm.py:71-72
    }
    if (!wrote_out)
        for (int64_t i = 0; i < N; ++i) labels_out[i] = prev[(size_t)i];
    if (iters_run) *iters_run = it;
    fit_closer.ok = true;
    return CHB_OK;
}

int chb_topm_per_bin(chb_ctx *h, const int64_t *labels, int64_t B, int m, const int64_t *query_idx,
                     int64_t Q, int64_t *nbr_idx, double *nbr_dist, int32_t *nbr_cnt)
{
    if (!h || !labels || !query_idx || !nbr_idx || !nbr_cnt) return fail(CHB_EINVAL, "null argument");
    if (Q < 0) return fail(CHB_EINVAL, "Q < 0");
    HIPCHK(hipSetDevice(h->dev));
    std::vector<int64_t> lab((size_t)h->N);
    for (int64_t i = 0; i < h->N; ++i) lab[(size_t)i] = (labels[i] >= 0 && labels[i] < B) ? labels[i] : -1;
    int rc = fit_begin_impl(h, B, lab.data(), m);
    if (rc) return rc;
    for (int64_t i = 0; i < Q; ++i)
        if (query_idx[i] < 0 || query_idx[i] >= h->N) return fail(CHB_EINVAL, "query index out of range");
    const int Kmax = (int)std::min<int64_t>(std::max<int64_t>(Q, 1), 4096);
    rc = ensure_batch_buffers(h, Kmax);
    if (rc) return rc;
    hipStream_t s = h->stream;
    std::vector<double> hd;
    std::vector<int> hi, hc;
    int64_t t0 = 0;
    while (t0 < Q) {
        // a chunk must not contain the same sample twice
        std::unordered_set<int64_t> seen;
        int K = 0;
        while (t0 + K < Q && K < Kmax && seen.insert(query_idx[t0 + K]).second) ++K;
        std::vector<int> v = to_i32(query_idx + t0, (size_t)K);
        h->bq_cur = h->bq.p;
        HIPCHK(hipMemcpyAsync(h->bq_cur, v.data(), sizeof(int) * K, hipMemcpyHostToDevice, s));
        rc = batch_begin_dev(h, K, 0, K, true);
        if (rc) return rc;
        // every other query of the chunk is an ordinary member: code "pos != i"
        launch_bucket_batch(h->lab_old.p, nullptr, h->bq_cur, K, h->B, h->cnt2.p, h->bin_ptr2.p,
                            h->cursor2.p, h->memb2_id.p, h->memb2_code.p, nullptr, nullptr, nullptr, nullptr, s);
        TopmArgs a{};
        a.X = h->X.p; a.Dp = h->Dp; a.bq = h->bq_cur; a.pos_begin = 0; a.pos_end = K;
        a.bin_ptr = h->bin_ptr2.p; a.memb_id = h->memb2_id.p; a.memb_code = h->memb2_code.p;
        a.B = h->B; a.m = h->m; a.Kcap = h->Kcap;
        a.in = h->L0(); a.out = h->L1();
        {
            Timed t(h, "topm_update", (double)K);
            if (h->m > kMaxM) launch_topm_generic(a, s); else launch_topm(a, s);
        }
        HIPCHK(hipGetLastError());
        const size_t nB = (size_t)h->B, nm = (size_t)m, cap = (size_t)h->Kcap;
        hd.resize(nB * cap * nm); hi.resize(nB * cap * nm); hc.resize(nB * cap);
        HIPCHK(hipMemcpyAsync(hd.data(), h->l1d.p, sizeof(double) * hd.size(), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(hi.data(), h->l1i.p, sizeof(int) * hi.size(), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(hc.data(), h->l1c.p, sizeof(int) * hc.size(), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        for (int i = 0; i < K; ++i)
            for (size_t c = 0; c < nB; ++c) {
                const size_t slot = c * cap + (size_t)i;
                const size_t o = ((size_t)(t0 + i) * nB + c);
                nbr_cnt[o] = hc[slot];
                for (size_t e = 0; e < nm; ++e) {
                    nbr_idx[o * nm + e] = hi[slot * nm + e];
                    if (nbr_dist) nbr_dist[o * nm + e] = hd[slot * nm + e];
                }
            }
        rc = batch_commit_dev(h, h->lab_old.p);  // labels unchanged
        if (rc) return rc;
        HIPCHK(hipStreamSynchronize(s));
        t0 += K;
    }
    h->fit_open = false;
    return CHB_OK;
}

static int hull_indexed(chb_ctx *h, const double *Xdev, int D, int Dp, int64_t nrows,
                        const int64_t *query_idx, int64_t P, const int64_t *hull_idx, int m_max,
                        double *dist, double *alpha)
{
    if (m_max < 1 || m_max > CHB_MAX_NEIGHBORS) return fail(CHB_EUNSUPPORTED, "m_max must be in [1, 64]");
    if (P <= 0) return CHB_OK;
    if (m_max > kMaxM && !hull_generic_supported())
        return fail(CHB_EUNSUPPORTED, "more than 16 hull vertices need 68 KB of LDS per workgroup, which this device does not grant");
    hipStream_t s = h->stream;
    // compact each vertex list (padding may sit anywhere at the ABI) and remember the slots
    std::vector<int> q((size_t)P), hx((size_t)P * m_max, -1), slot((size_t)P * m_max, -1), hn((size_t)P, 0);
    for (int64_t p = 0; p < P; ++p) {
        if (query_idx[p] < 0 || query_idx[p] >= nrows) return fail(CHB_EINVAL, "query index out of range");
        q[(size_t)p] = (int)query_idx[p];
        int n = 0;
        for (int a = 0; a < m_max; ++a) {
            const int64_t id = hull_idx[p * m_max + a];
            if (id < 0) continue;
            if (id >= nrows) return fail(CHB_EINVAL, "hull index out of range");
            hx[(size_t)p * m_max + n] = (int)id;
            slot[(size_t)p * m_max + n] = a;
            ++n;
        }
        hn[(size_t)p] = n;
    }
    HIPCHK(h->xq.ensure((size_t)P));
    HIPCHK(h->xhull.ensure((size_t)P * m_max));
    HIPCHK(h->xcnt.ensure((size_t)P));
    HIPCHK(hipMemcpyAsync(h->xcnt.p, hn.data(), sizeof(int) * P, hipMemcpyHostToDevice, s));
    HIPCHK(h->xdist.ensure((size_t)P));
    HIPCHK(h->xalpha.ensure((size_t)P * m_max));
    HIPCHK(hipMemcpyAsync(h->xq.p, q.data(), sizeof(int) * P, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(h->xhull.p, hx.data(), sizeof(int) * P * m_max, hipMemcpyHostToDevice, s));
    {
        Timed t(h, "hull_qp", (double)P);
        if (m_max > kMaxM)
            launch_hull_generic_indexed(Xdev, D, Dp, h->xq.p, h->xhull.p, h->xcnt.p, (int)P, m_max, h->metric,
                                        h->xdist.p, h->xalpha.p, s);
        else
            launch_hull_qp_indexed(Xdev, D, Dp, h->xq.p, h->xhull.p, h->xcnt.p, (int)P, m_max, h->metric, h->xdist.p,
                                   h->xalpha.p, s);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(dist, h->xdist.p, sizeof(double) * P, hipMemcpyDeviceToHost, s));
    std::vector<double> al;
    if (alpha) {
        al.resize((size_t)P * m_max);
        HIPCHK(hipMemcpyAsync(al.data(), h->xalpha.p, sizeof(double) * P * m_max, hipMemcpyDeviceToHost, s));
    }
    HIPCHK(hipStreamSynchronize(s));
    if (alpha) {
        for (size_t i = 0; i < (size_t)P * m_max; ++i) alpha[i] = 0.0;
        for (int64_t p = 0; p < P; ++p)
            for (int a = 0; a < m_max; ++a) {
                const int sl = slot[(size_t)p * m_max + a];
                if (sl >= 0) alpha[p * m_max + sl] = al[(size_t)p * m_max + a];
            }
    }
    return CHB_OK;
}

int chb_hull_distance_batch(chb_ctx *h, const int64_t *query_idx, int64_t P, const int64_t *hull_idx,
                            int m_max, double *dist, double *alpha)
{
    if (!h || !query_idx || !hull_idx || !dist) return fail(CHB_EINVAL, "null argument");
    if (!h->X.p) return fail(CHB_ESTATE, "chb_set_samples has not been called");
    HIPCHK(hipSetDevice(h->dev));
    return hull_indexed(h, h->X.p, h->D, h->Dp, h->N, query_idx, P, hull_idx, m_max, dist, alpha);
}

int chb_hull_distance_points(chb_ctx *h, const double *x, const double *pts, int m, int64_t D,
                             double *dist, double *alpha)
{
    if (!h || !x || !dist || (m > 0 && !pts)) return fail(CHB_EINVAL, "null argument");
    if (m < 0 || D <= 0) return fail(CHB_EINVAL, "bad m or D");
    if (m == 0) { *dist = INFINITY; return CHB_OK; }
    if (m > CHB_MAX_NEIGHBORS) return fail(CHB_EUNSUPPORTED, "more than 64 hull vertices");
    HIPCHK(hipSetDevice(h->dev));
    const int Dp = (int)((D + kKChunk - 1) / kKChunk) * kKChunk;
    std::vector<double> rows((size_t)(m + 1) * Dp, 0.0);
    memcpy(rows.data(), x, sizeof(double) * D);
    for (int a = 0; a < m; ++a) memcpy(rows.data() + (size_t)(a + 1) * Dp, pts + (size_t)a * D, sizeof(double) * D);
    HIPCHK(h->xpts.ensure(rows.size()));
    HIPCHK(hipMemcpyAsync(h->xpts.p, rows.data(), sizeof(double) * rows.size(), hipMemcpyHostToDevice, h->stream));
    int64_t q = 0;
    std::vector<int64_t> idx((size_t)m);
    for (int a = 0; a < m; ++a) idx[(size_t)a] = a + 1;
    return hull_indexed(h, h->xpts.p, (int)D, Dp, m + 1, &q, 1, idx.data(), m, dist, alpha);
}

int chb_find_nearest_from_row(chb_ctx *h, int64_t c, const int64_t *labels, const double *row,
                              int64_t N, int m, int64_t *out_idx, int32_t *out_cnt)
{
    if (!h || !labels || !row || !out_idx || !out_cnt) return fail(CHB_EINVAL, "null argument");
    if (N <= 0 || N >= (1LL << 31)) return fail(CHB_EINVAL, "bad N");
    if (m < 1) return fail(CHB_EINVAL, "m must be >= 1");
    HIPCHK(hipSetDevice(h->dev));
    hipStream_t s = h->stream;
    std::vector<int> lab((size_t)N);
    for (int64_t i = 0; i < N; ++i) lab[(size_t)i] = (labels[i] < -1 || labels[i] > 0x7ffffff0) ? -2 : (int)labels[i];
    if (c < 0 || c > 0x7ffffff0) { *out_cnt = 0; for (int i = 0; i < m; ++i) out_idx[i] = -1; return CHB_OK; }
    DevBuf<int> dl, di;
    DevBuf<double> dr;
    hipError_t e = dl.ensure((size_t)N);
    if (e == hipSuccess) e = dr.ensure((size_t)N);
    if (e == hipSuccess) e = di.ensure((size_t)m + 1);
    if (e == hipSuccess) e = hipMemcpyAsync(dl.p, lab.data(), sizeof(int) * N, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(dr.p, row, sizeof(double) * N, hipMemcpyHostToDevice, s);
    std::vector<int> out((size_t)m + 1);
    if (e == hipSuccess) {
        launch_select_row(dl.p, dr.p, (int)N, (int)c, m, di.p, di.p + m, s);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out.data(), di.p, sizeof(int) * (m + 1), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    dl.release(); di.release(); dr.release();
    if (e != hipSuccess) return fail(CHB_EHIP, hipGetErrorString(e));
    for (int i = 0; i < m; ++i) out_idx[i] = out[(size_t)i];
    *out_cnt = out[(size_t)m];
    return CHB_OK;
}

int chb_set_metric(chb_ctx *h, int metric)
{
    if (!h) return fail(CHB_EINVAL, "null context");
    if (metric != CHB_METRIC_CONVEX && metric != CHB_METRIC_AFFINE) return fail(CHB_EINVAL, "unknown metric");
    h->metric = metric;
    return CHB_OK;
}

int chb_comm_unique_id(char *out128)
{
    if (!out128) return fail(CHB_EINVAL, "null argument");
    if (!rccl()) return fail(CHB_EUNSUPPORTED, "librccl could not be loaded");
    ncclUniqueId id;
    NCCLCHK(rccl()->GetUniqueId(&id));
    memcpy(out128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return CHB_OK;
}

int chb_comm_init(chb_ctx *h, const char *id128, int rank, int world)
{
    if (!h || !id128) return fail(CHB_EINVAL, "null argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(CHB_EINVAL, "bad rank/world");
    if (!rccl()) return fail(CHB_EUNSUPPORTED, "librccl could not be loaded");
    HIPCHK(hipSetDevice(h->dev));
    if (h->comm) { (void)rccl()->CommDestroy(h->comm); h->comm = nullptr; }
    ncclUniqueId id;
    memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    NCCLCHK(rccl()->CommInitRank(&h->comm, world, id, rank));
    h->rank = rank; h->world = world;
    h->Kcap = 0;
    return CHB_OK;
}

int chb_comm_init_hook(chb_ctx *h, int rank, int world, chb_allgather_fn fn, void *user)
{
    if (!h || !fn) return fail(CHB_EINVAL, "null argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(CHB_EINVAL, "bad rank/world");
    HIPCHK(hipSetDevice(h->dev));
    if (h->comm && rccl()) { (void)rccl()->CommDestroy(h->comm); h->comm = nullptr; }
    h->hook = fn; h->hook_user = user;
    h->rank = rank; h->world = world;
    h->Kcap = 0;
    return CHB_OK;
}

int chb_comm_destroy(chb_ctx *h)
{
    if (!h) return CHB_OK;
    h->hook = nullptr; h->hook_user = nullptr;
    if (h->comm && rccl()) {
        (void)hipSetDevice(h->dev);
        (void)hipStreamSynchronize(h->stream);
        (void)rccl()->CommDestroy(h->comm);
    }
    h->comm = nullptr; h->rank = 0; h->world = 1;
    return CHB_OK;
}

int chb_kmer_dim(int k)
{
    const int n = kmer_canonical_table(k, nullptr);
    return n > 0 ? n : fail(CHB_EUNSUPPORTED, "k must be in [1, 7]");
}

int chb_kmer_frequencies(chb_ctx *h, const unsigned char *seq, const int64_t *offsets, int64_t n, int k,
                         double *freq_out, uint32_t *counts_out)
{
    if (!h || !offsets || !freq_out) return fail(CHB_EINVAL, "null argument");
    if (n < 0 || n >= (1LL << 31)) return fail(CHB_EINVAL, "bad contig count");
    std::vector<unsigned short> table;
    const int dim = kmer_canonical_table(k, &table);
    if (dim <= 0) return fail(CHB_EUNSUPPORTED, "k must be in [1, 7]");
    if (n == 0) return CHB_OK;
    if (offsets[0] != 0) return fail(CHB_EINVAL, "offsets[0] must be 0");
    const int64_t total = offsets[n];
    if (total > 0 && !seq) return fail(CHB_EINVAL, "seq is null");
    HIPCHK(hipSetDevice(h->dev));
    // one work item per kmer_chunk_windows() windows of a contig
    const int cw = kmer_chunk_windows();
    std::vector<int> chunk_ptr((size_t)n + 1);
    std::vector<long long> off64((size_t)n + 1);
    int64_t items = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (offsets[i + 1] < offsets[i]) return fail(CHB_EINVAL, "offsets must be non-decreasing");
        chunk_ptr[(size_t)i] = (int)items;
        const int64_t nwin = offsets[i + 1] - offsets[i] - k + 1;
        if (nwin > 0) items += (nwin + cw - 1) / cw;
        if (items >= (1LL << 31) - 1) return fail(CHB_EUNSUPPORTED, "too many bases for one call");
        off64[(size_t)i] = offsets[i];
    }
    chunk_ptr[(size_t)n] = (int)items;
    off64[(size_t)n] = total;
    DevBuf<unsigned char> dseq;
    DevBuf<long long> doff;
    DevBuf<int> dptr;
    DevBuf<unsigned short> dtab;
    DevBuf<unsigned int> dcnt;
    DevBuf<double> dfreq;
    hipStream_t s = h->stream;
    HIPCHK(dseq.ensure((size_t)std::max<int64_t>(total, 1) + 16));
    HIPCHK(doff.ensure((size_t)n + 1));
    HIPCHK(dptr.ensure((size_t)n + 1));
    HIPCHK(dtab.ensure(table.size()));
    HIPCHK(dcnt.ensure((size_t)n * dim));
    HIPCHK(dfreq.ensure((size_t)n * dim));
    if (total > 0) HIPCHK(hipMemcpyAsync(dseq.p, seq, (size_t)total, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(doff.p, off64.data(), sizeof(long long) * (n + 1), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dptr.p, chunk_ptr.data(), sizeof(int) * (n + 1), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dtab.p, table.data(), sizeof(unsigned short) * table.size(), hipMemcpyHostToDevice, s));
    {
        Timed t(h, "kmer_count", (double)total);
        launch_kmer_count(dseq.p, doff.p, dptr.p, (int)n, (int)items, k, dim, dtab.p, dcnt.p, dfreq.p, s);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(freq_out, dfreq.p, sizeof(double) * (size_t)n * dim, hipMemcpyDeviceToHost, s));
    if (counts_out)
        HIPCHK(hipMemcpyAsync(counts_out, dcnt.p, sizeof(uint32_t) * (size_t)n * dim, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    dseq.release(); doff.release(); dptr.release(); dtab.release(); dcnt.release(); dfreq.release();
    return CHB_OK;
}

int chb_profile_enable(chb_ctx *h, int on)
{
    if (!h) return fail(CHB_EINVAL, "null context");
    drain_profile(h);
    h->prof = on < 0 ? 0 : (on > 2 ? 1 : on);
    return CHB_OK;
}

int chb_profile_reset(chb_ctx *h)
{
    if (!h) return fail(CHB_EINVAL, "null context");
    drain_profile(h);
    h->prof_acc.clear();
    return CHB_OK;
}

int chb_profile_get(chb_ctx *h, const char *kernel, double *total_ms, int64_t *launches, double *work_units)
{
    if (!h || !kernel) return fail(CHB_EINVAL, "null argument");
    (void)hipStreamSynchronize(h->stream);
    drain_profile(h);
    ProfEntry e;
    auto it = h->prof_acc.find(kernel);
    if (it != h->prof_acc.end()) e = it->second;
    if (total_ms) *total_ms = e.ms;
    if (launches) *launches = e.launches;
    if (work_units) *work_units = e.work;
    return CHB_OK;
}

int chb_counter(chb_ctx *h, const char *name, int64_t *out)
{
    if (!h || !name || !out) return fail(CHB_EINVAL, "null argument");
    *out = 0;
    if (!strcmp(name, "prefilter_overflow")) {
        if (h->overflow.p && h->overflow_total_valid) {
            int v = 0;
            HIPCHK(hipMemcpyAsync(&v, h->overflow.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            *out = v;
        }
        return CHB_OK;
    }
    if (!strcmp(name, "lookahead_batches")) { *out = h->stats_lookahead; return CHB_OK; }
    if (!strcmp(name, "lookahead_failed")) { *out = h->stats_lookahead_failed; return CHB_OK; }
    if (!strcmp(name, "exchanges")) { *out = h->xseq; return CHB_OK; }
    if (!strcmp(name, "pack_incremental_batches")) { *out = h->stats_pp_batches; return CHB_OK; }
    if (!strcmp(name, "pack_builds")) { *out = h->stats_pp_builds; return CHB_OK; }
    if (!strcmp(name, "pool_batches")) { *out = h->stats_pool_batches; return CHB_OK; }
    if (!strcmp(name, "pool_state")) { *out = h->pool_state; return CHB_OK; }
    if (!strcmp(name, "pool_candidates")) { *out = h->pool_cand; return CHB_OK; }
    if (!strcmp(name, "pool_pairs")) { *out = h->pool_pairs; return CHB_OK; }
    if (!strcmp(name, "shortlist_short")) {   // pairs of the last fit that broke the shortlist stage's contract (0, or the fit failed)
        if (h->short_cnt.p) {
            int v = 0;
            HIPCHK(hipMemcpyAsync(&v, h->short_cnt.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            *out = v;
        }
        return CHB_OK;
    }
    if (!strcmp(name, "shortlist_sum_last_batch") || !strcmp(name, "shortlist_max_last_batch")) {
        if (h->cand_cnt.p && h->K > 0) {
            std::vector<int> v((size_t)h->Kcap * h->B);
            HIPCHK(hipMemcpyAsync(v.data(), h->cand_cnt.p, sizeof(int) * v.size(), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            int64_t sum = 0, mx = 0;
            for (int c = 0; c < h->B; ++c)
                for (int i = 0; i < h->K; ++i) {
                    const int x = v[(size_t)c * h->Kcap + i];
                    sum += x; if (x > mx) mx = x;
                }
            *out = name[10] == 's' ? sum : mx;
        }
        return CHB_OK;
    }
    if (!strncmp(name, "shortlist_le", 12)) {   // "shortlist_le<N>_last_batch": pairs of the last batch with <= N candidates
        const int lim = atoi(name + 12);
        if (h->cand_cnt.p && h->K > 0) {
            std::vector<int> v((size_t)h->Kcap * h->B);
            HIPCHK(hipMemcpyAsync(v.data(), h->cand_cnt.p, sizeof(int) * v.size(), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            int64_t n = 0;
            for (int c = 0; c < h->B; ++c)
                for (int i = 0; i < h->K; ++i) n += v[(size_t)c * h->Kcap + i] <= lim;
            *out = n;
        }
        return CHB_OK;
    }
    if (!strcmp(name, "slow_pairs_last_round")) {   // pairs the fused kernel left to the exact path
        if (h->n_slow.p && h->fused) {
            int v = 0;
            HIPCHK(hipMemcpyAsync(&v, h->n_slow.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            *out = v;
        }
        return CHB_OK;
    }
    if (!strcmp(name, "fused_enabled")) { *out = h->fused ? 1 : 0; return CHB_OK; }
    if (!strcmp(name, "segment_batches")) { *out = h->stats_seg_batches; return CHB_OK; }
    // tile skipping of the last fit: its verdict (0 undecided, 1 kept on, -1 turned off) and the sampled wave-tile counters
    if (!strcmp(name, "batch_size")) { *out = h->last_batch; return CHB_OK; }   // (speculative batch size of the last fit)
    if (!strcmp(name, "tile_skip_state")) { *out = h->skip_state; return CHB_OK; }
    if (!strcmp(name, "tile_skipped")) { *out = h->skip_skipped; return CHB_OK; }
    if (!strcmp(name, "tile_seen")) { *out = h->skip_seen; return CHB_OK; }
    if (!strcmp(name, "tile_unloaded")) { *out = h->skip_unloaded; return CHB_OK; }
    if (!strcmp(name, "last_batch_k")) { *out = h->K; return CHB_OK; }
    if (!strcmp(name, "prefilter_enabled")) { *out = (h->use_prefilter && h->shadow_ok) ? 1 : 0; return CHB_OK; }
    return fail(CHB_EINVAL, "unknown counter");
}

int chb_fit_stats(chb_ctx *h, int64_t *out4)
{
    if (!h || !out4) return fail(CHB_EINVAL, "null argument");
    for (int i = 0; i < 4; ++i) out4[i] = h->stats[i];
    return CHB_OK;
}

}  // extern "C"
