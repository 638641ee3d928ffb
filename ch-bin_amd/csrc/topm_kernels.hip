// Distance tiles and per-(query, bin) nearest-member selection for gfx950.
//
// Replaces, without ever materialising the N x N matrix:
//   distance_matrix.py:33-44  cdist(arr, arr, "euclidean")         (reference; scipy C)
//   algorithm.py:51           distance_row[:] = distance_matrix[i]  (row fetch)
//   distance_matrix.py:47-62  find_nearest_from_cluster             (np.where over all N + argpartition)
//
// One 256-thread workgroup owns (64 batch queries) x (one bin): it streams the bin's members in tiles
// of 64 rows, accumulates a 64x64 tile of squared distances with a 4x4 register micro-tile per
// thread (fp64 VALU; LDS-staged transposed k-chunks, double buffered), and keeps each query's
// current m best members of THIS bin in the registers of the 16 lanes that share the query
// (lane tx holds list entry tx), so selection needs no LDS and no barrier.
//
// Numerics: every squared distance is sum_k (q_k - p_k)^2 accumulated sequentially in k with
// separate multiply and add (this file is compiled with -ffp-contract=off), then a correctly
// rounded sqrt: bit-identical to scipy's cdist (the CPU checker restates the same loop).
// Ordering of members is (distance, sample index) ascending -- total, so the result does not
// depend on the order in which members are streamed.
#include "chb_internal.h"

#include <limits.h>
#include <math.h>

#include <algorithm>

#pragma clang fp contract(off)

namespace chb {
namespace {

constexpr double kInf = __builtin_huge_val();

__device__ __forceinline__ bool lex_less(double d0, int i0, double d1, int i1)
{
    return d0 < d1 || (d0 == d1 && i0 < i1);
}

// s-domain (squared distance) admission bound for a list whose m-th entry has distance e:
// sqrt(s) <= e  implies  s <= e*e*(1 + 2^-50)
__device__ __forceinline__ double tau_from(double e) { return e * e * (1.0 + 0x1p-50); }

// Offer the (up to) NC candidates held by every lane of a W-lane group to the group's sorted
// list (lane t of the group holds entry t).  Wave-synchronous; all 64 lanes must call it together.
template <int W, int NC>
__device__ __forceinline__ void select_into(double (&s)[NC], const int (&mid)[NC], double &ld, int &li,
                                            int &lc, double &tau, int m, int tx, int gbase)
{
#pragma unroll
    for (int j = 0; j < NC; ++j)
        if (!(s[j] <= tau)) s[j] = kInf;
    for (;;) {
        double bs = s[0];
        int bi = mid[0], bj = 0;
#pragma unroll
        for (int j = 1; j < NC; ++j)
            if (lex_less(s[j], mid[j], bs, bi)) { bs = s[j]; bi = mid[j]; bj = j; }
        double gs = bs;
        int gi = bi;
#pragma unroll
        for (int off = W / 2; off >= 1; off >>= 1) {
            double os = __shfl_xor(gs, off, W);
            int oi = __shfl_xor(gi, off, W);
            if (lex_less(os, oi, gs, gi)) { gs = os; gi = oi; }
        }
        const bool have = gs < kInf;
        if (!__any(have)) break;
        if (have && bs == gs && bi == gi) {
#pragma unroll
            for (int j = 0; j < NC; ++j)
                if (j == bj) s[j] = kInf;
        }
        const double d = sqrt(gs);
        const bool lt = (tx < lc) && lex_less(ld, li, d, gi);
        const unsigned long long bal = __ballot(lt);
        const int pos = __popcll((bal >> gbase) & ((W == 64) ? ~0ull : ((1ull << (W & 63)) - 1ull)));
        const double ud = __shfl_up(ld, 1, W);
        const int ui = __shfl_up(li, 1, W);
        const bool ins = have && pos < m;
        if (ins) {
            if (tx == pos) { ld = d; li = gi; }
            else if (tx > pos) { ld = ud; li = ui; }
            lc = lc + 1 < m ? lc + 1 : m;
        }
        const double e = __shfl(ld, m - 1, W);
        if (ins && lc >= m) {
            tau = tau_from(e);
#pragma unroll
            for (int j = 0; j < NC; ++j)
                if (!(s[j] <= tau)) s[j] = kInf;
        }
    }
}

// PW = false: top-m selection per (query tile, bin).  PW = true: write the distance tile itself
// (chb_pairwise_distance); "bins" are then contiguous member ranges and bq is null.
template <bool PW>
__global__ __launch_bounds__(256) void tile_kernel(TopmArgs a, int nqt, int total, int pw_n,
                                                   int pw_group, double *pw_out, int *flags, const int *flaglist,
                                                   const int *nflag, Gate gate)
{
    CHB_GATE(gate);
    __shared__ __attribute__((aligned(16))) double sQ[2][kKChunk][kLdsStride];
    __shared__ __attribute__((aligned(16))) double sP[2][kKChunk][kLdsStride];
    __shared__ int sMid[2][kPTile];
    __shared__ int sMcode[2][kPTile];

    // fallback launch (flaglist != nullptr): a few workgroups walk the listed work items; otherwise one work
    // item per workgroup
    const int nfl = flaglist != nullptr ? *nflag : 0;
    for (int witer = 0;; ++witer) {
    int W;
    if (flaglist != nullptr) {
        const int i = (int)blockIdx.x + witer * (int)gridDim.x;
        if (i >= nfl) return;
        W = flaglist[i];
    } else {
        // XCD-aware order: blocks b and b+8 share an XCD (and its L2), so hand each XCD a contiguous
        // range of work items; consecutive items share a bin, i.e. the same member rows.
        const int per = (total + 7) >> 3;
        W = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
        if (W >= total) return;
    }
    const int c = W / nqt, qt = W - c * nqt;

    const int tid = threadIdx.x, lane = tid & 63;
    const int tx = tid & 15, ty = tid >> 4;
    const int gbase = lane & 48;
    const int srow = tid >> 2, skp = tid & 3;

    int mb, nmem;
    if (PW) {
        mb = c * pw_group;
        nmem = pw_n - mb < pw_group ? pw_n - mb : pw_group;
    } else {
        mb = a.bin_ptr[c];
        nmem = a.bin_cnt != nullptr ? a.bin_cnt[c] : a.bin_ptr[c + 1] - mb;
    }
    const int pos0 = a.pos_begin + qt * kQTile;
    const int m = a.m;

    // list state of my 4 queries: this lane holds entry #tx
    double ld[4], tau[4];
    int li[4], lc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ld[i] = kInf; li[i] = INT_MAX; lc[i] = 0; tau[i] = kInf;
        if (!PW && a.in.d != nullptr) {
            const int qpos = pos0 + 4 * ty + i;
            if (qpos < a.pos_end) {
                const size_t slot = (size_t)c * a.Kcap + qpos;
                lc[i] = a.in.cnt[slot];
                if (tx < lc[i]) { ld[i] = a.in.d[slot * m + tx]; li[i] = a.in.idx[slot * m + tx]; }
                if (lc[i] >= m) tau[i] = tau_from(a.in.d[slot * m + m - 1]);
            }
        }
    }

    // staging role: thread (srow, skp) moves feature columns [2*skp, 2*skp+1] of one row per chunk
    int sq = pos0 + srow;
    if (sq >= a.pos_end) sq = a.pos_end - 1;
    const int qid = PW ? sq : a.bq[sq];
    const double *qrow = a.X + (size_t)qid * a.Dp + 2 * skp;
    const double *prow = a.X + 2 * skp;
    const int nch = a.Dp / kKChunk;
    const int ntile = (nmem + kPTile - 1) / kPTile;
    const int nsteps = ntile * nch;

    double2 rq, rp;
    int pmid = -1, pcode = 0;
    int nt = 0, nc = 0;  // (tile, chunk) of the step being prefetched
    auto prefetch = [&]() {
        if (nc == 0) {
            const int e = nt * kPTile + srow;
            if (e < nmem) {
                pmid = PW ? mb + e : a.memb_id[mb + e];
                pcode = (!PW && a.memb_code) ? a.memb_code[mb + e] : 0;
            } else {
                pmid = -1; pcode = 0;
            }
            prow = a.X + (size_t)(pmid < 0 ? 0 : pmid) * a.Dp + 2 * skp;
        }
        rq = *reinterpret_cast<const double2 *>(qrow + nc * kKChunk);
        rp = *reinterpret_cast<const double2 *>(prow + nc * kKChunk);
    };
    auto stash = [&](int buf) {
        sQ[buf][2 * skp][srow] = rq.x;
        sQ[buf][2 * skp + 1][srow] = rq.y;
        sP[buf][2 * skp][srow] = rp.x;
        sP[buf][2 * skp + 1][srow] = rp.y;
        if (nc == 0 && skp == 0) { sMid[nt & 1][srow] = pmid; sMcode[nt & 1][srow] = pcode; }
        if (++nc == nch) { nc = 0; ++nt; }
    };

    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;

    if (nsteps > 0) { prefetch(); stash(0); }
    __syncthreads();
    if (flags != nullptr && tid == 0) flags[W] = 0;   // served (the flag only de-duplicates the list)

    int ct = 0, cc = 0;  // (tile, chunk) of the step being computed
    for (int step = 0; step < nsteps; ++step) {
        const int buf = step & 1;
        const bool has_next = step + 1 < nsteps;
        if (has_next) prefetch();
        // keep the global loads of the next chunk in flight across the whole compute block: the
        // scheduler must neither sink them nor hoist the LDS stores that consume them
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < kKChunk; ++k) {
            const double2 qa = *reinterpret_cast<const double2 *>(&sQ[buf][k][4 * ty]);
            const double2 qb = *reinterpret_cast<const double2 *>(&sQ[buf][k][4 * ty + 2]);
            const double2 pa = *reinterpret_cast<const double2 *>(&sP[buf][k][2 * tx]);
            const double2 pb = *reinterpret_cast<const double2 *>(&sP[buf][k][32 + 2 * tx]);
            const double q[4] = {qa.x, qa.y, qb.x, qb.y};
            const double p[4] = {pa.x, pa.y, pb.x, pb.y};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const double df = q[i] - p[j];
                    acc[i][j] = acc[i][j] + df * df;
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (has_next) stash(buf ^ 1);
        __syncthreads();
        if (++cc == nch) {
            // tile finished: offer its 64 members to the lists (or write the distances out)
            int mid[4], code[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ml = (j < 2) ? 2 * tx + j : 32 + 2 * tx + (j - 2);
                mid[j] = sMid[ct & 1][ml];
                code[j] = sMcode[ct & 1][ml];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int qpos = pos0 + 4 * ty + i;
                const bool qvalid = qpos < a.pos_end;
                if (PW) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (qvalid && mid[j] >= 0)
                            pw_out[(size_t)(qpos - a.pos_begin) * pw_n + mid[j]] = sqrt(acc[i][j]);
                } else {
                    double s[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        bool ok = qvalid && mid[j] >= 0;
                        if (code[j] > 0) ok = ok && (qpos > code[j] - 1);
                        else if (code[j] <= -(1 << 30)) ok = ok && (qpos != -(1 << 30) - code[j]);
                        else if (code[j] < 0) ok = ok && (qpos < -code[j] - 1);
                        s[j] = ok ? acc[i][j] : kInf;
                    }
                    select_into<16, 4>(s, mid, ld[i], li[i], lc[i], tau[i], m, tx, gbase);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
            }
            cc = 0; ++ct;
        }
    }

    if (!PW) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int qpos = pos0 + 4 * ty + i;
            if (qpos < a.pos_end) {
                const size_t slot = (size_t)c * a.Kcap + qpos;
                if (a.out.d != nullptr) {
                    if (tx < m) {
                        a.out.d[slot * m + tx] = tx < lc[i] ? ld[i] : kInf;
                        a.out.idx[slot * m + tx] = tx < lc[i] ? li[i] : -1;
                    }
                    if (tx == 0) a.out.cnt[slot] = lc[i];
                }
                if (a.cand_out != nullptr) {
                    // the exact top-m IS a valid shortlist (fused selection path)
                    if (tx < lc[i]) a.cand_out[slot * a.cand_cap + tx] = li[i];
                    if (tx == 0) a.cand_cnt_out[slot] = lc[i];
                    const double e_m = __shfl(ld[i], m - 1, 16);
                    if (tx == 0 && a.tau_out != nullptr) {
                        float t = INFINITY;
                        if (lc[i] >= m) {
                            t = (float)(e_m * a.S);
                            if ((double)t < e_m * a.S) t = nextafterf(t, INFINITY);
                            t *= 1.0f + 1e-6f;
                        }
                        a.tau_out[slot] = t;
                    }
                }
            }
        }
    }
    if (flaglist == nullptr) return;
    __syncthreads();   // the staging buffers are reused by the next listed work item
    }
}

// Stage 2 of the two-stage selection: exact distances on the shortlist of one (position, bin)
// pair per 16-lane group (4 pairs per wavefront), one candidate per lane.  Each lane must add its
// row's squared differences strictly in feature order (cdist rounding), so lanes cannot split a
// row; instead the wavefront fetches 16-feature chunks of its 64 candidate rows with coalesced
// 128-byte segments (8 lanes per row) into a private LDS slab, and every lane then walks its own
// row there (row stride 18 doubles: conflict-free ds_read_b128).  No barriers: the slab is per
// wavefront; the next chunk's loads are in flight during the arithmetic.  Then the same
// (distance, index) top-m list as the brute-force kernel, optionally seeded with a list (`in`).
constexpr int kRsChunk = 16;
constexpr int kRsStride = 18;

// W lanes per (position, bin) pair: 8 when the list fits 8 lanes (m <= 8: the shortlists average
// ~7 candidates, so 16-lane groups would idle half their lanes), else 16.
template <int W, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void rescore_kernel(RescoreArgs a, int npairs, Gate gate)
{
    CHB_GATE(gate);
    constexpr int G = 64 / W;   // pairs per wavefront
    __shared__ __attribute__((aligned(16))) double slab[WAVES][64 + G][kRsStride];

    const int lane = threadIdx.x & 63, gl = lane & (W - 1), gbase = lane & ~(W - 1), w = threadIdx.x >> 6;
    const int grp = lane / W;
    // active list: *n_active pairs, walked grid-stride (the grid is capped by the launcher)
    if (a.active != nullptr) npairs = *a.n_active;
    for (int wg0 = ((int)blockIdx.x * WAVES + w) * G; wg0 < npairs; wg0 += (int)gridDim.x * WAVES * G) {
    const int gidx = wg0 + grp;
    const bool pvalid = gidx < npairs;
    int pair = gidx;
    if (a.active != nullptr) pair = pvalid ? a.active[gidx] : 0;
    int pos = a.pos_begin, c = 0, cnt = 0, cnt1 = 0;
    if (pvalid) {
        pos = a.pos_begin + pair / a.B;
        c = pair - (pair / a.B) * a.B;
    }
    const size_t slot = (size_t)c * a.Kcap + pos;
    if (pvalid) {
        cnt1 = a.cand_cnt[slot];
        cnt = cnt1 + (a.cand2 != nullptr ? a.cand2_cnt[slot] : 0);
    }
    const int qid = a.bq[pos];
    const int m = a.m;

    double ld = kInf, tau = kInf;
    int li = INT_MAX, lc = 0;
    if (pvalid && a.in.d != nullptr) {
        lc = a.in.cnt[slot];
        if (gl < lc) { ld = a.in.d[slot * m + gl]; li = a.in.idx[slot * m + gl]; }
        if (lc >= m) tau = tau_from(a.in.d[slot * m + m - 1]);
    }

    int cmax = cnt;
#pragma unroll
    for (int off = W; off < 64; off <<= 1) cmax = max(cmax, __shfl_xor(cmax, off, 64));
    cmax = __builtin_amdgcn_readfirstlane(cmax);
    const int srow = lane >> 3, spc = lane & 7;   // staging role: row srow + 8 i, 16-byte piece spc
    const int nchunks = (a.Dp + kRsChunk - 1) / kRsChunk;
    // lanes with srow < G also stage the query row of lane group srow
    const int qsel = __shfl(qid, (srow & (G - 1)) * W, 64);
    const double *qp = a.X + (size_t)qsel * a.Dp + 2 * spc;
    for (int base = 0; base < cmax; base += W) {
        const int ci = base + gl;
        const bool have = ci < cnt;
        const int id = have ? (ci < cnt1 ? a.cand[slot * a.cand_cap + ci] : a.cand2[slot * a.cand2_cap + (ci - cnt1)]) : qid;
        const double *rp[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) rp[i] = a.X + (size_t)__shfl(id, srow + 8 * i, 64) * a.Dp + 2 * spc;
        double sacc = 0.0;
        double2 v[8], qv;
        auto fetch = [&](int ch) {
            const int k0 = ch * kRsChunk;
            const bool kin = k0 + 2 * spc < a.Dp;   // Dp % 8 == 0: the last chunk may be half empty
#pragma unroll
            for (int i = 0; i < 8; ++i)
                v[i] = kin ? *reinterpret_cast<const double2 *>(rp[i] + k0) : double2{0.0, 0.0};
            qv = double2{0.0, 0.0};
            if (srow < G && kin) qv = *reinterpret_cast<const double2 *>(qp + k0);
        };
        fetch(0);
        for (int ch = 0; ch < nchunks; ++ch) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                *reinterpret_cast<double2 *>(&slab[w][srow + 8 * i][2 * spc]) = v[i];
            if (srow < G) *reinterpret_cast<double2 *>(&slab[w][64 + srow][2 * spc]) = qv;
            __builtin_amdgcn_wave_barrier();
            if (ch + 1 < nchunks) fetch(ch + 1);   // next chunk's loads fly during the arithmetic
            __builtin_amdgcn_sched_barrier(0);
            const double *mine = &slab[w][lane][0];
            const double *qrow = &slab[w][64 + grp][0];
#pragma unroll
            for (int k = 0; k < kRsChunk; k += 2) {
                const double2 pv = *reinterpret_cast<const double2 *>(mine + k);
                const double2 qq = *reinterpret_cast<const double2 *>(qrow + k);
                const double d0 = qq.x - pv.x;
                sacc = sacc + d0 * d0;
                const double d1 = qq.y - pv.y;
                sacc = sacc + d1 * d1;
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_wave_barrier();
        }
        double s1[1] = {have ? sacc : kInf};
        const int id1[1] = {have ? id : INT_MAX};
        select_into<W, 1>(s1, id1, ld, li, lc, tau, m, gl, gbase);
    }
    if (pvalid) {
        if (gl < m) {
            a.out.d[slot * m + gl] = gl < lc ? ld : kInf;
            a.out.idx[slot * m + gl] = gl < lc ? li : -1;
        }
        if (gl == 0) a.out.cnt[slot] = lc;
    }
    }   // grid-stride loop
}

// ---------------------------------------------------------------------------------------------
// num_neighbors > 16 (the reference puts no cap on AlgoNumNeighbors: algorithm.py:17,
// cli/clustering.py:118-120): a plain form of the same selection with no tiling at all -- one wavefront per
// (bin, query), one member per lane, every lane sums its own row in feature order (cdist rounding), the list of
// up to 64 entries lives one entry per lane.  Orders of magnitude slower than the tiled kernel; it exists so
// that such a configuration runs instead of being refused.
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void select_generic_kernel(TopmArgs a, int nq, Gate gate)
{
    CHB_GATE(gate);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long item = (long long)blockIdx.x * WAVES + w;
    if (item >= (long long)nq * a.B) return;
    const int c = (int)(item / nq), qpos = a.pos_begin + (int)(item - (long long)c * nq);
    const size_t slot = (size_t)c * a.Kcap + qpos;
    const int m = a.m;
    const double *xq = a.X + (size_t)a.bq[qpos] * a.Dp;
    double ld = kInf, tau = kInf;
    int li = INT_MAX, lc = 0;
    if (a.in.d != nullptr) {
        lc = a.in.cnt[slot];
        if (lane < lc) { ld = a.in.d[slot * m + lane]; li = a.in.idx[slot * m + lane]; }
        if (lc >= m) tau = tau_from(a.in.d[slot * m + m - 1]);
    }
    const int mb = a.bin_ptr[c], nmem = a.bin_cnt != nullptr ? a.bin_cnt[c] : a.bin_ptr[c + 1] - mb;
    for (int base = 0; base < nmem; base += 64) {
        const int e = base + lane;
        bool ok = e < nmem;
        int mid = ok ? a.memb_id[mb + e] : 0;
        if (mid < 0) { ok = false; mid = 0; }   // (a hole of the persistent base pack)
        if (ok && a.memb_code != nullptr) {
            const int code = a.memb_code[mb + e];
            if (code > 0) ok = qpos > code - 1;
            else if (code <= -(1 << 30)) ok = qpos != -(1 << 30) - code;
            else if (code < 0) ok = qpos < -code - 1;
        }
        const double *row = a.X + (size_t)mid * a.Dp;
        double acc = 0.0;
        for (int k = 0; k < a.Dp; ++k) {
            const double df = xq[k] - row[k];
            acc = acc + df * df;
        }
        double s1[1] = {ok ? acc : kInf};
        const int id1[1] = {ok ? mid : INT_MAX};
        select_into<64, 1>(s1, id1, ld, li, lc, tau, m, lane, 0);
    }
    if (a.out.d != nullptr && lane < m) {
        a.out.d[slot * m + lane] = lane < lc ? ld : kInf;
        a.out.idx[slot * m + lane] = lane < lc ? li : -1;
    }
    if (a.out.d != nullptr && lane == 0) a.out.cnt[slot] = lc;
}

}  // namespace

void launch_topm_generic(const TopmArgs &a, hipStream_t s)
{
    const int nq = a.pos_end - a.pos_begin;
    if (nq <= 0 || a.B <= 0) return;
    constexpr int WV = 4;
    const long long items = (long long)nq * a.B;
    hipLaunchKernelGGL((select_generic_kernel<WV>), dim3((unsigned)((items + WV - 1) / WV)), dim3(64 * WV), 0, s, a, nq, g_gate);
}

void launch_topm(const TopmArgs &a, hipStream_t s)
{
    const int nq = a.pos_end - a.pos_begin;
    if (nq <= 0 || a.B <= 0) return;
    const int nqt = (nq + kQTile - 1) / kQTile;
    const int total = nqt * a.B;
    const int grid = ((total + 7) / 8) * 8;
    hipLaunchKernelGGL(tile_kernel<false>, dim3(grid), dim3(256), 0, s, a, nqt, total, 0, 0,
                       (double *)nullptr, (int *)nullptr, (const int *)nullptr, (const int *)nullptr, g_gate);
}

void launch_topm_flagged(const TopmArgs &a, int *flags64, const int *flaglist, const int *nflag, hipStream_t s)
{
    const int nq = a.pos_end - a.pos_begin;
    if (nq <= 0 || a.B <= 0) return;
    const int nqt = (nq + kQTile - 1) / kQTile;
    const int total = nqt * a.B;
    const int grid = std::min(total, 256);   // the listed work items are few (usually none)
    hipLaunchKernelGGL(tile_kernel<false>, dim3(grid), dim3(256), 0, s, a, nqt, total, 0, 0,
                       (double *)nullptr, flags64, flaglist, nflag, g_gate);
}

void launch_rescore(const RescoreArgs &a, hipStream_t s)
{
    const int npairs = (a.pos_end - a.pos_begin) * a.B;
    if (npairs <= 0) return;
    const int cap = a.active != nullptr ? 512 : INT_MAX;   // listed pairs: grid-stride over *n_active
    if (a.m <= 8) {
        constexpr int WV = 2, PB = WV * 8;      // 8 pairs per wavefront
        hipLaunchKernelGGL((rescore_kernel<8, WV>), dim3(std::min((npairs + PB - 1) / PB, cap)), dim3(64 * WV), 0, s, a, npairs, g_gate);
    } else {
        constexpr int WV = 4, PB = WV * 4;
        hipLaunchKernelGGL((rescore_kernel<16, WV>), dim3(std::min((npairs + PB - 1) / PB, cap)), dim3(64 * WV), 0, s, a, npairs, g_gate);
    }
}

void launch_pairwise(const double *X, int N, int Dp, int r0, int r1, double *out, hipStream_t s)
{
    if (r1 <= r0) return;
    TopmArgs a{};
    a.X = X; a.Dp = Dp; a.bq = nullptr; a.pos_begin = r0; a.pos_end = r1;
    a.m = 1; a.Kcap = 0; a.B = 0;
    const int group = 1024;
    const int ngroups = (N + group - 1) / group;
    const int nqt = (r1 - r0 + kQTile - 1) / kQTile;
    const int total = nqt * ngroups;
    const int grid = ((total + 7) / 8) * 8;
    hipLaunchKernelGGL(tile_kernel<true>, dim3(grid), dim3(256), 0, s, a, nqt, total, N, group, out,
                       (int *)nullptr, (const int *)nullptr, (const int *)nullptr, g_gate);
}

}  // namespace chb
