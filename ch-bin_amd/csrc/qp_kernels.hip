// Batched point-to-convex-hull distance for gfx950.
//
// Replaces hull_distance.py:7-35 (convex_hull_distance: P = 2 X X^T, q = -2 X x, sum(alpha) = 1,
// alpha >= 0, ||alpha X - x||) together with solve_qp.py:18-51 / :96-132 (nearest-PD repair,
// quadprog's Goldfarb-Idnani solve, cvxopt fallback) for every (contig, bin) pair of a batch.
//
// The QP min ||sum_a alpha_a x_a - x||^2 over the simplex is the minimum-norm point of
// conv{y_a = x_a - x}.  It is solved in Gram space:
//   phase 1: the shifted Gram Q_ab = <y_a, y_b>.  m <= 8: 16 lanes per problem, fp64 vector FMAs over
//     interleaved feature pairs, one reduce-scatter (hull_qp_kernel).  m > 8: one
//     v_mfma_f64_16x16x4_f64 tile per problem (hull_qp16_kernel).
//   phase 2: Wolfe's minimum-norm-point active-set method.  m <= 8: one problem per lane on the m x m
//     Gram held in registers (fixed-size, mask-driven, fully unrolled so nothing is dynamically
//     indexed); m > 8: one problem per 16 lanes.  Affine sub-problems by LDL^T (or an updated inverse)
//     of the lifted Gram Q_SS + s*11^T, which is positive definite exactly when the support is
//     affinely independent -- so duplicate / collinear vertices (singular 2XX^T, the case the
//     reference repairs with nearest-PD jitter and a cvxopt fallback) need no special handling.
//   distance = sqrt(alpha^T Q alpha).
// The result is the same projection distance the reference's QP defines (unique even when alpha
// is not); agreement with the CPU oracle's Goldfarb-Idnani restatement is ~1e-13 relative.
#include "chb_internal.h"
#include <cstdlib>
#include <type_traits>
#include <vector>

#include <math.h>

#include <algorithm>

#ifndef CHB_QP_UNROLL
#define CHB_QP_UNROLL 1
#endif

namespace chb {
namespace {

constexpr double kInf = __builtin_huge_val();

template <int M>
struct Sym {
    static constexpr int NP = M * (M + 1) / 2;
    __device__ static constexpr int at(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }
};

// Affine minimiser on support S: solve (Q_SS + s 11^T) b = 1, beta = b / sum(b).
// Rows/columns outside S are replaced by identity.  Returns false when a pivot collapses
// (support numerically affinely dependent).
template <int M>
__device__ __forceinline__ bool solve_affine(const double (&Q)[Sym<M>::NP], double s, unsigned S,
                                             double (&beta)[M])
{
    double L[M * (M - 1) / 2 + 1];
    double dg[M];
    bool ok = true;
#pragma unroll
    for (int k = 0; k < M; ++k) {
        const bool ink = (S >> k) & 1u;
        const double akk = ink ? Q[Sym<M>::at(k, k)] + s : 1.0;
        double dk = akk;
#pragma unroll
        for (int j = 0; j < k; ++j) dk -= L[k * (k - 1) / 2 + j] * L[k * (k - 1) / 2 + j] * dg[j];
        if (ink && !(dk > 1e-13 * akk)) ok = false;
        dg[k] = dk;
        const double inv = 1.0 / dk;
#pragma unroll
        for (int i = k + 1; i < M; ++i) {
            const bool ini = (S >> i) & 1u;
            double v = (ink && ini) ? Q[Sym<M>::at(i, k)] + s : 0.0;
#pragma unroll
            for (int j = 0; j < k; ++j) v -= L[i * (i - 1) / 2 + j] * L[k * (k - 1) / 2 + j] * dg[j];
            L[i * (i - 1) / 2 + k] = v * inv;
        }
    }
    if (!ok) return false;
    double z[M];
#pragma unroll
    for (int i = 0; i < M; ++i) {
        double v = ((S >> i) & 1u) ? 1.0 : 0.0;
#pragma unroll
        for (int j = 0; j < i; ++j) v -= L[i * (i - 1) / 2 + j] * z[j];
        z[i] = v;
    }
#pragma unroll
    for (int i = 0; i < M; ++i) z[i] = z[i] / dg[i];
    double sum = 0.0;
#pragma unroll
    for (int i = M - 1; i >= 0; --i) {
        double v = z[i];
#pragma unroll
        for (int j = i + 1; j < M; ++j) v -= L[j * (j - 1) / 2 + i] * beta[j];
        beta[i] = v;
        sum += v;
    }
    const double inv = 1.0 / sum;
#pragma unroll
    for (int i = 0; i < M; ++i) beta[i] *= inv;
    return sum > 0.0;
}

// Minimum of alpha^T Q alpha over the simplex on the first m (<= M) vertices.  Returns the value
// (squared hull distance) and the weights.
template <int M>
__device__ __forceinline__ double min_norm_point(const double (&Q)[Sym<M>::NP], int m,
                                                 double (&alpha)[M])
{
    double scale = 0.0;
    int i0 = 0;
    double best = kInf;
#pragma unroll
    for (int i = 0; i < M; ++i) {
        alpha[i] = 0.0;
        if (i < m) {
            const double qii = Q[Sym<M>::at(i, i)];
            scale = fmax(scale, qii);
            if (qii < best) { best = qii; i0 = i; }
        }
    }
    if (!(scale > 0.0)) {  // every vertex coincides with the query (or NaN input)
#pragma unroll
        for (int i = 0; i < M; ++i) alpha[i] = (i == 0) ? 1.0 : 0.0;
        return scale == 0.0 ? 0.0 : scale;
    }
    unsigned S = 1u << i0, banned = 0u;
#pragma unroll
    for (int i = 0; i < M; ++i) alpha[i] = (i == i0) ? 1.0 : 0.0;
    const double tol = 1.4210854715202004e-14 * scale;  // 64 eps * scale

    for (int it = 0; it < 3 * M + 8; ++it) {
        double g[M];
        double val = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            double gi = 0.0;
#pragma unroll
            for (int j = 0; j < M; ++j) gi += Q[Sym<M>::at(i, j)] * alpha[j];
            g[i] = gi;
            val += alpha[i] * gi;
        }
        int jb = -1;
        double gmin = kInf;
#pragma unroll
        for (int i = 0; i < M; ++i)
            if (i < m && !(((S | banned) >> i) & 1u) && g[i] < gmin) { gmin = g[i]; jb = i; }
        if (jb < 0 || !(gmin < val - tol)) break;
        S |= 1u << jb;
        for (int mi = 0; mi <= M; ++mi) {
            double beta[M];
            if (!solve_affine<M>(Q, scale, S, beta)) {
                S &= ~(1u << jb);
                banned |= 1u << jb;
                break;
            }
            bool allpos = true;
#pragma unroll
            for (int i = 0; i < M; ++i)
                if (((S >> i) & 1u) && !(beta[i] > 0.0)) allpos = false;
            if (allpos) {
#pragma unroll
                for (int i = 0; i < M; ++i) alpha[i] = ((S >> i) & 1u) ? beta[i] : 0.0;
                break;
            }
            double theta = 1.0;
            int kr = -1;
#pragma unroll
            for (int i = 0; i < M; ++i)
                if (((S >> i) & 1u) && !(beta[i] > 0.0)) {
                    const double den = alpha[i] - beta[i];
                    const double r = den > 0.0 ? alpha[i] / den : 0.0;
                    if (kr < 0 || r < theta) { theta = r; kr = i; }
                }
#pragma unroll
            for (int i = 0; i < M; ++i) {
                const double v = alpha[i] + theta * (beta[i] - alpha[i]);
                alpha[i] = (((S >> i) & 1u) && i != kr) ? v : 0.0;
            }
            S &= ~(1u << kr);
            if (kr == jb) banned |= 1u << jb;
        }
    }
    double val = 0.0;
#pragma unroll
    for (int i = 0; i < M; ++i) {
        double gi = 0.0;
#pragma unroll
        for (int j = 0; j < M; ++j) gi += Q[Sym<M>::at(i, j)] * alpha[j];
        val += alpha[i] * gi;
    }
    return val;
}

// Distance^2 to the AFFINE hull of the first m vertices (hull_distance.py:69-87 affine_hull_distance
// and :38-66 affine_hull_distance_qp: same quantity): the affine minimiser over a maximal affinely
// independent subset, built greedily -- a vertex whose addition collapses a pivot of the lifted
// Gram lies in the affine hull of those already taken and is skipped.
template <int M>
__device__ __forceinline__ double affine_min_norm(const double (&Q)[Sym<M>::NP], int m, double (&alpha)[M])
{
    double scale = 0.0;
#pragma unroll
    for (int i = 0; i < M; ++i) {
        alpha[i] = 0.0;
        if (i < m) scale = fmax(scale, Q[Sym<M>::at(i, i)]);
    }
    if (!(scale > 0.0)) {
#pragma unroll
        for (int i = 0; i < M; ++i) alpha[i] = (i == 0) ? 1.0 : 0.0;
        return scale == 0.0 ? 0.0 : scale;
    }
    unsigned S = 0u;
    double beta[M];
#pragma unroll
    for (int k = 0; k < M; ++k) {
        if (k < m) {
            double trial[M];
            if (solve_affine<M>(Q, scale, S | (1u << k), trial)) {
                S |= 1u << k;
#pragma unroll
                for (int i = 0; i < M; ++i) beta[i] = trial[i];
            }
        }
    }
    if (S == 0u) {   // cannot happen for scale > 0 (a single vertex is always independent)
#pragma unroll
        for (int i = 0; i < M; ++i) alpha[i] = (i == 0) ? 1.0 : 0.0;
        return Q[0];
    }
    double val = 0.0;
#pragma unroll
    for (int i = 0; i < M; ++i) alpha[i] = ((S >> i) & 1u) ? beta[i] : 0.0;
#pragma unroll
    for (int i = 0; i < M; ++i) {
        double gi = 0.0;
#pragma unroll
        for (int j = 0; j < M; ++j) gi += Q[Sym<M>::at(i, j)] * alpha[j];
        val += alpha[i] * gi;
    }
    return val;
}

using f64x4 = __attribute__((ext_vector_type(4))) double;

// Sum v[e] over the 16 lanes of a group and leave entry e = l16 on lane l16: a reduce-scatter in
// four halving steps (15 exchanges instead of the 64 of sixteen butterfly reductions).
__device__ __forceinline__ double reduce_scatter16(const double (&v)[16], int l16)
{
    double t8[8], t4[4], t2[2];
    const bool b8 = l16 & 8, b4 = l16 & 4, b2 = l16 & 2, b1 = l16 & 1;
#pragma unroll
    for (int e = 0; e < 8; ++e) t8[e] = (b8 ? v[e + 8] : v[e]) + __shfl_xor(b8 ? v[e] : v[e + 8], 8, 64);
#pragma unroll
    for (int e = 0; e < 4; ++e) t4[e] = (b4 ? t8[e + 4] : t8[e]) + __shfl_xor(b4 ? t8[e] : t8[e + 4], 4, 64);
#pragma unroll
    for (int e = 0; e < 2; ++e) t2[e] = (b2 ? t4[e + 2] : t4[e]) + __shfl_xor(b2 ? t4[e] : t4[e + 2], 2, 64);
    return (b1 ? t2[1] : t2[0]) + __shfl_xor(b1 ? t2[0] : t2[1], 1, 64);
}

// INDEXED = false: problems are (batch position, bin) pairs whose vertices come from the top-m
// lists.  INDEXED = true: explicit (query sample, compacted vertex index list, count) problems.
template <int M, int WAVES, bool INDEXED>
__global__ __launch_bounds__(64 * WAVES, (M <= 5 ? 4 : 1)) void hull_qp_kernel(QpArgs a, int nprob, const int *xq,
                                                              const int *xhull, const int *xn,
                                                              int xm, double *xdist, double *xalpha, Gate gate)
{
    CHB_GATE(gate);
    constexpr int NP = Sym<M>::NP;
    static_assert(M <= 8, "m > 8 runs on hull_qp16_kernel");
    __shared__ double sQ[WAVES][NP][65];    // (65: the 16 lanes of a group store 16 entries of ONE problem column -- a stride of 64 doubles would put them all on one bank)
    __shared__ int sN[WAVES][64];

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int m = INDEXED ? xm : a.m;
    // active mode: the problems are the listed pairs (device-side count), walked grid-stride
    const int *act = INDEXED ? nullptr : a.active;
    if (act != nullptr) nprob = *a.n_active;
    // a listed (usually short) problem set is spread as thinly as possible -- four problems per wavefront,
    // one pass -- because a wavefront's sixteen passes are a serial chain of memory round trips
    const int npass = act != nullptr ? 1 : 16;
    const int ppw = 4 * npass;

    for (int g0 = (blockIdx.x * WAVES + w) * ppw; g0 < nprob; g0 += (int)gridDim.x * WAVES * ppw) {
    {
        // 16 lanes per problem, lane l16 over the features k = l16 (mod 16): every row is read in
        // full 128-byte lines, each lane accumulates all M (M + 1) / 2 products of its features, and
        // a reduce-scatter leaves one finished Gram entry per lane.  For M <= 8 this beats the
        // 16 x 16 matrix-core tile, most of which (the blocks between different problems) is waste.
        const int grp = lane >> 4, l16 = lane & 15;
        for (int pass = 0; pass < npass; ++pass) {
            const int pl = pass * 4 + grp;
            const int g = g0 + pl;
            const bool valid = g < nprob;
            int n = 0, idm = -1, qid = 0;
            size_t slot = 0;
            if (valid) {
                if (INDEXED) {
                    qid = xq[g];
                    n = xn[g];
                    if (l16 < n) idm = xhull[(size_t)g * m + l16];
                } else {
                    const int pr = act ? act[g] : g;
                    const int pos = a.pos_begin + pr / a.B, c = pr - (pr / a.B) * a.B;
                    qid = a.bq[pos];
                    slot = (size_t)c * a.Kcap + pos;
                    n = a.lists.cnt[slot];
                    if (l16 < n) idm = a.lists.idx[slot * m + l16];
                }
            }
            bool changed = valid;
            if (!INDEXED && a.prev.idx != nullptr && act == nullptr && valid)
                changed = a.prev.cnt[slot] != n || (l16 < n && a.prev.idx[slot * m + l16] != idm);
            const bool doit = ((__ballot(changed) >> (16 * grp)) & 0xFFFFull) != 0ull;
            const double *vrow[M];
#pragma unroll
            for (int v = 0; v < M; ++v) {
                const int idv = __shfl(idm, 16 * grp + v, 64);
                // a missing vertex reads the query row: y = 0, its Gram row/column stays 0 (never used)
                vrow[v] = a.X + (size_t)(idv >= 0 ? idv : qid) * a.Dp;
            }
            if (!valid) continue;
            if (!doit) {   // unchanged vertex list: keep the stored distance
                if (l16 == 0) sN[w][pl] = -1;
                continue;
            }
            const double *xrow = a.X + (size_t)qid * a.Dp;
            double acc[NP];
#pragma unroll
            for (int e = 0; e < NP; ++e) acc[e] = 0.0;
            if (n > 0) {
                // 32 features per step as 16-byte loads (lane l16: features 2 l16, 2 l16 + 1), then the
                // remaining < 32 features one double per lane
                const int Dmain = a.Dp & ~31;
#pragma unroll CHB_QP_UNROLL
                for (int k = 2 * l16; k < Dmain; k += 32) {
                    const double2 xk = *reinterpret_cast<const double2 *>(xrow + k);
                    double2 y[M];
#pragma unroll
                    for (int v = 0; v < M; ++v) {
                        const double2 pv = *reinterpret_cast<const double2 *>(vrow[v] + k);
                        y[v] = double2{pv.x - xk.x, pv.y - xk.y};
                    }
#pragma unroll
                    for (int i = 0; i < M; ++i)
#pragma unroll
                        for (int j = 0; j <= i; ++j) {
                            double acc_ij = acc[Sym<M>::at(i, j)];
                            acc_ij = fma(y[i].x, y[j].x, acc_ij);
                            acc[Sym<M>::at(i, j)] = fma(y[i].y, y[j].y, acc_ij);
                        }
                }
                for (int k = Dmain + l16; k < a.Dp; k += 16) {
                    const double xk = xrow[k];
                    double y[M];
#pragma unroll
                    for (int v = 0; v < M; ++v) y[v] = vrow[v][k] - xk;
#pragma unroll
                    for (int i = 0; i < M; ++i)
#pragma unroll
                        for (int j = 0; j <= i; ++j) acc[Sym<M>::at(i, j)] = fma(y[i], y[j], acc[Sym<M>::at(i, j)]);
                }
            }
#pragma unroll
            for (int e0 = 0; e0 < NP; e0 += 16) {
                double v16[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) v16[e] = e0 + e < NP ? acc[e0 + e] : 0.0;
                const double r = reduce_scatter16(v16, l16);
                if (e0 + l16 < NP) sQ[w][e0 + l16][pl] = r;
            }
            if (l16 == 0) sN[w][pl] = n;
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's LDS writes have landed

    const int g = g0 + lane;
    if (lane >= ppw || g >= nprob) continue;
    double Q[NP];
#pragma unroll
    for (int e = 0; e < NP; ++e) Q[e] = sQ[w][e][lane];
    const int n = sN[w][lane];
    if (n == -1) continue;   // unchanged vertex list: a.dist already holds this distance
    double alpha[M];
    double dist;
    if (n <= 0) {
        dist = kInf;
#pragma unroll
        for (int v = 0; v < M; ++v) alpha[v] = 0.0;
    } else {
        const double val = a.metric == 0 ? min_norm_point<M>(Q, n, alpha) : affine_min_norm<M>(Q, n, alpha);
        dist = sqrt(fmax(val, 0.0));
    }
    if (INDEXED) {
        xdist[g] = dist;
        if (xalpha) {
#pragma unroll
            for (int v = 0; v < M; ++v)
                if (v < m) xalpha[(size_t)g * m + v] = alpha[v];
        }
    } else {
        const int pr = act ? act[g] : g;
        const int pos = a.pos_begin + pr / a.B, c = pr - (pr / a.B) * a.B;
        a.dist[(size_t)pos * a.B + c] = dist;
    }
    }   // grid-stride loop
}

// ---------------------------------------------------------------------------------------------
// Fused selection + hull distance (FusedArgs in chb_internal.h).  16 lanes per (position, bin) pair as in
// hull_qp_kernel, but over the pair's SHORTLIST (<= C candidates): one sweep over the candidate rows
// accumulates the full C x C shifted Gram.  Its diagonal holds the squared distances of the candidates
// (fp64, fused sums: within ~D eps of the exact value), which rank them; the m x m block of the m
// nearest is the Gram of the hull QP -- no row is read twice.
// Exactness of the selected SET: cdist's (sequentially rounded distance, index) order and this
// kernel's order can only disagree between candidates whose squared distances agree to ~1e-14
// relative; a pair whose m-th and (m+1)-th candidates are closer than kTieRel (1e-11) relative is
// not decided here but handed to the exact path (`slow`).  The ORDER inside the selected set may
// differ from cdist's on such near-ties inside the set; the hull distance does not depend on it
// beyond rounding.
constexpr double kTieRel = 1e-11;

// OFF32: the rows are addressed as 32-bit byte offsets from the (uniform) base of X -- half the registers of
// 64-bit row pointers, which is what keeps the kernel free of spills at 3 wavefronts per SIMD (the launcher picks
// it when the sample matrix is smaller than 4 GiB).
template <int CC, int NR, bool OFF32>
__device__ __forceinline__ void gram_rows(const double *X, int Dp, int Dk, int qid, int idm, int gbase, int l16,
                                          double (&r)[NR])
{   // (Dp: row stride -- whole 128-byte lines; Dk: the columns swept, D rounded up to 8: the padding beyond is never read)
    constexpr int NP = CC * (CC + 1) / 2;
    static_assert(NP <= 16 * NR, "result registers");
    const char *Xb = reinterpret_cast<const char *>(X);
    const double *xrow = X + (size_t)qid * Dp;
    const unsigned xoff = (unsigned)qid * (unsigned)Dp * 8u;
    const double *vrow[OFF32 ? 1 : CC];
    unsigned voff[OFF32 ? CC : 1];
#pragma unroll
    for (int v = 0; v < CC; ++v) {
        const int idv = __shfl(idm, gbase + v, 64);
        // a missing candidate reads the query row: y = 0, its Gram row / column stays 0 (never used)
        if (OFF32) voff[v] = (unsigned)(idv >= 0 ? idv : qid) * (unsigned)Dp * 8u;
        else vrow[v] = X + (size_t)(idv >= 0 ? idv : qid) * Dp;
    }
    auto xat = [&](int k) { return OFF32 ? reinterpret_cast<const double *>(Xb + (size_t)(xoff + 8u * (unsigned)k)) : xrow + k; };
    auto vat = [&](int v, int k) {
        return OFF32 ? reinterpret_cast<const double *>(Xb + (size_t)(voff[OFF32 ? v : 0] + 8u * (unsigned)k)) : vrow[OFF32 ? 0 : v] + k;
    };
    double acc[NP];
#pragma unroll
    for (int e = 0; e < NP; ++e) acc[e] = 0.0;
    const int Dmain = Dk & ~31;
#pragma unroll 1
    for (int k = 2 * l16; k < Dmain; k += 32) {
        const double2 xk = *reinterpret_cast<const double2 *>(xat(k));
        double2 y[CC];
#pragma unroll
        for (int v = 0; v < CC; ++v) {
            const double2 pv = *reinterpret_cast<const double2 *>(vat(v, k));
            y[v] = double2{pv.x - xk.x, pv.y - xk.y};
        }
#pragma unroll
        for (int i = 0; i < CC; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                double t = acc[i * (i + 1) / 2 + j];
                t = fma(y[i].x, y[j].x, t);
                acc[i * (i + 1) / 2 + j] = fma(y[i].y, y[j].y, t);
            }
    }
    for (int k = Dmain + l16; k < Dk; k += 16) {
        const double xk = *xat(k);
        double y[CC];
#pragma unroll
        for (int v = 0; v < CC; ++v) y[v] = *vat(v, k) - xk;
#pragma unroll
        for (int i = 0; i < CC; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) acc[i * (i + 1) / 2 + j] = fma(y[i], y[j], acc[i * (i + 1) / 2 + j]);
    }
#pragma unroll
    for (int t = 0; t < NR; ++t) {
        double v16[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) v16[e] = 16 * t + e < NP ? acc[16 * t + e] : 0.0;
        r[t] = 16 * t < NP ? reduce_scatter16(v16, l16) : 0.0;
    }
}

#ifndef CHB_FUSED_C
#define CHB_FUSED_C 7
#endif
#ifndef CHB_FUSED_OCC
#define CHB_FUSED_OCC 3
#endif
#ifndef CHB_FUSED_WIDE
#define CHB_FUSED_WIDE 16
#endif
constexpr int kFusedWide = CHB_FUSED_WIDE;   // widest shortlist the fused kernel takes (one candidate per lane of a group)

// squared distances only: candidate v (v < 8) of every 16-lane group is the one held by lane gbase + v0 + v;
// on return lane l16 < 8 holds the squared distance of candidate v0 + l16
template <bool OFF32>
__device__ __forceinline__ double diag_rows8(const double *X, int Dp, int Dk, int qid, int idm, int gbase, int l16, int v0)
{
    const char *Xb = reinterpret_cast<const char *>(X);
    const double *xrow = X + (size_t)qid * Dp;
    const unsigned xoff = (unsigned)qid * (unsigned)Dp * 8u;
    const double *vrow[OFF32 ? 1 : 8];
    unsigned voff[OFF32 ? 8 : 1];
#pragma unroll
    for (int v = 0; v < 8; ++v) {
        const int idv = __shfl(idm, gbase + v0 + v, 64);
        if (OFF32) voff[v] = (unsigned)(idv >= 0 ? idv : qid) * (unsigned)Dp * 8u;
        else vrow[v] = X + (size_t)(idv >= 0 ? idv : qid) * Dp;
    }
    auto xat = [&](int k) { return OFF32 ? reinterpret_cast<const double *>(Xb + (size_t)(xoff + 8u * (unsigned)k)) : xrow + k; };
    auto vat = [&](int v, int k) {
        return OFF32 ? reinterpret_cast<const double *>(Xb + (size_t)(voff[OFF32 ? v : 0] + 8u * (unsigned)k)) : vrow[OFF32 ? 0 : v] + k;
    };
    double acc[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0;
    const int Dmain = Dk & ~31;
#pragma unroll 1
    for (int k = 2 * l16; k < Dmain; k += 32) {
        const double2 xk = *reinterpret_cast<const double2 *>(xat(k));
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            const double2 pv = *reinterpret_cast<const double2 *>(vat(v, k));
            const double dx = pv.x - xk.x, dy = pv.y - xk.y;
            acc[v] = fma(dy, dy, fma(dx, dx, acc[v]));
        }
    }
    for (int k = Dmain + l16; k < Dk; k += 16) {
        const double xk = *xat(k);
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            const double d = *vat(v, k) - xk;
            acc[v] = fma(d, d, acc[v]);
        }
    }
    return reduce_scatter16(acc, l16);
}

// Work order of the m <= 5 fused kernel.  STRIPED (a.stripe != 0, B >= 16): consecutive workgroups go round-robin over the
// 8 XCDs, each with its own L2; workgroup b (XCD b % 8) only takes pairs of the bins c = b % 8 (mod 8), walked position-major
// over that bin subset (64 consecutive pairs per wavefront = 64 / nbx positions x the XCD's nbx bins).  The candidate rows an
// XCD gathers then come from an eighth of the bins -- its 4 MiB L2 holds a far larger share of them -- while a query row is
// still shared by the nbx pairs of a position.  Otherwise: position-major over all bins (one position x 64 bins per
// wavefront at B = 64).
__device__ __forceinline__ bool fused_pair_of(const FusedArgs &a, int g0, int lane, int nprob, int &pos, int &c)
{
    if (!a.stripe) {
        const int g = g0 + lane;
        pos = a.pos_begin + g / a.B;
        c = g - (g / a.B) * a.B;
        return g < nprob;
    }
    const int x = (int)(blockIdx.x & 7);
    const int nbx = (a.B - x + 7) >> 3;                    // bins c = x (mod 8)
    const int t = g0 + lane;                               // pair number inside the XCD's list
    const int pq = t / nbx;
    pos = a.pos_begin + pq;
    c = x + 8 * (t - pq * nbx);
    return pq < a.pos_end - a.pos_begin;
}

template <int M, int C, int WAVES, bool OFF32>
__global__ __launch_bounds__(64 * WAVES, CHB_FUSED_OCC) void hull_select_qp_kernel(FusedArgs a, int nprob, Gate gate)
{
    CHB_GATE(gate);
    constexpr int NPM = Sym<M>::NP;            // Gram entries of the hull QP
    constexpr int NPC = C * (C + 1) / 2;       // ... of the candidate Gram
    constexpr int NR = (NPC + 15) / 16;
    constexpr int NCLS = C - M + 1;            // classes of the one-sweep form: widths M, M + 1, ..., C
    static_assert(C <= 8 && M <= C && C - M <= 3, "candidate Gram in registers");
    __shared__ double sQ[WAVES][NPM][65];   // (65: the 16 lanes of a group store 16 entries of ONE problem column -- a stride of 64 doubles would put them all on one bank)
    __shared__ double sT[WAVES][4][NR * 16];   // candidate Gram of the group's pair
    __shared__ double sTh[WAVES][4][2];        // squared distances of ranks m-1 and m
    __shared__ int sSel[WAVES][4][16];         // candidate slot of each rank
    __shared__ int sN[WAVES][64];
    __shared__ int sOrd[WAVES][64];            // problem (lane) of each work position, narrow shortlists first
    __shared__ int sSlowN, sSlowBase;

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int grp = lane >> 4, l16 = lane & 15, gbase = lane & 48;
    const int m = a.m;
    const int Dk = (a.D + 7) & ~7;   // columns swept (the rows are padded to whole 128-byte lines beyond)
    if (threadIdx.x == 0) sSlowN = 0;
    __syncthreads();
    // (striped: the wavefront's place in its XCD's own pair list; nprob is then the length of the longest of the 8 lists)
    const int g0 = ((a.stripe ? (int)(blockIdx.x >> 3) : (int)blockIdx.x) * WAVES + w) * 64;
    unsigned long long slowmask = 0ull;        // problems of this wavefront left to the exact path
    int pos = 0, c = 0;
    bool valid = false;

    if (g0 < nprob) {
        // ---- classification, one problem per lane
        valid = fused_pair_of(a, g0, lane, nprob, pos, c);
        int nb = 0, nu = 0, qid = 0;
        size_t slot = 0;
        bool changed = valid;
        if (valid) {
            qid = a.bq[pos];
            slot = (size_t)c * a.Kcap + pos;
            nb = a.cand_cnt[slot];
            if (a.candu != nullptr) nu = a.candu_cnt[slot];
            // unchanged set of batch-entry candidates (the base candidates are fixed during a batch): keep
            if (a.candp != nullptr && a.candp_cnt[slot] == nu) {
                if (nu == 0) changed = false;
                else if (nu <= 8) {
                    bool same = true;
                    for (int i = 0; i < nu; ++i) {
                        const int u = a.candu[slot * kCandCapU + i];
                        bool f = false;
                        for (int j = 0; j < nu; ++j) f = f || a.candp[slot * kCandCapU + j] == u;
                        same = same && f;
                    }
                    changed = !same;
                }
            }
        }
        if (a.short_cnt != nullptr) {
            // the shortlist stage's contract (a short list would be a wrong hull with no error, a wild index a fault)
            bool sh = false;
            if (valid) sh = nb < min(m, a.bin_size != nullptr ? a.bin_size[c] : a.bin_ptr[c + 1] - a.bin_ptr[c]);
            const unsigned long long shm = __ballot(sh);
            if (shm != 0ull && lane == 0) atomicAdd(a.short_cnt, __popcll(shm));
        }
        const int n = nb + nu;
        // class 0: nothing to compute here.  Classes 1 .. NCLS: ONE sweep over max(n, M) = M + k - 1 candidate
        // rows (full candidate Gram).  Class NCLS + 1 (C < n <= 16): a distance sweep over all candidates,
        // then a second sweep over the m selected rows (just read: they come out of L1 / L2).
        int cls = 0;
        if (changed && n > 0) {
            if (n <= C) cls = n <= M ? 1 : n - M + 1;
            else if (n <= kFusedWide) cls = NCLS + 1;
        }
        slowmask = __ballot(changed && n > kFusedWide);
        sN[w][lane] = (changed && n == 0) ? 0 : -1;   // -1: distance kept / written later
        int before = 0, nwork = 0, nnarrow = 0;
#pragma unroll
        for (int k = 1; k <= NCLS + 1; ++k) {
            const unsigned long long mk = __ballot(cls == k);
            if (cls == k) before = nwork + __popcll(mk & ((1ull << lane) - 1ull));
            nwork += __popcll(mk);
            if (k == NCLS) nnarrow = nwork;
        }
        if (cls > 0) sOrd[w][before] = lane;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xC07F);

        // ---- 16 lanes per problem, four problems per pass, narrow shortlists first.
        // (Shuffles stay outside any condition: a lane only delivers data while it is active.)
        for (int p0 = 0; p0 < nnarrow; p0 += 4) {
            const int op = p0 + grp;
            const bool has = op < nnarrow;
            const int pl = has ? sOrd[w][op] : 0;
            const int n_s = __shfl(n, pl, 64);
            const int n_g = has ? n_s : 0;
            const int nb_g = __shfl(nb, pl, 64);
            const int qid_g = __shfl(qid, pl, 64);
            const size_t slot_g = (size_t)__shfl(c, pl, 64) * a.Kcap + (size_t)__shfl(pos, pl, 64);
            int idm = -1;
            if (l16 < n_g)
                idm = l16 < nb_g ? a.cand[slot_g * kCandCap + l16] : a.candu[slot_g * kCandCapU + (l16 - nb_g)];
            if ((unsigned)idm >= (unsigned)a.n_samples && idm != -1) {   // never a row address from a wild index
                if (a.short_cnt != nullptr) atomicAdd(a.short_cnt, 1);
                idm = -1;
            }
            // the widest shortlist of the pass (the last problem: they are sorted)
            const int cw = __builtin_amdgcn_readfirstlane(__shfl(n, sOrd[w][min(p0 + 3, nnarrow - 1)], 64));
            double r[NR];
            if (cw <= M) gram_rows<M, NR, OFF32>(a.X, a.Dp, Dk, qid_g, idm, gbase, l16, r);
            else if (C >= M + 1 && cw == M + 1) gram_rows<(C >= M + 1 ? M + 1 : M), NR, OFF32>(a.X, a.Dp, Dk, qid_g, idm, gbase, l16, r);
            else if (C >= M + 2 && cw == M + 2) gram_rows<(C >= M + 2 ? M + 2 : M), NR, OFF32>(a.X, a.Dp, Dk, qid_g, idm, gbase, l16, r);
            else gram_rows<C, NR, OFF32>(a.X, a.Dp, Dk, qid_g, idm, gbase, l16, r);
            bool tie = false;
            if (has && n_g <= m) {
                // every candidate is a hull vertex: the candidate Gram is the hull's (any vertex order)
#pragma unroll
                for (int t = 0; t < NR; ++t)
                    if (16 * t + l16 < NPM) sQ[w][16 * t + l16][pl] = r[t];
                if (l16 == 0) sN[w][pl] = n_g;
            } else if (has) {
#pragma unroll
                for (int t = 0; t < NR; ++t) sT[w][grp][16 * t + l16] = r[t];
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_s_waitcnt(0xC07F);
                // rank of my candidate by (squared distance, index); the ids come through LDS as well
                // (no shuffle inside a condition)
                if (l16 < C) sSel[w][grp][8 + l16] = idm;
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_s_waitcnt(0xC07F);
                const double sv = l16 < n_g ? sT[w][grp][l16 * (l16 + 3) / 2] : kInf;
                int rank = 0;
#pragma unroll
                for (int u = 0; u < C; ++u) {
                    const double su = sT[w][grp][u * (u + 3) / 2];
                    const int iu = sSel[w][grp][8 + u];
                    rank += (u < n_g && u != l16 && (su < sv || (su == sv && iu < idm))) ? 1 : 0;
                }
                if (l16 < n_g) {
                    if (rank < m) sSel[w][grp][rank] = l16;
                    if (rank == m - 1) sTh[w][grp][0] = sv;
                    if (rank == m) sTh[w][grp][1] = sv;
                }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_s_waitcnt(0xC07F);
                const double t0 = sTh[w][grp][0], t1 = sTh[w][grp][1];
                tie = !(t1 - t0 > kTieRel * t1);   // (also catches NaN)
                if (!tie) {
#pragma unroll
                    for (int t = 0; 16 * t < NPM; ++t) {
                        const int e = 16 * t + l16;   // entry (i, j) of the hull Gram, ranks i >= j
                        int i = 0;
#pragma unroll
                        for (int q = 1; q < M; ++q) i += (q * (q + 1) / 2 <= e) ? 1 : 0;
                        const int j = e - i * (i + 1) / 2;
                        if (e < NPM) {
                            double val = 0.0;
                            if (i < m) {   // (j <= i)
                                const int vi = sSel[w][grp][i], vj = sSel[w][grp][j];
                                const int hi = vi > vj ? vi : vj, lo = vi > vj ? vj : vi;
                                val = sT[w][grp][hi * (hi + 1) / 2 + lo];
                            }
                            sQ[w][e][pl] = val;
                        }
                    }
                    if (l16 == 0) sN[w][pl] = m;
                }
            }
            if (l16 == 0 && tie) slowmask |= 1ull << pl;
        }

        // ---- wide shortlists (C < n <= 16): distances first, then the Gram of the m selected rows
        for (int p0 = nnarrow; p0 < nwork; p0 += 4) {
            const int op = p0 + grp;
            const bool has = op < nwork;
            const int pl = has ? sOrd[w][op] : 0;
            const int n_s = __shfl(n, pl, 64);
            const int n_g = has ? n_s : 0;
            const int nb_g = __shfl(nb, pl, 64);
            const int qid_g = __shfl(qid, pl, 64);
            const size_t slot_g = (size_t)__shfl(c, pl, 64) * a.Kcap + (size_t)__shfl(pos, pl, 64);
            int idm = -1;
            if (l16 < n_g)
                idm = l16 < nb_g ? a.cand[slot_g * kCandCap + l16] : a.candu[slot_g * kCandCapU + (l16 - nb_g)];
            if ((unsigned)idm >= (unsigned)a.n_samples && idm != -1) {   // never a row address from a wild index
                if (a.short_cnt != nullptr) atomicAdd(a.short_cnt, 1);
                idm = -1;
            }
            int nmax = n_g;
            nmax = max(nmax, __shfl_xor(nmax, 16, 64));
            nmax = max(nmax, __shfl_xor(nmax, 32, 64));
            nmax = __builtin_amdgcn_readfirstlane(nmax);
            double sv = diag_rows8<OFF32>(a.X, a.Dp, Dk, qid_g, idm, gbase, l16, 0);
            if (nmax > 8) {
                const double s_hi = diag_rows8<OFF32>(a.X, a.Dp, Dk, qid_g, idm, gbase, l16, 8);
                const double moved = __shfl(s_hi, gbase + (l16 & 7), 64);
                sv = l16 < 8 ? sv : moved;
            }
            if (l16 >= n_g) sv = kInf;
            // rank by (squared distance, index) among the group's n_g candidates
            int rank = 0;
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const double su = __shfl(sv, gbase + u, 64);
                const int iu = __shfl(idm, gbase + u, 64);
                rank += (u < n_g && u != l16 && (su < sv || (su == sv && iu < idm))) ? 1 : 0;
            }
            if (has && l16 < n_g) {
                if (rank < m) sSel[w][grp][rank] = idm;
                if (rank == m - 1) sTh[w][grp][0] = sv;
                if (rank == m) sTh[w][grp][1] = sv;
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xC07F);
            bool tie = false;
            int idm2 = -1;
            if (has) {   // (n_g > C >= m: both thresholds exist)
                const double t0 = sTh[w][grp][0], t1 = sTh[w][grp][1];
                tie = !(t1 - t0 > kTieRel * t1);
                if (l16 < m) idm2 = sSel[w][grp][l16];
            }
            double r[NR];
            gram_rows<M, NR, OFF32>(a.X, a.Dp, Dk, qid_g, idm2, gbase, l16, r);
            if (has && !tie) {
#pragma unroll
                for (int t = 0; t < NR; ++t)
                    if (16 * t + l16 < NPM) sQ[w][16 * t + l16][pl] = r[t];
                if (l16 == 0) sN[w][pl] = m;
            }
            if (l16 == 0 && tie) slowmask |= 1ull << pl;
        }
    }
    unsigned long long sm = slowmask;   // classification bits on every lane, tie bits on lanes 0, 16, 32, 48
    sm |= __shfl_xor(sm, 16, 64);
    sm |= __shfl_xor(sm, 32, 64);
    sm = __shfl(sm, 0, 64);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's LDS writes have landed

    // ---- phase 2: one problem per lane
    {
        const int n = valid ? sN[w][lane] : -1;
        if (valid && n != -1 && !((sm >> lane) & 1ull)) {   // else: distance kept, or written by the exact path
            double Q[NPM];
#pragma unroll
            for (int e = 0; e < NPM; ++e) Q[e] = sQ[w][e][lane];
            double alpha[M];
            double dist;
            if (n <= 0) {
                dist = kInf;
            } else {
                const double val = a.metric == 0 ? min_norm_point<M>(Q, n, alpha) : affine_min_norm<M>(Q, n, alpha);
                dist = sqrt(fmax(val, 0.0));
            }
            a.dist[(size_t)pos * a.B + c] = dist;
        }
    }

    // ---- the exact path's work list: one device-scope atomic per workgroup (at the very end: the barriers
    // below must not hold a wavefront back between its sweeps and its solver)
    const int nsl = __popcll(sm);
    int wbase = 0;
    if (lane == 0 && nsl > 0) wbase = atomicAdd(&sSlowN, nsl);
    __syncthreads();
    if (threadIdx.x == 0) sSlowBase = sSlowN > 0 ? atomicAdd(a.n_slow, sSlowN) : 0;
    __syncthreads();
    wbase = __shfl(wbase, 0, 64) + sSlowBase;
    if (nsl > 0 && ((sm >> lane) & 1ull)) {
        const int before = __popcll(sm & ((1ull << lane) - 1ull));
        a.slow[wbase + before] = (pos - a.pos_begin) * a.B + c;   // (position-major pair index, whatever the work order)
    }
}

// ---------------------------------------------------------------------------------------------
// m > 8 (the reference's function default is num_neighbors = 15, algorithm.py:17): a 16 x 16 Gram and
// the solver's factor do not fit one lane's registers (the per-lane form above spills ~2 KB), so a
// problem is spread over 16 lanes -- lane i owns vertex i, i.e. row i of Q and its weight alpha_i.
//   phase 1: one v_mfma_f64_16x16x4_f64 tile per problem (16 vertices fill it exactly), four problems
//     per wavefront one after the other, Gram rows handed to their lanes through LDS;
//   phase 2: the same Wolfe iteration as min_norm_point<M>, with the reductions (value, entering
//     vertex, ratio test) as 16-lane butterflies and the affine sub-problem solved through an
//     incrementally maintained inverse with one matrix row per lane (see Inv16).
// Groups follow their own control flow (all branches are uniform within a group of 16 lanes).

// Reductions over a 16-lane group = one DPP row: rotations inside the row are VALU moves (row_ror), no LDS
// crossbar and no lgkmcnt wait.  The rotation butterfly (8, 4, 2, 1) adds the same pairs as the xor butterfly at
// every level (the partial results are periodic), so every lane ends with the bit-identical value.
template <int N>
__device__ __forceinline__ int row_ror_i32(int v)
{
    // (every lane of the row is written: no `old` operand to initialise)
    return __builtin_amdgcn_mov_dpp(v, 0x120 + N, 0xF, 0xF, false);
}
template <int N>
__device__ __forceinline__ double row_ror_f64(double v)
{
    const int lo = row_ror_i32<N>(__double2loint(v)), hi = row_ror_i32<N>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double group_sum16(double v)
{
    v += row_ror_f64<8>(v);
    v += row_ror_f64<4>(v);
    v += row_ror_f64<2>(v);
    v += row_ror_f64<1>(v);
    return v;
}
__device__ __forceinline__ double group_max16(double v)
{
    v = fmax(v, row_ror_f64<8>(v));
    v = fmax(v, row_ror_f64<4>(v));
    v = fmax(v, row_ror_f64<2>(v));
    v = fmax(v, row_ror_f64<1>(v));
    return v;
}
__device__ __forceinline__ double group_min16(double v)
{
    v = fmin(v, row_ror_f64<8>(v));
    v = fmin(v, row_ror_f64<4>(v));
    v = fmin(v, row_ror_f64<2>(v));
    v = fmin(v, row_ror_f64<1>(v));
    return v;
}
// smallest key, ties to the lowest lane; key = +inf (or NaN) everywhere gives idx = -1
__device__ __forceinline__ void group_argmin16(double key, int lane, double &kmin, int &idx)
{
    kmin = group_min16(key);
    const unsigned hit = (unsigned)(__ballot(key == kmin && key < kInf) >> (lane & 48)) & 0xFFFFu;
    idx = hit ? __ffs(hit) - 1 : -1;
}

// The affine sub-problem (Q_SS + s 11^T) b = 1, beta = b / sum(b) is solved through the explicit
// inverse H of the lifted support Gram, kept up to date as vertices enter and leave (lane i holds
// row i of H, zeros outside the support).  Entering / leaving is a bordering / Schur update whose
// broadcasts are independent of each other -- a few LDS rounds deep, where an elimination from
// scratch is a 16-step dependent chain.  The pivot of the update is the Schur complement delta, the
// same quantity whose collapse marks an affinely dependent support in solve_affine<M>.
// developer variants (tools/m15_probe.py; tools/build_variant.sh <name> "-DCHB_DEV_QP16_STAT" or "-DCHB_DEV_CLK"):
// CHB_DEV_QP16_STAT = iteration statistics of the 16-lane solver (its atomics distort timings), CHB_DEV_CLK = cycle
// stamps of the fused 16-lane kernel's phases only
#if defined(CHB_DEV_QP16_STAT) || defined(CHB_DEV_CLK)
__device__ unsigned long long g_qp16_stats[16];
#endif
#ifdef CHB_DEV_CLK
__device__ unsigned long long g_qp16_clk[131072][8];   // per-wavefront cycle stamps of the last launch (no atomics)
#endif
#ifdef CHB_DEV_QP16_STAT
#define QP16_STAT(i, v) do { if (l16 == 0) atomicAdd(&g_qp16_stats[i], (unsigned long long)(v)); } while (0)
#else
#define QP16_STAT(i, v) do { } while (0)
#endif
#ifdef CHB_DEV_CLK
#define QP16_CLK(t) const unsigned long long t = __builtin_readcyclecounter()
#define QP16_CLK_ADD(i, v) do { if (lane == 0) atomicAdd(&g_qp16_stats[i], (unsigned long long)(v)); } while (0)
#define QP16_CLK_ACC(var, v) do { var += (unsigned long long)(v); } while (0)
#else
#define QP16_CLK(t) do { } while (0)
#define QP16_CLK_ADD(i, v) do { } while (0)
#define QP16_CLK_ACC(var, v) do { } while (0)
#endif

// 1 / x for normal, well-scaled x (the 16-lane solver works on a Gram normalised to O(1)): hardware estimate + two
// Newton steps, within an ulp or two of the correctly rounded quotient
__device__ __forceinline__ double fast_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

struct Inv16 {
    double H[16];
};

// Every lane of a 16-lane group publishes one value; afterwards out[j] is lane j's.  Through a 16-double LDS row
// of the group (one ds_write_b64 + eight ds_read_b128, broadcast reads) instead of sixteen shuffles of a double
// (32 ds_bpermute_b32): the LDS crossbar is what the 16-lane solver runs on.
__device__ __forceinline__ void group_allgather16(double *sv, int l16, double v, double (&out)[16])
{
    sv[l16] = v;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
    for (int j = 0; j < 16; j += 2) {
        const double2 t = *reinterpret_cast<const double2 *>(sv + j);
        out[j] = t.x; out[j + 1] = t.y;
    }
    __builtin_amdgcn_wave_barrier();
}

#ifndef CHB_QP16_OCC
#define CHB_QP16_OCC 3   // wavefronts per SIMD the 16-lane solver is compiled for (168 VGPRs)
#endif
static_assert(kFusedMaxDp <= 16 * 18, "the query row is staged in a 16 x kQ16Ld tile");
constexpr int kQ16Ld = 18;   // row stride of the 16 x 16 Gram tile in LDS (16-byte aligned rows)

// vertex v enters: false (and no change) when it is affinely dependent on the support.
// Qt = the group's LIFTED Gram tile in LDS (Q + s, row stride kQ16Ld), sv = its 16-double exchange row.
// Rows and columns of H outside the support are zero, so the products below need no support mask.
// small (out): the accepted pivot was below kSmallPivot x the vertex's own lifted norm -- the support is ill-conditioned
// (cond ~ 1 / that ratio), see solve16's rebuild.  reject: the pivot below which the vertex counts as dependent.
constexpr double kSmallPivot = 1e-4;
__device__ __forceinline__ bool inv16_insert(Inv16 &I, const double *Qt, double *sv, unsigned &S, int v, int l16, bool &small,
                                             double reject = 1e-13)
{
    // a = lifted row v (a broadcast read); u = H a
    double u = 0.0;
#pragma unroll
    for (int j = 0; j < 16; j += 2) {
        const double2 t = *reinterpret_cast<const double2 *>(Qt + v * kQ16Ld + j);
        u = fma(I.H[j], t.x, u);
        u = fma(I.H[j + 1], t.y, u);
    }
    const double a_own = Qt[v * kQ16Ld + l16], avv = Qt[v * kQ16Ld + v];
    double delta = avv - group_sum16(a_own * u);
    double ug[16];
    if (!(delta > 1e-6 * avv)) {
#ifdef CHB_DEV_QP16_STAT
        if (l16 == 0) atomicAdd(&g_qp16_stats[4], 1ull);
#endif
        // A small pivot is decided after one step of iterative refinement against the ORIGINAL rows
        // of Q: the stored inverse carries an error of eps * cond, which must not leak into the
        // test below (a dependent vertex has to come out at delta ~ eps * avv, as the Schur
        // complement of a factorisation does).
        const bool in = (S >> l16) & 1u;
        group_allgather16(sv, l16, u, ug);
        double r = a_own;
#pragma unroll
        for (int j = 0; j < 16; ++j) r = fma(-Qt[l16 * kQ16Ld + j], ug[j], r);   // u is zero outside the support
        r = in ? r : 0.0;
        group_allgather16(sv, l16, r, ug);
        double du = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) du = fma(I.H[j], ug[j], du);
        u += du;
        delta = avv - group_sum16(a_own * u);
    }
    small = !(delta > kSmallPivot * avv);
    if (!(delta > reject * avv)) return false;
    const double inv = fast_rcp(delta);
    // bordering: H' = H + w w^T / delta with w = (u on the support, -1 at v, 0 elsewhere)
    const double w = l16 == v ? -1.0 : u;
    group_allgather16(sv, l16, w, ug);
    const double f = w * inv;
#pragma unroll
    for (int j = 0; j < 16; ++j) I.H[j] = fma(f, ug[j], I.H[j]);
    S |= 1u << v;
    return true;
}

__device__ __forceinline__ bool inv16_insert(Inv16 &I, const double *Qt, double *sv, unsigned &S, int v, int l16)
{
    bool small;
    return inv16_insert(I, Qt, sv, S, v, l16, small);
}

// vertex r (in S) leaves
__device__ __forceinline__ void inv16_remove(Inv16 &I, double *sv, unsigned &S, int r, int l16)
{
    // row r of H = column r (H is symmetric): every lane contributes its own entry H[l16][r]
    double hrr = 0.0, own = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) own = (j == r) ? I.H[j] : own;
    double hr[16];
    group_allgather16(sv, l16, own, hr);
#pragma unroll
    for (int j = 0; j < 16; ++j) hrr = (j == r) ? hr[j] : hrr;
    const double f = own / hrr;
#pragma unroll
    for (int j = 0; j < 16; ++j) I.H[j] = (l16 == r || j == r) ? 0.0 : fma(-f, hr[j], I.H[j]);
    S &= ~(1u << r);
}

// beta_l16 of the affine minimiser on the current support (0 outside); false if the weights do not sum > 0
__device__ __forceinline__ bool inv16_beta(const Inv16 &I, double &beta)
{
    double b = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) b += I.H[j];
    const double sum = group_sum16(b);
    beta = b * fast_rcp(sum);
    return sum > 0.0;
}

// phase 2 of the 16-lane kernels: the hull (metric 0) or affine-hull distance SQUARED of the group's problem.
// Qt = the group's plain shifted Gram tile in LDS (rows / columns >= n finite, e.g. zero), sv = its exchange row;
// lane l16 owns vertex l16 (n <= 16 vertices, n > 0) and returns its weight in `alpha`.
__device__ __forceinline__ double solve16(double *Qt, double *sv, int n, int metric, int lane, double &alpha)
{
    const int l16 = lane & 15;
    double Qr[16];
#pragma unroll
    for (int j = 0; j < 16; j += 2) {
        const double2 t = *reinterpret_cast<const double2 *>(Qt + l16 * kQ16Ld + j);
        Qr[j] = t.x; Qr[j + 1] = t.y;
    }
    double ag[16];   // gathered weights
    const bool mine = l16 < n;
    double val = 0.0;
    alpha = 0.0;
    const double diag = Qt[l16 * kQ16Ld + l16];
    const double scale0 = group_max16(mine ? diag : 0.0);
    // The problem is normalised by a power of two (exact): the largest squared distance becomes `scale` in
    // [0.5, 1), so thresholds and reciprocals see O(1) numbers whatever the units of the data.
    int ex = 0;
    if (scale0 > 0.0 && scale0 < kInf) (void)frexp(scale0, &ex);
    ex = ex < -1000 ? -1000 : (ex > 1000 ? 1000 : ex);
    const double dn = ldexp(1.0, -ex);
    const double scale = scale0 * dn;
#pragma unroll
    for (int j = 0; j < 16; ++j) Qr[j] *= dn;
    // the tile in LDS becomes the lifted Gram Q + scale (what inv16_insert reads); Qr keeps the plain rows
#pragma unroll
    for (int j = 0; j < 16; j += 2)
        *reinterpret_cast<double2 *>(Qt + l16 * kQ16Ld + j) = double2{Qr[j] + scale, Qr[j + 1] + scale};
    __builtin_amdgcn_wave_barrier();
    if (!(scale > 0.0)) {   // every vertex coincides with the query (or NaN input)
        alpha = l16 == 0 ? 1.0 : 0.0;
        val = scale == 0.0 ? 0.0 : scale;
    } else if (metric == 0) {
        double best;
        int i0;
        group_argmin16(mine ? diag * dn : kInf, lane, best, i0);
        unsigned S = 0u, banned = 0u;
        Inv16 I;
#pragma unroll
        for (int j = 0; j < 16; ++j) I.H[j] = 0.0;
        (void)inv16_insert(I, Qt, sv, S, i0, l16);   // a single vertex is always independent
        alpha = l16 == i0 ? 1.0 : 0.0;
        // Round 5: the explicit inverse is only as good as the supports it has been through.  A vertex that enters with a
        // pivot of 1e-7 of its norm (four nearly coplanar points in three dimensions: a near-duplicate contig among the
        // neighbours) leaves H with entries of 1e7 and a relative error of eps * cond; the Schur update that takes a vertex
        // out again cancels all but 1 / cond of that magnitude, so H -- and with it every later weight vector -- is off by
        // eps * cond^2 (6e-2 in the case tools/solve16_cases.py found: a distance 1.2 % too large, a corral of eight
        // "independent" vertices in three dimensions).  Hence: once a small pivot has been accepted (`dirty`), every removal
        // REBUILDS H from the tile's rows for the vertices that remain -- |S| borderings, error eps * cond of the CURRENT
        // support, no history.  Well-conditioned problems (every benchmark configuration) never take that path.
        bool dirty = false;
        // H for the vertices in S from scratch; a vertex whose pivot comes out non-positive now (it was accepted on a
        // corrupted H) is dropped and its weight shared out.  false: nothing usable was left, the solver has been set back
        // to the nearest vertex alone (the caller leaves its minor cycle).
        auto rebuild = [&]() __attribute__((always_inline)) -> bool {
            const unsigned S2 = S;
            unsigned lost = 0u;
            S = 0u; dirty = false;
#pragma unroll
            for (int j = 0; j < 16; ++j) I.H[j] = 0.0;
            for (int v = 0; v < 16; ++v) {
                if (!((S2 >> v) & 1u)) continue;
                bool sm;
                if (inv16_insert(I, Qt, sv, S, v, l16, sm, 0.0)) dirty = dirty || sm;
                else lost |= 1u << v;
            }
            if (lost == 0u) return true;
            alpha = ((lost >> l16) & 1u) ? 0.0 : alpha;
            const double s1 = group_sum16(alpha);
            if (S != 0u && s1 > 0.0) { alpha *= fast_rcp(s1); return true; }
            S = 0u;
#pragma unroll
            for (int j = 0; j < 16; ++j) I.H[j] = 0.0;
            (void)inv16_insert(I, Qt, sv, S, i0, l16);
            alpha = l16 == i0 ? 1.0 : 0.0;
            return false;
        };
        const double tol = 1.4210854715202004e-14 * scale;  // 64 eps * scale
        QP16_STAT(0, 1);
        for (int it = 0; it < 3 * 16 + 8; ++it) {
            double gi = 0.0;
            QP16_STAT(1, 1);
            group_allgather16(sv, l16, alpha, ag);
#pragma unroll
            for (int j = 0; j < 16; ++j) gi = fma(Qr[j], ag[j], gi);
            val = group_sum16(alpha * gi);
            double gmin;
            int jb;
            group_argmin16((mine && !(((S | banned) >> l16) & 1u)) ? gi : kInf, lane, gmin, jb);
            if (jb < 0 || !(gmin < val - tol)) break;
            {
                bool sm;
                if (!inv16_insert(I, Qt, sv, S, jb, l16, sm)) {
                    banned |= 1u << jb;
                    continue;
                }
                dirty = dirty || sm;
            }
            for (int mi = 0; mi <= 16; ++mi) {
                double beta;
                if (!inv16_beta(I, beta)) {   // (degenerate weights: give the vertex up)
                    if ((S >> jb) & 1u) {
                        inv16_remove(I, sv, S, jb, l16);
                        if (dirty) (void)rebuild();
                    }
                    banned |= 1u << jb;
                    break;
                }
                const bool in = (S >> l16) & 1u;
                const bool bad = in && !(beta > 0.0);
                if (((__ballot(bad) >> (lane & 48)) & 0xFFFFull) == 0ull) {
                    alpha = in ? beta : 0.0;
                    break;
                }
                QP16_STAT(2, 1);
                const double den = alpha - beta;
                double theta;
                int kr;
                group_argmin16(bad ? (den > 0.0 ? alpha / den : 0.0) : kInf, lane, theta, kr);
                const double vnew = alpha + theta * (beta - alpha);
                alpha = (in && l16 != kr) ? vnew : 0.0;
                inv16_remove(I, sv, S, kr, l16);
                if (kr == jb) banned |= 1u << jb;
                if (dirty && !rebuild()) break;
            }
        }
        QP16_STAT(3, __popc(S));
        double gi = 0.0;
        group_allgather16(sv, l16, alpha, ag);
#pragma unroll
        for (int j = 0; j < 16; ++j) gi = fma(Qr[j], ag[j], gi);
        val = group_sum16(alpha * gi);
    } else {
        // distance to the AFFINE hull: greedy maximal affinely independent subset (affine_min_norm)
        unsigned S = 0u;
        Inv16 I;
#pragma unroll
        for (int j = 0; j < 16; ++j) I.H[j] = 0.0;
        for (int k = 0; k < n; ++k) (void)inv16_insert(I, Qt, sv, S, k, l16);
        double beta = 0.0;
        const bool okb = S != 0u && inv16_beta(I, beta);
        alpha = (okb && ((S >> l16) & 1u)) ? beta : 0.0;
        if (!okb) alpha = l16 == 0 ? 1.0 : 0.0;
        double gi = 0.0;
        group_allgather16(sv, l16, alpha, ag);
#pragma unroll
        for (int j = 0; j < 16; ++j) gi = fma(Qr[j], ag[j], gi);
        val = group_sum16(alpha * gi);
    }
    return ldexp(val, ex);   // back to the data's units (exact)
}

// One k-sweep of the matrix core over the 16 rows `idv` (lane (row, kq): row = lane & 15 supplies the row,
// kq = lane >> 4 its features 4 kq .. 4 kq + 3 of every 16) shifted by the query row q: the 16 x 16 Gram tile,
// lane (row, kq) ends with D[kq + 4 r][row] in acc[r].  TWO: a second row set idw and the tiles
// <rows, rows> (acc), <rows, rows2> (acx: acx[r] = <row kq + 4 r of the first set, row `row` of the second>),
// <rows2, rows2> (acw) from ONE read of every row.
template <bool TWO>
struct RowChunk16 {   // 32 features of the lane's rows: 2 x 4 doubles each
    double2 v[4], x[4], u[TWO ? 4 : 1];
};
// FULL: the whole chunk lies inside the row (no range checks; only a row's last chunk can be partial)
template <bool TWO, bool FULL>
__device__ __forceinline__ void load_chunk16(RowChunk16<TWO> &c, const double *vptr, const double *wptr,
                                             const double *qptr, int k0, int kq, int Dp)
{
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int kk = k0 + 16 * t + 4 * kq;
        const bool in = FULL || kk < Dp;   // Dp % 8 == 0 and kk % 4 == 0: all 4 in range
        c.v[2 * t] = in ? *reinterpret_cast<const double2 *>(vptr + k0 + 16 * t) : double2{0.0, 0.0};
        c.v[2 * t + 1] = in ? *reinterpret_cast<const double2 *>(vptr + k0 + 16 * t + 2) : double2{0.0, 0.0};
        c.x[2 * t] = in ? *reinterpret_cast<const double2 *>(qptr + k0 + 16 * t) : double2{0.0, 0.0};
        c.x[2 * t + 1] = in ? *reinterpret_cast<const double2 *>(qptr + k0 + 16 * t + 2) : double2{0.0, 0.0};
        if (TWO) {
            c.u[2 * t] = in ? *reinterpret_cast<const double2 *>(wptr + k0 + 16 * t) : double2{0.0, 0.0};
            c.u[2 * t + 1] = in ? *reinterpret_cast<const double2 *>(wptr + k0 + 16 * t + 2) : double2{0.0, 0.0};
        }
    }
}
template <bool TWO>
__device__ __forceinline__ void mfma_chunk16(const RowChunk16<TWO> &cur, f64x4 &acc, f64x4 &acx, f64x4 &acw)
{
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const double y0 = cur.v[t].x - cur.x[t].x;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y0, y0, acc, 0, 0, 0);
        if (TWO) {
            const double z0 = cur.u[t].x - cur.x[t].x;
            acx = __builtin_amdgcn_mfma_f64_16x16x4f64(y0, z0, acx, 0, 0, 0);
            acw = __builtin_amdgcn_mfma_f64_16x16x4f64(z0, z0, acw, 0, 0, 0);
        }
        const double y1 = cur.v[t].y - cur.x[t].y;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y1, y1, acc, 0, 0, 0);
        if (TWO) {
            const double z1 = cur.u[t].y - cur.x[t].y;
            acx = __builtin_amdgcn_mfma_f64_16x16x4f64(y1, z1, acx, 0, 0, 0);
            acw = __builtin_amdgcn_mfma_f64_16x16x4f64(z1, z1, acw, 0, 0, 0);
        }
    }
}
template <bool TWO>
__device__ __forceinline__ void gram_tile16(const double *X, int Dp, int Dk, int q, int idv, int idw, int kq, f64x4 &acc,
                                            f64x4 &acx, f64x4 &acw)
{
    const double *vptr = X + (size_t)(idv >= 0 ? idv : q) * Dp + 4 * kq;   // a missing vertex reads the query row: y = 0
    const double *wptr = X + (size_t)(idw >= 0 ? idw : q) * Dp + 4 * kq;
    const double *qptr = X + (size_t)q * Dp + 4 * kq;
    const int Dfull = Dk & ~31;
    int k0 = 0;
    for (; k0 < Dfull; k0 += 32) {
        RowChunk16<TWO> cur;
        load_chunk16<TWO, true>(cur, vptr, wptr, qptr, k0, kq, Dk);
        mfma_chunk16<TWO>(cur, acc, acx, acw);
    }
    if (k0 < Dk) {
        RowChunk16<TWO> cur;
        load_chunk16<TWO, false>(cur, vptr, wptr, qptr, k0, kq, Dk);
        mfma_chunk16<TWO>(cur, acc, acx, acw);
    }
}

constexpr int kExtraMax = 2;   // extra rows beside the tile (3 measured the same: 19 candidates are rare enough)
constexpr int kExtraNP = kExtraMax * (kExtraMax + 1) / 2;
// The sweep of the fused kernel for rows of up to 288 doubles: 16 rows plus NE <= kExtraMax EXTRA rows ide[0..NE)
// (the usual shape of a shortlist at m = 15: 17 or 18 candidates).  The matrix core forms the tile of the 16 rows;
// the few products with and between the extra rows are vector FMAs on the lanes' feature slices: ae[b] = <row `row`,
// extra b>, ee[] = <extra b, extra b'> (b' <= b, packed lower-triangular: ee[b (b + 1) / 2 + b']), complete on every
// lane after the reduction over the four feature slices (lanes row, row + 16, row + 32, row + 48).  The query row
// comes from LDS (xs, zero-padded to a multiple of 16 doubles), which leaves the registers to the candidate rows:
// CH features of every row are in flight per round trip (the whole row for NE = 0 and D <= 160).
// features per round trip for 0 / 1 / 2 extra rows (register budget of the kernel: 168 VGPRs)
#ifndef CHB_SW0
#define CHB_SW0 160
#define CHB_SW1 48
#define CHB_SW2 32
#endif
template <int NE, int CH, bool FULL>
__device__ __forceinline__ void gram_sweep16_chunk(const double *vptr, const double *const (&eptr)[kExtraMax],
                                                   const double *xs, int k0, int kq, int Dp, f64x4 &acc,
                                                   double (&ae)[kExtraMax], double (&ee)[kExtraNP])
{
    constexpr int NG = CH / 16;   // 16-feature groups: the lane holds 4 doubles of each row per group
    double2 v[NG][2], e[NE > 0 ? NE : 1][NG][2];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const bool in = FULL || k0 + 16 * g + 4 * kq < Dp;   // Dp % 8 == 0: all 4 in range
        v[g][0] = in ? *reinterpret_cast<const double2 *>(vptr + k0 + 16 * g) : double2{0.0, 0.0};
        v[g][1] = in ? *reinterpret_cast<const double2 *>(vptr + k0 + 16 * g + 2) : double2{0.0, 0.0};
#pragma unroll
        for (int b = 0; b < NE; ++b) {
            e[b][g][0] = in ? *reinterpret_cast<const double2 *>(eptr[b] + k0 + 16 * g) : double2{0.0, 0.0};
            e[b][g][1] = in ? *reinterpret_cast<const double2 *>(eptr[b] + k0 + 16 * g + 2) : double2{0.0, 0.0};
        }
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (FULL || k0 + 16 * g < Dp) {   // (wave-uniform)
            const double2 x0 = *reinterpret_cast<const double2 *>(xs + k0 + 16 * g + 4 * kq);
            const double2 x1 = *reinterpret_cast<const double2 *>(xs + k0 + 16 * g + 4 * kq + 2);
            const double y[4] = {v[g][0].x - x0.x, v[g][0].y - x0.y, v[g][1].x - x1.x, v[g][1].y - x1.y};
#pragma unroll
            for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y[t], y[t], acc, 0, 0, 0);
            double z[NE > 0 ? NE : 1][4];
#pragma unroll
            for (int b = 0; b < NE; ++b) {
                z[b][0] = e[b][g][0].x - x0.x; z[b][1] = e[b][g][0].y - x0.y;
                z[b][2] = e[b][g][1].x - x1.x; z[b][3] = e[b][g][1].y - x1.y;
#pragma unroll
                for (int t = 0; t < 4; ++t) ae[b] = fma(y[t], z[b][t], ae[b]);
#pragma unroll
                for (int b2 = 0; b2 <= b; ++b2)
#pragma unroll
                    for (int t = 0; t < 4; ++t) ee[b * (b + 1) / 2 + b2] = fma(z[b][t], z[b2][t], ee[b * (b + 1) / 2 + b2]);
            }
        }
    }
}
// (rows wider than the LDS tile that stages the query row -- round 5 -- are swept in WINDOWS of up to kFusedMaxDp columns:
//  c0 = first column of the window, Dk = its width, xs = the query row's columns c0 .. c0 + Dk; `first` clears the extra-row
//  sums, `last` reduces them.  The one-window call of the narrow rows is the defaults.)
template <int NE, int CH>
__device__ __forceinline__ void gram_sweep16(const double *X, int Dp, int Dk, int q, int idv, const int (&ide)[kExtraMax], int kq,
                                             const double *xs, f64x4 &acc, double (&ae)[kExtraMax],
                                             double (&ee)[kExtraNP], int c0 = 0, bool first = true, bool last = true)
{
    const double *vptr = X + (size_t)(idv >= 0 ? idv : q) * Dp + 4 * kq + c0;   // a missing vertex reads the query row: y = 0
    const double *eptr[kExtraMax];
#pragma unroll
    for (int b = 0; b < kExtraMax; ++b) eptr[b] = X + (size_t)(b < NE ? ide[b] : q) * Dp + 4 * kq + c0;
    if (first) {
#pragma unroll
        for (int b = 0; b < kExtraMax; ++b) ae[b] = 0.0;
#pragma unroll
        for (int e = 0; e < kExtraNP; ++e) ee[e] = 0.0;
    }
    int k0 = 0;
    for (; k0 + CH <= Dk; k0 += CH) gram_sweep16_chunk<NE, CH, true>(vptr, eptr, xs, k0, kq, Dk, acc, ae, ee);
    if (k0 < Dk) gram_sweep16_chunk<NE, CH, false>(vptr, eptr, xs, k0, kq, Dk, acc, ae, ee);
    if (NE > 0 && last) {
        // sum over the four feature slices
#pragma unroll
        for (int b = 0; b < NE; ++b) {
            ae[b] += __shfl_xor(ae[b], 16, 64);
            ae[b] += __shfl_xor(ae[b], 32, 64);
        }
#pragma unroll
        for (int e = 0; e < NE * (NE + 1) / 2; ++e) {
            ee[e] += __shfl_xor(ee[e], 16, 64);
            ee[e] += __shfl_xor(ee[e], 32, 64);
        }
    }
}

// List-based form: the vertex lists are given (a.lists / explicit problems).  Active mode (a.active): the listed
// pairs only, walked grid-stride.
template <bool INDEXED>
__global__ __launch_bounds__(256, CHB_QP16_OCC) void hull_qp16_kernel(QpArgs a, int nprob, const int *xq, const int *xhull,
                                                        const int *xn, int xm, double *xdist, double *xalpha, Gate gate)
{
    CHB_GATE(gate);
    __shared__ __attribute__((aligned(16))) double sQ[4][4][16][kQ16Ld];   // [wavefront][problem][row][col], padded
    __shared__ __attribute__((aligned(16))) double sV[4][4][16];           // exchange row of each 16-lane group
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int grp = lane >> 4, l16 = lane & 15;
    const int m = INDEXED ? xm : a.m;
    const int *act = INDEXED ? nullptr : a.active;
    if (act != nullptr) nprob = *a.n_active;
    for (int gw = (blockIdx.x * 4 + w) * 4; gw < nprob; gw += (int)gridDim.x * 16) {   // wave-uniform
        const int g = gw + grp;   // my group's problem
        const bool valid = g < nprob;
        int n = 0, idm = -1, qid = 0, pr = 0;
        size_t slot = 0;
        if (valid) {
            if (INDEXED) {
                qid = xq[g];
                n = xn[g];
                if (l16 < n) idm = xhull[(size_t)g * m + l16];
            } else {
                pr = act ? act[g] : g;
                const int pos = a.pos_begin + pr / a.B, c = pr - (pr / a.B) * a.B;
                qid = a.bq[pos];
                slot = (size_t)c * a.Kcap + pos;
                n = a.lists.cnt[slot];
                if (l16 < n) idm = a.lists.idx[slot * m + l16];
            }
        }
        bool changed = valid;
        if (!INDEXED && a.prev.idx != nullptr && act == nullptr && valid)
            changed = a.prev.cnt[slot] != n || (l16 < n && a.prev.idx[slot * m + l16] != idm);
        const bool doit = ((__ballot(changed) >> (16 * grp)) & 0xFFFFull) != 0ull;   // else: keep the stored distance

        // ---- phase 1: Gram of the shifted vertices, one matrix-core tile per problem
        const int row = lane & 15, kq = lane >> 4;
#pragma unroll 1
        for (int p = 0; p < 4; ++p) {
            const int np = __shfl(n, 16 * p, 64);
            const bool go = __shfl(doit ? 1 : 0, 16 * p, 64) != 0 && np > 0;
            const int id = __shfl(idm, 16 * p + row, 64);
            const int q = __shfl(qid, 16 * p, 64);
            if (!go) continue;   // wave-uniform
            f64x4 acc = {0.0, 0.0, 0.0, 0.0}, d0 = acc, d1 = acc;
            gram_tile16<false>(a.X, a.Dp, (a.D + 7) & ~7, q, id, -1, kq, acc, d0, d1);
#pragma unroll
            for (int r = 0; r < 4; ++r) sQ[w][p][kq + 4 * r][row] = acc[r];
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wavefront's LDS writes have landed
        if (!valid || !doit) continue;

        // ---- phase 2: one problem per 16-lane group
        double alpha = 0.0;
        const double val = n <= 0 ? kInf : solve16(&sQ[w][grp][0][0], &sV[w][grp][0], n, a.metric, lane, alpha);
        const double dist = n <= 0 ? kInf : sqrt(fmax(val, 0.0));
        if (INDEXED) {
            if (l16 == 0) xdist[g] = dist;
            if (xalpha && l16 < m) xalpha[(size_t)g * m + l16] = alpha;
        } else if (l16 == 0) {
            const int pos = a.pos_begin + pr / a.B, c = pr - (pr / a.B) * a.B;
            a.dist[(size_t)pos * a.B + c] = dist;
        }
    }
}

// Fused selection + hull distance for 5 < m <= 16 (the same contract as hull_select_qp_kernel: FusedArgs, tie
// rule kTieRel, pairs it does not decide are appended to `slow`).  One (position, bin) pair per 16-lane group.
// A pair's candidates (base shortlist, then the batch's own entries) are read ONCE:
//   n <= 16: one matrix-core tile (Gram of all candidates);
//   17 .. 20 (the usual excess at m = 15): the tile of candidates 0..15 plus the n - 16 extra rows by vector FMAs;
//   21 .. 32: two tiles and the block between them (ranked and scattered inside the sweep loop);
//   n > 32, or a near-tie between ranks m - 1 and m: exact path.
// The diagonal = the squared distances ranks the candidates; the entries between the m nearest become the hull's
// Gram tile (vertex slot = rank).  For n <= 18 that happens after the sweeps, the four pairs of the wavefront
// side by side (16 lanes each: lane i owns candidate i's row of the raw tile and moves it to its slot).
// WIDEROW: the instantiation for rows wider than kFusedMaxDp columns (its window loop would cost the ordinary one 24 spilled
// registers).
template <int WAVES, bool WIDEROW>
__global__ __launch_bounds__(64 * WAVES, CHB_QP16_OCC) void hull_select_qp16_kernel(FusedArgs a, int nprob, Gate gate)
{
    CHB_GATE(gate);
    __shared__ __attribute__((aligned(16))) double sQ[WAVES][4][16][kQ16Ld];
    __shared__ __attribute__((aligned(16))) double sV[WAVES][4][16];
    __shared__ __attribute__((aligned(16))) double sAE[WAVES][4][kExtraMax][16];   // <candidate i, extra b> of each pair
    __shared__ __attribute__((aligned(16))) double sEE[WAVES][4][kExtraNP];        // <extra b, extra b'>
    __shared__ __attribute__((aligned(16))) double sGD[WAVES][4][20];      // squared distances of a pair's candidates
    __shared__ __attribute__((aligned(16))) int sGI[WAVES][4][20];         // their sample indices
    __shared__ __attribute__((aligned(16))) int sGN[WAVES][4][20];         // their vertex slots (-1: not selected)
    __shared__ double sTh[WAVES][4][2];   // squared distances of ranks m - 1 and m
    __shared__ int sSlowN, sSlowBase;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int grp = lane >> 4, l16 = lane & 15;
    const int m = a.m;
    const int Dk = (a.D + 7) & ~7;   // columns swept (the rows are padded to whole 128-byte lines beyond)
    // rows wider than the tile that stages a pair's query row (kFusedMaxDp = 288 doubles; wide feature vectors, round 5):
    // the sweep of a pair then runs in windows of 288 columns, all 64 lanes staging each window's piece of the query row
    constexpr bool widerow = WIDEROW;   // (the launcher: Dk > kFusedMaxDp)
    // the two-tile form works on one pair at a time inside the sweep loop and borrows the wavefront's group arrays
    // (32 of the 80 entries each), which the groups only use after the loop
    double *sDw = &sGD[w][0][0];
    int *sIdw = &sGI[w][0][0], *sNeww = &sGN[w][0][0];
    if (threadIdx.x == 0) sSlowN = 0;
    __syncthreads();
    // (a.stripe: the work order of the m <= 5 kernel -- workgroup b only takes pairs of the bins c = b (mod 8), so that an
    //  XCD gathers candidate rows of an eighth of the bins; the wavefront's four pairs are four of those bins at one
    //  position while the XCD has that many.  nprob is then the length of the longest of the 8 lists)
    const int gw = ((a.stripe ? (int)(blockIdx.x >> 3) : (int)blockIdx.x) * WAVES + w) * 4;
    int pos = 0, c = 0;
    const bool valid = fused_pair_of(a, gw, grp, nprob, pos, c);
    int nb = 0, nu = 0, qid = 0;
    bool changed = valid;
    const size_t slot = (size_t)c * a.Kcap + pos;
    if (valid) {
        qid = a.bq[pos];
        nb = a.cand_cnt[slot];
        if (a.candu != nullptr) nu = a.candu_cnt[slot];
        // unchanged set of batch-entry candidates (the base candidates are fixed during a batch): keep the distance
        if (a.candp != nullptr && a.candp_cnt[slot] == nu) {
            bool same = true;
            for (int i = l16; i < nu; i += 16) {
                const int u = a.candu[slot * kCandCapU + i];
                bool f = false;
                for (int j = 0; j < nu; ++j) f = f || a.candp[slot * kCandCapU + j] == u;
                same = same && f;
            }
            changed = ((__ballot(!same) >> (lane & 48)) & 0xFFFFull) != 0ull;
        }
    }
    if (a.short_cnt != nullptr && valid && l16 == 0 &&
        nb < min(m, a.bin_size != nullptr ? a.bin_size[c] : a.bin_ptr[c + 1] - a.bin_ptr[c]))
        atomicAdd(a.short_cnt, 1);   // (the shortlist stage's contract, see FusedArgs)
    QP16_CLK(tc0);
#ifdef CHB_DEV_CLK
    unsigned long long c_ids = 0, c_sweep = 0, c_one = 0;
#endif
    const int n = nb + nu;
    const bool work = changed && n > 0 && n <= 32;
    bool slow = changed && n > 32;
    int nsel = n < m ? n : m;   // vertices of the hull problem

    // ---- before the sweeps, every group for its own pair (one round trip for the four pairs): the candidates'
    // sample indices (lane i: candidate i; lanes 0 .. kExtraMax - 1 also the extra candidates 16 + i) and the query
    // row, which goes through LDS -- the pair's own tile is free until its sweep ends and holds rows of up to
    // kFusedMaxDp doubles (the caller checks; the shortlist stage itself needs D <= 160)
    int idm = -1, idx = -1;
    if (work) {
        if (l16 < n) idm = l16 < nb ? a.cand[slot * kCandCap + l16] : a.candu[slot * kCandCapU + (l16 - nb)];
        if (l16 < kExtraMax && 16 + l16 < n)
            idx = 16 + l16 < nb ? a.cand[slot * kCandCap + 16 + l16] : a.candu[slot * kCandCapU + (16 + l16 - nb)];
        if (((unsigned)idm >= (unsigned)a.n_samples && idm != -1) || ((unsigned)idx >= (unsigned)a.n_samples && idx != -1)) {
            if (a.short_cnt != nullptr) atomicAdd(a.short_cnt, 1);   // never a row address from a wild index
            if ((unsigned)idm >= (unsigned)a.n_samples) idm = -1;
            if ((unsigned)idx >= (unsigned)a.n_samples) idx = -1;
        }
        if (!widerow) {
            double *xq = &sQ[w][grp][0][0];
            const int Dp16 = (Dk + 15) & ~15;
            for (int e = 2 * l16; e < Dp16; e += 32)
                *reinterpret_cast<double2 *>(xq + e) =
                    e < Dk ? *reinterpret_cast<const double2 *>(a.X + (size_t)qid * a.Dp + e) : double2{0.0, 0.0};
        }
    }
    __builtin_amdgcn_wave_barrier();

    // ---- phase 1: the wavefront's four pairs one after the other, all 64 lanes on one pair
    const int row = lane & 15, kq = lane >> 4;
#pragma unroll 1
    for (int p = 0; p < 4; ++p) {
        const int np = __shfl(n, 16 * p, 64);
        const int nbp = __shfl(nb, 16 * p, 64);
        const int q = __shfl(qid, 16 * p, 64);
        const bool go = __shfl(work ? 1 : 0, 16 * p, 64) != 0;
        const int idA = __shfl(idm, 16 * p + row, 64);
        int ide[kExtraMax];
#pragma unroll
        for (int b = 0; b < kExtraMax; ++b) ide[b] = __shfl(idx, 16 * p + b, 64);
        if (!go) continue;   // wave-uniform
        QP16_CLK(tp0);
        const size_t slot_p = (size_t)__shfl(c, 16 * p, 64) * a.Kcap + (size_t)__shfl(pos, 16 * p, 64);
        const bool two = np > 16 + kExtraMax;
        const int ne = (np > 16 && !two) ? np - 16 : 0;
        int idB = -1;
        double *Qp = &sQ[w][p][0][0];
        if (!two) {
            f64x4 aa = {0.0, 0.0, 0.0, 0.0};
            double ae[kExtraMax], ee[kExtraNP];
            QP16_CLK(tp1);
            QP16_CLK_ACC(c_ids, tp1 - tp0);
            if (!widerow) {
                if (ne == 0) gram_sweep16<0, CHB_SW0>(a.X, a.Dp, Dk, q, idA, ide, kq, Qp, aa, ae, ee);
                else if (ne == 1) gram_sweep16<1, CHB_SW1>(a.X, a.Dp, Dk, q, idA, ide, kq, Qp, aa, ae, ee);
                else gram_sweep16<2, CHB_SW2>(a.X, a.Dp, Dk, q, idA, ide, kq, Qp, aa, ae, ee);
            } else {
#pragma unroll 1
                for (int c0 = 0; c0 < Dk; c0 += kFusedMaxDp) {
                    const int cw = min(kFusedMaxDp, Dk - c0), cw16 = (cw + 15) & ~15;
                    __builtin_amdgcn_wave_barrier();   // (the previous window's reads of the tile are issued: LDS keeps a wavefront's order)
                    for (int e = 2 * lane; e < cw16; e += 128)
                        *reinterpret_cast<double2 *>(Qp + e) =
                            e < cw ? *reinterpret_cast<const double2 *>(a.X + (size_t)q * a.Dp + c0 + e) : double2{0.0, 0.0};
                    __builtin_amdgcn_wave_barrier();
                    const bool fst = c0 == 0, lst = c0 + kFusedMaxDp >= Dk;
                    if (ne == 0) gram_sweep16<0, CHB_SW1>(a.X, a.Dp, cw, q, idA, ide, kq, Qp, aa, ae, ee, c0, fst, lst);
                    else if (ne == 1) gram_sweep16<1, CHB_SW1>(a.X, a.Dp, cw, q, idA, ide, kq, Qp, aa, ae, ee, c0, fst, lst);
                    else gram_sweep16<2, CHB_SW2>(a.X, a.Dp, cw, q, idA, ide, kq, Qp, aa, ae, ee, c0, fst, lst);
                }
            }
            if (ne > 0) {
                if (kq == 0) {
#pragma unroll
                    for (int b = 0; b < kExtraMax; ++b) sAE[w][p][b][row] = ae[b];
                }
                if (lane == 0) {
#pragma unroll
                    for (int e = 0; e < kExtraNP; ++e) sEE[w][p][e] = ee[e];
                }
            }
            // the raw tile of candidates 0..15; selection (np > m) follows after the loop
#pragma unroll
            for (int r = 0; r < 4; ++r) Qp[(kq + 4 * r) * kQ16Ld + row] = aa[r];
            {
                QP16_CLK(tp2);
                QP16_CLK_ACC(c_sweep, tp2 - tp1);
                QP16_CLK_ACC(c_one, 1);
            }
            continue;
        }
        // ---- two tiles (up to 32 candidates): rank and scatter here, the accumulators are this pair's
        if (16 + row < np)
            idB = 16 + row < nbp ? a.cand[slot_p * kCandCap + 16 + row] : a.candu[slot_p * kCandCapU + (16 + row - nbp)];
        if ((unsigned)idB >= (unsigned)a.n_samples && idB != -1) {   // never a row address from a wild index
            if (a.short_cnt != nullptr) atomicAdd(a.short_cnt, 1);
            idB = -1;
        }
        f64x4 aa = {0.0, 0.0, 0.0, 0.0}, ab = aa, bb = aa;
        gram_tile16<true>(a.X, a.Dp, Dk, q, idA, idB, kq, aa, ab, bb);
        // squared distances = the diagonals (lane (row, kq = row & 3) holds D[row][row] in acc[row >> 2])
        if ((row & 3) == kq) {
            const int rr = row >> 2;
            const double dA = rr == 0 ? aa[0] : rr == 1 ? aa[1] : rr == 2 ? aa[2] : aa[3];
            const double dB = rr == 0 ? bb[0] : rr == 1 ? bb[1] : rr == 2 ? bb[2] : bb[3];
            sDw[row] = dA; sIdw[row] = idA;
            sDw[16 + row] = dB; sIdw[16 + row] = idB;
        }
        __builtin_amdgcn_wave_barrier();
        // rank of candidate lane & 31 by (squared distance, index): each half of the wavefront scans 16 entries
        {
            const int cI = lane & 31, u0 = (lane >> 5) * 16;
            const double dc = cI < np ? sDw[cI] : kInf;
            const int ic = sIdw[cI];
            int rank = 0;
#pragma unroll
            for (int t = 0; t < 16; t += 2) {
                const int u = u0 + t;
                const double2 dd = *reinterpret_cast<const double2 *>(&sDw[u]);
                const int2 ii = *reinterpret_cast<const int2 *>(&sIdw[u]);
                rank += (u < np && u != cI && (dd.x < dc || (dd.x == dc && ii.x < ic))) ? 1 : 0;
                rank += (u + 1 < np && u + 1 != cI && (dd.y < dc || (dd.y == dc && ii.y < ic))) ? 1 : 0;
            }
            rank += __shfl_xor(rank, 32, 64);
            if (lane < 32) {
                sNeww[cI] = (cI < np && rank < m) ? rank : -1;
                if (cI < np && rank == m - 1) sTh[w][p][0] = dc;
                if (cI < np && rank == m) sTh[w][p][1] = dc;
            }
        }
        __builtin_amdgcn_wave_barrier();
        const double t0 = sTh[w][p][0], t1 = sTh[w][p][1];
        const bool tie = !(t1 - t0 > kTieRel * t1);   // (also catches NaN)
        if (tie) {
            if (grp == p) slow = true;
            continue;
        }
        // exactly m candidates are selected: the scatter below writes every entry between vertex slots < m once;
        // rows / columns m .. 15 of the tile are cleared (the solver multiplies them by zero weights)
        for (int v = m; v < 16; ++v)
            if (lane < 16) { Qp[v * kQ16Ld + lane] = 0.0; Qp[lane * kQ16Ld + v] = 0.0; }
        const int njA = sNeww[row], njB = sNeww[16 + row];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int niA = sNeww[kq + 4 * r], niB = sNeww[16 + kq + 4 * r];
            if (niA >= 0 && njA >= 0) Qp[niA * kQ16Ld + njA] = aa[r];
            if (niA >= 0 && njB >= 0) { Qp[niA * kQ16Ld + njB] = ab[r]; Qp[njB * kQ16Ld + niA] = ab[r]; }
            if (niB >= 0 && njB >= 0) Qp[niB * kQ16Ld + njB] = bb[r];
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wavefront's LDS writes have landed

    QP16_CLK(tc1);
    // ---- selection for m < n <= 18, one pair per 16-lane group: lane i owns candidate i (and, i < n - 16, extra i)
    if (work && n > m && n <= 16 + kExtraMax) {
        double *Qt = &sQ[w][grp][0][0];
        const int ne = n > 16 ? n - 16 : 0, nA = n - ne;
        const int ide = l16 < ne ? idx : -1;
        if (l16 >= nA) idm = -1;
        const double dA = l16 < nA ? Qt[l16 * kQ16Ld + l16] : kInf;
        const double dE = l16 < ne ? sEE[w][grp][l16 * (l16 + 1) / 2 + l16] : kInf;
        sGD[w][grp][l16] = dA; sGI[w][grp][l16] = idm;
        if (l16 < 4) { sGD[w][grp][16 + l16] = dE; sGI[w][grp][16 + l16] = ide; }   // (+inf / -1 beyond the extras)
        __builtin_amdgcn_wave_barrier();
        int rA = 0, rE = 0;
#pragma unroll
        for (int u = 0; u < 20; u += 2) {
            const double2 dd = *reinterpret_cast<const double2 *>(&sGD[w][grp][u]);
            const int2 ii = *reinterpret_cast<const int2 *>(&sGI[w][grp][u]);
            // (absent candidates carry +inf: they never precede a real one)
            rA += (u != l16 && (dd.x < dA || (dd.x == dA && ii.x < idm))) ? 1 : 0;
            rA += (u + 1 != l16 && (dd.y < dA || (dd.y == dA && ii.y < idm))) ? 1 : 0;
            rE += (u != 16 + l16 && (dd.x < dE || (dd.x == dE && ii.x < ide))) ? 1 : 0;
            rE += (u + 1 != 16 + l16 && (dd.y < dE || (dd.y == dE && ii.y < ide))) ? 1 : 0;
        }
        const int newA = (l16 < nA && rA < m) ? rA : -1;
        const int newE = (l16 < ne && rE < m) ? rE : -1;
        sGN[w][grp][l16] = newA;
        if (l16 < 4) sGN[w][grp][16 + l16] = newE;
        if (l16 < nA && rA == m - 1) sTh[w][grp][0] = dA;
        if (l16 < nA && rA == m) sTh[w][grp][1] = dA;
        if (l16 < ne && rE == m - 1) sTh[w][grp][0] = dE;
        if (l16 < ne && rE == m) sTh[w][grp][1] = dE;
        // my candidate's row of the raw tile, its products with the extras, and (extra lanes) the extras' own block
        double Qr[16], qe[kExtraMax], qx[kExtraMax];
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
            const double2 t = *reinterpret_cast<const double2 *>(Qt + l16 * kQ16Ld + j);
            Qr[j] = t.x; Qr[j + 1] = t.y;
        }
#pragma unroll
        for (int b = 0; b < kExtraMax; ++b) {
            qe[b] = sAE[w][grp][b][l16];
            const int hi = min(max(l16, b), kExtraMax - 1), lo = min(l16, b);
            qx[b] = sEE[w][grp][hi * (hi + 1) / 2 + lo];   // <extra l16, extra b> (used by lanes < ne)
        }
        __builtin_amdgcn_wave_barrier();
        const double t0 = sTh[w][grp][0], t1 = sTh[w][grp][1];
        if (!(t1 - t0 > kTieRel * t1)) {   // near-tie at the selection boundary (also catches NaN)
            slow = true;
        } else {
            int nj[20];
#pragma unroll
            for (int u = 0; u < 20; u += 4) {
                const int4 t = *reinterpret_cast<const int4 *>(&sGN[w][grp][u]);
                nj[u] = t.x; nj[u + 1] = t.y; nj[u + 2] = t.z; nj[u + 3] = t.w;
            }
            // rows / columns m .. 15 of the tile are cleared (the solver multiplies them by zero weights)
            for (int v = m; v < 16; ++v) { Qt[v * kQ16Ld + l16] = 0.0; Qt[l16 * kQ16Ld + v] = 0.0; }
            if (newA >= 0) {
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (nj[j] >= 0) Qt[newA * kQ16Ld + nj[j]] = Qr[j];
#pragma unroll
                for (int b = 0; b < kExtraMax; ++b)
                    if (b < ne && nj[16 + b] >= 0) {
                        Qt[newA * kQ16Ld + nj[16 + b]] = qe[b];
                        Qt[nj[16 + b] * kQ16Ld + newA] = qe[b];
                    }
            }
            if (newE >= 0) {
#pragma unroll
                for (int b = 0; b < kExtraMax; ++b)
                    if (b < ne && nj[16 + b] >= 0) Qt[newE * kQ16Ld + nj[16 + b]] = qx[b];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }

    QP16_CLK(tc2);
    // ---- phase 2: one pair per 16-lane group
    if (valid && changed && !slow) {   // else: distance kept, or written by the exact path
        double alpha = 0.0;
        double dist = kInf;
        if (n > 0) dist = sqrt(fmax(solve16(&sQ[w][grp][0][0], &sV[w][grp][0], nsel, a.metric, lane, alpha), 0.0));
        if (l16 == 0) a.dist[(size_t)pos * a.B + c] = dist;
    }
    QP16_CLK(tc3);

    // ---- the exact path's work list: one device-scope atomic per workgroup (at the very end: the barriers
    // below must not hold a wavefront back between its sweeps and its solver)
    {
        const unsigned long long bal = __ballot(slow && l16 == 0);
        const int nsl = __popcll(bal);
        int wbase = 0;
        if (lane == 0 && nsl > 0) wbase = atomicAdd(&sSlowN, nsl);
        __syncthreads();
        if (threadIdx.x == 0) sSlowBase = sSlowN > 0 ? atomicAdd(a.n_slow, sSlowN) : 0;
        __syncthreads();
        wbase = __shfl(wbase, 0, 64) + sSlowBase;
        if (slow && l16 == 0)   // (position-major pair index, whatever the work order)
            a.slow[wbase + __popcll(bal & ((1ull << lane) - 1ull))] = (pos - a.pos_begin) * a.B + c;
    }
#ifdef CHB_DEV_CLK
    {
        const unsigned wid = blockIdx.x * WAVES + w;
        if (lane == 0 && wid < 131072u) {
            unsigned long long *o = g_qp16_clk[wid];
            o[0] = tc1 - tc0; o[1] = tc2 - tc1; o[2] = tc3 - tc2; o[3] = c_ids; o[4] = c_sweep; o[5] = c_one; o[6] = 1; o[7] = tc0;
        }
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// num_neighbors > 16 (no cap in the reference: algorithm.py:17, cli/clustering.py:118-120): the plain form,
// ONE wavefront per problem, up to 64 vertices.  Lane i owns vertex i: row i of the shifted Gram Q and of the
// inverse H of the lifted support Gram (both in LDS, [64][65] doubles) and its weight alpha_i.  The same Wolfe
// iteration and the same bordering / Schur updates as hull_qp16_kernel, with loops over the vertex count and
// LDS rows where that kernel has 16-entry register arrays and shuffles.  Slow by design.
constexpr int kGenN = 64, kGenLd = 65;

struct GenLds {
    double Q[kGenN][kGenLd];
    double H[kGenN][kGenLd];
    double va[kGenN], vu[kGenN];
};

__device__ __forceinline__ double wave_sum64(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// smallest key, ties to the lowest lane; all keys +inf gives idx = -1
__device__ __forceinline__ void wave_argmin64(double key, int lane, double &kmin, int &idx)
{
    kmin = key;
    idx = key < kInf ? lane : -1;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double ok = __shfl_xor(kmin, off, 64);
        const int oi = __shfl_xor(idx, off, 64);
        if (oi >= 0 && (idx < 0 || ok < kmin || (ok == kmin && oi < idx))) { kmin = ok; idx = oi; }
    }
}
__device__ __forceinline__ void gen_sync()
{
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);
}

// vertex v enters the support: false (and no change) when it is affinely dependent on it
// (small / reject: as inv16_insert)
__device__ __forceinline__ bool gen_insert(GenLds &L, int n, double s, unsigned long long &S, int v, int lane, bool &small,
                                           double reject = 1e-13)
{
    const bool in = (S >> lane) & 1ull;
    const double a_own = (lane < n) ? L.Q[v][lane] + s : 0.0;
    const double avv = L.Q[v][v] + s;
    L.va[lane] = in ? a_own : 0.0;
    gen_sync();
    double u = 0.0;
    if (in)
        for (int j = 0; j < n; ++j) u = fma(L.H[lane][j], L.va[j], u);   // (H is zero outside the support)
    double delta = avv - wave_sum64(in ? a_own * u : 0.0);
    if (!(delta > 1e-6 * avv)) {
        // small pivot: one step of iterative refinement against the original rows of Q (see inv16_insert)
        L.vu[lane] = in ? u : 0.0;
        gen_sync();
        double r = a_own;
        if (in)
            for (int j = 0; j < n; ++j)
                if ((S >> j) & 1ull) r = fma(-(L.Q[lane][j] + s), L.vu[j], r);
        gen_sync();
        L.vu[lane] = in ? r : 0.0;
        gen_sync();
        double du = 0.0;
        if (in)
            for (int j = 0; j < n; ++j) du = fma(L.H[lane][j], L.vu[j], du);
        u += du;
        delta = avv - wave_sum64(in ? a_own * u : 0.0);
        gen_sync();
    }
    small = !(delta > kSmallPivot * avv);
    if (!(delta > reject * avv)) return false;
    const double inv = 1.0 / delta;
    L.vu[lane] = in ? u : 0.0;
    gen_sync();
    if (in) {
        const double ui = u * inv;
        for (int j = 0; j < n; ++j)
            if ((S >> j) & 1ull) L.H[lane][j] = fma(ui, L.vu[j], L.H[lane][j]);
        L.H[lane][v] = -ui;
    }
    if (lane == v) {
        for (int j = 0; j < n; ++j) L.H[v][j] = ((S >> j) & 1ull) ? -L.vu[j] * inv : 0.0;
        L.H[v][v] = inv;
    }
    S |= 1ull << v;
    gen_sync();
    return true;
}

__device__ __forceinline__ bool gen_insert(GenLds &L, int n, double s, unsigned long long &S, int v, int lane)
{
    bool small;
    return gen_insert(L, n, s, S, v, lane, small);
}

// vertex r (in the support) leaves
__device__ __forceinline__ void gen_remove(GenLds &L, int n, unsigned long long &S, int r, int lane)
{
    const bool in = (S >> lane) & 1ull;
    const double hrr = L.H[r][r];
    if (in && lane != r) {
        const double f = L.H[lane][r] / hrr;
        for (int j = 0; j < n; ++j)
            if (j != r) L.H[lane][j] = fma(-f, L.H[r][j], L.H[lane][j]);
        L.H[lane][r] = 0.0;
    }
    gen_sync();
    if (lane == r)
        for (int j = 0; j < n; ++j) L.H[r][j] = 0.0;
    S &= ~(1ull << r);
    gen_sync();
}

// weight of my vertex in the affine minimiser on the support; false if the weights do not sum > 0
__device__ __forceinline__ bool gen_beta(GenLds &L, int n, unsigned long long S, int lane, double &beta)
{
    double b = 0.0;
    if ((S >> lane) & 1ull)
        for (int j = 0; j < n; ++j) b += L.H[lane][j];
    const double sum = wave_sum64(b);
    beta = b / sum;
    return sum > 0.0;
}

template <bool INDEXED>
__global__ __launch_bounds__(64) void hull_generic_kernel(QpArgs a, int nprob, const int *xq, const int *xhull,
                                                        const int *xn, int xm, double *xdist, double *xalpha, Gate gate)
{
    CHB_GATE(gate);
    extern __shared__ __attribute__((aligned(16))) unsigned char gen_smem[];
    GenLds &L = *reinterpret_cast<GenLds *>(gen_smem);
    const int lane = threadIdx.x;
    const int g = blockIdx.x;
    if (g >= nprob) return;
    const int m = INDEXED ? xm : a.m;
    int n = 0, id = -1, qid = 0, pos = 0, c = 0;
    if (INDEXED) {
        qid = xq[g];
        n = xn[g];
        if (lane < n) id = xhull[(size_t)g * m + lane];
    } else {
        pos = a.pos_begin + g / a.B; c = g - (g / a.B) * a.B;
        qid = a.bq[pos];
        const size_t slot = (size_t)c * a.Kcap + pos;
        n = a.lists.cnt[slot];
        if (lane < n) id = a.lists.idx[slot * m + lane];
        if (a.prev.idx != nullptr) {   // unchanged vertex list: keep the stored distance
            const bool changed = a.prev.cnt[slot] != n || (lane < n && a.prev.idx[slot * m + lane] != id);
            if (__ballot(changed) == 0ull) return;
        }
    }
    const bool mine = lane < n;
    double alpha = 0.0, val = kInf;
    if (n > 0) {
        // ---- shifted Gram: lane i computes row i (its own row against every vertex j <= i, mirrored)
        const double *xrow = a.X + (size_t)qid * a.Dp;
        const double *vi = a.X + (size_t)(mine ? id : qid) * a.Dp;
        for (int j = 0; j < n; ++j) {
            const int idj = __shfl(id, j, 64);
            const double *vj = a.X + (size_t)idj * a.Dp;
            double acc = 0.0;
            for (int k = 0; k < a.Dp; ++k) {
                const double xk = xrow[k];
                acc = fma(vi[k] - xk, vj[k] - xk, acc);
            }
            if (mine) L.Q[lane][j] = acc;
        }
        for (int j = 0; j < kGenN; ++j) L.H[lane][j] = 0.0;
        gen_sync();
        const double diag = mine ? L.Q[lane][lane] : 0.0;
        double scale = diag;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) scale = fmax(scale, __shfl_xor(scale, off, 64));
        unsigned long long S = 0ull, banned = 0ull;
        if (!(scale > 0.0)) {   // every vertex coincides with the query (or NaN input)
            alpha = lane == 0 ? 1.0 : 0.0;
            val = scale == 0.0 ? 0.0 : scale;
        } else if (a.metric == 0) {
            double best;
            int i0;
            wave_argmin64(mine ? diag : kInf, lane, best, i0);
            (void)gen_insert(L, n, scale, S, i0, lane);   // a single vertex is always independent
            alpha = lane == i0 ? 1.0 : 0.0;
            // (as solve16: once a small pivot has been accepted, every removal rebuilds the inverse for what remains)
            bool dirty = false;
            auto rebuild = [&]() -> bool {
                const unsigned long long S2 = S;
                unsigned long long lost = 0ull;
                S = 0ull; dirty = false;
                for (int j = 0; j < kGenN; ++j) L.H[lane][j] = 0.0;
                gen_sync();
                for (int v = 0; v < n; ++v) {
                    if (!((S2 >> v) & 1ull)) continue;
                    bool sm;
                    if (gen_insert(L, n, scale, S, v, lane, sm, 0.0)) dirty = dirty || sm;
                    else lost |= 1ull << v;
                }
                if (lost == 0ull) return true;
                alpha = ((lost >> lane) & 1ull) ? 0.0 : alpha;
                const double s1 = wave_sum64(alpha);
                if (S != 0ull && s1 > 0.0) { alpha /= s1; return true; }
                S = 0ull;
                for (int j = 0; j < kGenN; ++j) L.H[lane][j] = 0.0;
                gen_sync();
                (void)gen_insert(L, n, scale, S, i0, lane);
                alpha = lane == i0 ? 1.0 : 0.0;
                return false;
            };
            const double tol = 1.4210854715202004e-14 * scale;  // 64 eps * scale
            for (int it = 0; it < 3 * kGenN + 8; ++it) {
                L.va[lane] = alpha;
                gen_sync();
                double gi = 0.0;
                if (mine)
                    for (int j = 0; j < n; ++j) gi = fma(L.Q[lane][j], L.va[j], gi);
                val = wave_sum64(alpha * gi);
                gen_sync();
                double gmin;
                int jb;
                wave_argmin64((mine && !(((S | banned) >> lane) & 1ull)) ? gi : kInf, lane, gmin, jb);
                if (jb < 0 || !(gmin < val - tol)) break;
                {
                    bool sm;
                    if (!gen_insert(L, n, scale, S, jb, lane, sm)) {
                        banned |= 1ull << jb;
                        continue;
                    }
                    dirty = dirty || sm;
                }
                for (int mi = 0; mi <= n; ++mi) {
                    double beta;
                    if (!gen_beta(L, n, S, lane, beta)) {   // (degenerate weights: give the vertex up)
                        if ((S >> jb) & 1ull) {
                            gen_remove(L, n, S, jb, lane);
                            if (dirty) (void)rebuild();
                        }
                        banned |= 1ull << jb;
                        break;
                    }
                    const bool in = (S >> lane) & 1ull;
                    const bool bad = in && !(beta > 0.0);
                    if (__ballot(bad) == 0ull) {
                        alpha = in ? beta : 0.0;
                        break;
                    }
                    const double den = alpha - beta;
                    double theta;
                    int kr;
                    wave_argmin64(bad ? (den > 0.0 ? alpha / den : 0.0) : kInf, lane, theta, kr);
                    const double vnew = alpha + theta * (beta - alpha);
                    alpha = (in && lane != kr) ? vnew : 0.0;
                    gen_remove(L, n, S, kr, lane);
                    if (kr == jb) banned |= 1ull << jb;
                    if (dirty && !rebuild()) break;
                }
            }
            L.va[lane] = alpha;
            gen_sync();
            double gi = 0.0;
            if (mine)
                for (int j = 0; j < n; ++j) gi = fma(L.Q[lane][j], L.va[j], gi);
            val = wave_sum64(alpha * gi);
        } else {
            // distance to the AFFINE hull: greedy maximal affinely independent subset (affine_min_norm)
            for (int k = 0; k < n; ++k) (void)gen_insert(L, n, scale, S, k, lane);
            double beta = 0.0;
            const bool okb = S != 0ull && gen_beta(L, n, S, lane, beta);
            alpha = (okb && ((S >> lane) & 1ull)) ? beta : 0.0;
            if (!okb) alpha = lane == 0 ? 1.0 : 0.0;
            L.va[lane] = alpha;
            gen_sync();
            double gi = 0.0;
            if (mine)
                for (int j = 0; j < n; ++j) gi = fma(L.Q[lane][j], L.va[j], gi);
            val = wave_sum64(alpha * gi);
        }
    }
    const double dist = n <= 0 ? kInf : sqrt(fmax(val, 0.0));
    if (INDEXED) {
        if (lane == 0) xdist[g] = dist;
        if (xalpha && lane < m) xalpha[(size_t)g * m + lane] = alpha;
    } else if (lane == 0) {
        a.dist[(size_t)pos * a.B + c] = dist;
    }
}

template <bool INDEXED>
void launch_generic(const QpArgs &a, int nprob, int m, const int *xq, const int *xhull, const int *xn, double *xdist,
                    double *xalpha, hipStream_t s)
{
    if (nprob <= 0) return;
    // per launch (cheap, and the attribute is per device: a process may hold contexts on several); whether the
    // device grants it at all is asked up front by hull_generic_supported()
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(hull_generic_kernel<INDEXED>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(GenLds));
    hipLaunchKernelGGL((hull_generic_kernel<INDEXED>), dim3(nprob), dim3(64), sizeof(GenLds), s, a, nprob, xq, xhull, xn, m,
                       xdist, xalpha, g_gate);
}

template <int M, int WV, bool INDEXED>
void launch_one(const QpArgs &a, int nprob, int m, const int *xq, const int *xhull, const int *xn,
                double *xdist, double *xalpha, hipStream_t s)
{
    constexpr int PPW = 64;
    const int nwaves = (nprob + PPW - 1) / PPW;
    int grid = (nwaves + WV - 1) / WV;
    if (!INDEXED && a.active != nullptr) grid = std::min(16 * grid, 1024);   // listed pairs: grid-stride, 4 per wavefront
    hipLaunchKernelGGL((hull_qp_kernel<M, WV, INDEXED>), dim3(grid), dim3(64 * WV), 0, s, a, nprob,
                       xq, xhull, xn, m, xdist, xalpha, g_gate);
}

template <bool INDEXED>
void dispatch(const QpArgs &a, int nprob, int m, const int *xq, const int *xhull, const int *xn,
              double *xdist, double *xalpha, hipStream_t s)
{
    if (nprob <= 0) return;
    if (m <= 4) launch_one<4, 4, INDEXED>(a, nprob, m, xq, xhull, xn, xdist, xalpha, s);
    else if (m <= 5) launch_one<5, 4, INDEXED>(a, nprob, m, xq, xhull, xn, xdist, xalpha, s);
    else if (m <= 8) launch_one<8, 2, INDEXED>(a, nprob, m, xq, xhull, xn, xdist, xalpha, s);
    else
    {
        int grid = (nprob + 15) / 16;
        if (!INDEXED && a.active != nullptr) grid = std::min(grid, 1024);   // listed pairs (device-side count): grid-stride
        hipLaunchKernelGGL((hull_qp16_kernel<INDEXED>), dim3(grid), dim3(256), 0, s, a, nprob, xq, xhull,
                           xn, m, xdist, xalpha, g_gate);
    }
}

}  // namespace

void launch_hull_qp(const QpArgs &a, hipStream_t s)
{
    const int nprob = (a.pos_end - a.pos_begin) * a.B;
    dispatch<false>(a, nprob, a.m, nullptr, nullptr, nullptr, nullptr, nullptr, s);
}

// (rows of any width: the m <= 5 kernel sweeps them from registers, the 16-lane kernel stages the query row in LDS in
//  windows of kFusedMaxDp columns)
bool fused_supported(int m, int Dp) { (void)Dp; return m >= 1 && m <= 16; }

void launch_hull_select_qp(const FusedArgs &a, hipStream_t s)
{
    const int nprob = (a.pos_end - a.pos_begin) * a.B;
    if (nprob <= 0) return;
    constexpr int WV = 4;
    FusedArgs as = a;
    static const bool stripe_off = getenv("CHB_FUSED_STRIPE") != nullptr && atoi(getenv("CHB_FUSED_STRIPE")) == 0;
    as.stripe = (!stripe_off && a.B >= 16) ? 1 : 0;
    if (a.m > 5) {   // 16 lanes per pair, four pairs per wavefront
        int np16 = nprob, grid16 = (nprob + 4 * WV - 1) / (4 * WV);
        if (as.stripe) {
            np16 = (a.pos_end - a.pos_begin) * ((a.B + 7) / 8);
            grid16 = 8 * ((np16 + 4 * WV - 1) / (4 * WV));
        }
        if (((a.D + 7) & ~7) > kFusedMaxDp)
            hipLaunchKernelGGL((hull_select_qp16_kernel<WV, true>), dim3(grid16), dim3(64 * WV), 0, s, as, np16, g_gate);
        else
            hipLaunchKernelGGL((hull_select_qp16_kernel<WV, false>), dim3(grid16), dim3(64 * WV), 0, s, as, np16, g_gate);
        return;
    }
    int np5 = nprob, grid = (nprob + 64 * WV - 1) / (64 * WV);
    if (as.stripe) {
        // the longest of the 8 XCD lists: XCD 0 has ceil(B / 8) bins
        np5 = (a.pos_end - a.pos_begin) * ((a.B + 7) / 8);
        grid = 8 * ((np5 + 64 * WV - 1) / (64 * WV));
    }
    // (rows as 32-bit byte offsets while the sample matrix is smaller than 4 GiB: see gram_rows; CHB_FUSED_PTR64=1
    //  selects the 64-bit-pointer instantiation regardless, for the tests)
    static const bool ptr64 = getenv("CHB_FUSED_PTR64") != nullptr && atoi(getenv("CHB_FUSED_PTR64")) != 0;
    if (!ptr64 && (unsigned long long)a.n_samples * (unsigned long long)a.Dp * 8ull < (1ull << 32))
        hipLaunchKernelGGL((hull_select_qp_kernel<5, CHB_FUSED_C, WV, true>), dim3(grid), dim3(64 * WV), 0, s, as, np5, g_gate);
    else
        hipLaunchKernelGGL((hull_select_qp_kernel<5, CHB_FUSED_C, WV, false>), dim3(grid), dim3(64 * WV), 0, s, as, np5, g_gate);
}

bool hull_generic_supported()
{
    const hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void *>(hull_generic_kernel<false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(GenLds));
    const hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(hull_generic_kernel<true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(GenLds));
    if (e0 != hipSuccess || e1 != hipSuccess) { (void)hipGetLastError(); return false; }
    return true;
}

void launch_hull_generic(const QpArgs &a, hipStream_t s)
{
    launch_generic<false>(a, (a.pos_end - a.pos_begin) * a.B, a.m, nullptr, nullptr, nullptr, nullptr, nullptr, s);
}

void launch_hull_generic_indexed(const double *X, int D, int Dp, const int *q, const int *hull_idx,
                                 const int *hull_cnt, int P, int m_max, int metric, double *dist,
                                 double *alpha, hipStream_t s)
{
    QpArgs a{};
    a.X = X; a.D = D; a.Dp = Dp; a.m = m_max; a.metric = metric;
    launch_generic<true>(a, P, m_max, q, hull_idx, hull_cnt, dist, alpha, s);
}

void launch_hull_qp_indexed(const double *X, int D, int Dp, const int *q, const int *hull_idx,
                            const int *hull_cnt, int P, int m_max, int metric, double *dist,
                            double *alpha, hipStream_t s)
{
    QpArgs a{};
    a.X = X; a.D = D; a.Dp = Dp; a.m = m_max; a.metric = metric;
    dispatch<true>(a, P, m_max, q, hull_idx, hull_cnt, dist, alpha, s);
}

}  // namespace chb

#if defined(CHB_DEV_QP16_STAT) || defined(CHB_DEV_CLK)
// developer variants only: [0] problems solved, [1] major iterations, [2] ratio-test removals, [3] final support
// sizes (sum), [4] small-pivot refinements of inv16_insert
#ifdef CHB_DEV_CLK
// sums of the per-wavefront stamps of the last launch: [0] sweeps [1] selection [2] slow list + solver [3] id loads
// [4] one-tile sweeps [5] one-tile pairs [6] wavefronts [7] span (last start - first start)
extern "C" int chb_dev_qp16_clk(unsigned long long *out8)
{
    std::vector<unsigned long long> h(131072 * 8);
    if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(chb::g_qp16_clk), h.size() * sizeof(unsigned long long)) != hipSuccess) return -1;
    for (int k = 0; k < 8; ++k) out8[k] = 0;
    unsigned long long lo = ~0ull, hi = 0;
    for (size_t i = 0; i < 131072; ++i) {
        if (!h[i * 8 + 6]) continue;
        for (int k = 0; k < 7; ++k) out8[k] += h[i * 8 + k];
        lo = std::min(lo, h[i * 8 + 7]); hi = std::max(hi, h[i * 8 + 7]);
    }
    out8[7] = hi > lo ? hi - lo : 0;
    const std::vector<unsigned long long> z(131072 * 8, 0ull);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(chb::g_qp16_clk), z.data(), z.size() * sizeof(unsigned long long));
    return 0;
}
#endif
extern "C" int chb_dev_qp16_stats(unsigned long long *out, int reset)
{
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(chb::g_qp16_stats), 16 * sizeof(unsigned long long)) != hipSuccess)
        return -1;
    if (reset) {
        const unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(chb::g_qp16_stats), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
