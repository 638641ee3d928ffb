// Batched point-to-convex-hull distance for gfx950.
//
// Replaces hull_distance.py:7-35 (convex_hull_distance: P = 2 X X^T, q = -2 X x, sum(alpha) = 1,
// alpha >= 0, ||alpha X - x||) together with solve_qp.py:18-51 / :96-132 (nearest-PD repair,
// quadprog's Goldfarb-Idnani solve, cvxopt fallback) for every (contig, bin) pair of a batch.
//
// The QP min ||sum_a alpha_a x_a - x||^2 over the simplex is the minimum-norm point of
// conv{y_a = x_a - x}.  It is solved in Gram space:
//   phase 1 (fp64 matrix core): every lane streams one hull-vertex row, forms y_a = x_a - x, and
//     v_mfma_f64_16x16x4_f64 accumulates the shifted Gram Q_ab = <y_a, y_b> of 16/M problems per
//     tile (no cross-lane reduction); the diagonal blocks go to LDS.  ~64 problems per wavefront.
//   phase 2 (one problem per lane): Wolfe's minimum-norm-point active-set method on the m x m Gram
//     held in registers (fixed-size, mask-driven, fully unrolled so nothing is dynamically
//     indexed); affine sub-problems by LDL^T of the lifted Gram Q_SS + s*11^T, which is positive
//     definite exactly when the support is affinely independent -- so duplicate / collinear
//     vertices (singular 2XX^T, the case the reference repairs with nearest-PD jitter and a cvxopt
//     fallback) need no special handling.
//   distance = sqrt(alpha^T Q alpha).
// The result is the same projection distance the reference's QP defines (unique even when alpha
// is not); agreement with the CPU oracle's Goldfarb-Idnani restatement is ~1e-13 relative.
#include "chb_internal.h"

#include <math.h>

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

// INDEXED = false: problems are (batch position, bin) pairs whose vertices come from the top-m
// lists.  INDEXED = true: explicit (query sample, compacted vertex index list, count) problems.
//
// Phase 1 maps the shifted Gram onto the fp64 matrix core: v_mfma_f64_16x16x4_f64 computes
// D = A(16x4) * B(4x16) + C with lane l supplying A[l&15][l>>4] and B[l>>4][l&15] -- for a Gram
// A = Y and B = Y^T are the SAME register.  The 16 rows of a tile are the vertices of 16/M
// consecutive problems (3 problems of 5 vertices at the default AlgoNumNeighbors); the diagonal
// M x M blocks of the 16x16 product are their Gram matrices, and the k-reduction over the feature
// dimension happens inside the MFMA accumulator: no cross-lane reduction at all.  Each lane streams
// ONE vertex row (every 4th feature), 8 loads in flight per operand.
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

template <int M, int WAVES, bool INDEXED>
__global__ __launch_bounds__(64 * WAVES, (M <= 5 ? 4 : 1)) void hull_qp_kernel(QpArgs a, int nprob, const int *xq,
                                                              const int *xhull, const int *xn,
                                                              int xm, double *xdist, double *xalpha)
{
    constexpr int NP = Sym<M>::NP;
    constexpr bool VGRAM = M <= 8;   // Gram by vector FMAs (16 lanes per problem) instead of the matrix core
    constexpr int PPT = 16 / M;    // problems per MFMA tile
    constexpr int NT = 64 / PPT;   // tiles per wavefront
    constexpr int PPW = VGRAM ? 64 : PPT * NT;  // problems per wavefront (one per lane in phase 2)
    __shared__ double sQ[WAVES][NP][64];
    __shared__ int sN[WAVES][64];

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int g0 = (blockIdx.x * WAVES + w) * PPW;
    if (g0 >= nprob) return;
    const int m = INDEXED ? xm : a.m;
    const int row = lane & 15, kq = lane >> 4;
    const int rp = row / M, rv = row - rp * M;

    if constexpr (VGRAM) {
        // 16 lanes per problem, lane l16 over the features k = l16 (mod 16): every row is read in
        // full 128-byte lines, each lane accumulates all M (M + 1) / 2 products of its features, and
        // a reduce-scatter leaves one finished Gram entry per lane.  For M <= 8 this beats the
        // 16 x 16 matrix-core tile, most of which (the blocks between different problems) is waste.
        const int grp = lane >> 4, l16 = lane & 15;
        for (int pass = 0; pass < 16; ++pass) {
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
                    const int pos = a.pos_begin + g / a.B, c = g - (g / a.B) * a.B;
                    qid = a.bq[pos];
                    slot = (size_t)c * a.Kcap + pos;
                    n = a.lists.cnt[slot];
                    if (l16 < n) idm = a.lists.idx[slot * m + l16];
                }
            }
            bool changed = valid;
            if (!INDEXED && a.prev.idx != nullptr && valid)
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
    } else
    for (int t = 0; t < NT; ++t) {
        const int gt = g0 + t * PPT;
        if (gt >= nprob) break;
        const int g = gt + rp;
        const bool valid = rp < PPT && g < nprob;
        int n = 0, id = 0, qid = 0;
        if (valid) {
            if (INDEXED) {
                qid = xq[g];
                n = xn[g];
                if (rv < n) id = xhull[(size_t)g * m + rv];
            } else {
                const int pos = a.pos_begin + g / a.B, c = g - (g / a.B) * a.B;
                qid = a.bq[pos];
                const size_t slot = (size_t)c * a.Kcap + pos;
                n = a.lists.cnt[slot];
                if (rv < n) id = a.lists.idx[slot * m + rv];
            }
        }
        const bool live = valid && rv < n;
        if (!INDEXED && a.prev.idx != nullptr) {
            // skip the tile when every one of its problems has the vertex list of the previous round
            bool changed = false;
            if (valid) {
                const int pos = a.pos_begin + g / a.B, c = g - (g / a.B) * a.B;
                const size_t slot = (size_t)c * a.Kcap + pos;
                changed = a.prev.cnt[slot] != n || (rv < n && a.prev.idx[slot * m + rv] != id);
            }
            if (!__any(changed)) {
                if (valid && rv == 0) sN[w][t * PPT + rp] = -1;   // keep the stored distance
                continue;
            }
        }
        // The Gram sum over features is order-free, so feature k is assigned to MFMA step / k-slot
        // as k = 16 t + 4 kq + s: every lane then reads 32 contiguous bytes per 4 steps and the 4
        // lanes of a row cover one full 128-byte line.
        const double *vptr = a.X + (size_t)id * a.Dp + 4 * kq;
        const double *qptr = a.X + (size_t)qid * a.Dp + 4 * kq;
        f64x4 acc = {0.0, 0.0, 0.0, 0.0};
        for (int k0 = 0; k0 < a.Dp; k0 += 32) {
            double2 v[4], x[4];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int kk = k0 + 16 * t + 4 * kq;
                const bool in = live && kk < a.Dp;   // Dp % 8 == 0 and kk % 4 == 0: all 4 in range
                v[2 * t] = in ? *reinterpret_cast<const double2 *>(vptr + k0 + 16 * t) : double2{0.0, 0.0};
                v[2 * t + 1] = in ? *reinterpret_cast<const double2 *>(vptr + k0 + 16 * t + 2) : double2{0.0, 0.0};
                x[2 * t] = in ? *reinterpret_cast<const double2 *>(qptr + k0 + 16 * t) : double2{0.0, 0.0};
                x[2 * t + 1] = in ? *reinterpret_cast<const double2 *>(qptr + k0 + 16 * t + 2) : double2{0.0, 0.0};
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const double y0 = v[s].x - x[s].x;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y0, y0, acc, 0, 0, 0);
                const double y1 = v[s].y - x[s].y;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y1, y1, acc, 0, 0, 0);
            }
        }
        // lane holds D[kq + 4r][row]; keep the lower triangles of the diagonal blocks
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = kq + 4 * r;
            const int pi = i / M, vi = i - pi * M;
            if (pi == rp && pi < PPT && vi >= rv && gt + pi < nprob)
                sQ[w][Sym<M>::at(vi, rv)][t * PPT + pi] = acc[r];
        }
        if (valid && rv == 0) sN[w][t * PPT + rp] = n;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's LDS writes have landed

    const int g = g0 + lane;
    if (lane >= PPW || g >= nprob) return;
    double Q[NP];
#pragma unroll
    for (int e = 0; e < NP; ++e) Q[e] = sQ[w][e][lane];
    const int n = sN[w][lane];
    if (n == -1) return;   // unchanged vertex list: a.dist already holds this distance
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
        const int pos = a.pos_begin + g / a.B, c = g - (g / a.B) * a.B;
        a.dist[(size_t)pos * a.B + c] = dist;
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

__device__ __forceinline__ double group_sum16(double v)
{
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) v += __shfl_xor(v, off, 16);
    return v;
}
// smallest key, ties to the lowest lane; key = +inf everywhere gives idx = -1
__device__ __forceinline__ void group_argmin16(double key, int l16, double &kmin, int &idx)
{
    kmin = key;
    idx = key < kInf ? l16 : -1;
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) {
        const double ok = __shfl_xor(kmin, off, 16);
        const int oi = __shfl_xor(idx, off, 16);
        if (oi >= 0 && (idx < 0 || ok < kmin || (ok == kmin && oi < idx))) { kmin = ok; idx = oi; }
    }
}

// The affine sub-problem (Q_SS + s 11^T) b = 1, beta = b / sum(b) is solved through the explicit
// inverse H of the lifted support Gram, kept up to date as vertices enter and leave (lane i holds
// row i of H, zeros outside the support).  Entering / leaving is a bordering / Schur update whose
// broadcasts are independent of each other -- a few shuffle rounds deep, where an elimination from
// scratch is a 16-step dependent chain.  The pivot of the update is the Schur complement delta, the
// same quantity whose collapse marks an affinely dependent support in solve_affine<M>.
struct Inv16 {
    double H[16];
};

// vertex v enters: false (and no change) when it is affinely dependent on the support
__device__ __forceinline__ bool inv16_insert(Inv16 &I, const double (&Qr)[16], double s, unsigned &S, int v,
                                             int l16, int gbase)
{
    // a_j = Q[v][j] + s for j in S (row v of Q lives on lane v); u = H a
    double u = 0.0, a_own = 0.0, avv = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const double aj = __shfl(Qr[j], gbase + v, 64) + s;
        if ((S >> j) & 1u) u = fma(I.H[j], aj, u);
        a_own = (j == l16) ? aj : a_own;
        avv = (j == v) ? aj : avv;
    }
    const bool in = (S >> l16) & 1u;
    double delta = avv - group_sum16(in ? a_own * u : 0.0);
    if (!(delta > 1e-6 * avv)) {
        // A small pivot is decided after one step of iterative refinement against the ORIGINAL rows
        // of Q: the stored inverse carries an error of eps * cond, which must not leak into the
        // test below (a dependent vertex has to come out at delta ~ eps * avv, as the Schur
        // complement of a factorisation does).
        double r = a_own;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const double uj = __shfl(in ? u : 0.0, gbase + j, 64);
            if ((S >> j) & 1u) r = fma(-(Qr[j] + s), uj, r);
        }
        r = in ? r : 0.0;
        double du = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const double rj = __shfl(r, gbase + j, 64);
            if ((S >> j) & 1u) du = fma(I.H[j], rj, du);
        }
        u += du;
        delta = avv - group_sum16(in ? a_own * u : 0.0);
    }
    if (!(delta > 1e-13 * avv)) return false;
    const double inv = 1.0 / delta;
    const double ui = in ? u : 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const double uj = __shfl(ui, gbase + j, 64);
        double h;
        if (l16 == v) h = (j == v) ? inv : -uj * inv;                      // the new row
        else h = (j == v) ? -ui * inv : fma(ui * inv, uj, I.H[j]);         // old rows + the new column
        I.H[j] = h;
    }
    S |= 1u << v;
    if (!in && l16 != v) {
#pragma unroll
        for (int j = 0; j < 16; ++j) I.H[j] = 0.0;
    }
    return true;
}

// vertex r (in S) leaves
__device__ __forceinline__ void inv16_remove(Inv16 &I, unsigned &S, int r, int l16, int gbase)
{
    double hrr = 0.0, own = 0.0;
    double hr[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        hr[j] = __shfl(I.H[j], gbase + r, 64);
        hrr = (j == r) ? hr[j] : hrr;
        own = (j == r) ? I.H[j] : own;
    }
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
    beta = b / sum;
    return sum > 0.0;
}

template <bool INDEXED>
__global__ __launch_bounds__(256, 3) void hull_qp16_kernel(QpArgs a, int nprob, const int *xq, const int *xhull,
                                                        const int *xn, int xm, double *xdist, double *xalpha)
{
    __shared__ double sQ[4][4][16][17];   // [wavefront][problem][row][col], padded
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int grp = lane >> 4, l16 = lane & 15, gbase = lane & ~15;
    const int m = INDEXED ? xm : a.m;
    const int g = (blockIdx.x * 4 + w) * 4 + grp;   // my group's problem
    const bool valid = g < nprob;
    int n = 0, idm = -1, qid = 0;
    size_t slot = 0;
    if (valid) {
        if (INDEXED) {
            qid = xq[g];
            n = xn[g];
            if (l16 < n) idm = xhull[(size_t)g * m + l16];
        } else {
            const int pos = a.pos_begin + g / a.B, c = g - (g / a.B) * a.B;
            qid = a.bq[pos];
            slot = (size_t)c * a.Kcap + pos;
            n = a.lists.cnt[slot];
            if (l16 < n) idm = a.lists.idx[slot * m + l16];
        }
    }
    bool changed = valid;
    if (!INDEXED && a.prev.idx != nullptr && valid)
        changed = a.prev.cnt[slot] != n || (l16 < n && a.prev.idx[slot * m + l16] != idm);
    const bool doit = ((__ballot(changed) >> (16 * grp)) & 0xFFFFull) != 0ull;   // else: keep the stored distance

    // ---- phase 1: Gram of the shifted vertices, one matrix-core tile per problem
    const int row = lane & 15, kq = lane >> 4;
#pragma unroll 1
    for (int p = 0; p < 4; ++p) {
        const int np = __shfl(n, 16 * p, 64);
        const bool go = __shfl(doit ? 1 : 0, 16 * p, 64) != 0 && np > 0;
        if (!go) continue;   // wave-uniform
        const int id = __shfl(idm, 16 * p + row, 64);
        const int q = __shfl(qid, 16 * p, 64);
        const bool live = id >= 0;
        const double *vptr = a.X + (size_t)(live ? id : q) * a.Dp + 4 * kq;
        const double *qptr = a.X + (size_t)q * a.Dp + 4 * kq;
        f64x4 acc = {0.0, 0.0, 0.0, 0.0};
        for (int k0 = 0; k0 < a.Dp; k0 += 32) {
            double2 v[4], x[4];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int kk = k0 + 16 * t + 4 * kq;
                const bool in = kk < a.Dp;   // Dp % 8 == 0 and kk % 4 == 0: all 4 in range
                v[2 * t] = in ? *reinterpret_cast<const double2 *>(vptr + k0 + 16 * t) : double2{0.0, 0.0};
                v[2 * t + 1] = in ? *reinterpret_cast<const double2 *>(vptr + k0 + 16 * t + 2) : double2{0.0, 0.0};
                x[2 * t] = in ? *reinterpret_cast<const double2 *>(qptr + k0 + 16 * t) : double2{0.0, 0.0};
                x[2 * t + 1] = in ? *reinterpret_cast<const double2 *>(qptr + k0 + 16 * t + 2) : double2{0.0, 0.0};
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const double y0 = v[t].x - x[t].x;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y0, y0, acc, 0, 0, 0);
                const double y1 = v[t].y - x[t].y;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y1, y1, acc, 0, 0, 0);
            }
        }
        // lane holds D[kq + 4 r][row]
#pragma unroll
        for (int r = 0; r < 4; ++r) sQ[w][p][kq + 4 * r][row] = acc[r];
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wavefront's LDS writes have landed
    if (!valid || !doit) return;

    // ---- phase 2: one problem per 16-lane group
    double Qr[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) Qr[j] = sQ[w][grp][l16][j];
    const bool mine = l16 < n;
    double alpha = 0.0, val = 0.0;
    if (n <= 0) {
        val = kInf;
    } else {
        double diag = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) diag = (j == l16) ? Qr[j] : diag;
        double scale = mine ? diag : 0.0;
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) scale = fmax(scale, __shfl_xor(scale, off, 16));
        if (!(scale > 0.0)) {   // every vertex coincides with the query (or NaN input)
            alpha = l16 == 0 ? 1.0 : 0.0;
            val = scale == 0.0 ? 0.0 : scale;
        } else if (a.metric == 0) {
            double best;
            int i0;
            group_argmin16(mine ? diag : kInf, l16, best, i0);
            unsigned S = 0u, banned = 0u;
            Inv16 I;
#pragma unroll
            for (int j = 0; j < 16; ++j) I.H[j] = 0.0;
            (void)inv16_insert(I, Qr, scale, S, i0, l16, gbase);   // a single vertex is always independent
            alpha = l16 == i0 ? 1.0 : 0.0;
            const double tol = 1.4210854715202004e-14 * scale;  // 64 eps * scale
            for (int it = 0; it < 3 * 16 + 8; ++it) {
                double gi = 0.0;
#pragma unroll
                for (int j = 0; j < 16; ++j) gi = fma(Qr[j], __shfl(alpha, gbase + j, 64), gi);
                val = group_sum16(alpha * gi);
                double gmin;
                int jb;
                group_argmin16((mine && !(((S | banned) >> l16) & 1u)) ? gi : kInf, l16, gmin, jb);
                if (jb < 0 || !(gmin < val - tol)) break;
                if (!inv16_insert(I, Qr, scale, S, jb, l16, gbase)) {
                    banned |= 1u << jb;
                    continue;
                }
                for (int mi = 0; mi <= 16; ++mi) {
                    double beta;
                    if (!inv16_beta(I, beta)) {   // (degenerate weights: give the vertex up)
                        if ((S >> jb) & 1u) inv16_remove(I, S, jb, l16, gbase);
                        banned |= 1u << jb;
                        break;
                    }
                    const bool in = (S >> l16) & 1u;
                    const bool bad = in && !(beta > 0.0);
                    if (((__ballot(bad) >> (16 * grp)) & 0xFFFFull) == 0ull) {
                        alpha = in ? beta : 0.0;
                        break;
                    }
                    const double den = alpha - beta;
                    double theta;
                    int kr;
                    group_argmin16(bad ? (den > 0.0 ? alpha / den : 0.0) : kInf, l16, theta, kr);
                    const double vnew = alpha + theta * (beta - alpha);
                    alpha = (in && l16 != kr) ? vnew : 0.0;
                    inv16_remove(I, S, kr, l16, gbase);
                    if (kr == jb) banned |= 1u << jb;
                }
            }
            double gi = 0.0;
#pragma unroll
            for (int j = 0; j < 16; ++j) gi = fma(Qr[j], __shfl(alpha, gbase + j, 64), gi);
            val = group_sum16(alpha * gi);
        } else {
            // distance to the AFFINE hull: greedy maximal affinely independent subset (affine_min_norm)
            unsigned S = 0u;
            Inv16 I;
#pragma unroll
            for (int j = 0; j < 16; ++j) I.H[j] = 0.0;
            for (int k = 0; k < n; ++k) (void)inv16_insert(I, Qr, scale, S, k, l16, gbase);
            double beta = 0.0;
            const bool okb = S != 0u && inv16_beta(I, beta);
            alpha = (okb && ((S >> l16) & 1u)) ? beta : 0.0;
            if (!okb) alpha = l16 == 0 ? 1.0 : 0.0;
            double gi = 0.0;
#pragma unroll
            for (int j = 0; j < 16; ++j) gi = fma(Qr[j], __shfl(alpha, gbase + j, 64), gi);
            val = group_sum16(alpha * gi);
        }
    }
    const double dist = n <= 0 ? kInf : sqrt(fmax(val, 0.0));
    if (INDEXED) {
        if (l16 == 0) xdist[g] = dist;
        if (xalpha && l16 < m) xalpha[(size_t)g * m + l16] = alpha;
    } else if (l16 == 0) {
        const int pos = a.pos_begin + g / a.B, c = g - (g / a.B) * a.B;
        a.dist[(size_t)pos * a.B + c] = dist;
    }
}

template <int M, int WV, bool INDEXED>
void launch_one(const QpArgs &a, int nprob, int m, const int *xq, const int *xhull, const int *xn,
                double *xdist, double *xalpha, hipStream_t s)
{
    constexpr int PPW = M <= 8 ? 64 : (16 / M) * (64 / (16 / M));
    const int nwaves = (nprob + PPW - 1) / PPW;
    const int grid = (nwaves + WV - 1) / WV;
    hipLaunchKernelGGL((hull_qp_kernel<M, WV, INDEXED>), dim3(grid), dim3(64 * WV), 0, s, a, nprob,
                       xq, xhull, xn, m, xdist, xalpha);
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
        hipLaunchKernelGGL((hull_qp16_kernel<INDEXED>), dim3((nprob + 15) / 16), dim3(256), 0, s, a, nprob, xq, xhull,
                           xn, m, xdist, xalpha);
}

}  // namespace

void launch_hull_qp(const QpArgs &a, hipStream_t s)
{
    const int nprob = (a.pos_end - a.pos_begin) * a.B;
    dispatch<false>(a, nprob, a.m, nullptr, nullptr, nullptr, nullptr, nullptr, s);
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
