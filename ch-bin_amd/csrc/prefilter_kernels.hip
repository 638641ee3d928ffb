// Stage 1 of the two-stage EXACT nearest-member selection: an fp16 matrix-core shortlist.
//
// The reference selects the m nearest members of every bin from an exact distance row
// (distance_matrix.py:47-62 over distance_matrix.py:33-44).  Computing every one of the N^2
// distances in unfused fp64 is what bounds the sweep, yet only ~m of the ~N/B members of a bin can
// ever be selected.  This file produces, for each (batch position j, bin c), a SHORTLIST that is
// guaranteed to contain the exact top-m; topm_kernels.hip:rescore_kernel then evaluates the exact
// cdist-rounded distance on the shortlist only.  The final result is bit-identical to the
// brute-force path (and is checked against it in tests/).
//
// Shadows.  S is one power of two for the whole fit (all features end up within +-2^14, the
// comfortable middle of the fp16 range; multiplying by it is exact) and mu_g the global mean.
//   member p of bin c :  zh_p = fp16((x_p - mu_c) S)      mu_c = centre of bin c (fixed during a fit)
//                        rho_p = ||zh_p - (x_p - mu_c) S||   measured exactly when the row is built
//                        bias_p = ||zh_p||^2 + 2 <(mu_c - mu_g) S, zh_p>
//   query j (any bin) :  qh_j = fp16((x_j - mu_g) S),  rho_j = ||qh_j - (x_j - mu_g) S||   (once per fit)
//   query j vs bin c  :  N_jc = ||(x_j - mu_c) S||^2 in fp64 (query_norms_kernel, once per fit, for every sample)
// Members are centred on their own bin, so their rounding error is relative to the within-bin spread
// whatever the absolute scale of the features; the query keeps ONE row for all bins.
//
// Guarantee.  With z_j = (x_j - mu_c) S exact,
//     T(j,p) := ||z_j - zh_p||^2 = N_jc + bias_p - 2 <qh_j, zh_p> - 2 <(x_j - mu_g) S - qh_j, zh_p>
// and the matrix core returns acc = -bias_p / 2 + <qh_j, zh_p> (v_mfma_f32_32x32x16_f16: fp16
// products are exact in fp32), so that | T - (N_jc - 2 acc) | <= E(j,c) + 2 rho_j ||zh_p|| with
//     E = 2 r_c + g (Bmax_c + 2 ||qh_j|| snb_c)
// Base members (round 3): -bias_p / 2 rides INSIDE the dot product.  The shadow rows have at least three spare
// columns (D + 3 <= Dz); a member row carries three fp16 pieces h1 + h2 + h3 ~ -bias_p / 2^(kBiasExp + 1) there,
// every query row the constant 2^kBiasExp, so the matrix core adds (h1 + h2 + h3) 2^kBiasExp = -bias_p / 2 - r_p
// with the residual r_p measured exactly when the row is built (r_c = its maximum over the bin; S keeps every
// centred feature inside +-2^11, so |bias_p / 2| < 2^29.4 and the pieces never overflow).  The tile loop then reads
// nothing but the fragments: no bias / norm columns through LDS, no accumulator start values.  The batch's own
// entries (update mode) keep the explicit start value -- they also carry their eligibility columns.
// (accumulation error: g = 2.5e-5 exceeds the worst case (n - 1) u of summing
// the n <= 161 fp32 terms in ANY order with one-ulp truncating adds, u = 2^-23 -- round-to-nearest
// halves it, and the observed error is two orders of magnitude smaller; the terms are the start
// value and the Dz <= 160 exact products; snb_c = max ||zh_p||, Bmax_c = max (||zh_p||^2 + 2 |<..>|)
// over the bin).  The last term is Cauchy-Schwarz on the query's rounding error, damped by the small
// norm of the bin-centred member.  Update mode folds it per member into the accumulator's start value,
// -bias_p / 2 -+ rho_j ||zh_p||, minus for the upper bounds of sweep 0, plus for the lower bounds of
// sweep 1; base mode uses the largest norm of the member's 32-row TILE, d = rho_j sn_tile (P.tsn): one constant per
// (query, tile) that moves the two thresholds instead of every accumulator -- per tile and not per bin, so that a
// stray member (a contig far from the centre of the bin it sits in for the moment) loosens its own tile only.
// By the triangle inequality | S d(j,p) - sqrt(T) | <= rho_p <= rho_bin, hence for every member
//     LB(j,p) <= S d(j,p) <= UB(j,p),   UB/LB = sqrt(N_jc - 2 acc' +- E) +- rho_bin .
// Sweep 0 over a bin learns tau = (an upper bound of) the m-th smallest UB -- at least m members
// are provably within tau -- and sweep 1 shortlists every member with LB <= tau.  Any member of
// the true top-m has d <= tau, hence LB <= tau: it is on the list.  A relative slack of 1e-6
// covers fp32 rounding of the bound arithmetic itself and the fp64 rounding of the exact distances.
// If a shortlist overflows its capacity the (query tile, bin) is flagged and recomputed by the
// brute-force tile kernel, so correctness never depends on the data.
#include "chb_internal.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>

#include <algorithm>

namespace chb {
namespace {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

#ifndef CHB_SL_DEFER
#define CHB_SL_DEFER 0   // 1: a tile's selection code runs one tile late, under the next tile's LDS reads (round 4 experiment:
                         // no gain at 100k x 136 x 64, +21 % / +45 % on the tile-skipping builds, whose registers it spills)
#endif
#ifndef CHB_SL_DMAREP
#define CHB_SL_DMAREP 1   // developer experiment: issue every tile's DMA this many times
#endif
constexpr int kBiasCols = 3;         // spare shadow columns that carry -bias / 2 of a base member
constexpr int kBiasExp = 14;         // ... as (h1 + h2 + h3) 2^kBiasExp; the query rows hold 2^kBiasExp there
constexpr float kGamma = 2.5e-5f;
constexpr float kSlack = 1e-6f;
#ifndef CHB_SL_WAVES
#define CHB_SL_WAVES 4   // wavefronts per workgroup of the shortlist kernel (developer experiment: 8)
#endif
constexpr int kPfW = CHB_SL_WAVES;
constexpr int kPfQ = 32 * kPfW;  // batch positions per workgroup (32 per wavefront)
constexpr int kPfP = 32;   // members per tile
constexpr int kWideMaxSlices = 4;   // wide rows: at most this many 144-column slices (shortlist_wide_kernel)

__device__ __forceinline__ float round_up_f32(double v)
{
    float f = (float)v;
    if ((double)f < v) f = nextafterf(f, INFINITY);
    return f;
}

// fp16 shadow of a scaled feature; `back` is the value the matrix core will see
__device__ __forceinline__ unsigned short f16_shadow(double zs, double &back)
{
    float f = (float)zs;
    f = fminf(fmaxf(f, -65504.f), 65504.f);   // (S keeps real data far inside; NaN-free whatever comes)
    const _Float16 hv = (_Float16)f;
    back = (double)(float)hv;
    return __builtin_bit_cast(unsigned short, hv);
}

// maximum of non-negative floats through an integer atomic on the float bits (NaN never enters)
__device__ __forceinline__ void bound_max(float *slot, float v)
{
    if (v == v && v > 0.f) atomicMax(reinterpret_cast<unsigned int *>(slot), __float_as_uint(v));
}

// ---------------------------------------------------------------------------------------------
// fit-level preparation

// partial column sums of rows [r0, r1) -> part[blockIdx.x][Dp]; summed in block order by the
// second kernel (deterministic)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const double *X, int N, int Dp, int rows_per,
                                                             double *part)
{
    const int r0 = blockIdx.x * rows_per, r1 = min(N, r0 + rows_per);
    for (int k = threadIdx.x; k < Dp; k += 256) {
        double s = 0.0;
        for (int r = r0; r < r1; ++r) s += X[(size_t)r * Dp + k];
        part[(size_t)blockIdx.x * Dp + k] = s;
    }
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const double *part, int nblk, int N, int Dp,
                                                           double *mu_g)
{
    for (int k = threadIdx.x; k < Dp; k += 256) {
        double s = 0.0;
        for (int b = 0; b < nblk; ++b) s += part[(size_t)b * Dp + k];
        mu_g[k] = N > 0 ? s / (double)N : 0.0;
    }
}
// *rmax (float bits, pre-zeroed) = max |x - mu_g| over everything
__global__ __launch_bounds__(256) void absmax_kernel(const double *X, int N, int Dp, const double *mu_g,
                                                     unsigned int *rmax)
{
    float v = 0.f;
    const long long tot = (long long)N * Dp;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < tot; i += (long long)gridDim.x * 256) {
        const double d = fabs(X[i] - mu_g[i % Dp]);
        v = fmaxf(v, d < 3e38 ? round_up_f32(d) : 3e38f);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(rmax, __float_as_uint(v));
}

// Query-side rows of ALL samples (any sample can become a batch position), relative to the global
// mean; gq[p] = {||qh||^2 rounded up, rho rounded up}.  One wavefront per sample.
__global__ __launch_bounds__(256) void global_shadow_kernel(const double *X, int N, int D, int Dp,
                                                            const double *mu_g, double S, unsigned short *Gs,
                                                            int Dz, float2 *gq)
{
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= N) return;
    const double *x = X + (size_t)p * Dp;
    double n2 = 0.0, e2 = 0.0;
    for (int k = lane; k < Dz; k += 64) {
        unsigned short hb = 0;
        if (k < D) {
            const double z = (x[k] - mu_g[k]) * S;
            double zh;
            hb = f16_shadow(z, zh);
            n2 += zh * zh;
            e2 += (zh - z) * (zh - z);
        } else if (k < D + kBiasCols) {
            hb = (unsigned short)((kBiasExp + 15) << 10);   // fp16 2^kBiasExp: multiplies a member's bias pieces
        }
        Gs[(size_t)p * Dz + k] = hb;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        n2 += __shfl_xor(n2, off, 64);
        e2 += __shfl_xor(e2, off, 64);
    }
    if (lane == 0)
        gq[p] = make_float2(round_up_f32(n2 * (1.0 + 1e-12)), round_up_f32(sqrt(e2) * (1.0 + 1e-9) + 1e-300));
}

// centers[c][k] = mean over the members of bin c (0 for an empty bin); one block per bin, a thread
// per feature, members in CSR order
__global__ __launch_bounds__(256) void bin_center_kernel(const double *X, int D, int Dp,
                                                         const int *memb_id, const int *bin_ptr,
                                                         double *centers)
{
    const int c = blockIdx.x;
    const int b = bin_ptr[c], e = bin_ptr[c + 1];
    for (int k = threadIdx.x; k < Dp; k += 256) {
        double s = 0.0;
        if (k < D)
            for (int i = b; i < e; ++i) s += X[(size_t)memb_id[i] * Dp + k];
        centers[(size_t)c * Dp + k] = e > b ? s / (double)(e - b) : 0.0;
    }
}

// Member-side shadow of one sample relative to bin c's centre, computed by one wavefront.
// Returns (on every lane) {bias, rho, ||zh||^2, ||zh||^2 + 2 |<(mu_c - mu_g) S, zh>|}.
// bias_cols: the row also gets -bias / 2 as three fp16 pieces in columns D .. D + 2 (base members); *resid then
// receives | -bias / 2 - (h1 + h2 + h3) 2^kBiasExp |, rounded up.  Without: those columns stay zero.
__device__ __forceinline__ float4 member_shadow_row(const double *x, const double *mu, const double *mu_g,
                                                    double S, int D, int Dz, int lane, unsigned short *zrow,
                                                    bool bias_cols = false, float *resid = nullptr,
                                                    unsigned short *zrow2 = nullptr)
{   // (zrow2: a second copy of the row -- the persistent base pack's)
    double n2 = 0.0, e2 = 0.0, sp = 0.0;
    for (int k = lane; k < Dz; k += 64) {
        unsigned short hb = 0;
        if (k < D) {
            const double z = (x[k] - mu[k]) * S;
            double zh;
            hb = f16_shadow(z, zh);
            n2 += zh * zh;
            e2 += (zh - z) * (zh - z);
            sp += ((mu[k] - mu_g[k]) * S) * zh;
        }
        zrow[k] = hb;
        if (zrow2 != nullptr) zrow2[k] = hb;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        n2 += __shfl_xor(n2, off, 64);
        e2 += __shfl_xor(e2, off, 64);
        sp += __shfl_xor(sp, off, 64);
    }
    if (bias_cols) {
        const double v = -0.5 * (n2 + 2.0 * sp) * (1.0 / (double)(1 << kBiasExp));   // (exact scaling)
        double b1, b2, b3;
        const unsigned short h1 = f16_shadow(v, b1);
        const unsigned short h2 = f16_shadow(v - b1, b2);
        const unsigned short h3 = f16_shadow(v - b1 - b2, b3);
        if (lane == 0) {
            zrow[D] = h1; zrow[D + 1] = h2; zrow[D + 2] = h3;
            if (zrow2 != nullptr) { zrow2[D] = h1; zrow2[D + 1] = h2; zrow2[D + 2] = h3; }
        }
        if (resid != nullptr)
            *resid = round_up_f32(fabs(v - b1 - b2 - b3) * (double)(1 << kBiasExp) * (1.0 + 1e-12) + 1e-300);
    }
    float4 o;
    o.x = (float)(n2 + 2.0 * sp);
    o.y = round_up_f32(sqrt(e2) * (1.0 + 1e-9) + 1e-300);
    o.z = round_up_f32(n2 * (1.0 + 1e-12));
    o.w = round_up_f32((n2 + 2.0 * fabs(sp)) * (1.0 + 1e-12));
    return o;
}

// Per-sample member rows relative to the centre of the sample's CURRENT bin (labels[p] >= 0), with the bias
// columns filled; ms[p] = {residual of the bias pieces, rho, ||zh||^2, amax}.
// ids == nullptr: all samples 0..n-1; otherwise the n listed samples (the batch just committed).
__global__ __launch_bounds__(256) void sample_shadow_kernel(const double *X, int D, int Dp, const int *ids,
                                                            int n, int *labels, int B,
                                                            const double *centers, const double *mu_g, double S,
                                                            unsigned short *Zs, int Dz, float4 *ms,
                                                            const int *new_lab, int *inb, Gate gate)
{
    CHB_GATE(gate);
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int p = ids ? ids[i] : i;
    int c;
    if (new_lab) {   // batch commit: the sample's final label goes out, its batch mark is cleared
        c = new_lab[i];
        if (lane == 0) { labels[p] = c; inb[p] = -1; }
    } else {
        c = labels[p];
    }
    if (c < 0 || c >= B) return;
    float resid = 0.f;
    float4 o = member_shadow_row(X + (size_t)p * Dp, centers + (size_t)c * Dp, mu_g, S, D, Dz, lane,
                                 Zs + (size_t)p * Dz, true, &resid);
    o.x = resid;   // (-bias / 2 itself sits in the row's bias columns; .x carries how far the pieces miss it)
    if (lane == 0) ms[p] = o;
}

// bin of padded row r (pad_ptr ascending, pad_ptr[B] > r)
__device__ __forceinline__ int bin_of_row(const int *pad_ptr, int B, int r)
{
    int lo = 0, hi = B - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (pad_ptr[mid] <= r) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// Gathers the per-sample member rows of the CSR-ordered base members into the padded layout;
// 16 lanes per row.  Padding rows: zero features, bias = +inf (never selectable).
__device__ __forceinline__ void pack_rows_block(const unsigned short *Zs, int D, int Dz,
                                                const int *memb_id, const int *bin_ptr,
                                                const int *pad_ptr, int B, const MemberPack &P, int block, int nblocks)
{
    // the two CSR offset tables go through LDS (up to 2048 bins): the bin search of every row is then a chain of LDS
    // reads instead of six dependent global loads in front of the row's own gather
    constexpr int kLdsBins = 2048;
    __shared__ int s_pad[kLdsBins + 1], s_bin[kLdsBins + 1];
    if (B <= kLdsBins) {
        for (int b = threadIdx.x; b <= B; b += blockDim.x) { s_pad[b] = pad_ptr[b]; s_bin[b] = bin_ptr[b]; }
        __syncthreads();
        pad_ptr = s_pad; bin_ptr = s_bin;
    }
    const int total = pad_ptr[B];
    const int cpr = Dz >> 3;
    const int l16 = threadIdx.x & 15;
    for (int r = block * 16 + (threadIdx.x >> 4); r < total; r += nblocks * 16) {
        const int c = bin_of_row(pad_ptr, B, r);
        const int e = r - pad_ptr[c];
        const bool real = e < bin_ptr[c + 1] - bin_ptr[c];
        const int id = real ? memb_id[bin_ptr[c] + e] : 0;
        for (int cc = l16; cc < cpr; cc += 16) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (real) v = *reinterpret_cast<const uint4 *>(Zs + (size_t)id * Dz + cc * 8);
            else if (cc == (D >> 3)) {
                // padding row: first bias piece = fp16 -inf, so that its accumulator comes out as -inf
                const unsigned hw = 0xFC00u << (16 * (D & 1));
                const int wd = (D & 7) >> 1;
                if (wd == 0) v.x = hw; else if (wd == 1) v.y = hw; else if (wd == 2) v.z = hw; else v.w = hw;
            }
            *reinterpret_cast<uint4 *>(P.Z + (size_t)r * Dz + cc * 8) = v;
        }
        // (the base pack has no per-row columns beside Z: -bias / 2 sits in the row, the bounds are per bin / per tile)
    }
}

// Update mode: the batch's own entries (CSR over bins, eligibility code per entry, see
// aux_kernels.hip) relative to the centre of the bin they are listed under, into the padded layout,
// with the code rewritten as (s, b): "eligible for position q  <=>  s q + b >= 0".
// Block (bin, y); one wavefront per entry.
__global__ __launch_bounds__(256) void pack_centered_kernel(const double *X, int D, int Dp, const int *memb_id,
                                                            const int *memb_code, const int *bin_ptr,
                                                            const int *pad_ptr, const double *centers,
                                                            const double *mu_g, double S, int Dz, MemberPack P, Gate gate)
{
    CHB_GATE(gate);
    const int c = blockIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int b0 = bin_ptr[c], cnt = bin_ptr[c + 1] - b0, r0 = pad_ptr[c];
    const int padded = pad_ptr[c + 1] - r0;
    float m_rho = 0.f, m_nrm = 0.f, m_amax = 0.f;   // this wavefront's share of the bin's bounds
    for (int e = blockIdx.y * 4 + w; e < padded; e += gridDim.y * 4) {
        const int r = r0 + e;
        if (e < cnt) {
            const float4 o = member_shadow_row(X + (size_t)memb_id[b0 + e] * Dp, centers + (size_t)c * Dp, mu_g,
                                               S, D, Dz, lane, P.Z + (size_t)r * Dz);
            m_rho = fmaxf(m_rho, o.y); m_nrm = fmaxf(m_nrm, o.z); m_amax = fmaxf(m_amax, o.w);
            if (lane == 0) {
                P.bias[r] = o.x;
                P.sn[r] = sqrtf(o.z) * (1.0f + 1e-6f);
                const int code = memb_code[b0 + e];
                float sv = 0.f, bv = 0.f;
                if (code > 0) { sv = 1.f; bv = -(float)code; }               // q > code - 1
                else if (code <= -(1 << 30)) { sv = 0.f; bv = -1.f; }        // "q != i" has no affine form:
                                                                             // never routed here (chb_api.hip)
                else if (code < 0) { sv = -1.f; bv = (float)(-code - 2); }   // q < -code - 1
                P.cs[r] = sv; P.cb[r] = bv;
            }
        } else {
            for (int k = lane; k < Dz; k += 64) P.Z[(size_t)r * Dz + k] = 0;
            if (lane == 0) {
                P.bias[r] = INFINITY; P.sn[r] = 0.f;
                P.cs[r] = 0.f; P.cb[r] = 0.f;
            }
        }
    }
    // the bin's bounds {max rho, max ||zh||^2, max amax} accumulate in P.bb[c] (zeroed by the batch-CSR kernel of this
    // round; the shortlist kernel takes the root of .y): the four wavefronts meet in LDS, three atomics per block --
    // a same-address device atomic costs ~0.25 us, so never one per wavefront
    __shared__ float sbd[3][4];
    if (lane == 0) { sbd[0][w] = m_rho; sbd[1][w] = m_nrm; sbd[2][w] = m_amax; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const float v = fmaxf(fmaxf(sbd[threadIdx.x][0], sbd[threadIdx.x][1]), fmaxf(sbd[threadIdx.x][2], sbd[threadIdx.x][3]));
        bound_max(reinterpret_cast<float *>(P.bb) + 4 * c + threadIdx.x, v);
    }
}

// shell_inv[c] = nsh / (1.25 x the largest ||zh|| among the bin's members): the per-fit unit of the CSR's shell key
// (member_key in aux_kernels.hip), from the initially labelled members; 0 for an empty bin.  One block per bin.
__global__ __launch_bounds__(256) void shell_scale_kernel(const float4 *ms, const int *memb_id, const int *bin_ptr, int nsh,
                                                          float *shell_inv)
{
    __shared__ float red[256];
    const int c = blockIdx.x;
    const int b0 = bin_ptr[c], cnt = bin_ptr[c + 1] - b0;
    float u = 0.f;
    for (int e = threadIdx.x; e < cnt; e += 256) {
        const float z = ms[memb_id[b0 + e]].z;
        u = fmaxf(u, z == z ? z : 0.f);
    }
    red[threadIdx.x] = u;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) shell_inv[c] = red[0] > 0.f ? (float)nsh / (1.25f * sqrtf(red[0])) : 0.f;
}

// bb[c] = {largest rounding distance, largest ||zh|| (rounded up), largest ||zh||^2 + 2|<..>|, largest bias residual}
// over the members of bin c, from the per-sample rows the pack is gathered FROM (ms[memb_id[..]]): independent of the
// pack kernel's output, so that both can share one launch
// tsn[tile] (tiles of the padded layout, first tile of the bin = pad0 / 32): the largest ||zh|| of each 32-row tile --
// the query-rounding term of the shortlist kernel is taken per TILE, so that one stray member (a contig far from the
// centre of the bin it sits in for the moment) loosens the bounds of its own tile only, not of the whole bin
// suffix: tsn[t] = the largest ||zh|| of tile t AND of every later tile of the bin (tile skipping: the members come in
// shells of decreasing norm, and a bound that never grows along the bin makes "nobody needs this tile" final for the rest)
__device__ __forceinline__ void bin_bounds_from_source(const float4 *ms, const int *memb_id, const int *bin_ptr, int c,
                                                       float4 *bb, float *tsn, int pad0, bool suffix)
{
    __shared__ float red[4][256];
    const int b0 = bin_ptr[c], cnt = bin_ptr[c + 1] - b0;
    float v = 0.f, u = 0.f, a = 0.f, rs = 0.f;
    for (int e0 = 0; e0 < cnt; e0 += 256) {   // (whole wavefronts: the tile maximum is a half-wavefront reduction)
        const int e = e0 + (int)threadIdx.x;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < cnt) o = ms[memb_id[b0 + e]];
        rs = fmaxf(rs, o.x); v = fmaxf(v, o.y); u = fmaxf(u, o.z); a = fmaxf(a, o.w);
        float tn = o.z;
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) tn = fmaxf(tn, __shfl_xor(tn, off, 64));
        if ((threadIdx.x & 31) == 0 && e < cnt) tsn[pad0 / 32 + e / 32] = sqrtf(tn) * (1.0f + 1e-6f);
    }
    red[0][threadIdx.x] = v; red[1][threadIdx.x] = u; red[2][threadIdx.x] = a; red[3][threadIdx.x] = rs;
    __syncthreads();
    if (suffix && threadIdx.x < 64) {
        // one wavefront, from the last tile backwards in chunks of 64 (the block's own writes: visible behind the barrier)
        const int lane = (int)threadIdx.x, nt = (cnt + 31) / 32;
        float *tb = tsn + pad0 / 32;
        float carry = 0.f;
        for (int hi = nt; hi > 0; hi -= 64) {
            const int t = hi - 64 + lane;
            float x = t >= 0 ? tb[t] : 0.f;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const float o = __shfl_down(x, off, 64);
                if (lane + off < 64) x = fmaxf(x, o);
            }
            x = fmaxf(x, carry);
            if (t >= 0) tb[t] = x;
            carry = __shfl(x, 0, 64);
        }
    }
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off)
            for (int q = 0; q < 4; ++q)
                red[q][threadIdx.x] = fmaxf(red[q][threadIdx.x], red[q][threadIdx.x + off]);
        __syncthreads();
    }
    // {largest rho, largest ||zh|| (rounded up), largest amax, largest residual of the bias pieces}
    if (threadIdx.x == 0) bb[c] = make_float4(red[0][0], sqrtf(red[1][0]) * (1.0f + 1e-6f), red[2][0], red[3][0]);
}

// qn[j][c] = ||(x_j - mu_c) S||^2 in fp64, rounded {up, down}, for every sample j and bin c: the centres are fixed for
// the whole fit, so this is computed ONCE per fit (fit_begin_impl) instead of per batch for the batch's positions (up to
// round 4 these tiles rode in every batch-start launch: 12 us on the critical path of each of the 13 batches of a sweep
// at 100k x 64, and the same work again every sweep).  A 32 samples x 64 bins tile per block, features staged through
// LDS 16 at a time (already scaled by S: exact, and keeps tiny feature scales away from underflow); each thread owns a
// 2 x 4 micro-tile.
struct QnArgs {
    const double *X;
    int D, Dp;
    int N, B;
    const double *centers;
    double S;
    float2 *qn;                 // [N][B]
    unsigned long long *ckey;   // optional, [N] (pre-set to ~0): 64-bit minimum of {N_jc bits, bin} over the bins = the
                                // sample's nearest bin centre (the tile-skipping shortlist launch's query order)
};

__device__ __forceinline__ void query_norms_tile(const QnArgs &q, int tile_x, int tile_y)
{
    const double *X = q.X; const int D = q.D, Dp = q.Dp;
    const int pos_begin = 0, pos_end = q.N, B = q.B;
    const double *centers = q.centers; const double S = q.S; float2 *qn = q.qn;
    constexpr int KC = 16;
    __shared__ double xs[32][KC + 1], cs[64][KC + 1];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int p0 = pos_begin + tile_x * 32, c0 = tile_y * 64;
    // staging roles: centre row tid / 4 with four consecutive features, query row tid / 8 with two
    const int crow_i = tid >> 2, ck = (tid & 3) * 4;
    const int xrow_i = tid >> 3, xk = (tid & 7) * 2;
    const double *xrow = p0 + xrow_i < pos_end ? X + (size_t)(p0 + xrow_i) * Dp : nullptr;
    const double *crow = c0 + crow_i < B ? centers + (size_t)(c0 + crow_i) * Dp : nullptr;
    double acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
    // the global loads of the next 16 features are in flight while the current ones are used (the kernel is a
    // chain of memory round trips otherwise: 9 of them at D = 136)
    double cnx[4], xnx[2];
#pragma unroll
    for (int t = 0; t < 4; ++t) cnx[t] = (crow && ck + t < D) ? crow[ck + t] : 0.0;
#pragma unroll
    for (int t = 0; t < 2; ++t) xnx[t] = (xrow && xk + t < D) ? xrow[xk + t] : 0.0;
    for (int k0 = 0; k0 < D; k0 += KC) {
#pragma unroll
        for (int t = 0; t < 4; ++t) cs[crow_i][ck + t] = cnx[t] * S;
#pragma unroll
        for (int t = 0; t < 2; ++t) xs[xrow_i][xk + t] = xnx[t] * S;
        if (k0 + KC < D) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int k = k0 + KC + ck + t;
                cnx[t] = (crow && k < D) ? crow[k] : 0.0;
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int k = k0 + KC + xk + t;
                xnx[t] = (xrow && k < D) ? xrow[k] : 0.0;
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) {
            double xv[2], cv[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) xv[i] = xs[2 * ty + i][kk];
#pragma unroll
            for (int j = 0; j < 4; ++j) cv[j] = cs[tx + 16 * j][kk];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const double z = xv[i] - cv[j];
                    acc[i][j] += z * z;
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pos = p0 + 2 * ty + i, c = c0 + tx + 16 * j;
            if (pos < pos_end && c < B)
                qn[(size_t)pos * B + c] = make_float2(round_up_f32(acc[i][j] * (1.0 + 1e-12)),
                                                     (float)(acc[i][j] * (1.0 - 1e-6)));
        }
    if (q.ckey != nullptr) {
        // nearest centre of my two positions among this tile's 64 bins: 4 bins here, 16 lanes (tx) per position
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            unsigned long long key = ~0ull;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = c0 + tx + 16 * j;
                if (c < B && acc[i][j] == acc[i][j]) {
                    const unsigned long long k = ((unsigned long long)__float_as_uint((float)acc[i][j]) << 32) | (unsigned)c;
                    key = k < key ? k : key;
                }
            }
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) {
                const unsigned long long o = __shfl_xor(key, off, 64);
                key = o < key ? o : key;
            }
            const int pos = p0 + 2 * ty + i;
            if (tx == 0 && pos < pos_end && key != ~0ull) atomicMin(&q.ckey[pos - pos_begin], key);
        }
    }
}

// qord[0 .. nq) = the positions [pos_begin, pos_end) sorted by their nearest bin centre (counting sort in one workgroup;
// positions without a key last): the order in which the shortlist kernel seats its queries, so that the 32 queries of a
// wavefront (and mostly the 128 of a workgroup) look at a bin from the same side and agree on the tiles they can skip.
// home[b] = the query tile (of kPfQ seats) where the positions nearest to bin b start: the shortlist launch runs the
// query tiles of a bin from there on, wrapping round -- the long work items of a bin (its own neighbourhood) first, the
// short ones (far queries, most of their tiles skipped) last, so that the launch does not end on long ones.
__device__ __forceinline__ void query_order_body(const unsigned long long *ckey, const int *bq, int pos_begin, int nq, int B,
                                                 int *qord, int *home)
{
    extern __shared__ int sh[];   // [B + 1] counts -> cursors, [1024] scan partials
    int *cnt = sh, *part = sh + B + 1;
    const int tid = threadIdx.x;
    for (int b = tid; b <= B; b += 1024) cnt[b] = 0;
    __syncthreads();
    for (int i = tid; i < nq; i += 1024) {
        const unsigned long long k = ckey[bq[pos_begin + i]];
        const unsigned c = (unsigned)(k & 0xffffffffu);
        atomicAdd(&cnt[(k != ~0ull && c < (unsigned)B) ? (int)c : B], 1);
    }
    __syncthreads();
    const int per = (B + 1 + 1023) / 1024;
    const int b0 = min(B + 1, tid * per), b1 = min(B + 1, b0 + per);
    int s = 0;
    for (int b = b0; b < b1; ++b) s += cnt[b];
    part[tid] = s;
    __syncthreads();
    if (tid < 64) {
        int loc = 0;
        for (int i = 0; i < 16; ++i) loc += part[16 * tid + i];
        int inc = loc;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(inc, off, 64);
            if (tid >= off) inc += o;
        }
        int run = inc - loc;
        for (int i = 0; i < 16; ++i) {
            const int v = part[16 * tid + i];
            part[16 * tid + i] = run;
            run += v;
        }
    }
    __syncthreads();
    int run = part[tid];
    for (int b = b0; b < b1; ++b) {
        const int c = cnt[b];
        cnt[b] = run;
        if (b < B) home[b] = run / kPfQ;
        run += c;
    }
    __syncthreads();
    for (int i = tid; i < nq; i += 1024) {
        const unsigned long long k = ckey[bq[pos_begin + i]];
        const unsigned c = (unsigned)(k & 0xffffffffu);
        qord[atomicAdd(&cnt[(k != ~0ull && c < (unsigned)B) ? (int)c : B], 1)] = pos_begin + i;
    }
}

__global__ __launch_bounds__(1024) void query_order_kernel(const unsigned long long *ckey, const int *bq, int pos_begin,
                                                           int nq, int B, int *qord, int *home, Gate gate)
{
    CHB_GATE(gate);
    query_order_body(ckey, bq, pos_begin, nq, B, qord, home);
}

// the same for EVERY batch of a sweep in one launch (the fit loop knows the sweep's batches when it uploads the
// permutation): block b orders the positions [geo[b].y, geo[b].z) of the batch that starts at perm + geo[b].x;
// qord_all[geo[b].x + position index], home_all[b * B + bin]
__global__ __launch_bounds__(1024) void query_order_sweep_kernel(const unsigned long long *ckey, const int *perm, const int4 *geo,
                                                                 int B, int *qord_all, int *home_all)
{
    const int4 g = geo[blockIdx.x];
    if (g.z > g.y)
        query_order_body(ckey, perm + g.x, g.y, g.z - g.y, B, qord_all + g.x + g.y, home_all + (size_t)blockIdx.x * B);
}

__global__ __launch_bounds__(256) void query_norms_kernel(QnArgs q, int nqx)
{
    query_norms_tile(q, (int)blockIdx.x % nqx, (int)blockIdx.x / nqx);
}

// One launch for two independent pieces of a batch start (each used to be a launch of its own, each too small to fill
// the chip): blocks [0, npack) gather the base members' shadow rows into the padded pack (pack_rows_block); blocks
// [npack, npack + B) reduce the per-bin bounds from the same source rows.
__global__ __launch_bounds__(256) void pack_build_kernel(const unsigned short *Zs, const float4 *ms, int D, int Dz,
                                                         const int *memb_id, const int *bin_ptr, const int *pad_ptr,
                                                         int B, MemberPack P, int npack, bool shells, Gate gate)
{
    CHB_GATE(gate);
    const int b = blockIdx.x;
    if (b < npack) pack_rows_block(Zs, D, Dz, memb_id, bin_ptr, pad_ptr, B, P, b, npack);
    else bin_bounds_from_source(ms, memb_id, bin_ptr, b - npack, P.bb, P.tsn, pad_ptr[b - npack], shells);
}


// ---------------------------------------------------------------------------------------------
// the persistent base pack
//
// algorithm.py:50,60 removes ONE contig from its bin and puts it back.  Up to round 3 every batch start rebuilt the CSR
// and the padded member pack of ALL labelled contigs (count / scan / fill over N labels, then a gather of every shadow
// row: 29 MB at 100k x 136, four launches).  Here the pack lives across the batches of a fit (when the shortlist launch
// does not skip tiles: the shell order of the tile-skipping builds needs the rebuild):
//   * every bin owns a REGION of the row arena with room to grow (capacity a multiple of 32 rows; rows past `fill` are
//     padding rows, i.e. never selectable);
//   * a batch start turns the rows of the batch's own members into HOLES (first bias piece = fp16 -inf, the padding rows'
//     encoding; sample -1) -- they are not base members while the batch is open;
//   * a commit writes every member's row back: in place when its label is the one it was removed under (the usual case
//     from sweep 2 on), else appended to its new bin's region; a region that runs full is moved to one twice as large
//     (holes squeezed out) by the fix kernel that follows every commit;
//   * per-tile and per-bin bounds only grow between two builds (valid, at worst looser);
//   * the arena's fill mark rides home with the batch's verdict; the host rebuilds the pack from the labels (compaction)
//     when half of the arena is gone, and at every fit start.
// The order of a bin's rows depends on the order of the appends (atomics) -- as the rebuild's fill already did; the
// selection is exact whatever the order.
constexpr unsigned short kF16NegInf = 0xFC00u;

__device__ __forceinline__ void atomic_max_nonneg(float *p, float v)   // (non-negative floats: their bits order like ints)
{
    if (v > *p) atomicMax(reinterpret_cast<int *>(p), __float_as_int(v));
}
__device__ __forceinline__ void pack_padding_row(unsigned short *zrow, int D, int Dz, int l16)
{
    const int cpr = Dz >> 3;
    for (int cc = l16; cc < cpr; cc += 16) {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (cc == (D >> 3)) {
            const unsigned hw = (unsigned)kF16NegInf << (16 * (D & 1));
            const int wd = (D & 7) >> 1;
            if (wd == 0) v.x = hw; else if (wd == 1) v.y = hw; else if (wd == 2) v.z = hw; else v.w = hw;
        }
        *reinterpret_cast<uint4 *>(zrow + cc * 8) = v;
    }
}

// regions from the compact CSR: capacity = twice the bin's members, and at least `grow` rows -- the host's estimate of what
// a bin will hold once the still unlabelled contigs are in (moving a region is one workgroup's copy: rare, but up to
// 100 us for a 1500-row bin; measured before the estimate existed: a move in most batches of sweep 1); one block
__global__ __launch_bounds__(256) void pack_state_layout_kernel(PackState ps, const int *bin_ptr, int B, int grow)
{
    __shared__ int part[257];
    const int per = (B + 255) / 256;
    const int b0 = min(B, (int)threadIdx.x * per), b1 = min(B, b0 + per);
    auto cap_of = [&](int b) { const int c = bin_ptr[b + 1] - bin_ptr[b]; return (max(2 * c, max(grow, 64)) + 31) / 32 * 32; };
    int sp = 0;
    for (int b = b0; b < b1; ++b) sp += cap_of(b);
    part[threadIdx.x] = sp;
    __syncthreads();
    if (threadIdx.x == 0) {   // (256 partial sums, once per build)
        int run = 0;
        for (int t = 0; t < 256; ++t) { const int v = part[t]; part[t] = run; run += v; }
        part[256] = run;
        ps.ctl[0] = run; ps.ctl[1] = 0;
        if (run > ps.arena_rows) ps.ctl[2] = 1;   // (cannot happen by the host's sizes; fails the fit rather than writing past the arena)
    }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int b = b0; b < b1; ++b) {
        const int c = bin_ptr[b + 1] - bin_ptr[b];
        ps.start[b] = run; ps.cap[b] = cap_of(b); ps.fill[b] = c; ps.live[b] = c; ps.nt[b] = (c + 31) / 32;
        run += cap_of(b);
    }
}

// blocks [0, ngather): rows of the regions (members from the CSR, then padding); blocks [ngather, ngather + B): bounds
__global__ __launch_bounds__(256) void pack_state_build_kernel(PackState ps, MemberPack P, const unsigned short *Zs,
                                                               const float4 *ms, int D, int Dz, const int *memb_id,
                                                               const int *bin_ptr, int B, int ngather)
{
    const int b = blockIdx.x;
    if (b >= ngather) {
        const int c = b - ngather;
        bin_bounds_from_source(ms, memb_id, bin_ptr, c, P.bb, P.tsn, ps.start[c], false);
        // the tiles of the region past the members: no norm yet
        const int t0 = ps.start[c] / 32 + (bin_ptr[c + 1] - bin_ptr[c] + 31) / 32, t1 = (ps.start[c] + ps.cap[c]) / 32;
        for (int t = t0 + (int)threadIdx.x; t < t1; t += 256) P.tsn[t] = 0.f;
        return;
    }
    const int total = min(ps.ctl[0], ps.arena_rows);   // (a layout past the arena has raised ctl[2]: the fit fails, nothing is written beyond)
    const int cpr = Dz >> 3, l16 = threadIdx.x & 15;
    for (int r = b * 16 + (int)(threadIdx.x >> 4); r < total; r += ngather * 16) {
        const int c = bin_of_row(ps.start, B, r);   // (start ascending; the last region ends at `total`)
        const int e = r - ps.start[c];
        const bool real = e < bin_ptr[c + 1] - bin_ptr[c];
        if (real) {
            const int id = memb_id[bin_ptr[c] + e];
            for (int cc = l16; cc < cpr; cc += 16)
                *reinterpret_cast<uint4 *>(P.Z + (size_t)r * Dz + cc * 8) =
                    *reinterpret_cast<const uint4 *>(Zs + (size_t)id * Dz + cc * 8);
            if (l16 == 0) { ps.memb[r] = id; ps.row[id] = r; }
        } else {
            pack_padding_row(P.Z + (size_t)r * Dz, D, Dz, l16);
            if (l16 == 0) ps.memb[r] = -1;
        }
    }
}

// batch start in one launch: block 0 = the plan (tiles per bin, statistics, segment plan: what scan_kernel does for the
// rebuilt pack), blocks [1, 1 + nopen) open the batch
__global__ __launch_bounds__(256) void pack_state_start_kernel(PackState ps, MemberPack P, int D, int Dz, const int *labels,
                                                               int *inb, const int *bq, int K, int *lab_old, int B,
                                                               SegPlan seg, int *stats, int *zero_me, Gate gate)
{
    CHB_GATE(gate);
    const int b = blockIdx.x;
    if (b == 0) {
        __shared__ int s_max, s_tot, s_ng, s_ni;
        if (threadIdx.x == 0) { s_max = 0; s_tot = 0; s_ng = 0; s_ni = 0; ps.ctl[1] = 0; if (zero_me != nullptr) *zero_me = 0; }
        __syncthreads();
        int loc_tot = 0, loc_max = 0;
        for (int c = threadIdx.x; c < B; c += 256) {
            const int nt = (ps.fill[c] + 31) / 32;
            ps.nt[c] = nt;
            loc_tot += nt; loc_max = max(loc_max, nt);
        }
        atomicAdd(&s_tot, loc_tot); atomicMax(&s_max, loc_max);
        __syncthreads();
        const long long tot_tiles = s_tot;
        if (seg.gflag != nullptr)
            for (int c = threadIdx.x; c < B; c += 256) {
                const int ntile = (ps.fill[c] + 31) / 32;
                int g = -1;
                if (seg.launch && ntile > kSegMinTiles && (long long)ntile * B > 4 * tot_tiles) {
                    const int len = max(kSegLenTiles, (ntile + 15) / 16);
                    const int ns = (ntile + len - 1) / len;
                    g = atomicAdd(&s_ng, 1);
                    if (g < seg.gcap) {
                        const int i0 = atomicAdd(&s_ni, ns);
                        for (int sgi = 0; sgi < ns; ++sgi)
                            seg.items[i0 + sgi] = make_int4(c, sgi * len, min(ntile, (sgi + 1) * len),
                                                            (g << 8) | (sgi << 4) | (ns - 1));
                    } else {
                        g = -1;
                    }
                }
                seg.gflag[c] = g;
            }
        __syncthreads();
        if (threadIdx.x == 0) {
            if (seg.gflag != nullptr && seg.nseg != nullptr) *seg.nseg = s_ni;
            if (stats != nullptr) { stats[0] = s_max; stats[1] = s_tot; stats[2] = 0; stats[3] = 0; stats[4] = 0; stats[5] = ps.ctl[0]; stats[6] = 0; stats[7] = 0; }
        }
        return;
    }
    {
        extern __shared__ int gone[];   // [B] members this block takes out of each bin (one global atomic per block and bin)
        for (int c = threadIdx.x; c < B; c += 256) gone[c] = 0;
        __syncthreads();
        const int i = (b - 1) * 256 + (int)threadIdx.x;
        if (i < K) {
            const int p = bq[i];
            const int l = labels[p];
            lab_old[i] = l;
            inb[p] = i;
            if (l >= 0 && l < B) {
                const int r = ps.row[p];
                if (r >= 0) {
                    P.Z[(size_t)r * Dz + D] = kF16NegInf;   // a hole: never selectable while the batch is open
                    ps.memb[r] = -1;
                    atomicAdd(&gone[l], 1);
                }
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < B; c += 256)
            if (gone[c] > 0) atomicSub(&ps.live[c], gone[c]);
    }
}

// Where every committed sample's row goes, decided for 256 samples per block so that a bin's row counter sees ONE atomic
// per block instead of one per sample (128 same-address device atomics in a row cost ~30 us): dest[i] = 2 * row + fresh
// (fresh: a new row of the bin), or -1 (no label, or the bin's region is full: the sample is on the overflow list).
__global__ __launch_bounds__(256) void pack_state_slots_kernel(PackState ps, const int *ids, int n, int B, const int *new_lab,
                                                               const int *lab_old, int *dest, Gate gate)
{
    CHB_GATE(gate);
    extern __shared__ int app[];   // [B] per bin: appends of the block (low half; then: their first slot) | rows restored in place (high half)
    for (int b = threadIdx.x; b < B; b += 256) app[b] = 0;
    __syncthreads();
    const int i = blockIdx.x * 256 + (int)threadIdx.x;
    int c = -1, r0 = -1, rank = -1;
    bool restore = false;
    if (i < n) {
        c = new_lab[i];
        if (c >= 0 && c < B) {
            r0 = ps.row[ids[i]];
            restore = lab_old[i] == c && r0 >= 0;
            if (restore) atomicAdd(&app[c], 0x10000);
            else rank = atomicAdd(&app[c], 1) & 0xffff;
        } else {
            c = -1;
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < B; b += 256) {
        const int na = app[b] & 0xffff, nr = app[b] >> 16;   // (at most 256 of either per block)
        int base = 0;
        if (na > 0) base = atomicAdd(&ps.fill[b], na);
        const int within = min(max(ps.cap[b] - base, 0), na);   // (the overflowed ones join `live` in the fix kernel)
        if (nr + within > 0) atomicAdd(&ps.live[b], nr + within);
        app[b] = base;
    }
    __syncthreads();
    if (i < n) {
        int d = -1;
        if (c >= 0) {
            if (restore) d = 2 * r0;
            else {
                const int slot = app[c] + rank;
                if (slot < ps.cap[c]) d = 2 * (ps.start[c] + slot) + 1;
                else ps.ovf[atomicAdd(&ps.ctl[1], 1)] = i;
            }
        }
        dest[i] = d;
    }
}

// batch commit: sample_shadow_kernel's commit form + the member's row back into the pack
__global__ __launch_bounds__(256) void pack_state_commit_kernel(PackState ps, MemberPack P, const double *X, int D, int Dp,
                                                                const int *ids, int n, int *labels, int B,
                                                                const double *centers, const double *mu_g, double S,
                                                                unsigned short *Zs, int Dz, float4 *ms, const int *new_lab,
                                                                const int *dest_of, int *inb, Gate gate)
{
    CHB_GATE(gate);
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int p = ids[i];
    const int c = new_lab[i];
    if (lane == 0) { labels[p] = c; inb[p] = -1; }
    if (c < 0 || c >= B) {
        if (lane == 0) ps.row[p] = -1;   // (its old row, if any, stays a hole)
        return;
    }
    const int d = dest_of[i];
    const int dest = d >= 0 ? d >> 1 : -1, fresh = d >= 0 ? d & 1 : 0;
    float resid = 0.f;
    float4 o = member_shadow_row(X + (size_t)p * Dp, centers + (size_t)c * Dp, mu_g, S, D, Dz, lane,
                                 Zs + (size_t)p * Dz, true, &resid, dest >= 0 ? P.Z + (size_t)dest * Dz : nullptr);
    o.x = resid;
    if (lane == 0) {
        ms[p] = o;
        if (dest >= 0) {
            ps.memb[dest] = p; ps.row[p] = dest;
            if (fresh) {
                // a new row of the bin: its tile's and its bin's bounds (bin_bounds_from_source's quantities) can only grow
                atomic_max_nonneg(&P.tsn[dest >> 5], sqrtf(o.z) * (1.0f + 2e-6f));
                float *bb = reinterpret_cast<float *>(&P.bb[c]);
                atomic_max_nonneg(bb + 0, o.y);
                atomic_max_nonneg(bb + 1, sqrtf(o.z) * (1.0f + 2e-6f));
                atomic_max_nonneg(bb + 2, o.w);
                atomic_max_nonneg(bb + 3, o.x);
            }
        }
    }
}

// behind every commit, one block per bin: a bin whose region ran full (fill > cap: the commit's appends past the
// capacity are on the overflow list) moves to a region of three times its size -- members squeezed together, the overflowed
// samples appended, the rest padding.  Exits at once for every other bin.
__global__ __launch_bounds__(256) void pack_state_fix_kernel(PackState ps, MemberPack P, const unsigned short *Zs,
                                                             const float4 *ms, int D, int Dz, const int *ids,
                                                             const int *new_lab, Gate gate)
{
    CHB_GATE(gate);
    const int c = blockIdx.x;
    const int cap = ps.cap[c], fill0 = ps.fill[c];
    if (fill0 <= cap) return;   // (nobody writes the bin's records in this launch then: every wavefront of the block sees the same)
    __shared__ int s_new, s_ncap, s_cnt, s_src[256];
    const int nov = ps.ctl[1];
    const int old = ps.start[c];
    const int tid = threadIdx.x, l16 = tid & 15, cpr = Dz >> 3;
    // every wavefront has read the OLD capacity, start and fill before thread 0 changes any of them (a wavefront that
    // started late would otherwise take the early return above against the new capacity and miss the barriers below)
    __syncthreads();
    if (tid == 0) {
        const int want = ps.live[c] + (fill0 - cap);
        const int nc = (3 * want + 64 + 31) / 32 * 32;   // (a bin that outgrew the estimate keeps growing: fewer moves)
        int st = atomicAdd(&ps.ctl[0], nc);
        if (st + nc > ps.arena_rows) { ps.ctl[2] = 1; st = -1; }   // (the host sizes the arena so that this cannot happen)
        s_new = st; s_ncap = nc; s_cnt = 0;
    }
    __syncthreads();
    const int nst = s_new;
    if (nst < 0) { if (tid == 0) ps.fill[c] = cap; return; }
    const int ncap = s_ncap;
    // members of the old region, in row order, 256 rows at a time
    for (int r0 = 0; r0 < cap; r0 += 256) {
        const int r = r0 + tid;
        const int id = r < cap ? ps.memb[old + r] : -1;
        const unsigned long long bal = __ballot(id >= 0);
        __shared__ int s_wave[4];
        if ((tid & 63) == 0) s_wave[tid >> 6] = __popcll(bal);
        __syncthreads();
        int before = __popcll(bal & ((1ull << (tid & 63)) - 1ull));
        for (int w = 0; w < (tid >> 6); ++w) before += s_wave[w];
        const int chunk = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        const int base = s_cnt;
        if (id >= 0) s_src[before] = old + r;
        __syncthreads();
        // (a flat loop over the chunk's 16-byte pieces: independent iterations, many loads in flight -- a loop of 16 rows
        //  at a time made a 1500-row bin's move take 100 us)
#pragma unroll 4
        for (int idx = tid; idx < chunk * cpr; idx += 256) {
            const int j = idx / cpr, cc = idx - j * cpr;
            *reinterpret_cast<uint4 *>(P.Z + (size_t)(nst + base + j) * Dz + cc * 8) =
                *reinterpret_cast<const uint4 *>(P.Z + (size_t)s_src[j] * Dz + cc * 8);
        }
        if (tid < chunk) { const int sid = ps.memb[s_src[tid]]; ps.memb[nst + base + tid] = sid; ps.row[sid] = nst + base + tid; }
        __syncthreads();
        if (tid == 0) s_cnt = base + chunk;
        __syncthreads();
    }
    // the commit's overflowed appends of this bin
    for (int k0 = 0; k0 < nov; k0 += 256) {
        const int k = k0 + tid;
        int i = -1;
        if (k < nov) { i = ps.ovf[k]; if (new_lab[i] != c) i = -1; }
        const unsigned long long bal = __ballot(i >= 0);
        __shared__ int s_wave2[4];
        if ((tid & 63) == 0) s_wave2[tid >> 6] = __popcll(bal);
        __syncthreads();
        int before = __popcll(bal & ((1ull << (tid & 63)) - 1ull));
        for (int w = 0; w < (tid >> 6); ++w) before += s_wave2[w];
        const int chunk = s_wave2[0] + s_wave2[1] + s_wave2[2] + s_wave2[3];
        const int base = s_cnt;
        if (i >= 0) s_src[before] = ids[i];
        __syncthreads();
#pragma unroll 4
        for (int idx = tid; idx < chunk * cpr; idx += 256) {
            const int j = idx / cpr, cc = idx - j * cpr;
            *reinterpret_cast<uint4 *>(P.Z + (size_t)(nst + base + j) * Dz + cc * 8) =
                *reinterpret_cast<const uint4 *>(Zs + (size_t)s_src[j] * Dz + cc * 8);
        }
        if (tid < chunk) { const int sid = s_src[tid]; ps.memb[nst + base + tid] = sid; ps.row[sid] = nst + base + tid; }
        __syncthreads();
        if (tid == 0) s_cnt = base + chunk;
        __syncthreads();
    }
    const int cnt = s_cnt;
    for (int r = cnt + (tid >> 4); r < ncap; r += 16) {
        pack_padding_row(P.Z + (size_t)(nst + r) * Dz, D, Dz, l16);
        if (l16 == 0) ps.memb[nst + r] = -1;
    }
    // the new tiles' norms from the members' own records (a tile = 32 rows = half a wavefront)
    for (int r0 = 0; r0 < ncap; r0 += 256) {
        const int r = r0 + tid;
        float tn = 0.f;
        if (r < cnt) tn = ms[ps.memb[nst + r]].z;
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) tn = fmaxf(tn, __shfl_xor(tn, off, 64));
        if ((tid & 31) == 0 && r < ncap) P.tsn[(nst + r) >> 5] = sqrtf(tn) * (1.0f + 2e-6f);
    }
    if (tid == 0) { ps.start[c] = nst; ps.cap[c] = ncap; ps.fill[c] = cnt; ps.live[c] = cnt; }
}

// ---------------------------------------------------------------------------------------------
// threshold pools (round 5)
//
// The base shortlist launch used to stream every bin TWICE per query tile: a threshold sweep that learns tau(j, c) (the m-th
// smallest upper bound over the bin's members) and the admission sweep.  But ANY m base members of c bound the m-th nearest
// distance from above, and which members are near a query is mostly decided by where the query's own bin lies: with
// x_j = mu_h + e_j and x_p = mu_c + e_p the part of d^2(j, p) that varies over p is dominated by -2 <mu_h - mu_c, e_p>
// (86 % of its variance on the benchmark generator, tools/pool_tau_probe.py).  So for every ordered pair (bin c, home bin
// h) the 32 base members of c NEAREST TO THE CENTRE OF h are kept as one tile of shadow rows; a query whose nearest centre
// is h learns tau(j, c) from that one tile (5.0-5.1 candidates per pair where the exact threshold gives 5.0; benchmark
// configs[2] / [3] / [4]), and the bin itself is streamed once.  The query's own bin (h == c) keeps the two sweeps: a
// bin's members nearest to its own centre say little about one particular member's neighbourhood.
//   * key of a member p for home h: ||(x_p - mu_h) S||^2 from the fit's table qn; a pool holds the 32 smallest keys among
//     the members it has been offered (build: all labelled samples; later: every batch's ARRIVALS -- a member that
//     arrives replaces the slot with the largest key if its own is smaller);
//   * validity needs only that a usable slot is a base member of c NOW: the slots of the open batch's members are holes
//     while the batch is open (pool_open_kernel), and a commit makes a hole usable again if the sample stayed in c, else
//     empties the slot (pool_update_kernel<false>); ok[c * B + h] = usable rows, and a (query tile, bin) whose pools hold
//     fewer than m of them falls back to the two sweeps;
//   * a tile's rows are copies of the samples' own shadow rows Zs (bias pieces included), empty slots and holes carry the
//     padding rows' -inf bias piece.
// One wavefront per pool: lanes 0 .. 31 hold the slots.
constexpr int kPoolWin = 4096;   // candidates staged per pass of pool_update_kernel

// the slot with the largest key (lowest lane among equals), on every lane
__device__ __forceinline__ void pool_worst(float key, int lane, float &wkey, int &wlane)
{
    float k = lane < kPoolRows ? key : -INFINITY;
    int l = lane;
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) {
        const float ok = __shfl_xor(k, off, 64);
        const int ol = __shfl_xor(l, off, 64);
        if (ok > k || (ok == k && ol < l)) { k = ok; l = ol; }
    }
    wkey = __shfl(k, 0, 64); wlane = __shfl(l, 0, 64);
}

// BUILD: candidates = all members of bin c (CSR), pools start empty.  Otherwise (commit): holes are resolved first (`holes`:
// the batch may have held labelled samples), the candidates are the batch's arrivals of bin c.
// Block (c, y): homes h = y * hb + w, w + 4, ... < min(B, (y + 1) * hb); one wavefront per pool, lanes 0 .. 31 = its slots.
// (Tried in round 5: one block per bin with one HOME per lane, so that a candidate's keys are one coalesced load of its qn
//  row -- 64 .. 200 blocks of long serial loops instead of B x B short wavefronts: 11.9 against 0.58 ms per sweep at
//  100k x 136 x 64.  The parallel form stays.)
template <bool BUILD>
__global__ __launch_bounds__(256) void pool_update_kernel(PoolState ps, const unsigned short *Zs, const float4 *ms,
                                                          const float2 *qn, int D, int Dz, int B, int hb,
                                                          const int *memb_id, const int *bin_ptr,   // BUILD
                                                          const int *ids, int n, const int *new_lab, const int *lab_old,
                                                          const int *labels, int holes, Gate gate)
{
    CHB_GATE(gate);
    __shared__ int s_ids[kPoolWin];
    __shared__ int s_n;
    const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int h_begin = blockIdx.y * hb, h_end = min(B, h_begin + hb);
    const int cpr = Dz >> 3;
    const int src_n = BUILD ? bin_ptr[c + 1] - bin_ptr[c] : n;
    const int nwin = max(1, (src_n + kPoolWin - 1) / kPoolWin);
    for (int win = 0; win < nwin; ++win) {
        __syncthreads();
        if (tid == 0) s_n = 0;
        __syncthreads();
        const int w0 = win * kPoolWin, w1 = min(src_n, w0 + kPoolWin);
        if (BUILD) {
            for (int i = w0 + tid; i < w1; i += 256) s_ids[i - w0] = memb_id[bin_ptr[c] + i];
            if (tid == 0) s_n = w1 - w0;
        } else {
            // (four positions per load: the scan of the batch's labels is every block's first ~K / 256 dependent loads)
            for (int i = w0 + 4 * tid; i < w1; i += 1024) {
                if (i + 4 <= w1) {
                    const int4 nl = *reinterpret_cast<const int4 *>(new_lab + i), lo = *reinterpret_cast<const int4 *>(lab_old + i);
                    if (nl.x == c && lo.x != c) s_ids[atomicAdd(&s_n, 1)] = ids[i];
                    if (nl.y == c && lo.y != c) s_ids[atomicAdd(&s_n, 1)] = ids[i + 1];
                    if (nl.z == c && lo.z != c) s_ids[atomicAdd(&s_n, 1)] = ids[i + 2];
                    if (nl.w == c && lo.w != c) s_ids[atomicAdd(&s_n, 1)] = ids[i + 3];
                } else {
                    for (int k = i; k < w1; ++k)
                        if (new_lab[k] == c && lab_old[k] != c) s_ids[atomicAdd(&s_n, 1)] = ids[k];
                }
            }
        }
        __syncthreads();
        const int ncand = s_n;
        if (!BUILD && ncand == 0 && !(win == 0 && holes)) continue;   // (nothing arrives in this bin, no hole to look for)
        for (int h = h_begin + w; h < h_end; h += 4) {
            const size_t slot = ((size_t)c * B + h) * kPoolRows + (size_t)(lane & 31);
            float key = INFINITY, sn = 0.f;
            int id = -1, hole = 0, changed = 0;
            if (BUILD && win == 0) {
                // empty pool: every row a padding row (finite features, first bias piece -inf)
                for (int r = lane >> 4; r < kPoolRows; r += 4)
                    pack_padding_row(ps.Z + (((size_t)c * B + h) * kPoolRows + r) * Dz, D, Dz, lane & 15);
            } else if (lane < kPoolRows) {
                key = ps.key[slot]; id = ps.id[slot]; sn = ps.sn[slot];
                if (holes) hole = ps.hole[slot];
            }
            if (!BUILD && win == 0 && lane < kPoolRows && hole) {
                if (id >= 0 && labels[id] == c) {
                    ps.Z[slot * Dz + D] = Zs[(size_t)id * Dz + D];   // usable again (the row itself never changed)
                } else {
                    id = -1; key = INFINITY; sn = 0.f;               // left the bin: the slot is empty (its row stays -inf)
                }
                hole = 0; changed |= 2;
            }
            float wkey; int wlane;
            pool_worst(key, lane, wkey, wlane);
            for (int base = 0; base < ncand; base += 64) {
                const int cid = base + lane < ncand ? s_ids[base + lane] : -1;
                const float ck = cid >= 0 ? qn[(size_t)cid * B + h].x : INFINITY;
                unsigned long long mask = __ballot(cid >= 0 && ck < wkey);
                while (mask) {
                    const int b = __ffsll((long long)mask) - 1;
                    mask &= mask - 1;
                    const float bk = __shfl(ck, b, 64);
                    const int bid = __shfl(cid, b, 64);
                    if (bk < wkey) {   // (the bar may have dropped since the ballot)
                        if (lane == wlane) { key = bk; id = bid; changed |= 1; }
                        pool_worst(key, lane, wkey, wlane);
                    }
                }
            }
            // the rows of the slots that changed hands
            const unsigned long long chm = __ballot((changed & 1) != 0 && lane < kPoolRows);
            if (changed & 1) sn = sqrtf(ms[id].z) * (1.0f + 2e-6f);
            for (unsigned long long mm = chm; mm; mm &= mm - 1) {
                const int r = __ffsll((long long)mm) - 1;
                const int rid = __shfl(id, r, 64);
                if (lane < cpr)
                    *reinterpret_cast<uint4 *>(ps.Z + (((size_t)c * B + h) * kPoolRows + r) * Dz + lane * 8) =
                        *reinterpret_cast<const uint4 *>(Zs + (size_t)rid * Dz + lane * 8);
            }
            const bool wr = BUILD || __ballot(changed != 0) != 0ull;   // (a pool nothing happened to is not written back)
            if (wr) {
                float tn = lane < kPoolRows ? sn : 0.f;
#pragma unroll
                for (int off = 16; off >= 1; off >>= 1) tn = fmaxf(tn, __shfl_xor(tn, off, 64));
                if (lane < kPoolRows) { ps.key[slot] = key; ps.id[slot] = id; ps.sn[slot] = sn; ps.hole[slot] = hole; }
                if (lane == 0) ps.tsn[(size_t)c * B + h] = tn;
            }
        }
    }
}

// batch open: one thread per slot
__global__ __launch_bounds__(256) void pool_open_kernel(PoolState ps, const int *inb, int D, int Dz, long long nslot, int *zero_me,
                                                        Gate gate)
{
    CHB_GATE(gate);
    const long long slot = (long long)blockIdx.x * 256 + threadIdx.x;
    if (slot == 0 && zero_me != nullptr) *zero_me = 0;   // (the counter of the second-chance launch's overflow list)
    bool usable = false;
    if (slot < nslot) {
        const int id = ps.id[slot];
        const bool inbatch = id >= 0 && inb[id] >= 0;
        if (inbatch) { ps.hole[slot] = 1; ps.Z[(size_t)slot * Dz + D] = kF16NegInf; }
        usable = id >= 0 && !inbatch;
    }
    const unsigned long long bal = __ballot(usable);
    const int lane = threadIdx.x & 63;
    if ((lane & 31) == 0 && slot < nslot) ps.ok[slot >> 5] = __popc((unsigned)(bal >> (lane & 32)));
}

// ---------------------------------------------------------------------------------------------
// the shortlist kernel
//
// Workgroup = 128 batch positions (32 per wavefront: members are the A operand, queries the B
// operand, so each lane owns one query column) x a run of consecutive bins.  The query fragments
// are loaded once; the members of the run's bins stream through LDS as 32-row tiles.
//   * Dz / 16 (9 or 10 matrix-core steps) is a compile-time constant: all fragment reads of a tile
//     are in flight together and the steps issue back to back;
//   * every global read inside the tile loop is an LDS-DMA (shadow rows in 1-KiB pieces, the
//     member bias / eligibility as 4-byte pieces): no ordinary load whose use would make the
//     compiler drain the DMA queue.  Three LDS buffers, two tiles in flight, one raw s_barrier and
//     one COUNTED s_waitcnt vmcnt(n) per tile (n = this wavefront's DMA instructions per tile).
//     The tile stream runs across sweep and bin boundaries without a bubble;
//   * -bias/2 is the accumulator's start value, so the matrix core returns -t/2 directly: sweep 0
//     is a 16-way maximum, sweep 1 sixteen compares;
//   * shortlist hits are parked per wavefront in LDS as (query, member offset) words through a
//     ballot + mbcnt compaction (no atomics, no divergent store loop) and written out at the end of
//     the bin; the order inside a shortlist is irrelevant (rescore_kernel orders exactly).
// UPD = false: base members of the bin, two sweeps (tau learned, then shortlist).
// UPD = true : the batch's own entries against a FIXED tau = the exact m-th distance of the already
//              known list `seed`; entries that cannot displace a list entry never touch fp64.

// parked shortlist entries per wavefront (32 queries; a pool that fills up sends the wavefront's
// queries to the brute-force fallback).  512 entries keep the workgroup at 37-40 KB of LDS, i.e.
// four workgroups per CU; lists for m > 8 are longer and get 1024 (three per CU).
__host__ __device__ constexpr int shortlist_pool_entries(int ml) { return ml <= 8 ? 512 : 1024; }

__device__ __forceinline__ void wait_vmcnt(int n)   // n: wave-uniform
{
    if (n <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (n == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (n == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (n == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (n == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}

// LDS accesses of the tile loop are issued from inline asm: while an LDS-DMA is in flight the
// compiler puts s_waitcnt vmcnt(0) in front of every LDS access it can see (it cannot tell the
// buffers apart), which would drain the two tiles kept in flight.  The asm reads are completed by
// an explicit s_waitcnt lgkmcnt(0) before their results are used.
template <int OFF>
__device__ __forceinline__ f16x8 lds_read_frag(unsigned addr)
{
    f16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
template <int OFF>
__device__ __forceinline__ f32x4 lds_read_f4(unsigned addr)
{
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
__device__ __forceinline__ void lds_write_u32(unsigned addr, unsigned v)
{
    asm volatile("ds_write_b32 %0, %1" : : "v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void *p)
{
    return (unsigned)(unsigned long)(__attribute__((address_space(3))) const void *)p;
}

template <int ML>
__device__ __forceinline__ void list_insert_desc(float (&l)[ML], float v)
{
#pragma unroll
    for (int i = 0; i < ML; ++i) {
        const float hi = fmaxf(l[i], v);
        v = fminf(l[i], v);
        l[i] = hi;
    }
}

// SEG (base mode only) -- bins much larger than the rest are not streamed by one workgroup per query tile (a bin of
// 40k members is a 1,245-tile stream, twice, while the other workgroups have long finished) but cut into up to 16
// SEGMENTS that run as work items of their own, in two launches:
//   SEG = 0: the ordinary launch; skips the bins the batch's plan (a.seg.gflag) marks as segmented;
//   SEG = 1: one sweep over the segment's tiles, the m best accumulators of every (query, segment) go to a.seg.lists;
//   SEG = 2: merges the lists of ALL segments of the bin (members of different segments are distinct, so the m-th best
//            of the union is exactly what one workgroup streaming the whole bin would have found), derives tau and
//            shortlists the segment's members; segments append to the same (query, bin) shortlist through its
//            global counter.
//
// SKIP (base mode, SEG = 0 only; launched when a.skip is set) -- exact tile skipping for data whose bins are not
// isotropic (several coverage columns).  The host side has ordered every bin's members in shells of decreasing norm
// (CSR key (bin, shell), aux_kernels.hip), made P.tsn a suffix maximum and seated the queries in the order of their nearest
// bin centre (a.qord).  Every member p of tile t then satisfies  S d(j, p) >= ||z_jc|| - ||zh_p|| - rho_p >= zn_lo(j) - tsn[t];
// a tile whose bound exceeds an upper bound of query j's m-th nearest distance (sweep 0: of the lane half's running list;
// sweep 1: the admission bound) holds none of j's top m.  A wavefront none of whose queries needs tile t skips its
// fragment reads, matrix-core and selection work; because the bounds only fall along a bin and the thresholds only
// tighten, the tiles a workgroup can still need are a prefix of the run, and the run ends -- for the issue side of the
// tile stream and, two tiles later, for the consumers -- when the "somebody needs it" flag of the tile about to be
// requested is down (flags: four LDS words, posted three tiles ahead).  These builds take no per-tile-best shortcut (the m
// nearest crowd into few tiles under the shell order) and flush their pools in the middle of a bin.  No asm value of
// theirs lives across a loop iteration: the tile bounds sit in one VGPR (64 tiles, v_readlane) -- see DESIGN.md section 5
// for what the first version's scalar-register window did.
//
// The tile-skipping builds' flush in the MIDDLE of a bin, as a real function call: inlined, its code costs the 128-VGPR
// builds registers inside the tile loop; as a call the live registers are saved around it on this rare path only.
// pool / cnt / qpos: this wavefront's parked entries, per-query counters and positions (LDS, through generic pointers).
__device__ __noinline__ void shortlist_flush_call(const unsigned *pool, int npark, int *cnt, const int *qpos, int *cand_bin,
                                                  int cand_cap, const int *memb, int lane)
{
    for (int i = lane; i < npark; i += 64) {
        const unsigned en = pool[i];
        const int qc = (int)(en >> 27), e = (int)(en & ((1u << 27) - 1u));
        const int off = atomicAdd(&cnt[qc], 1);
        if (off < cand_cap) cand_bin[(size_t)qpos[qc] * cand_cap + off] = memb[e];
    }
}

// POOL (base mode, SEG = 0; launched when a.pool.Z is set) -- threshold pools, see "threshold pools" above.  The queries
// are seated by their nearest bin centre (a.qord); a workgroup's seats then span the home bins h_lo .. h_lo + n_home - 1
// (usually one to three).  For a bin c outside that range whose pools (c, h_lo ..) all hold at least m usable rows the
// threshold sweep streams those n_home POOL TILES instead of the bin's tiles -- a lane takes part only in the tile of its
// own home bin, a wavefront only computes the tiles of its own lanes' homes -- and the admission sweep streams the bin as
// before: ntile + n_home tiles instead of 2 ntile.  Every other (workgroup, bin) keeps the two sweeps.
//
// WORK (base mode, the plain two-sweep build) -- the second chance of the pool launches.  A threshold that comes from a pool
// tile is an upper bound of the m-th nearest distance, but for a few (query, bin) pairs a loose one (a query far out in its
// bin: 0.05 % of the pairs at 500k x 140 x 128 admit more than the 128 candidates a shortlist holds).  Those pairs' work
// items (64 positions x a bin) are on the overflow list; before the fp64 brute-force kernel gets them, this launch runs the
// exact two-sweep selection for them alone: one workgroup per listed item (a.worklist / a.nwork, read on the device; items
// beyond the grid go straight on to the brute-force list), seats 0 .. 63 = the item's positions.
template <int ML, bool UPD, int KS, int SEG = 0, bool SKIP = false, bool POOL = false, bool WORK = false>
__global__ __launch_bounds__(64 * kPfW, (ML <= 5 && (KS == 9 || (!UPD && SEG != 2))) ? 4 : 3) void shortlist_kernel(ShortlistArgs a, int nqt, int nchunk,
                                                                         int bpw, int *flags64, int nqt64, Gate gate)
{
    CHB_GATE(gate);
    static_assert(!WORK || (!UPD && SEG == 0 && !SKIP && !POOL), "the work-list form is the plain base build");
    static_assert(SEG == 0 || !UPD, "segments exist for base members only");
    static_assert(!POOL || (!UPD && SEG == 0), "threshold pools serve the ordinary base launch");
#ifdef CHB_DEV_KNOBS
    const unsigned long long dbg_t0 = wall_clock64();
    int dbg_tiles = 0;
#endif
    constexpr int CPR = 2 * KS;              // 16-byte chunks per shadow row
    constexpr int ROWB = 32 * KS;            // bytes per shadow row
    constexpr int TILEB = kPfP * ROWB;       // one member tile
    constexpr int METAB = 512;               // floats [0,32) bias | [32,64) ||zh|| (base) or s | [64,96) b | [96,128) ||zh|| (update)
    // (kFour: the 160-column base builds for m <= 5 drop what they do not use -- the bias / norm columns' slots, the
    //  segment flush's bases -- and read their fragments in two halves through the same registers: 4 instead of 3
    //  workgroups per CU)
    constexpr bool kFour = KS == 10 && !UPD && ML <= 5 && SEG != 2;
    constexpr bool kDefer = CHB_SL_DEFER != 0 && !UPD;   // base mode: a tile's selection code runs under the next tile's LDS reads
    constexpr bool kHalf9 = kDefer && KS == 9 && ML <= 5;   // ... and the 144-column builds read their fragments in two halves too
    constexpr int BUFB = kFour ? TILEB : TILEB + METAB;
    constexpr int NBUF = 3;
    constexpr int kPoolW = shortlist_pool_entries(ML);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned *sPool = reinterpret_cast<unsigned *>(smem + NBUF * BUFB);   // [kPfW][kPoolW]
    int *sCnt = reinterpret_cast<int *>(sPool + kPfW * kPoolW);           // [kPfQ]
    float *sTau = reinterpret_cast<float *>(sCnt + kPfQ);                 // [kPfQ] tau of the bin (parked: no register)
    int *sGb = reinterpret_cast<int *>(sTau + kPfQ);                      // [kPfQ] SEG = 2: global base of a flush

    // consecutive workgroups alternate over the 8 XCDs: give each XCD a contiguous range of work
    // items (bins-major), so that the member tiles of a bin stay in one L2
    // (SEG > 0: the work items are the plan's segment items, their number is read from the device)
    const int nitem = SEG == 0 ? nchunk : min(*a.seg.nseg, a.seg.cap);
    const int total = nqt * nitem;
    const int per = (total + 7) >> 3;
    int witem = -1;
    if (WORK) {
        const int nw = *a.nwork;
        if (blockIdx.x == 0)   // (more items than workgroups: the rest keeps its flag and goes to the brute-force kernel)
            for (int i = (int)gridDim.x + (int)threadIdx.x; i < nw; i += (int)blockDim.x) a.flaglist[atomicAdd(a.nflag, 1)] = a.worklist[i];
        if ((int)blockIdx.x >= nw) return;
        witem = a.worklist[blockIdx.x];
        if (threadIdx.x == 0) flags64[witem] = 0;   // (set again below if the exact selection overflows as well)
        __syncthreads();
    } else if ((int)(blockIdx.x >> 3) >= per) return;
    const int W = WORK ? 0 : (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (!WORK && W >= total) return;
    const int chunk = WORK ? witem / nqt64 : W / nqt;
    int qt = WORK ? 0 : W - chunk * nqt;
    // (tile skipping, one bin per workgroup: the bin's query tiles start at its own neighbourhood's -- long items first)
    if (SKIP && a.home != nullptr && bpw == 1) { qt += a.home[chunk]; qt = qt >= nqt ? qt - nqt : qt; }
    int c0 = chunk * bpw, c1 = min(a.B, c0 + bpw);
    int seg_tb = 0, seg_te = 0x3fffffff;   // tile window inside the bin (segments)
    int seg_g = 0, seg_i = 0, seg_n = 1;   // giant-bin slot, segment number, segments of the bin
    if (SEG != 0) {
        const int4 itv = a.seg.items[chunk];
        c0 = itv.x; c1 = c0 + 1; seg_tb = itv.y; seg_te = itv.z;
        seg_g = itv.w >> 8; seg_i = (itv.w >> 4) & 15; seg_n = (itv.w & 15) + 1;
    }

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned smem_base = lds_addr(smem);
    const unsigned pool_base = lds_addr(sPool) + (unsigned)(w * kPoolW * 4);
    const int col = lane & 31, h = lane >> 5;
    const int pos0 = WORK ? a.pos_begin + (witem - chunk * nqt64) * kQTile : a.pos_begin + qt * kPfQ;
    const int m = a.m;

    // my query: lane (col, h) owns B[k = 16 s + 8 h + j][col] of its global-centred row
    // (seated by position, or -- a.qord -- in the order of their nearest bin centre)
    const int qseat = pos0 + 32 * w + col;
    const bool qvalid = qseat < a.pos_end && (!WORK || 32 * w + col < kQTile);
    // (kSeated: only the tile-skipping launches -- and the segment launches that may accompany them -- seat their queries
    //  out of position order; the other builds keep seat == position and need no table)
    constexpr bool kSeated = SKIP || SEG != 0 || POOL;
    const int qpos = (kSeated && a.qord != nullptr && qvalid) ? a.qord[qseat - a.pos_begin] : qseat;
    int *sQpos = reinterpret_cast<int *>(kFour ? sGb : sGb + kPfQ);   // [kPfQ] position of every seat of the workgroup (for the flush)
    // [4] tile skipping: "somebody in the workgroup needs tile t" for t = 0 .. 3 (mod 4); accessed by LDS address only
    const unsigned need_base = lds_addr(sQpos + kPfQ);
    if (kSeated && h == 0) sQpos[32 * w + col] = qpos;
    f16x8 qreg[KS];
    float nq, rg;
    const int sidx = a.bq[qvalid ? qpos : a.pos_end - 1];
    {
        const unsigned short *zq = a.Gs + (size_t)sidx * (KS * 16) + h * 8;
#pragma unroll
        for (int sx = 0; sx < KS; ++sx) qreg[sx] = *reinterpret_cast<const f16x8 *>(zq + sx * 16);
        const float2 g2 = a.gq[sidx];
        nq = g2.x; rg = g2.y;
    }
    const float snq = sqrtf(nq) * (1.0f + kSlack);
    const float qposf = (float)qpos;
    // threshold pools: my query's home bin (its nearest centre), the workgroup's range of home bins and this wavefront's
    int hq = -1, h_lo = 0, n_home = 0x3fffffff, hw_lo = 0, hw_hi = -1;
    if (POOL) {
        if (qvalid) {
            const unsigned long long k = a.ckey[sidx];
            hq = (k != ~0ull && (unsigned)(k & 0xffffffffu) < (unsigned)a.B) ? (int)(k & 0xffffffffu) : -1;
        }
        int lo = qvalid ? (hq < 0 ? -0x40000000 : hq) : 0x7fffffff, hi = qvalid ? (hq < 0 ? 0x40000000 : hq) : -0x7fffffff;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { lo = min(lo, __shfl_xor(lo, off, 64)); hi = max(hi, __shfl_xor(hi, off, 64)); }
        hw_lo = __builtin_amdgcn_readfirstlane(lo); hw_hi = __builtin_amdgcn_readfirstlane(hi);
        int *sHome = sQpos + kPfQ + 4;   // [2 kPfW] behind the tile-skipping flags
        if (lane == 0) { sHome[2 * w] = hw_lo; sHome[2 * w + 1] = hw_hi; }
        __syncthreads();
        int glo = 0x7fffffff, ghi = -0x7fffffff;
#pragma unroll
        for (int ww = 0; ww < kPfW; ++ww) { glo = min(glo, sHome[2 * ww]); ghi = max(ghi, sHome[2 * ww + 1]); }
        // (a query without a key, or seats spanning many homes: no pool for this workgroup)
        if (ghi >= glo && glo >= 0 && ghi - glo < kPoolMaxHomes && ghi < a.B) { h_lo = glo; n_home = ghi - glo + 1; }
        h_lo = __builtin_amdgcn_readfirstlane(h_lo); n_home = __builtin_amdgcn_readfirstlane(n_home);
    }
    // does this workgroup take bin c's threshold from the pools (c, h_lo ..)?  (wave-uniform; the issue side and the
    // consumers of the tile stream ask at different times and must get the same answer: a.pool.ok is constant here)
    // (decided ONCE per workgroup, for all its bins, and kept as a bit mask: the issue side and the consumers then test a bit.
    //  With the decision as a function that loops over the pools' counters at every use, hipcc 7.2 mis-scheduled the
    //  assignments behind the loop in two of its inlined copies -- a dropped base pointer (GPU memory fault), then, in the
    //  copy inside the tile loop, short shortlists for the second bin of a workgroup: experiments 12 and 32 of round 5)
    unsigned long long pool_bins = 0ull;
    if (POOL && n_home <= kPoolMaxHomes && c1 - c0 <= 64) {
        for (int c = c0; c < c1; ++c) {
            int bad = (c >= h_lo && c < h_lo + n_home) ? 1 : 0;
            for (int t = 0; t < n_home; ++t) bad |= a.pool.ok[(size_t)c * a.B + h_lo + t] < a.m ? 1 : 0;
            pool_bins |= (unsigned long long)(bad ^ 1) << (c - c0);
        }
    }
    {
        const unsigned lo32 = __builtin_amdgcn_readfirstlane((unsigned)pool_bins), hi32 = __builtin_amdgcn_readfirstlane((unsigned)(pool_bins >> 32));
        pool_bins = ((unsigned long long)hi32 << 32) | lo32;
    }
    auto pool_mode_of = [&](int c) -> bool { return POOL && ((pool_bins >> (c - c0)) & 1ull) != 0ull; };

    // DMA roles: wavefront w moves the 1-KiB pieces w, w+4, w+8 of a tile (lane l of piece i fills
    // LDS chunk 64 i + l from the global chunk the swizzle maps there: chunk cs of row r sits at
    // cs ^ f(r), f = (r >> 4) & 1 for 18 chunks per row, (r >> 2) & 3 for 20, which makes the
    // ds_read_b128 fragment reads conflict-free); wavefront 3 also moves the bias (+ s) column,
    // wavefront 2 the b column in update mode
    int src_off[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int pch = (w + kPfW * j) * 64 + lane;
        const int r = pch / CPR, cs = pch - r * CPR;
        const int f = KS == 9 ? ((r >> 4) & 1) : ((r >> 2) & 3);
        src_off[j] = (r * CPR + (cs ^ f)) * 16;
    }
    int n_w = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) n_w += (w + kPfW * j < KS) ? CHB_SL_DMAREP : 0;
    n_w += (UPD && w == 3) ? 1 : 0;   // (base mode moves no bias / norm columns: -bias / 2 rides in the rows)
    n_w += (UPD && w == 2) ? 1 : 0;
    const unsigned char *zall = reinterpret_cast<const unsigned char *>(a.P.Z);

    // ---- issue side of the tile stream: (bin ic, sweep isw, tile it), two tiles ahead of the consumer
    int ic = c0, it = 0, isw = (UPD || SEG == 2) ? 1 : 0, ibuf = 0;
    int irow0 = 0, int_ = 0;   // first padded row and tile count of bin ic
    bool ivalid = false;
    // (POOL: the current run's source and extent -- the pool tiles (ic, h_lo ..) in the threshold sweep of a pool-mode bin,
    //  else the bin's own tiles, whose first row and count are kept in brow0 / bnt.  The pool tiles live in a buffer of their
    //  own; the run's source is kept as ONE 64-bit byte offset from the pack's base -- with a second base pointer selected
    //  per run, hipcc 7.2 dropped the assignment of the pool's base on the path behind pool_mode_of's loop (found in the
    //  ISA after a GPU memory fault: the pool tile's row offset was applied to the pack))
    const long long pool_delta = (long long)reinterpret_cast<unsigned long long>(a.pool.Z) -
                                 (long long)reinterpret_cast<unsigned long long>(a.P.Z);
    long long ioff = 0;   // byte offset of the current run's first tile from zall
    int brow0 = 0, bnt = 0;
    // move to the first / next non-empty bin
#define CHB_SL_ISSUE_SEEK()                                                                        \
    {                                                                                              \
        ivalid = false;                                                                            \
        while (ic < c1) {                                                                          \
            irow0 = a.P.pad_ptr[ic];                                                               \
            int_ = a.P.nt != nullptr ? a.P.nt[ic] : (a.P.pad_ptr[ic + 1] - irow0) / kPfP;          \
            if (SEG == 0 && !UPD && a.seg.gflag != nullptr && a.seg.gflag[ic] >= 0) int_ = 0;     \
            if (SEG != 0) { irow0 += seg_tb * kPfP; int_ = min(int_, seg_te) - seg_tb; }           \
            if (int_ > 0) { ivalid = true; break; }                                                \
            ++ic;                                                                                  \
        }                                                                                          \
        if (POOL && ivalid) {                                                                      \
            brow0 = irow0; bnt = int_;                                                             \
            const bool pm_ = pool_mode_of(ic);                                                     \
            ioff = pm_ ? pool_delta + (long long)((ic * a.B + h_lo) * kPoolRows) * ROWB : (long long)irow0 * ROWB; \
            int_ = pm_ ? n_home : int_;                                                            \
        }                                                                                          \
    }
#ifdef CHB_DEV_KNOBS
#define CHB_SL_BOUNDS_DMA()                                                                        \
        if (a.viol != nullptr) {                                                                   \
            const long long bo_ = POOL ? ioff + (long long)it * kPfP * ROWB : (long long)row_ * ROWB; \
            const bool in_pack_ = bo_ >= 0 && bo_ + (long long)kPfP * ROWB <= a.viol_rows * ROWB;  \
            const bool in_pool_ = POOL && bo_ >= pool_delta && bo_ + (long long)kPfP * ROWB <= pool_delta + a.viol_pool_rows * ROWB; \
            if (!in_pack_ && !in_pool_) {                                                          \
                if (tid == 0 && atomicCAS(&a.viol[0], 0, 1) == 0) {                                \
                    a.viol[1] = ic; a.viol[2] = it; a.viol[3] = (int)(bo_ / ROWB); a.viol[4] = isw; a.viol[5] = int_; a.viol[6] = (int)(pool_delta >> 8); a.viol[7] = (int)blockIdx.x; \
                }                                                                                  \
                src_ = zall;                                                                       \
            }                                                                                      \
        }
#else
#define CHB_SL_BOUNDS_DMA()
#endif
#define CHB_SL_ISSUE()                                                                             \
    {                                                                                              \
        unsigned char *dst_ = smem + ibuf * BUFB;                                                  \
        const size_t row_ = (size_t)irow0 + (size_t)it * kPfP;                                     \
        const unsigned char *src_ = POOL ? zall + (ioff + (long long)it * (kPfP * ROWB)) : zall + row_ * ROWB; \
        CHB_SL_BOUNDS_DMA()                                                                        \
        _Pragma("unroll") for (int rep_ = 0; rep_ < CHB_SL_DMAREP; ++rep_)                         \
        _Pragma("unroll") for (int j = 0; j < 3; ++j)                                              \
            if (w + kPfW * j < KS)                                                                 \
                __builtin_amdgcn_global_load_lds(src_ + src_off[j],                                \
                    (__attribute__((address_space(3))) void *)(dst_ + (w + kPfW * j) * 1024), 16, 0, 0); \
        if (UPD && w == 3) {                                                                       \
            const float *p_ = (h ? a.P.cs : a.P.bias) + row_ + col;                                \
            __builtin_amdgcn_global_load_lds(p_, (__attribute__((address_space(3))) void *)(dst_ + TILEB), 4, 0, 0); \
        }                                                                                          \
        if (UPD && w == 2) {                                                                       \
            const float *p_ = (h ? a.P.sn : a.P.cb) + row_ + col;                                  \
            __builtin_amdgcn_global_load_lds(p_, (__attribute__((address_space(3))) void *)(dst_ + TILEB + 256), 4, 0, 0); \
        }                                                                                          \
        if (++ibuf == NBUF) ibuf = 0;                                                              \
        ++n_issued;                                                                                \
        if (++it == int_) CHB_SL_ISSUE_NEXTRUN()                                                   \
    }
    // on to the next (bin, sweep) of the stream: at the end of a run of tiles, or -- tile skipping -- earlier
#define CHB_SL_ISSUE_NEXTRUN()                                                                     \
    {                                                                                              \
        it = 0;                                                                                    \
        if (!UPD && SEG == 0 && isw == 0) { isw = 1; if (POOL) { ioff = (long long)brow0 * ROWB; irow0 = brow0; int_ = bnt; } } \
        else { isw = (UPD || SEG == 2) ? 1 : 0; ++ic; CHB_SL_ISSUE_SEEK() }                        \
    }
    int n_issued = 0, n_consumed = 0;
    CHB_SL_ISSUE_SEEK()
    if (ivalid) CHB_SL_ISSUE()
    if (ivalid) CHB_SL_ISSUE()

    // fragment addressing: chunk (2 sx + h) of row `col`, XOR-swizzled as the DMA laid it out
    int fbase0, fbase1;
    if (KS == 9) {
        const int f = (col >> 4) & 1;
        fbase0 = col * ROWB + ((h ^ f) << 4);
        fbase1 = fbase0 + 32;
    } else {
        const int f0 = (col >> 2) & 1, f1 = (col >> 3) & 1;
        fbase0 = col * ROWB + (f1 << 5) + ((h ^ f0) << 4);
        fbase1 = col * ROWB + ((1 ^ f1) << 5) + ((h ^ f0) << 4);
    }
    const unsigned ent0 = ((unsigned)col << 27) | (unsigned)(4 * h);

    // parked entries -> the queries' shortlists (entry = query of this wavefront, member offset in the
    // bin); the per-query counters keep running, so the pool can be emptied in the middle of a bin
    // (SEG = 2: the shortlist is shared with the bin's other segments -- the parked entries are counted per query
    //  first, one global atomic per query reserves their places, then they are written behind that base)
#ifdef CHB_DEV_KNOBS
#define CHB_SL_BOUNDS_FLUSH()                                                                      \
            if (a.viol != nullptr && ((long long)mb_ + e >= a.viol_members || e < 0)) {            \
                if (atomicCAS(&a.viol[0], 0, 2) == 0) { a.viol[1] = c; a.viol[2] = e; a.viol[3] = mb_; a.viol[4] = qc; a.viol[7] = (int)blockIdx.x; } \
                continue;                                                                          \
            }
#else
#define CHB_SL_BOUNDS_FLUSH()
#endif
#define CHB_SL_FLUSH()                                                                             \
    {                                                                                              \
        const int mb_ = a.bin_ptr[c];                                                              \
        const int npark_ = wcnt < kPoolW ? wcnt : kPoolW;                                          \
        if (SEG == 2) {                                                                            \
            for (int i = lane; i < npark_; i += 64)                                                \
                atomicAdd(&sCnt[32 * w + (int)(sPool[w * kPoolW + i] >> 27)], 1);                  \
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");                                 \
            if (h == 0) {                                                                          \
                const int nq_ = sCnt[32 * w + col];                                                \
                int gb_ = 0;                                                                       \
                if (nq_ > 0 && qvalid) gb_ = atomicAdd(&a.cand_cnt[slot], nq_);                    \
                seg_over = seg_over || gb_ + nq_ > a.cand_cap;                                     \
                sGb[32 * w + col] = gb_;                                                           \
                sCnt[32 * w + col] = 0;                                                            \
            }                                                                                      \
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");                                 \
        }                                                                                          \
        for (int i = lane; i < npark_; i += 64) {                                                  \
            const unsigned en = sPool[w * kPoolW + i];                                             \
            const int qc = (int)(en >> 27), e = (int)(en & ((1u << 27) - 1u));                     \
            int off = atomicAdd(&sCnt[32 * w + qc], 1);                                            \
            if (SEG == 2) off += sGb[32 * w + qc];                                                 \
            CHB_SL_BOUNDS_FLUSH()                                                                  \
            if (off < a.cand_cap)                                                                  \
                a.cand[(kSeated ? (size_t)c * a.Kcap + sQpos[32 * w + qc] : (size_t)c * a.Kcap + pos0 + 32 * w + qc) * a.cand_cap + off] = a.memb_id[mb_ + e]; \
        }                                                                                          \
        if (SEG == 2) {                                                                            \
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");                                 \
            if (h == 0) sCnt[32 * w + col] = 0;                                                    \
        }                                                                                          \
    }
    int cbuf = 0;
    int wt_seen = 0, wt_skipped = 0, wt_unloaded = 0;   // tiles this wavefront met / skipped / found not loaded (statistics)
    int pool_cand = 0, pool_pairs = 0;                  // POOL: candidates admitted for / pairs of this lane's query (statistics)
    for (int c = c0; c < c1; ++c) {
        const int row0 = a.P.pad_ptr[c];
        int ntile = a.P.nt != nullptr ? a.P.nt[c] : (a.P.pad_ptr[c + 1] - row0) / kPfP;
        if (SEG == 0 && !UPD && a.seg.gflag != nullptr && a.seg.gflag[c] >= 0) continue;   // the segment launches' bin
        if (SEG != 0) ntile = min(ntile, seg_te) - seg_tb;
        bool seg_over = false;   // SEG = 2: a reservation went past the shortlist's capacity
        const size_t slot = (size_t)c * a.Kcap + (qvalid ? qpos : a.pos_end - 1);
        // ---- per-(query, bin) bounds
        float4 bb = a.P.bb[c];                            // {rho_bin, snb, Bmax, -}
        if (UPD) bb.y = sqrtf(bb.y) * (1.0f + 1e-6f);     // (the batch-entry pack accumulates the largest ||zh||^2)
        const float2 qn2 = a.qn[(size_t)sidx * a.B + c];  // N_jc {up, down}
        // base mode: bias exact up to the measured residual of its pieces (2 r_c on T = N - 2 acc); update mode: bias
        // rounded to fp32 (2^-24 Bmax)
        const float E = ((UPD ? 1.2e-7f * bb.z : 2.0f * bb.w) + a.gamma * (1.001f * bb.z + 2.0f * snq * bb.y)) *
                        (1.0f + 4.0f * kSlack);
        // base mode: the query's rounding error against the largest member norm of each TILE (P.tsn) -- one constant per
        // (query, tile) that shifts the thresholds (update mode: per member, in the accumulator's start value)
        const float rgq = rg * (1.0f + 4.0f * kSlack);
        const float *tsn_c = UPD ? nullptr : a.P.tsn + (row0 >> 5) + seg_tb;

        const float nj_hi = (qn2.x + E) * (1.0f + kSlack);
        const float nj_lo = (qn2.y - E) * (qn2.y > E ? (1.0f - kSlack) : (1.0f + kSlack));
        const float rsum = bb.x * (1.0f + kSlack);
        // tile skipping (base mode, a.skip): every member p of a tile has  S d(j, p) >= ||z_j|| - ||zh_p|| - rho_p
        // >= zn_lo - sn_tile - rho_bin  (triangle inequality through the bin's centre; sn_tile = P.tsn); a tile whose bound
        // exceeds an upper bound of the m-th nearest distance holds no member of the top m.  Sweep 1: the bound hi the
        // admission test uses; sweep 0: the running bound tau_run of the list so far (+inf until it is full).
        const float zn_lo = sqrtf(fmaxf(nj_lo, 0.f)) * (1.0f - 4.0f * kSlack) - rsum;
        float tau_run = INFINITY, hi_s1 = INFINITY;
#ifdef CHB_DEV_KNOBS
        const bool kSkipNever = (a.skip & 2) != 0;   // (CHB_SKIP_NEVER: the skipping build with every tile needed, for A/B timing)
        const bool kNoExit = (a.skip & 4) != 0, kNoWaveSkip = (a.skip & 8) != 0, kNoMidFlush = (a.skip & 16) != 0;
#else
        constexpr bool kSkipNever = false;
#endif
        constexpr bool can_skip = SKIP;   // (its own build: the flags and bounds cost registers the 128-VGPR builds lack)
        // sweep 0: the m largest accumulator values (= m smallest t) seen by this lane half,
        // descending; the first ML - m slots are pinned at +inf so that the m-th largest is lb[ML - 1]
        float lb[ML];
#pragma unroll
        for (int i = 0; i < ML; ++i) lb[i] = i < ML - m ? INFINITY : -INFINITY;
        float thr_s = -INFINITY;
        // sweep 1 admits a member iff its accumulator >= thr2  (t <= C2, t = -2 acc)
        float thr2 = qvalid ? -FLT_MAX : INFINITY;
        if (UPD && qvalid) {
            // the m-th distance among the base members in shadow units: exact from the list `seed`, or
            // the upper bound the base stage left in tau_in (+inf: fewer than m base members)
            float tau = INFINITY;
            if (a.tau_in != nullptr) tau = a.tau_in[slot];
            else if (a.seed.cnt[slot] >= m) tau = round_up_f32(a.seed.d[slot * m + m - 1] * a.S) * (1.0f + kSlack);
            if (tau < INFINITY) {
                const float hi = tau * (1.0f + 4.0f * kSlack) + rsum;
                thr2 = -0.5f * (hi * hi * (1.0f + 4.0f * kSlack) - nj_lo);
            }
        }
        if (h == 0) sCnt[32 * w + col] = 0;
        // (per-tile bests assume that the m nearest members sit in different tiles, i.e. a random member order; with the
        //  shell order of tile skipping they crowd into the same few tiles, so every value competes then)
        // (... nor when the top m are likely to share tile halves anyway -- m^2 / (4 tiles) of them do: at m = 15 and bins of
        //  25-49 tiles the per-tile-best tau is so loose that the hull kernel gathers 17 instead of ~15.5 rows per pair:
        //  39.5 against 43.5 ms per sweep with every value competing, 39.1 with the three best per tile half (kmax below);
        //  at m <= 8 the cheaper sweep 0 wins, at m = 12 it is a tie)
        const bool pmode = POOL && pool_mode_of(c);   // this bin's threshold comes from the pool tiles (c, h_lo ..)
        const bool tile_best = !SKIP && (SEG != 0 || (ntile >= a.tile_best_min && 4 * ntile >= m * m)) && !(!UPD && a.skip != 0);
        // (in between: a bin with enough tiles for the shortcut but crowded tile halves lets its tile_k2 best values per tile
        //  half compete -- a bounded loop)
        const int kmax = (ML > 8 && !SKIP && !tile_best && SEG == 0 && ntile >= a.tile_best_min && !(!UPD && a.skip != 0)) ? a.tile_k2 : 0x7fffffff;
        int wcnt = 0;   // entries parked by this wavefront and not yet written out (wave-uniform)
        if (SEG == 2) {
            // the m best accumulators of every segment of this bin (phase-1 launch): their union's m-th best
            float *sl = a.seg.lists + ((size_t)seg_g * 16 * a.Kcap + (qvalid ? qpos : a.pos_end - 1)) * ML;
            for (int sg = 0; sg < seg_n; ++sg)
#pragma unroll
                for (int i = 0; i < ML; ++i)
                    list_insert_desc<ML>(lb, i >= ML - m ? sl[(size_t)sg * a.Kcap * ML + i] : -INFINITY);
        }

        for (int sweep = (UPD || SEG == 2) ? 1 : 0; sweep < (SEG == 1 ? 1 : 2); ++sweep) {
            const float rgs = (sweep ? rg : -rg) * (1.0f + kSlack);
            // POOL: the threshold sweep of a pool-mode bin runs over the n_home pool tiles; their norms are a.pool.tsn's
            const bool psweep = POOL && pmode && sweep == 0;
            const float *tsn_run = psweep ? a.pool.tsn + (size_t)c * a.B + h_lo : tsn_c;
            // tile skipping: the tiles' norm bounds of this run, 64 of them across the lanes of a VGPR (an ordinary load,
            // once per sweep and then every 61 tiles: its wait drains the tile DMA queue, which at that rate costs nothing;
            // the table has slack behind); a bound is then one v_readlane
            float v_tsn = 0.f;
            int tbase = 0;
            // (the pool sweep of a tile-skipping build skips nothing: it takes the plain builds' path through the tile loop --
            //  no bounds table, no flags, the tile's norm by a scalar load)
            const bool skp = can_skip && !psweep;
            if (SKIP && !psweep && ntile > 0) {
                v_tsn = tsn_run[lane];
                asm volatile("" : "+v"(v_tsn));   // (waited for here, not at a use inside the tile loop)
                // (the "somebody needs tile t" flags of this sweep; nobody reads the previous sweep's any more: its last
                //  tiles' successors in the stream were this sweep's first tiles, which are always loaded)
                if (tid < 4) lds_write_u32(need_base + 4u * (unsigned)tid, 0u);
            }
            if (!UPD && sweep == 1) {
                // end of sweep 0: m-th smallest t over BOTH lane halves -> tau -> thr2
                // (SEG = 2: both halves hold the same merged list of all segments already)
                float mg[ML];
#pragma unroll
                for (int i = 0; i < ML; ++i) mg[i] = lb[i];
                if (SEG != 2) {
#pragma unroll
                    for (int i = 0; i < ML; ++i) {
                        const float o = __shfl_xor(lb[i], 32, 64);
                        list_insert_desc<ML>(mg, i >= ML - m ? o : -INFINITY);   // (not the pinned slots twice)
                    }
                }
                const float ms = mg[ML - 1];
                float tau = INFINITY;
                if (qvalid && ms > -INFINITY) {
                    // tau = m-th smallest upper bound; at least m members are provably within it
                    // (base mode: the list holds accumulators already pushed DOWN by their tile's query-rounding term,
                    //  i.e. upper bounds of t; a member is admitted when its accumulator pushed UP by that term
                    //  reaches thr2: acc >= thr2 - d_tile)
                    const float thr = -2.0f * ms;
                    tau = sqrtf(fmaxf(thr + nj_hi, 0.f)) * (1.0f + 4.0f * kSlack) + rsum;
                    const float hi = tau * (1.0f + 4.0f * kSlack) + rsum;
                    thr2 = -0.5f * (hi * hi * (1.0f + 4.0f * kSlack) - nj_lo);
                    hi_s1 = hi * (1.0f + 4.0f * kSlack);
                }
                // for the fused selection path: tau bounds the m-th distance among these members -- the update
                // stage's threshold and (smallest over the bins) the label guess
                if (h == 0) sTau[32 * w + col] = tau;
            }
            // Tile skipping (SKIP builds).  The tiles' norm bounds never grow along a bin (P.tsn holds suffix maxima, and
            // the thresholds only tighten), so the tiles a query can still need are a PREFIX of the run: every wavefront
            // posts "one of my queries needs tile ct + 3" (LDS flags, made visible by the next barrier); when the flag of
            // the tile about to be requested (two ahead) is down, nobody needs it or any later tile: the run ends two
            // tiles from here, for the issue side (on to the next sweep / bin at once) and for the consumers alike.
            int nt_run = psweep ? n_home : ntile;
            // Everything this (bin, sweep) has loaded so far -- bounds, thresholds, spilled values -- is waited for HERE, by a
            // wait the compiler knows (the builtin, not asm): otherwise its own waitcnt pass finds those loads still pending
            // on the path into the loop and puts `s_waitcnt vmcnt(0)` in front of their first use INSIDE the tile loop, on
            // every iteration (the loop header merges the pre-loop state) -- which drains the two member tiles kept in
            // flight once per tile (round 3's builds did, in the selection code of sweep 1).  Once per run costs nothing.
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
            // The selection code of a tile ("epilogue").  kDefer (base mode): it runs one tile LATE -- between the issue of the
            // next tile's fragment reads and their wait -- so that a wavefront's matrix-core chain runs under its own trip
            // round the barrier, the DMA issue and the next reads instead of being waited for right behind its last
            // instruction, and the selection's vector work hides the LDS latency (MI355X_MICROARCH.md, "Two waves that run
            // the SAME program ...": a deferred epilogue; here for every wavefront, the accumulators are simply left alone
            // until the next tile's products are about to overwrite them -- no second set of registers).
            f32x16 acc;
            bool pend = false;
            float dlt_p = 0.f;
            int ct_p = 0;
            auto epilogue = [&](const float dlt, const int ctp) __attribute__((always_inline)) {
                if (!UPD && sweep == 0) {
                    // acc = -t/2: UB is monotone in t, so the m LARGEST accumulators are kept (pushed down by the
                    // tile's term: upper bounds of t); per tile a 16-way maximum and one branch
                    float mx = acc[0];
#pragma unroll
                    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, acc[r]);
                    if (POOL && psweep && hq != h_lo + ctp) mx = -INFINITY;   // (another home bin's pool tile)
                    if (tile_best && !psweep) {
                        // Large bins: only the BEST of the 16 values enters the list.  The m-th best
                        // of per-(tile, lane half) bests is the m-th best of m distinct members, i.e.
                        // still a valid tau; it is the exact m-th unless two of the top m share a tile
                        // half (probability ~ m^2 / (4 ntile)), and then one rank looser.
                        if (mx - dlt > thr_s) {
                            list_insert_desc<ML>(lb, mx - dlt);
                            thr_s = lb[ML - 1];
                            if (can_skip && thr_s > -INFINITY)
                                tau_run = (sqrtf(fmaxf(-2.0f * thr_s + nj_hi, 0.f)) * (1.0f + 4.0f * kSlack) + rsum) * (1.0f + 4.0f * kSlack);
                        }
                    } else
                    // (the counted form only in the builds for m > 8, the only ones that can reach kmax: in the 128-register
                    //  builds the counter costs the shortlist kernel 1.5 %)
                    for (int left = ML > 8 ? (psweep ? 0x7fffffff : kmax) : 1; left > 0 && mx - dlt > thr_s; left -= ML > 8 ? 1 : 0) {
                        list_insert_desc<ML>(lb, mx - dlt);
                        thr_s = lb[ML - 1];
                        if (can_skip && thr_s > -INFINITY)
                            tau_run = (sqrtf(fmaxf(-2.0f * thr_s + nj_hi, 0.f)) * (1.0f + 4.0f * kSlack) + rsum) * (1.0f + 4.0f * kSlack);
                        float nx = -INFINITY;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            acc[r] = acc[r] == mx ? -INFINITY : acc[r];   // (all copies of a tie at once:
                            nx = fmaxf(nx, acc[r]);                       //  harmless, thr only gets looser)
                        }
                        mx = nx;
                    }
                } else {
                    // make room (wave-uniform).  Only in the builds for m > 8 (update mode: ML = 2), whose
                    // shortlists are long, and in the tile-skipping builds, whose workgroups hold queries of ONE
                    // neighbourhood (in their own bin all of them park candidates); in the 128-VGPR builds the extra
                    // code pushes query fragments into scratch inside this loop
                    // (and in the pool builds: a query whose nearest centre says little about where it lies -- a bin made of
                    //  several clusters -- gets a loose threshold from the pool tile and parks many candidates)
                    if ((SKIP || POOL) && ML <= 8) {
#ifdef CHB_DEV_KNOBS
                        if (wcnt >= kPoolW / 2 && wcnt <= kPoolW && !kNoMidFlush) {
#else
                        if (wcnt >= kPoolW / 2 && wcnt <= kPoolW) {
#endif
                            shortlist_flush_call(sPool + w * kPoolW, wcnt, sCnt + 32 * w, sQpos + 32 * w,
                                                 a.cand + (size_t)c * a.Kcap * a.cand_cap, a.cand_cap,
                                                 a.memb_id + a.bin_ptr[c], lane);
                            wcnt = 0;
                        }
                    } else if ((ML > 8 || (UPD && ML > 1)) && wcnt >= kPoolW / 2 && wcnt <= kPoolW) {
                        CHB_SL_FLUSH()
                        wcnt = 0;
                    }
                    const unsigned ebase = ent0 + (unsigned)((ctp + seg_tb) * kPfP);
                    const float thr_t = thr2 - dlt;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const bool hit = acc[r] >= thr_t;
                        const unsigned long long bal = __ballot(hit);
                        if (bal) {
                            const int before = __builtin_amdgcn_mbcnt_hi(
                                (unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                            const int pos = wcnt + before;
                            if (hit && pos < kPoolW)
                                lds_write_u32(pool_base + 4u * (unsigned)pos,
                                              ebase + (unsigned)((r & 3) + 8 * (r >> 2)));
                            wcnt += __popcll(bal);
                        }
                    }
                }
            };
            for (int ct = 0; ct < nt_run; ++ct) {
                wait_vmcnt(n_issued - n_consumed > 1 ? n_w : 0);   // my pieces of this tile have landed
                if (skp) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (and my flag writes)
                __builtin_amdgcn_s_barrier();   // everybody's have; the buffer two tiles ahead is free again
                if (skp && ct + 2 < nt_run && ct + 2 >= 3) {
                    // (the issue side stands at tile ct + 2 of this very run)
                    // (asm read: the compiler would order an ordinary LDS read behind the tile DMA -- vmcnt(0))
                    int nd;
                    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)"
                                 : "=v"(nd) : "v"(need_base + 4u * (unsigned)((ct + 2) & 3)) : "memory");
#ifdef CHB_DEV_KNOBS
                    if (kNoExit) nd = 1;
#endif
                    if (__builtin_amdgcn_readfirstlane(nd) == 0) {
                        nt_run = ct + 2;
                        CHB_SL_ISSUE_NEXTRUN()
                    }
                }
                if (ivalid) CHB_SL_ISSUE()
                if (skp && tid == 0) lds_write_u32(need_base + 4u * (unsigned)((ct + 1) & 3), 0u);   // (read one tile ago; next written in two)

                const unsigned tb = smem_base + (unsigned)(cbuf * BUFB);
                // rows held by this lane: (r & 3) + 8 (r >> 2) + 4 h
                const unsigned ma = tb + (unsigned)(TILEB + 16 * h);
                float tsn_t = 0.f, tsn_3 = 0.f;   // largest member norm of this tile (and of the rest of the bin) / three tiles on
                if (SKIP && !psweep) {
                    if (ct + 3 - tbase >= 64) { tbase = ct; v_tsn = tsn_run[ct + lane]; asm volatile("" : "+v"(v_tsn)); }   // (waited for here, not at a use)
                    tsn_t = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v_tsn), ct - tbase));
                    tsn_3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v_tsn), ct + 3 - tbase));
                }
                bool do_tile = true;
                if (skp) {
                    const float thr_now = sweep == 0 ? tau_run : hi_s1;
                    // does anybody of this wavefront need tile ct + 3?  (with the threshold as it stands: it only tightens)
                    if (ct + 3 < nt_run) {
                        const bool need3 = (qvalid && !(zn_lo - tsn_3 * (1.0f + kSlack) > thr_now)) || kSkipNever;
                        if (__ballot(need3) != 0ull && lane == 0) lds_write_u32(need_base + 4u * (unsigned)((ct + 3) & 3), 1u);
                    }
                    // and this tile: no fragment reads, no matrix core, no selection code for a wavefront none of whose
                    // queries can find a member of its top m here (the barriers stay: they are the workgroup's)
                    const bool need0 = (qvalid && !(zn_lo - tsn_t * (1.0f + kSlack) > thr_now)) || kSkipNever;
                    ++wt_seen;
#ifdef CHB_DEV_KNOBS
                    if (__ballot(need0) == 0ull && !kNoWaveSkip) { do_tile = false; ++wt_skipped; }
#else
                    if (__ballot(need0) == 0ull) { do_tile = false; ++wt_skipped; }
#endif
                }
                if (POOL && psweep && (h_lo + ct < hw_lo || h_lo + ct > hw_hi)) do_tile = false;   // none of my lanes' home
                if (!do_tile) {
                    if (kDefer && pend) { epilogue(dlt_p, ct_p); pend = false; }
                    ++n_consumed;
                    if (++cbuf == NBUF) cbuf = 0;
                    continue;
                }
#ifdef CHB_DEV_KNOBS
                ++dbg_tiles;
#endif
                if (UPD) {
                    f32x4 nv[4], nn[4], sv[4], bv[4];
#define CHB_SL_META(G)                                                                             \
                    nv[G] = lds_read_f4<32 * (G)>(ma);                                             \
                    nn[G] = lds_read_f4<384 + 32 * (G)>(ma);                                       \
                    sv[G] = lds_read_f4<128 + 32 * (G)>(ma); bv[G] = lds_read_f4<256 + 32 * (G)>(ma);
                    CHB_SL_META(0) CHB_SL_META(1) CHB_SL_META(2) CHB_SL_META(3)
#undef CHB_SL_META
                    // (asm reads are complete once the tied s_waitcnt below returns: no use can be scheduled
                    //  ahead of it.  Two phases -- bias column, then fragments -- keep the peak register count down.)
                    asm volatile("s_waitcnt lgkmcnt(0)"
                                 : "+v"(nv[0]), "+v"(nv[1]), "+v"(nv[2]), "+v"(nv[3]), "+v"(nn[0]), "+v"(nn[1]), "+v"(nn[2]),
                                   "+v"(nn[3])
                                 :
                                 : "memory");
                    asm volatile("" : "+v"(sv[0]), "+v"(sv[1]), "+v"(sv[2]), "+v"(sv[3]), "+v"(bv[0]), "+v"(bv[1]),
                                      "+v"(bv[2]), "+v"(bv[3]) : : "memory");
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            // -bias / 2, pushed up (lower bounds) by the query's rounding error against THIS member's norm
                            acc[4 * g + k] = fmaf(rgs, nn[g][k], -0.5f * nv[g][k]);
                            // a batch member counts for this query only on the right side of the visiting order
                            if (fmaf(sv[g][k], qposf, bv[g][k]) < 0.f) acc[4 * g + k] = -INFINITY;
                        }
                    }
                }
                // base mode (builds without tile skipping): the tile's largest member norm, a SCALAR load (an ordinary vector
                // load here would make the compiler drain the tile DMA queue); it is complete behind the fragment reads'
                // lgkmcnt(0) below
                if (!UPD && (!SKIP || psweep)) {
                    const unsigned long long ta = reinterpret_cast<unsigned long long>(tsn_run + ct);
                    const unsigned ta_lo = __builtin_amdgcn_readfirstlane((unsigned)ta);
                    const unsigned ta_hi = __builtin_amdgcn_readfirstlane((unsigned)(ta >> 32));
                    const unsigned long long ta_s = ((unsigned long long)ta_hi << 32) | ta_lo;
                    asm volatile("s_load_dword %0, %1, 0x0" : "=s"(tsn_t) : "s"(ta_s) : "memory");
                }
                if constexpr (kFour) {
                    // ten fragments through five registers: the second half is requested as the matrix core takes the first
                    const unsigned fa0 = tb + (unsigned)fbase0, fa1 = tb + (unsigned)fbase1;
                    f16x8 h0 = lds_read_frag<0>(fa0), h1 = lds_read_frag<0>(fa1), h2 = lds_read_frag<64>(fa0),
                          h3 = lds_read_frag<64>(fa1), h4 = lds_read_frag<128>(fa0);
                    if (kDefer && pend) epilogue(dlt_p, ct_p);   // (the previous tile's, under this tile's LDS reads)
                    if (SKIP && !POOL)
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(h4) : : "memory");
                    else
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(h4), "+s"(tsn_t) : : "memory");
                    // base members: -bias / 2 is part of the dot product (three bias columns of the row against the query's 2^kBiasExp) -- the
                    // tile loop touches nothing but the fragments
                    if (!UPD) {
                        _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                    }
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h0, qreg[0], acc, 0, 0, 0); h0 = lds_read_frag<128>(fa1);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1, qreg[1], acc, 0, 0, 0); h1 = lds_read_frag<192>(fa0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h2, qreg[2], acc, 0, 0, 0); h2 = lds_read_frag<192>(fa1);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h3, qreg[3], acc, 0, 0, 0); h3 = lds_read_frag<256>(fa0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h4, qreg[4], acc, 0, 0, 0); h4 = lds_read_frag<256>(fa1);
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(h4) : : "memory");
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h0, qreg[5], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1, qreg[6], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h2, qreg[7], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h3, qreg[8], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h4, qreg[9], acc, 0, 0, 0);
                } else if constexpr (kHalf9) {
                    // nine fragments through five registers (as kFour): the deferred selection code of the previous tile runs
                    // while the first five are on their way -- with all nine in flight it would not fit 128 registers
                    const unsigned fa0 = tb + (unsigned)fbase0;
                    f16x8 h0 = lds_read_frag<0>(fa0), h1 = lds_read_frag<32>(fa0), h2 = lds_read_frag<64>(fa0),
                          h3 = lds_read_frag<96>(fa0), h4 = lds_read_frag<128>(fa0);
                    if (pend) epilogue(dlt_p, ct_p);   // (the previous tile's, under this tile's LDS reads)
                    if (SKIP && !POOL)
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(h4) : : "memory");
                    else
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(h4), "+s"(tsn_t) : : "memory");
                    _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h0, qreg[0], acc, 0, 0, 0); h0 = lds_read_frag<160>(fa0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1, qreg[1], acc, 0, 0, 0); h1 = lds_read_frag<192>(fa0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h2, qreg[2], acc, 0, 0, 0); h2 = lds_read_frag<224>(fa0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h3, qreg[3], acc, 0, 0, 0); h3 = lds_read_frag<256>(fa0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h4, qreg[4], acc, 0, 0, 0);
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : : "memory");
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h0, qreg[5], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1, qreg[6], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h2, qreg[7], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h3, qreg[8], acc, 0, 0, 0);
                } else {
                f16x8 af[KS == 9 ? 9 : 10];
                {
                    const unsigned fa0 = tb + (unsigned)fbase0, fa1 = tb + (unsigned)fbase1;
                    if (KS == 9) {
                        af[0] = lds_read_frag<0>(fa0);   af[1] = lds_read_frag<32>(fa0);  af[2] = lds_read_frag<64>(fa0);
                        af[3] = lds_read_frag<96>(fa0);  af[4] = lds_read_frag<128>(fa0); af[5] = lds_read_frag<160>(fa0);
                        af[6] = lds_read_frag<192>(fa0); af[7] = lds_read_frag<224>(fa0); af[8] = lds_read_frag<256>(fa0);
                    } else {
                        af[0] = lds_read_frag<0>(fa0);   af[1] = lds_read_frag<0>(fa1);   af[2] = lds_read_frag<64>(fa0);
                        af[3] = lds_read_frag<64>(fa1);  af[4] = lds_read_frag<128>(fa0); af[5] = lds_read_frag<128>(fa1);
                        af[6] = lds_read_frag<192>(fa0); af[7] = lds_read_frag<192>(fa1); af[8] = lds_read_frag<256>(fa0);
                        af[9] = lds_read_frag<256>(fa1);
                    }
                }
                if (kDefer && pend) epilogue(dlt_p, ct_p);   // (the previous tile's, under this tile's LDS reads)
                if (SKIP && !POOL)
                    asm volatile("s_waitcnt lgkmcnt(0)"
                                 : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(af[4]), "+v"(af[5]),
                                   "+v"(af[6]), "+v"(af[7]), "+v"(af[8])
                                 :
                                 : "memory");
                else
                    asm volatile("s_waitcnt lgkmcnt(0)"
                                 : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(af[4]), "+v"(af[5]),
                                   "+v"(af[6]), "+v"(af[7]), "+v"(af[8]), "+s"(tsn_t)
                                 :
                                 : "memory");
                if (KS == 10) asm volatile("" : "+v"(af[KS - 1]) : : "memory");
                // base members: -bias / 2 is part of the dot product (three bias columns of the row against the query's 2^kBiasExp) -- the
                // tile loop touches nothing but the fragments
                if (!UPD) {
                    _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                }
#pragma unroll
                for (int sx = 0; sx < KS; ++sx)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[sx], qreg[sx], acc, 0, 0, 0);
                }

                // base mode: this tile's query-rounding term (a wave-uniform table read)
                const float dlt = UPD ? 0.f : rgq * tsn_t;
                if (kDefer) { dlt_p = dlt; ct_p = ct; pend = true; }
                else epilogue(dlt, ct);
                ++n_consumed;
                if (++cbuf == NBUF) cbuf = 0;
            }
            if (kDefer && pend) { epilogue(dlt_p, ct_p); pend = false; }   // the run's last tile
            if (skp) wt_unloaded += ntile - nt_run;
        }

        if (SEG == 1) {
            // ---- end of the segment (phase 1): its m best accumulators over BOTH lane halves go out; segment 0 also
            // empties the bin's shortlist counters for the phase-2 launch
            float mg[ML];
#pragma unroll
            for (int i = 0; i < ML; ++i) mg[i] = lb[i];
#pragma unroll
            for (int i = 0; i < ML; ++i) {
                const float o = __shfl_xor(lb[i], 32, 64);
                list_insert_desc<ML>(mg, i >= ML - m ? o : -INFINITY);
            }
            if (qvalid && h == 0) {
                float *sl = a.seg.lists + (((size_t)seg_g * 16 + seg_i) * a.Kcap + qpos) * ML;
#pragma unroll
                for (int i = 0; i < ML; ++i) sl[i] = mg[i];
                if (seg_i == 0) a.cand_cnt[slot] = 0;
            }
            continue;
        }
        // ---- end of the bin: write the parked entries out
        CHB_SL_FLUSH()
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const int ccount = (qvalid && h == 0) ? sCnt[32 * w + col] : 0;
        if (POOL) { pool_cand += ccount; pool_pairs += (qvalid && h == 0) ? 1 : 0; }
        if (qvalid && h == 0) {
            // (SEG = 2: the bin's counter is the sum of its segments' reservations; the last one past the capacity
            //  flags the pair, and the brute-force fallback then rewrites list and count)
            if (SEG != 2) a.cand_cnt[slot] = ccount < a.cand_cap ? ccount : a.cand_cap;
            if (!UPD && a.tau_out != nullptr && (SEG != 2 || seg_i == 0)) a.tau_out[slot] = sTau[32 * w + col];
            if (ccount > a.cand_cap || wcnt > kPoolW || seg_over) {
                atomicAdd(a.overflow, 1);
                const int fi = c * nqt64 + (qpos - a.pos_begin) / kQTile;
                if (atomicExch(&flags64[fi], 1) == 0) a.flaglist[atomicAdd(a.nflag, 1)] = fi;
            }
        }
    }
#ifdef CHB_DEV_KNOBS
    if (a.dbg != nullptr && lane == 0) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned long long *d = a.dbg + ((size_t)blockIdx.x * 4 + w) * 4;
        d[0] = dbg_t0; d[1] = wall_clock64(); d[2] = (unsigned long long)dbg_tiles; d[3] = hw;
    }
#endif
    // (statistics: about 64 workgroups spread evenly over the work items -- the first ones of a launch are the bins' own
    //  neighbourhoods since the query tiles are rotated, where least can be skipped)
    //  (an ODD stride: the work items are (bin, query tile) with the tile running fastest, usually over a power of two)
    if (POOL && a.pool_stat != nullptr && W % (max(1, total >> 6) | 1) == 0) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { pool_cand += __shfl_xor(pool_cand, off, 64); pool_pairs += __shfl_xor(pool_pairs, off, 64); }
        if (lane == 0) { atomicAdd(&a.pool_stat[0], pool_cand); atomicAdd(&a.pool_stat[1], pool_pairs); }
    }
    if (SKIP && a.skip_stat != nullptr && W % (max(1, total >> 6) | 1) == 0 && lane == 0) {
        atomicAdd(&a.skip_stat[0], wt_skipped);
        atomicAdd(&a.skip_stat[1], wt_seen);
        atomicAdd(&a.skip_stat[2], wt_unloaded);
    }
#undef CHB_SL_FLUSH
#undef CHB_SL_BOUNDS_DMA
#undef CHB_SL_BOUNDS_FLUSH
#undef CHB_SL_ISSUE
#undef CHB_SL_ISSUE_NEXTRUN
#undef CHB_SL_ISSUE_SEEK
}

// ---------------------------------------------------------------------------------------------
// WIDE rows (round 5): 157 < D <= 573 -- e.g. the 512 canonical 5-mers of KmerK = 5 plus coverage columns
// (ch_bin/core/features/kmer_count.py:110-125, config/default.ini:10 takes any k).  Up to round 4 such a fit fell back to
// brute-force fp64 selection (tile_kernel: 3 D flops per (query, member) pair).  Here a shadow row is NS = 2 .. 4 SLICES of 144
// columns (Dz = 144 NS; the bias pieces sit in the last slice, columns D .. D + 2, as in the narrow builds), a wavefront holds
// the 9 NS query fragments of its 32 queries in registers (36 NS VGPRs: these builds run at 2 wavefronts per SIMD), and the
// member tiles stream through LDS one slice at a time: the unit of the DMA pipeline is (tile, slice) = 32 rows x 288 bytes,
// laid out in LDS exactly like a tile of the 144-column builds; the accumulator is carried over the NS units of a tile and
// the selection code runs once per tile.  Bounds and thresholds are those of the narrow builds with the accumulation-error
// factor scaled by NS (n <= 577 fp32 terms: (n - 1) 2^-23 <= 6.9e-5 <= 4 x 2.5e-5) and the fit's scale S one binade lower
// (chb_api.hip: |x - mu_c| S < 2^10, so that -bias / 2 <= D 2^20 still fits the three fp16 pieces).  No tile skipping, pools,
// segments or work-list form: the plain two-sweep selection (and the one-sweep update mode for the batch's own entries).
template <int ML, bool UPD, int NS>
__global__ __launch_bounds__(64 * kPfW, 2) void shortlist_wide_kernel(ShortlistArgs a, int nqt, int nchunk, int bpw, int *flags64,
                                                                      int nqt64, Gate gate)
{
    CHB_GATE(gate);
    constexpr int KS = 9;                    // matrix-core steps per slice
    constexpr int CPR = 2 * KS;              // 16-byte chunks per slice of a row
    constexpr int SLB = 32 * KS;             // bytes per slice of a row (= the row stride of a unit in LDS)
    constexpr int ROWB = SLB * NS;           // bytes per shadow row in memory
    constexpr int TILEB = kPfP * SLB;        // one unit (tile, slice) in LDS
    constexpr int METAB = 512;               // update mode: floats [0,32) bias | [32,64) s | [64,96) b | [96,128) ||zh||
    constexpr int BUFB = TILEB + METAB;
    constexpr int NBUF = 3;
    constexpr int kPoolW = shortlist_pool_entries(ML);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned *sPool = reinterpret_cast<unsigned *>(smem + NBUF * BUFB);   // [kPfW][kPoolW]
    int *sCnt = reinterpret_cast<int *>(sPool + kPfW * kPoolW);           // [kPfQ]
    float *sTau = reinterpret_cast<float *>(sCnt + kPfQ);                 // [kPfQ]

    // work items as in shortlist_kernel: a contiguous range of (bin run, query tile) items per XCD
    const int total = nqt * nchunk;
    const int per = (total + 7) >> 3;
    if ((int)(blockIdx.x >> 3) >= per) return;
    const int W = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (W >= total) return;
    const int chunk = W / nqt, qt = W - chunk * nqt;
    const int c0 = chunk * bpw, c1 = min(a.B, c0 + bpw);

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned smem_base = lds_addr(smem);
    const unsigned pool_base = lds_addr(sPool) + (unsigned)(w * kPoolW * 4);
    const int col = lane & 31, h = lane >> 5;
    const int pos0 = a.pos_begin + qt * kPfQ;
    const int m = a.m;

    const int qpos = pos0 + 32 * w + col;
    const bool qvalid = qpos < a.pos_end;
    f16x8 qreg[KS * NS];
    float nq, rg;
    const int sidx = a.bq[qvalid ? qpos : a.pos_end - 1];
    {
        const unsigned short *zq = a.Gs + (size_t)sidx * (KS * NS * 16) + h * 8;
#pragma unroll
        for (int sx = 0; sx < KS * NS; ++sx) qreg[sx] = *reinterpret_cast<const f16x8 *>(zq + sx * 16);
        const float2 g2 = a.gq[sidx];
        nq = g2.x; rg = g2.y;
    }
    const float snq = sqrtf(nq) * (1.0f + kSlack);
    const float qposf = (float)qpos;

    // DMA roles: wavefront w moves the 1-KiB pieces w, w + 4, w + 8 of a unit; chunk cs of row r lands at cs ^ f(r),
    // f = (r >> 4) & 1 (conflict-free ds_read_b128 fragment reads); update mode: wavefront 3 also moves bias + s, 2 b + ||zh||
    int src_off[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int pch = (w + kPfW * j) * 64 + lane;
        const int r = pch / CPR, cs = pch - r * CPR;
        const int f = (r >> 4) & 1;
        src_off[j] = r * ROWB + (cs ^ f) * 16;
    }
    int n_w = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) n_w += (w + kPfW * j < KS) ? 1 : 0;
    n_w += (UPD && w == 3) ? 1 : 0;
    n_w += (UPD && w == 2) ? 1 : 0;
    const unsigned char *zall = reinterpret_cast<const unsigned char *>(a.P.Z);

    // ---- issue side of the unit stream: (bin ic, sweep isw, tile it, slice is), two units ahead of the consumer
    int ic = c0, it = 0, is = 0, isw = UPD ? 1 : 0, ibuf = 0;
    int irow0 = 0, int_ = 0;
    bool ivalid = false;
    int n_issued = 0, n_consumed = 0;
    auto issue_seek = [&]() __attribute__((always_inline)) {
        ivalid = false;
        while (ic < c1) {
            irow0 = a.P.pad_ptr[ic];
            int_ = a.P.nt != nullptr ? a.P.nt[ic] : (a.P.pad_ptr[ic + 1] - irow0) / kPfP;
            if (int_ > 0) { ivalid = true; break; }
            ++ic;
        }
    };
    auto issue = [&]() __attribute__((always_inline)) {
        unsigned char *dst_ = smem + ibuf * BUFB;
        const size_t row_ = (size_t)irow0 + (size_t)it * kPfP;
        const unsigned char *src_ = zall + row_ * ROWB + is * SLB;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (w + kPfW * j < KS)
                __builtin_amdgcn_global_load_lds(src_ + src_off[j],
                    (__attribute__((address_space(3))) void *)(dst_ + (w + kPfW * j) * 1024), 16, 0, 0);
        if (UPD && w == 3) {   // (the member columns ride with every slice of the tile: a fixed DMA count per unit)
            const float *p_ = (h ? a.P.cs : a.P.bias) + row_ + col;
            __builtin_amdgcn_global_load_lds(p_, (__attribute__((address_space(3))) void *)(dst_ + TILEB), 4, 0, 0);
        }
        if (UPD && w == 2) {
            const float *p_ = (h ? a.P.sn : a.P.cb) + row_ + col;
            __builtin_amdgcn_global_load_lds(p_, (__attribute__((address_space(3))) void *)(dst_ + TILEB + 256), 4, 0, 0);
        }
        if (++ibuf == NBUF) ibuf = 0;
        ++n_issued;
        if (++is == NS) {
            is = 0;
            if (++it == int_) {
                it = 0;
                if (!UPD && isw == 0) isw = 1;
                else { isw = UPD ? 1 : 0; ++ic; issue_seek(); }
            }
        }
    };
    issue_seek();
    if (ivalid) issue();
    if (ivalid) issue();

    // fragment addressing: chunk (2 sx + h) of row `col` of the unit, XOR-swizzled as the DMA laid it out
    const int fbase0 = col * SLB + ((h ^ ((col >> 4) & 1)) << 4);
    const unsigned ent0 = ((unsigned)col << 27) | (unsigned)(4 * h);

    int cbuf = 0;
    for (int c = c0; c < c1; ++c) {
        const int row0 = a.P.pad_ptr[c];
        const int ntile = a.P.nt != nullptr ? a.P.nt[c] : (a.P.pad_ptr[c + 1] - row0) / kPfP;
        const size_t slot = (size_t)c * a.Kcap + (qvalid ? qpos : a.pos_end - 1);
        // ---- per-(query, bin) bounds (as in shortlist_kernel)
        float4 bb = a.P.bb[c];                            // {rho_bin, snb, Bmax, bias residual}
        if (UPD) bb.y = sqrtf(bb.y) * (1.0f + 1e-6f);
        const float2 qn2 = a.qn[(size_t)sidx * a.B + c];  // N_jc {up, down}
        const float E = ((UPD ? 1.2e-7f * bb.z : 2.0f * bb.w) + a.gamma * (1.001f * bb.z + 2.0f * snq * bb.y)) *
                        (1.0f + 4.0f * kSlack);
        const float rgq = rg * (1.0f + 4.0f * kSlack);
        const float *tsn_c = UPD ? nullptr : a.P.tsn + (row0 >> 5);
        const float nj_hi = (qn2.x + E) * (1.0f + kSlack);
        const float nj_lo = (qn2.y - E) * (qn2.y > E ? (1.0f - kSlack) : (1.0f + kSlack));
        const float rsum = bb.x * (1.0f + kSlack);
        float lb[ML];
#pragma unroll
        for (int i = 0; i < ML; ++i) lb[i] = i < ML - m ? INFINITY : -INFINITY;
        float thr_s = -INFINITY;
        float thr2 = qvalid ? -FLT_MAX : INFINITY;
        if (UPD && qvalid) {
            float tau = INFINITY;
            if (a.tau_in != nullptr) tau = a.tau_in[slot];
            else if (a.seed.cnt[slot] >= m) tau = round_up_f32(a.seed.d[slot * m + m - 1] * a.S) * (1.0f + kSlack);
            if (tau < INFINITY) {
                const float hi = tau * (1.0f + 4.0f * kSlack) + rsum;
                thr2 = -0.5f * (hi * hi * (1.0f + 4.0f * kSlack) - nj_lo);
            }
        }
        if (h == 0) sCnt[32 * w + col] = 0;
        // (per-tile bests: only where the m nearest are unlikely to share a tile half, and never on a shell-ordered pack)
        const bool tile_best = ntile >= a.tile_best_min && 4 * ntile >= m * m && !(!UPD && a.skip != 0);
        const int kmax = (ML > 8 && !tile_best && ntile >= a.tile_best_min && !(!UPD && a.skip != 0)) ? a.tile_k2 : 0x7fffffff;
        int wcnt = 0;   // entries parked by this wavefront and not yet written out (wave-uniform)
        auto flush = [&]() __attribute__((always_inline)) {
            const int mb_ = a.bin_ptr[c];
            const int npark_ = wcnt < kPoolW ? wcnt : kPoolW;
            for (int i = lane; i < npark_; i += 64) {
                const unsigned en = sPool[w * kPoolW + i];
                const int qc = (int)(en >> 27), e = (int)(en & ((1u << 27) - 1u));
                const int off = atomicAdd(&sCnt[32 * w + qc], 1);
                if (off < a.cand_cap)
                    a.cand[((size_t)c * a.Kcap + pos0 + 32 * w + qc) * a.cand_cap + off] = a.memb_id[mb_ + e];
            }
        };

        for (int sweep = UPD ? 1 : 0; sweep < 2; ++sweep) {
            const float rgs = (sweep ? rg : -rg) * (1.0f + kSlack);
            if (!UPD && sweep == 1) {
                // end of sweep 0: m-th smallest t over BOTH lane halves -> tau -> thr2
                float mg[ML];
#pragma unroll
                for (int i = 0; i < ML; ++i) mg[i] = lb[i];
#pragma unroll
                for (int i = 0; i < ML; ++i) {
                    const float o = __shfl_xor(lb[i], 32, 64);
                    list_insert_desc<ML>(mg, i >= ML - m ? o : -INFINITY);
                }
                const float ms = mg[ML - 1];
                float tau = INFINITY;
                if (qvalid && ms > -INFINITY) {
                    const float thr = -2.0f * ms;
                    tau = sqrtf(fmaxf(thr + nj_hi, 0.f)) * (1.0f + 4.0f * kSlack) + rsum;
                    const float hi = tau * (1.0f + 4.0f * kSlack) + rsum;
                    thr2 = -0.5f * (hi * hi * (1.0f + 4.0f * kSlack) - nj_lo);
                }
                if (h == 0) sTau[32 * w + col] = tau;
            }
            // (everything loaded so far is waited for here, by a wait the compiler knows: see shortlist_kernel)
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
            for (int ct = 0; ct < ntile; ++ct) {
                f32x16 acc;
                float tsn_t = 0.f;
#pragma unroll
                for (int sl = 0; sl < NS; ++sl) {
                    wait_vmcnt(n_issued - n_consumed > 1 ? n_w : 0);   // my pieces of this unit have landed
                    __builtin_amdgcn_s_barrier();   // everybody's have; the buffer two units ahead is free again
                    if (ivalid) issue();
                    const unsigned tb = smem_base + (unsigned)(cbuf * BUFB);
                    if (sl == 0) {
                        if (UPD) {
                            const unsigned ma = tb + (unsigned)(TILEB + 16 * h);
                            f32x4 nv[4], nn[4], sv[4], bv[4];
#define CHB_SL_META(G)                                                                             \
                            nv[G] = lds_read_f4<32 * (G)>(ma);                                     \
                            nn[G] = lds_read_f4<384 + 32 * (G)>(ma);                               \
                            sv[G] = lds_read_f4<128 + 32 * (G)>(ma); bv[G] = lds_read_f4<256 + 32 * (G)>(ma);
                            CHB_SL_META(0) CHB_SL_META(1) CHB_SL_META(2) CHB_SL_META(3)
#undef CHB_SL_META
                            asm volatile("s_waitcnt lgkmcnt(0)"
                                         : "+v"(nv[0]), "+v"(nv[1]), "+v"(nv[2]), "+v"(nv[3]), "+v"(nn[0]), "+v"(nn[1]), "+v"(nn[2]),
                                           "+v"(nn[3])
                                         :
                                         : "memory");
                            asm volatile("" : "+v"(sv[0]), "+v"(sv[1]), "+v"(sv[2]), "+v"(sv[3]), "+v"(bv[0]), "+v"(bv[1]),
                                              "+v"(bv[2]), "+v"(bv[3]) : : "memory");
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
#pragma unroll
                                for (int k = 0; k < 4; ++k) {
                                    acc[4 * g + k] = fmaf(rgs, nn[g][k], -0.5f * nv[g][k]);
                                    if (fmaf(sv[g][k], qposf, bv[g][k]) < 0.f) acc[4 * g + k] = -INFINITY;
                                }
                            }
                        } else {
                            // the tile's largest member norm: a SCALAR load (complete behind the fragment reads' wait below)
                            const unsigned long long ta = reinterpret_cast<unsigned long long>(tsn_c + ct);
                            const unsigned ta_lo = __builtin_amdgcn_readfirstlane((unsigned)ta);
                            const unsigned ta_hi = __builtin_amdgcn_readfirstlane((unsigned)(ta >> 32));
                            const unsigned long long ta_s = ((unsigned long long)ta_hi << 32) | ta_lo;
                            asm volatile("s_load_dword %0, %1, 0x0" : "=s"(tsn_t) : "s"(ta_s) : "memory");
                            _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                        }
                    }
                    f16x8 af[KS];
                    {
                        const unsigned fa0 = tb + (unsigned)fbase0;
                        af[0] = lds_read_frag<0>(fa0);   af[1] = lds_read_frag<32>(fa0);  af[2] = lds_read_frag<64>(fa0);
                        af[3] = lds_read_frag<96>(fa0);  af[4] = lds_read_frag<128>(fa0); af[5] = lds_read_frag<160>(fa0);
                        af[6] = lds_read_frag<192>(fa0); af[7] = lds_read_frag<224>(fa0); af[8] = lds_read_frag<256>(fa0);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)"
                                 : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(af[4]), "+v"(af[5]),
                                   "+v"(af[6]), "+v"(af[7]), "+v"(af[8]), "+s"(tsn_t)
                                 :
                                 : "memory");
#pragma unroll
                    for (int sx = 0; sx < KS; ++sx)
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[sx], qreg[sl * KS + sx], acc, 0, 0, 0);
                    ++n_consumed;
                    if (++cbuf == NBUF) cbuf = 0;
                }
                // ---- the tile's selection code
                const float dlt = UPD ? 0.f : rgq * tsn_t;
                if (!UPD && sweep == 0) {
                    float mx = acc[0];
#pragma unroll
                    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, acc[r]);
                    if (tile_best) {
                        if (mx - dlt > thr_s) {
                            list_insert_desc<ML>(lb, mx - dlt);
                            thr_s = lb[ML - 1];
                        }
                    } else
                    for (int left = ML > 8 ? kmax : 1; left > 0 && mx - dlt > thr_s; left -= ML > 8 ? 1 : 0) {
                        list_insert_desc<ML>(lb, mx - dlt);
                        thr_s = lb[ML - 1];
                        float nx = -INFINITY;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            acc[r] = acc[r] == mx ? -INFINITY : acc[r];
                            nx = fmaxf(nx, acc[r]);
                        }
                        mx = nx;
                    }
                } else {
                    if ((ML > 8 || (UPD && ML > 1)) && wcnt >= kPoolW / 2 && wcnt <= kPoolW) {
                        flush();
                        wcnt = 0;
                    }
                    const unsigned ebase = ent0 + (unsigned)(ct * kPfP);
                    const float thr_t = thr2 - dlt;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const bool hit = acc[r] >= thr_t;
                        const unsigned long long bal = __ballot(hit);
                        if (bal) {
                            const int before = __builtin_amdgcn_mbcnt_hi(
                                (unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                            const int pos = wcnt + before;
                            if (hit && pos < kPoolW)
                                lds_write_u32(pool_base + 4u * (unsigned)pos, ebase + (unsigned)((r & 3) + 8 * (r >> 2)));
                            wcnt += __popcll(bal);
                        }
                    }
                }
            }
        }
        // ---- end of the bin: write the parked entries out
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (my asm pool writes, before the ordinary reads of the flush)
        flush();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const int ccount = (qvalid && h == 0) ? sCnt[32 * w + col] : 0;
        if (qvalid && h == 0) {
            a.cand_cnt[slot] = ccount < a.cand_cap ? ccount : a.cand_cap;
            if (!UPD && a.tau_out != nullptr) a.tau_out[slot] = sTau[32 * w + col];
            if (ccount > a.cand_cap || wcnt > kPoolW) {
                atomicAdd(a.overflow, 1);
                const int fi = c * nqt64 + (qpos - a.pos_begin) / kQTile;
                if (atomicExch(&flags64[fi], 1) == 0) a.flaglist[atomicAdd(a.nflag, 1)] = fi;
            }
        }
    }
}

// (four: the layout of the kFour builds -- no bias / norm column slots, the seat table in the segment bases' place)
static size_t shortlist_lds_bytes(int ks, int ml, bool four = false)
{
    if (four) return (size_t)3 * (kPfP * 32 * ks) + (size_t)kPfW * shortlist_pool_entries(ml) * 4 + 3 * kPfQ * 4 + 16 + 8 * kPfW;
    return (size_t)3 * (kPfP * 32 * ks + 512) + (size_t)kPfW * shortlist_pool_entries(ml) * 4 + 4 * kPfQ * 4 + 16 + 8 * kPfW;
}

template <int ML, bool UPD>
static void launch_sl(const ShortlistArgs &a, int *flags64, hipStream_t s)
{
    const int nq = a.pos_end - a.pos_begin;
    const int nqt = (nq + kPfQ - 1) / kPfQ;
    const int nqt64 = (nq + kQTile - 1) / kQTile;
    // bins per workgroup: long tile streams per workgroup, but enough workgroups for the 256 CUs x 4
    const long long units = (long long)nqt * a.B;
    // (tile skipping: the workgroups' run lengths differ by what they could skip -- one bin each, for an even finish:
    //  measured best at 8192 and at 16384 positions per batch)
    bool skip_build = false;
    if constexpr (!UPD) skip_build = a.skip != 0;
    int bpw = skip_build ? 1 : (int)std::max<long long>(1, units / ((UPD ? 1024 : 2048) * 4 / kPfW));
#ifdef CHB_DEV_KNOBS
    static int env_bpw = -2;
    if (env_bpw == -2) { const char *e = getenv("CHB_SL_BPW"); env_bpw = e ? atoi(e) : 0; }
    if (env_bpw > 0) bpw = env_bpw;
#endif
    bool pool_build = false;
    if constexpr (!UPD) pool_build = a.pool.Z != nullptr && a.qord != nullptr && a.ckey != nullptr;
    // (a pool-mode bin is ntile + a few tiles instead of 2 ntile: twice the bins per workgroup keep the streams as long)
    if (pool_build && !skip_build) bpw = (int)std::max<long long>(1, units / 1024);
#ifdef CHB_DEV_KNOBS
    if (env_bpw > 0) bpw = env_bpw;
#endif
    // (the tile-skipping builds are one bin per workgroup by construction -- their early run ends and, with the pools, the
    //  order of a bin's runs have only ever been exercised that way; the developer knob above does not reach them:
    //  experiment 29 of round 5 saw short shortlists from the skipping pool build with two and more bins per workgroup)
    if (skip_build) bpw = 1;
    bpw = std::min(bpw, a.B);
    const int nchunk = (a.B + bpw - 1) / bpw;
    const int total = nqt * nchunk;
    const int grid = ((total + 7) / 8) * 8;
    if (pool_build) {
        if constexpr (!UPD) {
            if (skip_build) {
                if (a.Dz == 144)
                    hipLaunchKernelGGL((shortlist_kernel<ML, false, 9, 0, true, true>), dim3(grid), dim3(64 * kPfW),
                                       shortlist_lds_bytes(9, ML), s, a, nqt, nchunk, bpw, flags64, nqt64, g_gate);
                else
                    hipLaunchKernelGGL((shortlist_kernel<ML, false, 10, 0, true, true>), dim3(grid), dim3(64 * kPfW),
                                       shortlist_lds_bytes(10, ML, ML <= 5), s, a, nqt, nchunk, bpw, flags64, nqt64, g_gate);
            } else if (a.Dz == 144)
                hipLaunchKernelGGL((shortlist_kernel<ML, false, 9, 0, false, true>), dim3(grid), dim3(64 * kPfW),
                                   shortlist_lds_bytes(9, ML), s, a, nqt, nchunk, bpw, flags64, nqt64, g_gate);
            else
                hipLaunchKernelGGL((shortlist_kernel<ML, false, 10, 0, false, true>), dim3(grid), dim3(64 * kPfW),
                                   shortlist_lds_bytes(10, ML, ML <= 5), s, a, nqt, nchunk, bpw, flags64, nqt64, g_gate);
        }
    } else if (skip_build) {
        if constexpr (!UPD) {
            if (a.Dz == 144)
                hipLaunchKernelGGL((shortlist_kernel<ML, false, 9, 0, true>), dim3(grid), dim3(64 * kPfW),
                                   shortlist_lds_bytes(9, ML), s, a, nqt, nchunk, bpw, flags64, nqt64, g_gate);
            else
                hipLaunchKernelGGL((shortlist_kernel<ML, false, 10, 0, true>), dim3(grid), dim3(64 * kPfW),
                                   shortlist_lds_bytes(10, ML, ML <= 5), s, a, nqt, nchunk, bpw, flags64, nqt64, g_gate);
        }
    } else if (a.Dz == 144)
        hipLaunchKernelGGL((shortlist_kernel<ML, UPD, 9>), dim3(grid), dim3(64 * kPfW), shortlist_lds_bytes(9, ML), s, a,
                           nqt, nchunk, bpw, flags64, nqt64, g_gate);
    else
        hipLaunchKernelGGL((shortlist_kernel<ML, UPD, 10>), dim3(grid), dim3(64 * kPfW), shortlist_lds_bytes(10, ML, !UPD && ML <= 5), s, a,
                           nqt, nchunk, bpw, flags64, nqt64, g_gate);
    if constexpr (!UPD) {
        // the segmented bins of this batch (usually none: the host only asks for these launches when the last batches'
        // bin sizes say a bin may qualify): phase 1 (m best accumulators per segment), phase 2 (shortlists)
        if (a.seg.gflag != nullptr && a.seg.launch) {
            const int gseg = ((nqt * a.seg.cap + 7) / 8) * 8;
            if (a.Dz == 144) {
                hipLaunchKernelGGL((shortlist_kernel<ML, false, 9, 1>), dim3(gseg), dim3(64 * kPfW), shortlist_lds_bytes(9, ML), s,
                                   a, nqt, 0, 1, flags64, nqt64, g_gate);
                hipLaunchKernelGGL((shortlist_kernel<ML, false, 9, 2>), dim3(gseg), dim3(64 * kPfW), shortlist_lds_bytes(9, ML), s,
                                   a, nqt, 0, 1, flags64, nqt64, g_gate);
            } else {
                hipLaunchKernelGGL((shortlist_kernel<ML, false, 10, 1>), dim3(gseg), dim3(64 * kPfW), shortlist_lds_bytes(10, ML, ML <= 5), s,
                                   a, nqt, 0, 1, flags64, nqt64, g_gate);
                hipLaunchKernelGGL((shortlist_kernel<ML, false, 10, 2>), dim3(gseg), dim3(64 * kPfW), shortlist_lds_bytes(10, ML), s,
                                   a, nqt, 0, 1, flags64, nqt64, g_gate);
            }
        }
    }
}

// the wide-row builds (Dz = 144 NS, NS = 2 .. 4): plain two-sweep base launch / one-sweep update launch
template <int ML, bool UPD>
static void launch_sl_wide(const ShortlistArgs &a, int *flags64, hipStream_t s)
{
    const int nq = a.pos_end - a.pos_begin;
    const int nqt = (nq + kPfQ - 1) / kPfQ;
    const int nqt64 = (nq + kQTile - 1) / kQTile;
    const long long units = (long long)nqt * a.B;
    int bpw = (int)std::max<long long>(1, units / (UPD ? 1024 : 2048));
    bpw = std::min(bpw, a.B);
    const int nchunk = (a.B + bpw - 1) / bpw;
    const int total = nqt * nchunk;
    const int grid = ((total + 7) / 8) * 8;
    const size_t lds = shortlist_lds_bytes(9, ML);
    switch (a.Dz / 144) {
    case 2:
        hipLaunchKernelGGL((shortlist_wide_kernel<ML, UPD, 2>), dim3(grid), dim3(64 * kPfW), lds, s, a, nqt, nchunk, bpw, flags64, nqt64, g_gate);
        break;
    case 3:
        hipLaunchKernelGGL((shortlist_wide_kernel<ML, UPD, 3>), dim3(grid), dim3(64 * kPfW), lds, s, a, nqt, nchunk, bpw, flags64, nqt64, g_gate);
        break;
    default:
        hipLaunchKernelGGL((shortlist_wide_kernel<ML, UPD, 4>), dim3(grid), dim3(64 * kPfW), lds, s, a, nqt, nchunk, bpw, flags64, nqt64, g_gate);
        break;
    }
}

template <int ML>
static void launch_sl_work(const ShortlistArgs &a, int *flags64, int grid, hipStream_t s)
{
    const int nq = a.pos_end - a.pos_begin;
    const int nqt = (nq + kPfQ - 1) / kPfQ, nqt64 = (nq + kQTile - 1) / kQTile;
    if (a.Dz == 144)
        hipLaunchKernelGGL((shortlist_kernel<ML, false, 9, 0, false, false, true>), dim3(grid), dim3(64 * kPfW),
                           shortlist_lds_bytes(9, ML), s, a, nqt, a.B, 1, flags64, nqt64, g_gate);
    else
        hipLaunchKernelGGL((shortlist_kernel<ML, false, 10, 0, false, false, true>), dim3(grid), dim3(64 * kPfW),
                           shortlist_lds_bytes(10, ML, ML <= 5), s, a, nqt, a.B, 1, flags64, nqt64, g_gate);
}

}  // namespace

int shortlist_list_len(int m) { return m <= 5 ? 5 : (m <= 8 ? 8 : 16); }

// 144 or 160 columns (the narrow builds), else 2 .. 4 slices of 144 (the wide builds, D <= 573), else 0 (no shortlist stage)
int shadow_row_elems(int D)
{
    if (D + kBiasCols <= 144) return 144;
    if (D + kBiasCols <= 160) return 160;
    const int ns = (D + kBiasCols + 143) / 144;
    return ns <= kWideMaxSlices ? 144 * ns : 0;
}

void launch_global_center(const double *X, int N, int D, int Dp, double *part, int part_blocks, double *mu_g,
                          unsigned int *rmax, hipStream_t s)
{
    if (N <= 0) return;
    const int rows_per = (N + part_blocks - 1) / part_blocks;
    const int nblk = (N + rows_per - 1) / rows_per;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk), dim3(256), 0, s, X, N, Dp, rows_per, part);
    hipLaunchKernelGGL(colsum_final_kernel, dim3(1), dim3(256), 0, s, part, nblk, N, Dp, mu_g);
    (void)hipMemsetAsync(rmax, 0, sizeof(unsigned int), s);
    hipLaunchKernelGGL(absmax_kernel, dim3(1024), dim3(256), 0, s, X, N, Dp, mu_g, rmax);
}

void launch_global_shadow(const double *X, int N, int D, int Dp, const double *mu_g, double S,
                          unsigned short *Gs, int Dz, void *gq, hipStream_t s)
{
    if (N > 0)
        hipLaunchKernelGGL(global_shadow_kernel, dim3((N + 3) / 4), dim3(256), 0, s, X, N, D, Dp, mu_g, S, Gs, Dz,
                           reinterpret_cast<float2 *>(gq));
}

void launch_shell_scale(const void *ms, const int *memb_id, const int *bin_ptr, int B, int nsh, float *shell_inv,
                        hipStream_t s)
{
    if (B > 0)
        hipLaunchKernelGGL(shell_scale_kernel, dim3(B), dim3(256), 0, s, reinterpret_cast<const float4 *>(ms), memb_id,
                           bin_ptr, nsh, shell_inv);
}

void launch_bin_centers(const double *X, int D, int Dp, const int *memb_id, const int *bin_ptr, int B,
                        double *centers, hipStream_t s)
{
    if (B > 0) hipLaunchKernelGGL(bin_center_kernel, dim3(B), dim3(256), 0, s, X, D, Dp, memb_id, bin_ptr, centers);
}

void launch_sample_shadow(const double *X, int D, int Dp, const int *ids, int n, int *labels,
                          int B, const double *centers, const double *mu_g, double S, unsigned short *Zs,
                          int Dz, void *ms, const int *new_lab, int *inb, hipStream_t s)
{
    if (n > 0)
        hipLaunchKernelGGL(sample_shadow_kernel, dim3((n + 3) / 4), dim3(256), 0, s, X, D, Dp, ids, n, labels,
                           B, centers, mu_g, S, Zs, Dz, reinterpret_cast<float4 *>(ms), new_lab, inb, g_gate);
}

void launch_pack_centered(const double *X, int D, int Dp, const int *memb_id, const int *memb_code,
                          const int *bin_ptr, int B, int rows_hint, const double *centers, const double *mu_g,
                          double S, int Dz, const MemberPack &P, hipStream_t s)
{
    if (B <= 0) return;
    int gy = ((rows_hint + 31 * B) / std::max(B, 1) + 15) / 16;   // ~4 entries per wavefront per bin
    gy = std::max(1, std::min(gy, 64));
    hipLaunchKernelGGL(pack_centered_kernel, dim3(B, gy), dim3(256), 0, s, X, D, Dp, memb_id, memb_code, bin_ptr,
                       P.pad_ptr, centers, mu_g, S, Dz, P, g_gate);   // (bounds: into P.bb, zeroed by the batch-CSR kernel)
}

void launch_query_order(const unsigned long long *ckey, const int *bq, int pos_begin, int pos_end, int B, int *qord, int *home,
                        hipStream_t s)
{
    if (pos_end > pos_begin)
        hipLaunchKernelGGL(query_order_kernel, dim3(1), dim3(1024), sizeof(int) * ((size_t)B + 1 + 1024), s, ckey, bq,
                           pos_begin, pos_end - pos_begin, B, qord, home, g_gate);
}

void launch_query_order_sweep(const unsigned long long *ckey, const int *perm, const void *geo, int nbatch, int B, int *qord_all,
                              int *home_all, hipStream_t s)
{
    if (nbatch > 0)
        hipLaunchKernelGGL(query_order_sweep_kernel, dim3(nbatch), dim3(1024), sizeof(int) * ((size_t)B + 1 + 1024), s, ckey, perm,
                           reinterpret_cast<const int4 *>(geo), B, qord_all, home_all);
}

void launch_query_norms(const double *X, int D, int Dp, int N, int B, const double *centers, double S, void *qn,
                        unsigned long long *ckey, hipStream_t s)
{
    if (N <= 0 || B <= 0) return;
    if (ckey != nullptr) (void)hipMemsetAsync(ckey, 0xFF, sizeof(unsigned long long) * (size_t)N, s);
    const int nqx = (N + 31) / 32, nqy = (B + 63) / 64;
    const QnArgs q{X, D, Dp, N, B, centers, S, reinterpret_cast<float2 *>(qn), ckey};
    hipLaunchKernelGGL(query_norms_kernel, dim3((unsigned)nqx * (unsigned)nqy), dim3(256), 0, s, q, nqx);
}

void launch_pack_build(const unsigned short *Zs, const void *ms, int D, int Dz, const int *memb_id, const int *bin_ptr,
                       int B, int rows_hint, const MemberPack &P, bool shells, hipStream_t s)
{
    if (B <= 0) return;
    const long long rows = (long long)rows_hint + 32LL * B;
    const int npack = (int)std::max<long long>(1, std::min<long long>((rows + 15) / 16, 16384));
    hipLaunchKernelGGL(pack_build_kernel, dim3(npack + B), dim3(256), 0, s, Zs, reinterpret_cast<const float4 *>(ms), D, Dz,
                       memb_id, bin_ptr, P.pad_ptr, B, P, npack, shells, g_gate);
}

void launch_pack_state_build(const PackState &ps, const MemberPack &P, const unsigned short *Zs, const void *ms, int D, int Dz,
                             const int *memb_id, const int *bin_ptr, int B, int N, int grow, hipStream_t s)
{
    if (B <= 0) return;
    launch_fill_i32(ps.row, -1, N, s);
    hipLaunchKernelGGL(pack_state_layout_kernel, dim3(1), dim3(256), 0, s, ps, bin_ptr, B, grow);
    const long long rows = 2LL * N + (long long)B * std::max(grow, 64) + 32LL * B;
    const int ngather = (int)std::max<long long>(1, std::min<long long>((rows + 15) / 16, 16384));
    hipLaunchKernelGGL(pack_state_build_kernel, dim3(ngather + B), dim3(256), 0, s, ps, P, Zs,
                       reinterpret_cast<const float4 *>(ms), D, Dz, memb_id, bin_ptr, B, ngather);
}

void launch_pack_state_start(const PackState &ps, const MemberPack &P, int D, int Dz, const int *labels, int *inb,
                             const int *open_bq, int open_K, int *open_lab_old, int B, const SegPlan *seg, int *stats,
                             int *zero_me, hipStream_t s)
{
    if (B <= 0) return;
    const int nopen = (open_K + 255) / 256;
    hipLaunchKernelGGL(pack_state_start_kernel, dim3(1 + nopen), dim3(256), (size_t)B * sizeof(int), s, ps, P, D, Dz, labels,
                       inb, open_bq, open_K, open_lab_old, B, seg ? *seg : SegPlan{}, stats, zero_me, g_gate);
}

void launch_pack_state_commit(const PackState &ps, const MemberPack &P, const double *X, int D, int Dp, const int *ids, int n,
                              int *labels, int B, const double *centers, const double *mu_g, double S, unsigned short *Zs,
                              int Dz, void *ms, const int *new_lab, const int *lab_old, int *inb, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(pack_state_slots_kernel, dim3((n + 255) / 256), dim3(256), (size_t)B * sizeof(int), s, ps, ids, n, B,
                       new_lab, lab_old, ps.dest, g_gate);
    hipLaunchKernelGGL(pack_state_commit_kernel, dim3((n + 3) / 4), dim3(256), 0, s, ps, P, X, D, Dp, ids, n, labels, B,
                       centers, mu_g, S, Zs, Dz, reinterpret_cast<float4 *>(ms), new_lab, ps.dest, inb, g_gate);
    hipLaunchKernelGGL(pack_state_fix_kernel, dim3(B), dim3(256), 0, s, ps, P, Zs, reinterpret_cast<const float4 *>(ms), D, Dz,
                       ids, new_lab, g_gate);
}

static int pool_home_blocks(int B, int *hb)
{
    int hy = std::max(1, std::min(2048 / std::max(B, 1), (B + 3) / 4));
    *hb = (B + hy - 1) / hy;
    return (B + *hb - 1) / *hb;
}

void launch_pool_build(const PoolState &ps, const unsigned short *Zs, const void *ms, const void *qn, int D, int Dz,
                       const int *memb_id, const int *bin_ptr, int B, hipStream_t s)
{
    if (B <= 0) return;
    int hb = 1;
    const int hy = pool_home_blocks(B, &hb);
    hipLaunchKernelGGL((pool_update_kernel<true>), dim3(B, hy), dim3(256), 0, s, ps, Zs, reinterpret_cast<const float4 *>(ms),
                       reinterpret_cast<const float2 *>(qn), D, Dz, B, hb, memb_id, bin_ptr, nullptr, 0, nullptr, nullptr, nullptr, 0,
                       g_gate);
}

void launch_pool_open(const PoolState &ps, const int *inb, int D, int Dz, int B, int *zero_me, hipStream_t s)
{
    const long long nslot = (long long)B * B * kPoolRows;
    if (nslot > 0)
        hipLaunchKernelGGL(pool_open_kernel, dim3((unsigned)((nslot + 255) / 256)), dim3(256), 0, s, ps, inb, D, Dz, nslot, zero_me,
                           g_gate);
}

void launch_pool_commit(const PoolState &ps, const unsigned short *Zs, const void *ms, const void *qn, int D, int Dz,
                        const int *ids, int n, const int *new_lab, const int *lab_old, const int *labels, int B, bool holes,
                        hipStream_t s)
{
    if (B <= 0 || n <= 0) return;
    int hb = 1;
    const int hy = pool_home_blocks(B, &hb);
    hipLaunchKernelGGL((pool_update_kernel<false>), dim3(B, hy), dim3(256), 0, s, ps, Zs, reinterpret_cast<const float4 *>(ms),
                       reinterpret_cast<const float2 *>(qn), D, Dz, B, hb, nullptr, nullptr, ids, n, new_lab, lab_old, labels,
                       holes ? 1 : 0, g_gate);
}

void launch_shortlist_worklist(const ShortlistArgs &a_, int *flags64, hipStream_t s)
{
    ShortlistArgs a = a_;
    a.gamma = kGamma; a.tile_best_min = 16; a.tile_k2 = 3;
    // (a.skip stays as it is: the plain build skips nothing, but the flag also says that the pack is in SHELL order, where the
    //  per-tile-best shortcut of the threshold sweep must not be taken -- the m nearest crowd into a few tiles)
    a.pool = PoolState{}; a.qord = nullptr; a.home = nullptr; a.skip_stat = nullptr; a.pool_stat = nullptr;
    a.seg = SegPlan{};   // (a segmented bin's pairs were never on the pool launch's list)
    const int nq = a.pos_end - a.pos_begin;
    if (nq <= 0 || a.B <= 0 || a.update || a.worklist == nullptr || a.Dz > 160) return;   // (wide rows: no pools, no second chance)
    const long long items = (long long)((nq + kQTile - 1) / kQTile) * a.B;
    const int grid = (int)std::min<long long>(items, 2048);
    if (a.m <= 5) launch_sl_work<5>(a, flags64, grid, s);
    else if (a.m <= 8) launch_sl_work<8>(a, flags64, grid, s);
    else launch_sl_work<16>(a, flags64, grid, s);
}

void launch_shortlist(const ShortlistArgs &a_, int *flags64, hipStream_t s)
{
    ShortlistArgs a = a_;
    // The accumulation-error factor is part of the proof that the shortlist contains the exact
    // top-m: the product library takes it from the constant only.
    a.gamma = kGamma; a.tile_best_min = 16; a.tile_k2 = 3;
#ifdef CHB_DEV_KNOBS   // developer builds (tools/): never below kGamma
    {
        static float g = -1.f; static int tb = -1;
        if (g < 0.f) { const char *e = getenv("CHB_SL_GAMMA"); g = e ? std::max((float)atof(e), kGamma) : kGamma; }
        if (tb < 0) { const char *e = getenv("CHB_SL_TILEBEST"); tb = e ? atoi(e) : 16; }
        a.gamma = g; a.tile_best_min = tb;
        { static int tk = -1; if (tk < 0) { const char *e = getenv("CHB_SL_TILEK"); tk = e ? std::max(1, atoi(e)) : 3; } a.tile_k2 = tk; }
    }
#endif
    const int nq = a.pos_end - a.pos_begin;
    if (nq <= 0 || a.B <= 0) return;
    if (a.Dz > 160) {
        // wide rows: NS slices of 144 columns, up to 144 NS + 1 fp32 terms per accumulator
        a.gamma *= (float)(a.Dz / 144);
        if (a.update) {
            if (a.m <= 8) launch_sl_wide<1, true>(a, flags64, s);
            else launch_sl_wide<2, true>(a, flags64, s);
        } else if (a.m <= 5) launch_sl_wide<5, false>(a, flags64, s);
        else if (a.m <= 8) launch_sl_wide<8, false>(a, flags64, s);
        else launch_sl_wide<16, false>(a, flags64, s);
        return;
    }
    if (a.update) {
        if (a.m <= 8) launch_sl<1, true>(a, flags64, s);
        else launch_sl<2, true>(a, flags64, s);   // (ML is unused in update mode: 2 marks the m > 8 build)
    } else if (a.m <= 5) {
        launch_sl<5, false>(a, flags64, s);
    } else if (a.m <= 8) {
        launch_sl<8, false>(a, flags64, s);
    } else {
        launch_sl<16, false>(a, flags64, s);
    }
}

}  // namespace chb
