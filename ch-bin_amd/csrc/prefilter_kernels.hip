// Stage 1 of the two-stage EXACT nearest-member selection: a bf16 matrix-core shortlist.
//
// The reference selects the m nearest members of every bin from an exact distance row
// (distance_matrix.py:47-62 over distance_matrix.py:33-44).  Computing every one of the N^2
// distances in unfused fp64 is what bounds the sweep, yet only ~m of the ~N/B members of a bin can
// ever be selected.  This file produces, for each (batch position j, bin c), a SHORTLIST that is
// guaranteed to contain the exact top-m; topm_kernels.hip:rescore_kernel then evaluates the exact
// cdist-rounded distance on the shortlist only.  The final result is bit-identical to the
// brute-force path (and is checked against it in tests/).
//
// Guarantee.  For the members of bin c and the queries examined against bin c let z = x - mu_c
// (mu_c = any fixed centre; the mean of the bin's current members is used, so member vectors are
// small and the bf16 rounding error is RELATIVE to the within-bin spread, whatever the absolute
// scale of the features -- coverage columns of magnitude 0.2 next to k-mer frequencies of 0.007
// included), zh = bf16(z) and rho = ||zh - z||_2, measured exactly when the shadow row is built
// (members: once per batch in pack_centered_kernel; queries: in the kernel prologue).  By the
// triangle inequality
//     | ||x_j - x_p|| - ||zh_j - zh_p|| |  <=  rho_j + rho_p .
// ||zh_j - zh_p||^2 = n_j + n_p - 2 <zh_j, zh_p> with n = ||zh||^2 exact and the dot product from
// v_mfma_f32_32x32x16_bf16 (bf16 products are exact in fp32; accumulation error <= g (n_j + n_p)
// with g = 1e-4, several times the worst-case fp32 summation bound for D <= 512).  Hence for every
// member p:  LB(j,p) <= d(j,p) <= UB(j,p).  The kernel keeps tau = (an upper bound of) the m-th
// smallest UB seen so far in the bin -- at least m members are provably within tau -- and
// shortlists every member with LB <= tau.  Any member of the true top-m has d <= tau, hence
// LB <= tau: it is on the list.  A relative slack of 1e-6 covers fp32 rounding of the bound
// arithmetic itself and the fp64 rounding of the exact distances.
// If a shortlist overflows its capacity the (query tile, bin) is flagged and recomputed by the
// brute-force tile kernel, so correctness never depends on the data.
#include "chb_internal.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>

#include <algorithm>

#ifndef CHB_PF_DEEP
#define CHB_PF_DEEP 0
#endif
#ifndef CHB_PF_WAVES
#define CHB_PF_WAVES 3
#endif

namespace chb {
namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr float kGamma = 1e-4f;
constexpr float kSlack = 1e-6f;
constexpr int kPfQ = 128;  // batch positions per workgroup (32 per wavefront)
constexpr int kPfP = 32;   // members per tile

__device__ __forceinline__ float round_up_f32(double v)
{
    float f = (float)v;
    if ((double)f < v) f = nextafterf(f, INFINITY);
    return f;
}

__device__ __forceinline__ unsigned short bf16_rn(float zf)
{
    unsigned int u = __float_as_uint(zf);
    u = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;   // round-to-nearest-even
    return (unsigned short)u;
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

// Shadow rows of the CSR-ordered members, relative to their bin's centre, stored contiguously in
// CSR order (the shortlist kernel streams plain sequential memory).  One wavefront per member.
__global__ __launch_bounds__(256) void pack_centered_kernel(const double *X, int D, int Dp,
                                                            const int *memb_id, const int *bin_ptr,
                                                            const double *centers, unsigned short *Zp,
                                                            int Dz, float *nrm_p, float *rho_p)
{
    const int c = blockIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const double *mu = centers + (size_t)c * Dp;
    for (int e = bin_ptr[c] + blockIdx.y * 4 + w; e < bin_ptr[c + 1]; e += gridDim.y * 4) {
        const double *x = X + (size_t)memb_id[e] * Dp;
        double n2 = 0.0, e2 = 0.0;
        for (int k = lane; k < Dz; k += 64) {
            unsigned short hb = 0;
            if (k < D) {
                const double z = x[k] - mu[k];
                hb = bf16_rn((float)z);
                const double zh = (double)__uint_as_float(((unsigned int)hb) << 16);
                n2 += zh * zh;
                e2 += (zh - z) * (zh - z);
            }
            Zp[(size_t)e * Dz + k] = hb;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            n2 += __shfl_xor(n2, off, 64);
            e2 += __shfl_xor(e2, off, 64);
        }
        if (lane == 0) {
            nrm_p[e] = round_up_f32(n2 * (1.0 + 1e-12));
            rho_p[e] = round_up_f32(sqrt(e2) * (1.0 + 1e-9) + 1e-300);
        }
    }
}

// Per-sample shadow rows relative to the centre of the sample's CURRENT bin (labels[p] >= 0).
// ids == nullptr: all samples 0..n-1; otherwise the n listed samples (the batch just committed).
// One wavefront per sample.
__global__ __launch_bounds__(256) void sample_shadow_kernel(const double *X, int D, int Dp, const int *ids,
                                                            int n, const int *labels, int B,
                                                            const double *centers, unsigned short *Zs,
                                                            int Dz, float *nrm_s, float *rho_s)
{
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int p = ids ? ids[i] : i;
    const int c = labels[p];
    if (c < 0 || c >= B) return;
    const double *mu = centers + (size_t)c * Dp;
    const double *x = X + (size_t)p * Dp;
    double n2 = 0.0, e2 = 0.0;
    for (int k = lane; k < Dz; k += 64) {
        unsigned short hb = 0;
        if (k < D) {
            const double z = x[k] - mu[k];
            hb = bf16_rn((float)z);
            const double zh = (double)__uint_as_float(((unsigned int)hb) << 16);
            n2 += zh * zh;
            e2 += (zh - z) * (zh - z);
        }
        Zs[(size_t)p * Dz + k] = hb;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        n2 += __shfl_xor(n2, off, 64);
        e2 += __shfl_xor(e2, off, 64);
    }
    if (lane == 0) {
        nrm_s[p] = round_up_f32(n2 * (1.0 + 1e-12));
        rho_s[p] = round_up_f32(sqrt(e2) * (1.0 + 1e-9) + 1e-300);
    }
}

// Gathers the per-sample shadow rows of the CSR-ordered members into contiguous storage.
__global__ __launch_bounds__(256) void pack_rows_kernel(const unsigned short *Zs, const float *nrm_s,
                                                        const float *rho_s, int Dz, const int *memb_id,
                                                        const int *bin_ptr, int B, unsigned short *Zp,
                                                        float *nrm_p, float *rho_p)
{
    const int total = bin_ptr[B];
    const int cpr = Dz >> 3;
    const long long nch = (long long)total * cpr;
    for (long long ch = (long long)blockIdx.x * blockDim.x + threadIdx.x; ch < nch;
         ch += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(ch / cpr), cc = (int)(ch - (long long)e * cpr);
        const int id = memb_id[e];
        *reinterpret_cast<uint4 *>(Zp + (size_t)e * Dz + cc * 8) =
            *reinterpret_cast<const uint4 *>(Zs + (size_t)id * Dz + cc * 8);
        if (cc == 0) { nrm_p[e] = nrm_s[id]; rho_p[e] = rho_s[id]; }
    }
}

// Query-side shadow rows, one per (batch position, bin): bf16(x_j - mu_c) plus the four scalars
// the bounds need {||zh||^2, rho, ||z||^2 rounded up, ||z||^2 rounded down}.  One wavefront per
// row, lanes along the features (coalesced); computed once per batch and shared by the base and
// the update shortlist launches.
__global__ __launch_bounds__(256) void query_shadow_kernel(const double *X, int D, int Dp, const int *bq,
                                                           int pos_begin, int pos_end, int B, int Kcap,
                                                           const double *centers, unsigned short *Zq,
                                                           int Dz, float4 *qs)
{
    // 16 lanes per row (4 rows per wavefront); a lane owns feature pairs (2 l + 32 t, 2 l + 32 t + 1)
    const int l16 = threadIdx.x & 15;
    const long long row = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const long long nrow = (long long)(pos_end - pos_begin) * B;
    const bool rvalid = row < nrow;
    const long long rr = rvalid ? row : 0;
    const int pos = pos_begin + (int)(rr / B), c = (int)(rr - (rr / B) * B);
    const double *x = X + (size_t)bq[pos] * Dp;
    const double *mu = centers + (size_t)c * Dp;
    const size_t slot = (size_t)c * Kcap + pos;
    double n2 = 0.0, e2 = 0.0, x2 = 0.0;
    for (int k = 2 * l16; k < Dz; k += 32) {
        unsigned int packed = 0u;
        if (k < Dp) {   // Dp is even and rows are zero padded to Dp: k + 1 < Dp as well
            const double2 xv = *reinterpret_cast<const double2 *>(x + k);
            const double2 mv = *reinterpret_cast<const double2 *>(mu + k);
            const double z0 = xv.x - mv.x, z1 = xv.y - mv.y;   // padding columns give exactly 0
            const unsigned short h0 = bf16_rn((float)z0), h1 = bf16_rn((float)z1);
            const double zh0 = (double)__uint_as_float(((unsigned int)h0) << 16);
            const double zh1 = (double)__uint_as_float(((unsigned int)h1) << 16);
            n2 += zh0 * zh0 + zh1 * zh1;
            e2 += (zh0 - z0) * (zh0 - z0) + (zh1 - z1) * (zh1 - z1);
            x2 += z0 * z0 + z1 * z1;
            packed = (unsigned int)h0 | ((unsigned int)h1 << 16);
        }
        if (rvalid) *reinterpret_cast<unsigned int *>(Zq + slot * Dz + k) = packed;
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) {
        n2 += __shfl_xor(n2, off, 16);
        e2 += __shfl_xor(e2, off, 16);
        x2 += __shfl_xor(x2, off, 16);
    }
    if (rvalid && l16 == 0) {
        float4 o;
        o.x = round_up_f32(n2 * (1.0 + 1e-12));
        o.y = round_up_f32(sqrt(e2) * (1.0 + 1e-9) + 1e-300);
        o.z = round_up_f32(x2 * (1.0 + 1e-12));
        o.w = (float)(x2 * (1.0 - 1e-6));
        qs[slot] = o;
    }
}

// rho_bound[c] = largest rounding distance, sn_bound[c] = largest ||zh|| among the (packed) members
// of bin c; one block per bin
__global__ __launch_bounds__(256) void bin_bounds_kernel(const float *rho_p, const float *nrm_p,
                                                         const int *bin_ptr, float *rho_out, float *sn_out)
{
    __shared__ float red[256], red2[256];
    const int c = blockIdx.x;
    float v = 0.f, u = 0.f;
    for (int e = bin_ptr[c] + threadIdx.x; e < bin_ptr[c + 1]; e += 256) {
        v = fmaxf(v, rho_p[e]);
        u = fmaxf(u, nrm_p[e]);
    }
    red[threadIdx.x] = v; red2[threadIdx.x] = u;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) {
            red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + off]);
            red2[threadIdx.x] = fmaxf(red2[threadIdx.x], red2[threadIdx.x + off]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        rho_out[c] = red[0];
        sn_out[c] = sqrtf(red2[0]) * (1.0f + 1e-6f);
    }
}

template <int ML>
__device__ __forceinline__ void list_insert(float (&l)[ML], float v)
{
#pragma unroll
    for (int i = 0; i < ML; ++i) {
        const float lo = fminf(l[i], v);
        v = fmaxf(l[i], v);
        l[i] = lo;
    }
}

// UPD = false: base members of the bin, tau learned on the fly (top-m of the upper bounds).
// UPD = true : the batch's own members (eligibility code per member, see aux_kernels.hip) against a
//              FIXED tau = the exact m-th distance of the already known list `seed`; members that
//              cannot displace a list entry are dropped without ever touching fp64.
// The query fragments (B operand) are built in the prologue (x_j - mu_c -> bf16, with the exact
// ||zh||^2 and rho of THIS query against THIS bin's centre) and live in registers for the whole
// kernel (Dz <= 160); LDS only holds the double-buffered member tile.
// This is the GENERIC form (any Dz <= 160, register-staged tiles, run-time step count); the two row
// shapes of real feature tables take shortlist_kernel below.
template <int ML, bool UPD>
__global__ __launch_bounds__(256, CHB_PF_WAVES) void prefilter_kernel(PrefilterArgs a, int nqt, int total,
                                                        int stride, int *flags64, int nqt64)
{
    constexpr int KSMAX = 10;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *sPz = smem;                               // [2][kPfP][stride]
    float *sPn = reinterpret_cast<float *>(sPz + (size_t)2 * kPfP * stride);  // [2][kPfP]
    float *sPr = sPn + 2 * kPfP;                             // [2][kPfP]
    int *sPid = reinterpret_cast<int *>(sPr + 2 * kPfP);     // [4][kPfP]: slot = tile & 3 (read one tile late)
    int *sPcode = sPid + 4 * kPfP;                           // [2][kPfP]

    const int per = (total + 7) >> 3;
    const int W = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (W >= total) return;
    const int c = W / nqt, qt = W - c * nqt;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    const int mb = a.bin_ptr[c];
    const int nmem = a.bin_ptr[c + 1] - mb;
    const int pos0 = a.pos_begin + qt * kPfQ;
    const int Dz = a.shm.Dz;
    const int cpr = Dz >> 3;            // 16-byte chunks per row
    const int ksteps = Dz >> 4;
    const int m = a.m;

    // my query
    const int qpos = pos0 + 32 * w + col;
    const bool qvalid = qpos < a.pos_end;
    // Query fragment: lane (col, h) owns B[k = 16 s + 8 h + j][col] of the precomputed shadow row
    // bf16(x_j - mu_c) of (this position, this bin)  (query_shadow_kernel).
    bf16x8 qreg[KSMAX];
    float nj, rq, njx_up, njx_dn;
    {
        const size_t qslot = (size_t)c * a.Kcap + (qvalid ? qpos : a.pos_end - 1);
        const unsigned short *zq = a.Zq + qslot * Dz + h * 8;
#pragma unroll
        for (int sx = 0; sx < KSMAX; ++sx) {
            if (sx < ksteps) qreg[sx] = *reinterpret_cast<const bf16x8 *>(zq + sx * 16);
            else qreg[sx] = qreg[0];
        }
        const float4 q4 = a.qs[qslot];
        nj = q4.x; rq = q4.y; njx_up = q4.z; njx_dn = q4.w;
    }
    // Bounds.  With z_j EXACT on the query side:
    //   ||z_j - zh_p||^2 = ||z_j||^2 + n_p - 2 <z_j, zh_p>,   <z_j, zh_p> = <zh_j, zh_p> + <z_j - zh_j, zh_p>,
    //   |<z_j - zh_j, zh_p>| <= rho_j sqrt(n_p) <= rho_j * snb   (snb = largest ||zh_p|| in the bin),
    // so the query's rounding error is damped by the (small) norm of the bin-centred member instead
    // of entering the distance at full size -- what keeps far bins, whose members are all almost
    // equally far, from flooding the shortlist.  The matrix-core accumulation error is bounded by
    // g (n_j + n_p) as before; the member's own rounding enters as +-rho_p <= rho_bin on d.
    const float Aq = (2.0f * rq * a.sn_bound[c] + kGamma * nj) * (1.0f + 4.0f * kSlack);
    const float nj_hi = (njx_up + Aq) * (1.0f + kSlack);
    const float nj_lo = (njx_dn - Aq) * (njx_dn > Aq ? (1.0f - kSlack) : (1.0f + kSlack));

    float ub[ML];
#pragma unroll
    for (int i = 0; i < ML; ++i) ub[i] = INFINITY;
    float tau = INFINITY;   // m-th smallest UB over both lane halves of this query
    int ccount = 0;
    if (UPD && qvalid) {
        const size_t sl = (size_t)c * a.Kcap + qpos;
        if (a.seed.cnt[sl] >= m) {
            const double e = a.seed.d[sl * m + m - 1];   // exact m-th distance so far
            tau = round_up_f32(e) * (1.0f + kSlack);
        }
    }

    const int ntile = (nmem + kPfP - 1) / kPfP;
    const int nchunk = kPfP * cpr;      // chunks per member tile (<= 4 per thread for Dz <= 256)
    // staging registers: up to 4 16-byte chunks per thread (Dz <= 256).  Two sets, so that the tile
    // after next is already in flight while the current one is being consumed (one global-memory
    // latency per tile would otherwise be exposed: a tile is only ~0.4 us of work per wavefront).
    uint4 cA0 = {0, 0, 0, 0}, cA1 = cA0, cA2 = cA0, cA3 = cA0;
    float nA = INFINITY;
    int idA = -1, codeA = 0;
#if CHB_PF_DEEP
    uint4 cB0 = cA0, cB1 = cA0, cB2 = cA0, cB3 = cA0;
    float nB = INFINITY;
    int idB = -1, codeB = 0;
#endif
    // (named scalars + macros: a struct passed by reference to a lambda ends up in scratch)
#define CHB_PF_FETCH_ONE(I, ST, TT)                                                               \
    {                                                                                              \
        const int ch = tid + 256 * (I);                                                            \
        if (ch < nchunk) {                                                                         \
            const int r = ch / cpr, cc = ch - r * cpr;                                             \
            const int e = (TT) * kPfP + r;                                                         \
            const int id = e < nmem ? mb + e : mb;                                                 \
            ST = *reinterpret_cast<const uint4 *>(a.shm.Z + (size_t)id * Dz + cc * 8);            \
        }                                                                                          \
    }
#define CHB_PF_STASH_ONE(I, ST, BB)                                                               \
    {                                                                                    \
        const int ch = tid + 256 * (I);                                                            \
        if (ch < nchunk) {                                                                         \
            const int r = ch / cpr, cc = ch - r * cpr;                                             \
            *reinterpret_cast<uint4 *>(sPz + ((size_t)(BB) * kPfP + r) * stride + cc * 16) = ST;   \
        }                                                                                          \
    }
#define CHB_PF_FETCH(TT, X)                                                                       \
    {                                                                                              \
        CHB_PF_FETCH_ONE(0, c##X##0, TT)                                                           \
        CHB_PF_FETCH_ONE(1, c##X##1, TT)                                                           \
        CHB_PF_FETCH_ONE(2, c##X##2, TT)                                                           \
        CHB_PF_FETCH_ONE(3, c##X##3, TT)                                                           \
        if (tid < kPfP) {                                                                          \
            const int e = (TT) * kPfP + tid;                                                       \
            if (e < nmem) {                                                                        \
                id##X = a.memb_id[mb + e];                                                         \
                n##X = a.shm.nrm[mb + e];                                                          \
                code##X = UPD ? a.memb_code[mb + e] : 0;                                           \
            } else {                                                                               \
                id##X = -1; n##X = INFINITY; code##X = 0;                                          \
            }                                                                                      \
        }                                                                                          \
    }
#define CHB_PF_STASH(BB, X, IDS)                                                                    \
    {                                                                                              \
        CHB_PF_STASH_ONE(0, c##X##0, BB)                                                           \
        CHB_PF_STASH_ONE(1, c##X##1, BB)                                                           \
        CHB_PF_STASH_ONE(2, c##X##2, BB)                                                           \
        CHB_PF_STASH_ONE(3, c##X##3, BB)                                                           \
        if (tid < kPfP) {                                                                          \
            sPn[(BB) * kPfP + tid] = n##X;                                                         \
            sPid[(IDS) * kPfP + tid] = id##X;                                                      \
            sPcode[(BB) * kPfP + tid] = code##X;                                                   \
        }                                                                                          \
    }

    const size_t slot = (size_t)c * a.Kcap + qpos;
    int *cand = a.cand + slot * kCandCap;

    // Two sweeps over the bin's members.  Sweep 0 (base mode only) learns tau = m-th smallest upper
    // bound; sweep 1 shortlists against that FINAL tau, so the list holds only the members whose
    // error interval reaches below it (about m + a handful) instead of everything that passed a
    // still-loose running threshold.  The matrix-core work is cheap enough to do twice.
    // rho_j + (largest rho of any member this kernel can meet): constant per (query, bin)
    const float rsum = a.rho_bound[c] * (1.0f + kSlack);
    unsigned pend_mask = 0u;   // candidates of the previous tile whose stores are still to be issued
    int pend_off = 0, pend_buf = 0;
    float thr = INFINITY;   // sweep 0: m-th smallest t1 so far
    float C2 = FLT_MAX;     // sweep 1: admission bound on t2
    if (UPD && tau < INFINITY) {
        const float hi = tau * (1.0f + 4.0f * kSlack) + rsum;
        C2 = hi * hi * (1.0f + 4.0f * kSlack) - nj_lo;
    }
    for (int sweep = UPD ? 1 : 0; sweep < 2; ++sweep) {
        if (!UPD && sweep == 1) {
            // m-th smallest t1 over BOTH lane halves of the query
            float mg[ML];
#pragma unroll
            for (int i = 0; i < ML; ++i) mg[i] = ub[i];
#pragma unroll
            for (int i = 0; i < ML; ++i) list_insert<ML>(mg, __shfl_xor(ub[i], 32, 64));
#pragma unroll
            for (int i = 0; i < ML; ++i)
                if (i == m - 1) thr = mg[i];
        }
        if (!UPD && sweep == 1 && thr < INFINITY) {
            // tau = m-th smallest upper bound; at least m members are provably within it
            tau = sqrtf(fmaxf(thr + nj_hi, 0.f)) * (1.0f + 4.0f * kSlack) + rsum;
            const float hi = tau * (1.0f + 4.0f * kSlack) + rsum;
            C2 = hi * hi * (1.0f + 4.0f * kSlack) - nj_lo;
        }
        auto flush_pending = [&]() {
            while (pend_mask) {
                const int r = __ffs(pend_mask) - 1;
                pend_mask &= pend_mask - 1u;
                const int prow = (r & 3) + 8 * (r >> 2) + 4 * h;
                if (pend_off < kCandCap) cand[pend_off] = sPid[pend_buf * kPfP + prow];
                ++pend_off;
            }
        };
        auto consume = [&](int buf, int idslot) {
        flush_pending();
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const unsigned char *pbase = sPz + ((size_t)buf * kPfP + col) * stride + h * 16;
#pragma unroll
        for (int sx = 0; sx < KSMAX; ++sx)
            if (sx < ksteps) {
                const bf16x8 af = *reinterpret_cast<const bf16x8 *>(pbase + sx * 32);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, qreg[sx], acc, 0, 0, 0);
            }

        // rows held by this lane: (r&3) + 8*(r>>2) + 4*h
        float np[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 v = *reinterpret_cast<const float4 *>(&sPn[buf * kPfP + 8 * g + 4 * h]);
            np[4 * g + 0] = v.x; np[4 * g + 1] = v.y; np[4 * g + 2] = v.z; np[4 * g + 3] = v.w;
        }
        if (UPD) {
            // a batch member counts for this query only on the right side of the visiting order
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int code = sPcode[buf * kPfP + (r & 3) + 8 * (r >> 2) + 4 * h];
                bool ok = true;
                if (code > 0) ok = qpos > code - 1;
                else if (code <= -(1 << 30)) ok = qpos != -(1 << 30) - code;
                else if (code < 0) ok = qpos < -code - 1;
                if (!ok) np[r] = INFINITY;
            }
        }

        if (sweep == 0) {
            // Learn tau.  UB'(p) = sqrt(t1_p + n_j(1+g)) + rho_bin is monotone in
            // t1_p = n_p(1+g) - 2<zh_j, zh_p>, so the m smallest t1 are kept (no sqrt per insert).
            // Each lane half keeps ITS m smallest and filters against its own m-th (`thr`); the two
            // halves of a query are merged once, after the sweep.  Per tile: the 16 values, their
            // minimum, ONE branch; the (rare) insertions happen smallest-first inside it.
            float t1v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) t1v[r] = fmaf(kGamma, np[r], fmaf(-2.0f, acc[r], np[r]));
            float mn = t1v[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mn = fminf(mn, t1v[r]);
            while (mn < thr) {
                list_insert<ML>(ub, mn);
#pragma unroll
                for (int i = 0; i < ML; ++i)
                    if (i == m - 1) thr = ub[i];
                float nx = INFINITY;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    t1v[r] = t1v[r] == mn ? INFINITY : t1v[r];   // consume (all copies of a tie at once:
                    nx = fminf(nx, t1v[r]);                     //  harmless, thr only gets looser)
                }
                mn = nx;
            }
        } else {
            // shortlist: LB' = sqrt(s - E) - rho_j - rho_bin <= tau  <=>  t2 <= C2 (constant)
            unsigned mask = 0u;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float t2 = fmaf(-kGamma, np[r], fmaf(-2.0f, acc[r], np[r]));
                if (t2 <= C2) mask |= 1u << r;
            }
            const int cnt = __popc(mask);
            const int pc = __shfl_xor(cnt, 32, 64);
            // The global stores of this tile's candidates are issued at the START of the next tile
            // (flush_pending): the barrier that ends a tile waits for vmcnt(0), and a store issued
            // right before it would expose its whole write latency every tile.
            pend_mask = qvalid ? mask : 0u;
            pend_off = ccount + (h ? pc : 0);
            pend_buf = idslot;
            ccount += cnt + pc;
        }

        };
        __syncthreads();
#if CHB_PF_DEEP
        if (ntile > 0) CHB_PF_FETCH(0, A)
        if (ntile > 1) CHB_PF_FETCH(1, B)
        if (ntile > 0) CHB_PF_STASH(0, A, 0)
        __syncthreads();
        for (int t = 0; t < ntile; t += 2) {
            // even tile t sits in buffer 0, tile t+1 is in flight in SB
            if (t + 2 < ntile) CHB_PF_FETCH(t + 2, A)
            __builtin_amdgcn_sched_barrier(0);
            consume(0, t & 3);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < ntile) CHB_PF_STASH(1, B, (t + 1) & 3)
            __syncthreads();
            if (t + 1 >= ntile) break;
            // odd tile t+1 sits in buffer 1, tile t+2 is in flight in SA
            if (t + 3 < ntile) CHB_PF_FETCH(t + 3, B)
            __builtin_amdgcn_sched_barrier(0);
            consume(1, (t + 1) & 3);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 2 < ntile) CHB_PF_STASH(0, A, (t + 2) & 3)
            __syncthreads();
        }
#else
        if (ntile > 0) { CHB_PF_FETCH(0, A) CHB_PF_STASH(0, A, 0) }
        __syncthreads();
        for (int t = 0; t < ntile; ++t) {
            if (t + 1 < ntile) CHB_PF_FETCH(t + 1, A)
            __builtin_amdgcn_sched_barrier(0);
            consume(t & 1, t & 3);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < ntile) CHB_PF_STASH((t + 1) & 1, A, (t + 1) & 3)
            __syncthreads();
        }
#endif
        flush_pending();
    }

    if (qvalid && h == 0) {
        a.cand_cnt[slot] = ccount < kCandCap ? ccount : kCandCap;
        if (UPD && ccount > 0 && a.active != nullptr)
            a.active[atomicAdd(a.n_active, 1)] = (qpos - a.pos_begin) * a.B + c;
        if (ccount > kCandCap) {
            atomicAdd(a.overflow, 1);
            flags64[(size_t)c * nqt64 + (qpos - a.pos_begin) / kQTile] = 1;
        }
    }
}


// ---------------------------------------------------------------------------------------------
// The streamlined form of the kernel above for the two row shapes of real feature tables
// (Dz = 144: 9 matrix-core steps, Dz = 160: 10).  Same bounds, same shortlist; what differs is
// how the work is fed:
//   * Dz / 16 is a compile-time constant, so all fragment reads of a tile are in flight together
//     and the matrix-core steps issue back to back;
//   * every global read inside the tile loop is an LDS-DMA (shadow rows in 1-KiB pieces, the
//     member norms / eligibility as 4-byte pieces): no ordinary load whose use would make the
//     compiler drain the DMA queue.  Three LDS buffers, two tiles in flight, one raw s_barrier and
//     one COUNTED s_waitcnt vmcnt(n) per tile (n = this wavefront's DMA instructions per tile);
//   * n_p (1 +- g) / 2 is folded into the accumulator's start value, so the matrix core returns
//     -t/2 directly: sweep 0 is a 16-way maximum, sweep 1 sixteen compares;
//   * shortlist hits are parked per wavefront in LDS as (query, member offset) words through a
//     ballot + mbcnt compaction (no atomics, no divergent store loop) and written out once, after
//     the last tile; the order inside a shortlist is irrelevant (rescore_kernel orders exactly).
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
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
}

// LDS accesses of the tile loop are issued from inline asm: while an LDS-DMA is in flight the
// compiler puts s_waitcnt vmcnt(0) in front of every LDS access it can see (it cannot tell the
// buffers apart), which would drain the two tiles kept in flight.  The asm reads are completed by
// lds_wait_all() before their results are used.
using f32x4 = __attribute__((ext_vector_type(4))) float;
template <int OFF>
__device__ __forceinline__ bf16x8 lds_read_frag(unsigned addr)
{
    bf16x8 v;
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

// eligibility code (aux_kernels.hip) -> (s, b) with "eligible for position q  <=>  s q + b >= 0"
__global__ __launch_bounds__(256) void code_affine_kernel(const int *memb_code, const int *bin_ptr, int B,
                                                          float *cs, float *cb)
{
    const int n = bin_ptr[B];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int code = memb_code[i];
        float sv = 0.f, bv = 0.f;
        if (code > 0) { sv = 1.f; bv = -(float)code; }                       // q > code - 1
        else if (code <= -(1 << 30)) { sv = 0.f; bv = -1.f; }                // "q != i" has no affine form:
                                                                             // never routed here (launcher)
        else if (code < 0) { sv = -1.f; bv = (float)(-code - 2); }           // q < -code - 1
        cs[i] = sv; cb[i] = bv;
    }
}

template <int ML, bool UPD, int KS>
__global__ __launch_bounds__(256, ML <= 8 ? 4 : 3) void shortlist_kernel(PrefilterArgs a, int nqt, int total,
                                                                       int *flags64, int nqt64)
{
    constexpr int CPR = 2 * KS;              // 16-byte chunks per shadow row
    constexpr int ROWB = 32 * KS;            // bytes per shadow row
    constexpr int TILEB = kPfP * ROWB;       // one member tile
    constexpr int METAB = 512;               // floats [0,32) n_p | [32,64) s | [64,96) b
    constexpr int BUFB = TILEB + METAB;
    constexpr int NBUF = 3;
    constexpr int kPoolW = shortlist_pool_entries(ML);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned *sPool = reinterpret_cast<unsigned *>(smem + NBUF * BUFB);   // [4][kPoolW]
    int *sCnt = reinterpret_cast<int *>(sPool + 4 * kPoolW);              // [kPfQ]

    const int per = (total + 7) >> 3;
    const int W = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (W >= total) return;
    const int c = W / nqt, qt = W - c * nqt;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned smem_base = lds_addr(smem);
    const unsigned pool_base = lds_addr(sPool) + (unsigned)(w * kPoolW * 4);
    const int col = lane & 31, h = lane >> 5;
    const int mb = a.bin_ptr[c];
    const int nmem = a.bin_ptr[c + 1] - mb;
    const int pos0 = a.pos_begin + qt * kPfQ;
    const int m = a.m;

    const int qpos = pos0 + 32 * w + col;
    const bool qvalid = qpos < a.pos_end;
    bf16x8 qreg[KS];
    float nj, rq, njx_up, njx_dn;
    {
        const size_t qslot = (size_t)c * a.Kcap + (qvalid ? qpos : a.pos_end - 1);
        const unsigned short *zq = a.Zq + qslot * (KS * 16) + h * 8;
#pragma unroll
        for (int sx = 0; sx < KS; ++sx) qreg[sx] = *reinterpret_cast<const bf16x8 *>(zq + sx * 16);
        const float4 q4 = a.qs[qslot];
        nj = q4.x; rq = q4.y; njx_up = q4.z; njx_dn = q4.w;
    }
    // bounds: see prefilter_kernel
    const float Aq = (2.0f * rq * a.sn_bound[c] + kGamma * nj) * (1.0f + 4.0f * kSlack);
    const float nj_hi = (njx_up + Aq) * (1.0f + kSlack);
    const float nj_lo = (njx_dn - Aq) * (njx_dn > Aq ? (1.0f - kSlack) : (1.0f + kSlack));
    const float rsum = a.rho_bound[c] * (1.0f + kSlack);
    const float qposf = (float)qpos;

    // sweep 0: the m largest accumulator values (= m smallest t1) seen by this lane half, descending;
    // the first ML - m slots are pinned at +inf so that the m-th largest is always lb[ML - 1]
    float lb[ML];
#pragma unroll
    for (int i = 0; i < ML; ++i) lb[i] = i < ML - m ? INFINITY : -INFINITY;
    float thr_s = -INFINITY;
    // sweep 1 admits a member iff its accumulator >= thr2  (t2 <= C2, t2 = -2 acc)
    float thr2 = qvalid ? -FLT_MAX : INFINITY;
    if (UPD && qvalid) {
        const size_t sl = (size_t)c * a.Kcap + qpos;
        if (a.seed.cnt[sl] >= m) {
            const float tau = round_up_f32(a.seed.d[sl * m + m - 1]) * (1.0f + kSlack);   // exact m-th distance so far
            const float hi = tau * (1.0f + 4.0f * kSlack) + rsum;
            thr2 = -0.5f * (hi * hi * (1.0f + 4.0f * kSlack) - nj_lo);
        }
    }
    if (h == 0) sCnt[32 * w + col] = 0;

    const int ntile = (nmem + kPfP - 1) / kPfP;
    const int nvt = UPD ? ntile : 2 * ntile;   // the tile stream: sweep 0 then sweep 1
    // DMA roles: wavefront w moves the 1-KiB pieces w, w+4, w+8 of a tile (lane l of piece i fills
    // LDS chunk 64 i + l from the global chunk the swizzle maps there); wavefront 3 also moves the
    // norms (+ s), wavefront 2 the b column in update mode
    int src_off[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int pch = (w + 4 * j) * 64 + lane;
        const int r = pch / CPR, cs = pch - r * CPR;
        const int f = KS == 9 ? ((r >> 4) & 1) : ((r >> 2) & 3);
        src_off[j] = (r * CPR + (cs ^ f)) * 16;
    }
    int n_w = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) n_w += (w + 4 * j < KS) ? 1 : 0;
    n_w += (w == 3) ? 1 : 0;
    n_w += (UPD && w == 2) ? 1 : 0;
    const unsigned char *zbase = reinterpret_cast<const unsigned char *>(a.shm.Z) + (size_t)mb * ROWB;

    int it = 0, ibuf = 0;   // next tile to issue and its buffer
#define CHB_SL_ISSUE()                                                                             \
    {                                                                                              \
        unsigned char *dst_ = smem + ibuf * BUFB;                                                  \
        const unsigned char *src_ = zbase + (size_t)it * TILEB;                                    \
        _Pragma("unroll") for (int j = 0; j < 3; ++j)                                              \
            if (w + 4 * j < KS)                                                                    \
                __builtin_amdgcn_global_load_lds(src_ + src_off[j],                                \
                    (__attribute__((address_space(3))) void *)(dst_ + (w + 4 * j) * 1024), 16, 0, 0); \
        if (w == 3) {                                                                              \
            const int e_ = it * kPfP + col;                                                        \
            const float *p_ = e_ < nmem ? a.shm.nrm + mb + e_ : a.inf_ptr;                         \
            if (UPD && h) p_ = a.code_s + mb + (e_ < nmem ? e_ : nmem - 1);                        \
            __builtin_amdgcn_global_load_lds(p_, (__attribute__((address_space(3))) void *)(dst_ + TILEB), 4, 0, 0); \
        }                                                                                          \
        if (UPD && w == 2) {                                                                       \
            const int e_ = it * kPfP + col;                                                        \
            const float *p_ = a.code_b + mb + (e_ < nmem ? e_ : nmem - 1);                         \
            __builtin_amdgcn_global_load_lds(p_, (__attribute__((address_space(3))) void *)(dst_ + TILEB + 256), 4, 0, 0); \
        }                                                                                          \
        if (++it == ntile) it = 0;                                                                 \
        if (++ibuf == NBUF) ibuf = 0;                                                              \
    }

    if (nvt > 0) CHB_SL_ISSUE()
    if (nvt > 1) CHB_SL_ISSUE()

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
    const float c0 = -0.5f * (1.0f + kGamma), c1 = -0.5f * (1.0f - kGamma);
    const unsigned ent0 = ((unsigned)col << 27) | (unsigned)(4 * h);

    int ct = 0, cbuf = 0, sweep = UPD ? 1 : 0;
    const bool tile_best = ntile >= 16;
    int wcnt = 0;   // entries parked by this wavefront (wave-uniform)
    for (int vt = 0; vt < nvt; ++vt) {
        wait_vmcnt(vt + 1 < nvt ? n_w : 0);   // my pieces of tile vt have landed
        __builtin_amdgcn_s_barrier();         // everybody's have; buffer (vt + 2) % 3 is free again
        if (vt + 2 < nvt) CHB_SL_ISSUE()
        if (!UPD && vt == ntile) {
            // end of sweep 0: m-th smallest t1 over BOTH lane halves -> tau -> thr2
            sweep = 1; ct = 0;
            float mg[ML];
#pragma unroll
            for (int i = 0; i < ML; ++i) mg[i] = lb[i];
#pragma unroll
            for (int i = 0; i < ML; ++i) {
                const float o = __shfl_xor(lb[i], 32, 64);
                list_insert_desc<ML>(mg, i >= ML - m ? o : -INFINITY);   // (not the pinned slots twice)
            }
            const float ms = mg[ML - 1];
            if (qvalid && ms > -INFINITY) {
                // tau = m-th smallest upper bound; at least m members are provably within it
                const float thr = -2.0f * ms;
                const float tau = sqrtf(fmaxf(thr + nj_hi, 0.f)) * (1.0f + 4.0f * kSlack) + rsum;
                const float hi = tau * (1.0f + 4.0f * kSlack) + rsum;
                thr2 = -0.5f * (hi * hi * (1.0f + 4.0f * kSlack) - nj_lo);
            }
        }

        const unsigned tb = smem_base + (unsigned)(cbuf * BUFB);
        // rows held by this lane: (r & 3) + 8 (r >> 2) + 4 h
        const unsigned ma = tb + (unsigned)(TILEB + 16 * h);
        f32x4 nv[4], sv[4], bv[4];
#define CHB_SL_META(G)                                                                             \
        nv[G] = lds_read_f4<32 * (G)>(ma);                                                         \
        if (UPD) { sv[G] = lds_read_f4<128 + 32 * (G)>(ma); bv[G] = lds_read_f4<256 + 32 * (G)>(ma); }
        CHB_SL_META(0) CHB_SL_META(1) CHB_SL_META(2) CHB_SL_META(3)
#undef CHB_SL_META
        bf16x8 af[KS == 9 ? 9 : 10];
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
        // every asm read above has landed once this returns (operands tied so that no use can be
        // scheduled ahead of the wait)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(nv[0]), "+v"(nv[1]), "+v"(nv[2]), "+v"(nv[3]), "+v"(af[0]), "+v"(af[1]), "+v"(af[2]),
                       "+v"(af[3]), "+v"(af[4]), "+v"(af[5]), "+v"(af[6]), "+v"(af[7]), "+v"(af[8])
                     :
                     : "memory");
        if (KS == 10) asm volatile("" : "+v"(af[KS - 1]) : : "memory");
        if (UPD)
            asm volatile("" : "+v"(sv[0]), "+v"(sv[1]), "+v"(sv[2]), "+v"(sv[3]), "+v"(bv[0]), "+v"(bv[1]),
                              "+v"(bv[2]), "+v"(bv[3]) : : "memory");
        f32x16 acc;
        const float cinit = sweep ? c1 : c0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            acc[4 * g + 0] = nv[g][0] * cinit; acc[4 * g + 1] = nv[g][1] * cinit;
            acc[4 * g + 2] = nv[g][2] * cinit; acc[4 * g + 3] = nv[g][3] * cinit;
            if (UPD) {
                // a batch member counts for this query only on the right side of the visiting order
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (fmaf(sv[g][k], qposf, bv[g][k]) < 0.f) acc[4 * g + k] = -INFINITY;
            }
        }
#pragma unroll
        for (int sx = 0; sx < KS; ++sx)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[sx], qreg[sx], acc, 0, 0, 0);

        if (!UPD && sweep == 0) {
            // acc = -t1/2 with t1 = n_p (1 + g) - 2 <zh_j, zh_p>: UB is monotone in t1, so the m
            // LARGEST accumulators are kept; per tile a 16-way maximum and one branch
            float mx = acc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, acc[r]);
            if (tile_best) {
                // Large bins: only the BEST of the 16 values enters the list.  The m-th best of
                // per-(tile, lane half) bests is the m-th best of m distinct members, i.e. still a
                // valid tau; it is the exact m-th unless two of the top m share a tile half
                // (probability ~ m^2 / (4 ntile)), and then one rank looser.  No rescan loop.
                if (mx > thr_s) {
                    list_insert_desc<ML>(lb, mx);
                    thr_s = lb[ML - 1];
                }
            } else
            while (mx > thr_s) {
                list_insert_desc<ML>(lb, mx);
                thr_s = lb[ML - 1];
                float nx = -INFINITY;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc[r] = acc[r] == mx ? -INFINITY : acc[r];   // (all copies of a tie at once:
                    nx = fmaxf(nx, acc[r]);                       //  harmless, thr only gets looser)
                }
                mx = nx;
            }
        } else {
            const unsigned ebase = ent0 + (unsigned)(ct * kPfP);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool hit = acc[r] >= thr2;
                const unsigned long long bal = __ballot(hit);
                if (bal) {
                    const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32),
                                                                  __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                    const int pos = wcnt + before;
                    if (hit && pos < kPoolW) lds_write_u32(pool_base + 4u * (unsigned)pos, ebase + (unsigned)((r & 3) + 8 * (r >> 2)));
                    wcnt += __popcll(bal);
                }
            }
        }
        if (++ct == ntile) ct = 0;
        if (++cbuf == NBUF) cbuf = 0;
    }
#undef CHB_SL_ISSUE

    // write the parked entries out: entry -> (query of this wavefront, member offset in the bin)
    const int npark = wcnt < kPoolW ? wcnt : kPoolW;
    for (int i = lane; i < npark; i += 64) {
        const unsigned en = sPool[w * kPoolW + i];
        const int qc = (int)(en >> 27), e = (int)(en & ((1u << 27) - 1u));
        const int off = atomicAdd(&sCnt[32 * w + qc], 1);
        if (off < kCandCap)
            a.cand[((size_t)c * a.Kcap + pos0 + 32 * w + qc) * kCandCap + off] = a.memb_id[mb + e];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const int ccount = (qvalid && h == 0) ? sCnt[32 * w + col] : 0;
    if (qvalid && h == 0) {
        const size_t slot = (size_t)c * a.Kcap + qpos;
        a.cand_cnt[slot] = ccount < kCandCap ? ccount : kCandCap;
        if (ccount > kCandCap || wcnt > kPoolW) {
            atomicAdd(a.overflow, 1);
            flags64[(size_t)c * nqt64 + (qpos - a.pos_begin) / kQTile] = 1;
        }
    }
    if (UPD && a.active != nullptr) {
        // pairs with a non-empty shortlist, appended with ONE atomic per wavefront
        const bool act = ccount > 0;
        const unsigned long long bal = __ballot(act);
        if (bal) {
            const int leader = __ffsll((long long)bal) - 1;
            int base = 0;
            if (lane == leader) base = atomicAdd(a.n_active, __popcll(bal));
            base = __shfl(base, leader, 64);
            const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
            if (act) a.active[base + before] = (qpos - a.pos_begin) * a.B + c;
        }
    }
}

}  // namespace

void launch_bin_centers(const double *X, int D, int Dp, const int *memb_id, const int *bin_ptr, int B,
                        double *centers, hipStream_t s)
{
    if (B > 0) hipLaunchKernelGGL(bin_center_kernel, dim3(B), dim3(256), 0, s, X, D, Dp, memb_id, bin_ptr, centers);
}

void launch_pack_centered(const double *X, int D, int Dp, const int *memb_id, const int *bin_ptr, int B,
                          int rows_hint, const double *centers, unsigned short *Zp, int Dz, float *nrm_p,
                          float *rho_p, hipStream_t s)
{
    if (B <= 0) return;
    int gy = (rows_hint / std::max(B, 1) + 15) / 16;   // ~4 members per wavefront per bin
    gy = std::max(1, std::min(gy, 64));
    hipLaunchKernelGGL(pack_centered_kernel, dim3(B, gy), dim3(256), 0, s, X, D, Dp, memb_id, bin_ptr,
                       centers, Zp, Dz, nrm_p, rho_p);
}

void launch_sample_shadow(const double *X, int D, int Dp, const int *ids, int n, const int *labels,
                          int B, const double *centers, unsigned short *Zs, int Dz, float *nrm_s,
                          float *rho_s, hipStream_t s)
{
    if (n > 0)
        hipLaunchKernelGGL(sample_shadow_kernel, dim3((n + 3) / 4), dim3(256), 0, s, X, D, Dp, ids, n, labels,
                           B, centers, Zs, Dz, nrm_s, rho_s);
}

void launch_pack_rows(const Shadow &src, const int *memb_id, const int *bin_ptr, int B, int n_max,
                      unsigned short *Zp, float *nrm_p, float *rho_p, hipStream_t s)
{
    if (n_max <= 0) return;
    long long nch = (long long)n_max * (src.Dz >> 3);
    int grid = (int)std::min<long long>((nch + 255) / 256, 8192);
    hipLaunchKernelGGL(pack_rows_kernel, dim3(grid), dim3(256), 0, s, src.Z, src.nrm, src.rho, src.Dz,
                       memb_id, bin_ptr, B, Zp, nrm_p, rho_p);
}

void launch_query_shadow(const double *X, int D, int Dp, const int *bq, int pos_begin, int pos_end, int B,
                         int Kcap, const double *centers, unsigned short *Zq, int Dz, void *qs,
                         hipStream_t s)
{
    const long long nrow = (long long)(pos_end - pos_begin) * B;
    if (nrow <= 0) return;
    hipLaunchKernelGGL(query_shadow_kernel, dim3((unsigned)((nrow + 15) / 16)), dim3(256), 0, s, X, D, Dp, bq,
                       pos_begin, pos_end, B, Kcap, centers, Zq, Dz, reinterpret_cast<float4 *>(qs));
}

void launch_bin_bounds(const float *rho_p, const float *nrm_p, const int *bin_ptr, int B, float *rho_out,
                       float *sn_out, hipStream_t s)
{
    if (B > 0) hipLaunchKernelGGL(bin_bounds_kernel, dim3(B), dim3(256), 0, s, rho_p, nrm_p, bin_ptr, rho_out, sn_out);
}

size_t prefilter_lds_bytes(int Dz)
{
    const int stride = Dz * 2 + 16;
    return (size_t)(2 * kPfP) * stride + 2 * kPfP * (4 + 4 + 4) + 4 * kPfP * 4;
}

static size_t shortlist_lds_bytes(int ks, int ml)
{
    return (size_t)3 * (kPfP * 32 * ks + 512) + (size_t)4 * shortlist_pool_entries(ml) * 4 + kPfQ * 4;
}

// the streamlined kernel exists for 18 or 20 16-byte chunks per row (the two row shapes its
// source-side swizzle is laid out for); CHB_PF_V2=0 forces the generic kernel
static bool use_shortlist_kernel(const PrefilterArgs &a)
{
    static int env = -1;
    if (env < 0) { const char *e = getenv("CHB_PF_V2"); env = e ? atoi(e) : 1; }
    return env != 0 && (a.shm.Dz == 144 || a.shm.Dz == 160) && a.inf_ptr != nullptr;
}

template <int ML, bool UPD>
static void launch_pf(const PrefilterArgs &a, int grid, size_t lds, int nqt, int total, int stride,
                      int *flags64, int nqt64, hipStream_t s)
{
    if (use_shortlist_kernel(a) && (!UPD || (a.code_s != nullptr && a.code_b != nullptr))) {
        if (UPD)
            hipLaunchKernelGGL(code_affine_kernel, dim3(64), dim3(256), 0, s, a.memb_code, a.bin_ptr, a.B,
                               a.code_s, a.code_b);
        if (a.shm.Dz == 144)
            hipLaunchKernelGGL((shortlist_kernel<ML, UPD, 9>), dim3(grid), dim3(256), shortlist_lds_bytes(9, ML), s, a,
                               nqt, total, flags64, nqt64);
        else
            hipLaunchKernelGGL((shortlist_kernel<ML, UPD, 10>), dim3(grid), dim3(256), shortlist_lds_bytes(10, ML), s, a,
                               nqt, total, flags64, nqt64);
        return;
    }
    hipLaunchKernelGGL((prefilter_kernel<ML, UPD>), dim3(grid), dim3(256), lds, s, a, nqt, total, stride, flags64,
                       nqt64);
}

void launch_prefilter(const PrefilterArgs &a, int *flags64, hipStream_t s)
{
    const int nq = a.pos_end - a.pos_begin;
    if (nq <= 0 || a.B <= 0) return;
    const int nqt = (nq + kPfQ - 1) / kPfQ;
    const int total = nqt * a.B;
    const int grid = ((total + 7) / 8) * 8;
    const int stride = a.shm.Dz * 2 + 16;
    const size_t lds = prefilter_lds_bytes(a.shm.Dz);
    const int nqt64 = (nq + kQTile - 1) / kQTile;
    if (a.memb_code != nullptr) {
        launch_pf<1, true>(a, grid, lds, nqt, total, stride, flags64, nqt64, s);
    } else if (a.m <= 5) {
        launch_pf<5, false>(a, grid, lds, nqt, total, stride, flags64, nqt64, s);
    } else if (a.m <= 8) {
        launch_pf<8, false>(a, grid, lds, nqt, total, stride, flags64, nqt64, s);
    } else {
        launch_pf<16, false>(a, grid, lds, nqt, total, stride, flags64, nqt64, s);
    }
}

}  // namespace chb
