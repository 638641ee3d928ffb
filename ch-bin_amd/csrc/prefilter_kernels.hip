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
// DMA  = true : member tiles go global -> LDS directly (global_load_lds_dwordx4, no staging
//               registers, no ds_write); needs packed rows and 18 chunks per row (Dz = 144).  The LDS
//               image is unpadded and XOR-swizzled through the SOURCE address (chunk c of row r sits
//               at c ^ f(r): f = (r >> 4) & 1 for 18 chunks per row (Dz = 144), f = (r >> 2) & 3 for 20
//               (Dz = 160)), which makes the ds_read_b128 fragment reads conflict-free.
template <int ML, bool UPD, bool DMA>
__global__ __launch_bounds__(256, CHB_PF_WAVES) void prefilter_kernel(PrefilterArgs a, int nqt, int total,
                                                        int stride, int *flags64, int nqt64)
{
    constexpr int KSMAX = 10;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *sPz = smem;                               // [2][kPfP][stride]
    float *sPn = reinterpret_cast<float *>(sPz + (size_t)2 * kPfP * stride);  // [2][kPfP]
    float *sPr = sPn + 2 * kPfP;                             // [2][kPfP]
    int *sPid = reinterpret_cast<int *>(sPr + 2 * kPfP);     // [2][kPfP]
    int *sPcode = sPid + 2 * kPfP;                           // [2][kPfP]

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
    // LDS-DMA roles: wave w issues the 1-KiB pieces w, w+4, w+8, w+12 of a tile; lane l of piece i
    // fills LDS chunk p = 64 i + l and fetches the global chunk that the swizzle maps there
    int dma_off[4] = {0, 0, 0, 0};
    if (DMA) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pch = (w + 4 * j) * 64 + lane;
            const int r = pch / cpr, cs = pch - r * cpr;
            const int f = (cpr & 15) == 2 ? ((r >> 4) & 1) : ((r >> 2) & 3);
            dma_off[j] = (r * cpr + (cs ^ f)) * 16;
        }
    }
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
    if (DMA) {                                                                                     \
        if (w + 4 * (I) < (cpr >> 1))                                                              \
            __builtin_amdgcn_global_load_lds(                                                      \
                reinterpret_cast<const unsigned char *>(a.shm.Z) +                                 \
                    ((size_t)(mb + (TT) * kPfP) * Dz) * 2 + dma_off[I],                            \
                (__attribute__((address_space(3))) void *)(sPz + (size_t)((TT) & 1) * kPfP * stride + \
                                                          (w + 4 * (I)) * 1024),                  \
                16, 0, 0);                                                                         \
    } else {                                                                                       \
        const int ch = tid + 256 * (I);                                                            \
        if (ch < nchunk) {                                                                         \
            const int r = ch / cpr, cc = ch - r * cpr;                                             \
            const int e = (TT) * kPfP + r;                                                         \
            const int id = e < nmem ? mb + e : mb;                                                 \
            ST = *reinterpret_cast<const uint4 *>(a.shm.Z + (size_t)id * Dz + cc * 8);            \
        }                                                                                          \
    }
#define CHB_PF_STASH_ONE(I, ST, BB)                                                               \
    if (!DMA) {                                                                                    \
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
#define CHB_PF_STASH(BB, X)                                                                       \
    {                                                                                              \
        CHB_PF_STASH_ONE(0, c##X##0, BB)                                                           \
        CHB_PF_STASH_ONE(1, c##X##1, BB)                                                           \
        CHB_PF_STASH_ONE(2, c##X##2, BB)                                                           \
        CHB_PF_STASH_ONE(3, c##X##3, BB)                                                           \
        if (tid < kPfP) {                                                                          \
            sPn[(BB) * kPfP + tid] = n##X;                                                         \
            sPid[(BB) * kPfP + tid] = id##X;                                                       \
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
    float thr = INFINITY;   // sweep 0: m-th smallest t1 so far
    float C2 = FLT_MAX;     // sweep 1: admission bound on t2
    if (UPD && tau < INFINITY) {
        const float hi = tau * (1.0f + 4.0f * kSlack) + rsum;
        C2 = hi * hi * (1.0f + 4.0f * kSlack) - nj_lo;
    }
    for (int sweep = UPD ? 1 : 0; sweep < 2; ++sweep) {
        if (!UPD && sweep == 1 && thr < INFINITY) {
            // tau = m-th smallest upper bound; at least m members are provably within it
            tau = sqrtf(fmaxf(thr + nj_hi, 0.f)) * (1.0f + 4.0f * kSlack) + rsum;
            const float hi = tau * (1.0f + 4.0f * kSlack) + rsum;
            C2 = hi * hi * (1.0f + 4.0f * kSlack) - nj_lo;
        }
        auto consume = [&](int buf) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const unsigned char *pbase = sPz + ((size_t)buf * kPfP + col) * stride + (DMA ? 0 : h * 16);
        const int swz = (cpr & 15) == 2 ? ((col >> 4) & 1) : ((col >> 2) & 3);
#pragma unroll
        for (int sx = 0; sx < KSMAX; ++sx)
            if (sx < ksteps) {
                const bf16x8 af = *reinterpret_cast<const bf16x8 *>(
                    pbase + (DMA ? ((2 * sx + h) ^ swz) * 16 : sx * 32));
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
            // Learn tau.  UB'(p) = sqrt(t1_p + n_j(1+g)) + rho_j + rho_bin is monotone in
            // t1_p = n_p(1+g) - 2<zh_j, zh_p>, so the m smallest t1 are kept (no sqrt per insert);
            // `thr` = m-th smallest t1 over both lane halves of this query.
            bool ins = false;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float t1 = fmaf(kGamma, np[r], fmaf(-2.0f, acc[r], np[r]));
                if (t1 < thr) {
                    list_insert<ML>(ub, t1);
                    ins = true;
                }
            }
            if (__any(ins)) {
                float mg[ML];
#pragma unroll
                for (int i = 0; i < ML; ++i) mg[i] = ub[i];
#pragma unroll
                for (int i = 0; i < ML; ++i) list_insert<ML>(mg, __shfl_xor(ub[i], 32, 64));
#pragma unroll
                for (int i = 0; i < ML; ++i)
                    if (i == m - 1) thr = mg[i];
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
            int off = ccount + (h ? pc : 0);
            ccount += cnt + pc;
            if (qvalid) {
                while (mask) {
                    const int r = __ffs(mask) - 1;
                    mask &= mask - 1u;
                    const int prow = (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (off < kCandCap) cand[off] = sPid[buf * kPfP + prow];
                    ++off;
                }
            }
        }

        };
        __syncthreads();
#if CHB_PF_DEEP
        if (ntile > 0) CHB_PF_FETCH(0, A)
        if (ntile > 1) CHB_PF_FETCH(1, B)
        if (ntile > 0) CHB_PF_STASH(0, A)
        __syncthreads();
        for (int t = 0; t < ntile; t += 2) {
            // even tile t sits in buffer 0, tile t+1 is in flight in SB
            if (t + 2 < ntile) CHB_PF_FETCH(t + 2, A)
            __builtin_amdgcn_sched_barrier(0);
            consume(0);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < ntile) CHB_PF_STASH(1, B)
            __syncthreads();
            if (t + 1 >= ntile) break;
            // odd tile t+1 sits in buffer 1, tile t+2 is in flight in SA
            if (t + 3 < ntile) CHB_PF_FETCH(t + 3, B)
            __builtin_amdgcn_sched_barrier(0);
            consume(1);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 2 < ntile) CHB_PF_STASH(0, A)
            __syncthreads();
        }
#else
        if (ntile > 0) { CHB_PF_FETCH(0, A) CHB_PF_STASH(0, A) }
        __syncthreads();
        for (int t = 0; t < ntile; ++t) {
            if (t + 1 < ntile) CHB_PF_FETCH(t + 1, A)
            __builtin_amdgcn_sched_barrier(0);
            consume(t & 1);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < ntile) CHB_PF_STASH((t + 1) & 1, A)
            __syncthreads();
        }
#endif
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
    return (size_t)(2 * kPfP) * stride + 2 * kPfP * (4 + 4 + 4 + 4);
}

template <int ML, bool UPD, bool DMA>
static void launch_pf3(const PrefilterArgs &a, int grid, size_t lds, int nqt, int total, int stride,
                       int *flags64, int nqt64, hipStream_t s)
{
    hipLaunchKernelGGL((prefilter_kernel<ML, UPD, DMA>), dim3(grid), dim3(256), lds, s, a, nqt, total,
                       stride, flags64, nqt64);
}

static bool use_dma(const PrefilterArgs &a)
{
    static int env = -1;
    if (env < 0) { const char *e = getenv("CHB_PF_DMA"); env = e ? atoi(e) : 1; }
    const int c16 = (a.shm.Dz >> 3) & 15;   // 16-byte chunks per row mod 16: swizzles exist for 2 and 4
    return env != 0 && (c16 == 2 || c16 == 4);
}

template <int ML, bool UPD>
static void launch_pf(const PrefilterArgs &a, int grid, size_t lds, int nqt, int total, int stride,
                      int *flags64, int nqt64, hipStream_t s)
{
    if (use_dma(a))
        launch_pf3<ML, UPD, true>(a, grid, lds, nqt, total, a.shm.Dz * 2, flags64, nqt64, s);
    else
        launch_pf3<ML, UPD, false>(a, grid, lds, nqt, total, stride, flags64, nqt64, s);
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
