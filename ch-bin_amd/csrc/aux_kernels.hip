// Label bookkeeping kernels of the speculative-batch sweep: membership buckets (CSR by bin),
// the strict-'>' argmin over bins (algorithm.py:47-60) and the first-changed-position reduction.
#include "chb_internal.h"

namespace chb {
thread_local Gate g_gate;
namespace {

constexpr double kInf = __builtin_huge_val();

__global__ void fill_i32_kernel(int *p, int v, int n, Gate gate)
{
    CHB_GATE(gate);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// batch end (without the shadow refresh): final labels out, marks cleared
__global__ void batch_close_kernel(int *labels, int *inb, const int *bq, const int *lab, int K)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < K) {
        const int p = bq[i];
        labels[p] = lab[i];
        inb[p] = -1;
    }
}

// ---- base members: every labelled sample that is not in the current batch

// CSR key of a base member.  Without shells (nsh == 1): its bin.  With shells: (bin, shell), the shell = the member's
// distance from its bin's centre in the shadow space, ||zh|| (ms[p].z = ||zh||^2), in units of shell_inv[bin], outermost
// shell first -- a bin's rows then come periphery first, and the members of a 32-row tile have similar norms, which is what
// lets the shortlist kernel skip tiles by the bound | ||z_query|| - ||zh_member|| | (prefilter_kernels.hip).  Any monotone
// bucketing is correct; shell_inv is a per-fit constant per bin.
__device__ __forceinline__ int member_key(int l, int p, const float4 *ms, const float *shell_inv, int nsh)
{
    if (nsh <= 1) return l;
    const float r = sqrtf(ms[p].z) * shell_inv[l];
    int sh = r == r ? (int)fminf(r, (float)(nsh - 1)) : 0;
    sh = sh < 0 ? 0 : sh;
    return l * nsh + (nsh - 1 - sh);
}

// bq != nullptr: the batch is opened in the same launch (labels remembered, members marked) -- the count then takes every labelled sample and the batch's own entries are subtracted again, so
// that neither part reads what the other writes.
__global__ void count_base_kernel(const int *labels, int *inb, int N, int B, int *cnt, const int *bq, int K,
                                  int *lab_old, const float4 *ms, const float *shell_inv, int nsh, Gate gate)
{
    CHB_GATE(gate);
    extern __shared__ int hist[];
    const int BK = B * nsh;
    for (int b = threadIdx.x; b < BK; b += blockDim.x) hist[b] = 0;
    __syncthreads();
    if (bq == nullptr) {
        for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < N; p += gridDim.x * blockDim.x) {
            const int l = labels[p];
            if (l >= 0 && l < B && inb[p] < 0) atomicAdd(&hist[member_key(l, p, ms, shell_inv, nsh)], 1);
        }
    } else {
        for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < N; p += gridDim.x * blockDim.x) {
            const int l = labels[p];
            if (l >= 0 && l < B) atomicAdd(&hist[member_key(l, p, ms, shell_inv, nsh)], 1);
        }
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < K; i += gridDim.x * blockDim.x) {
            const int p = bq[i];
            const int l = labels[p];
            lab_old[i] = l;
            inb[p] = i;
            if (l >= 0 && l < B) atomicSub(&hist[member_key(l, p, ms, shell_inv, nsh)], 1);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < BK; b += blockDim.x)
        if (hist[b]) atomicAdd(&cnt[b], hist[b]);
}

// exclusive scan of 256 per-thread partials (two of them at once) by the block's first wavefront, in place;
// returns the totals through tot / ptot (valid on every thread after the trailing barrier)
__device__ __forceinline__ void block_scan256(int *part, int *ppart, int *tot2)
{
    __syncthreads();
    if (threadIdx.x < 64) {
        const int l = threadIdx.x;
        int loc = 0, ploc = 0;
        for (int i = 0; i < 4; ++i) { loc += part[4 * l + i]; ploc += ppart[4 * l + i]; }
        int inc = loc, pinc = ploc;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(inc, off, 64), po = __shfl_up(pinc, off, 64);
            if (l >= off) { inc += o; pinc += po; }
        }
        int run = inc - loc, prun = pinc - ploc;
        for (int i = 0; i < 4; ++i) {
            const int v = part[4 * l + i], pv = ppart[4 * l + i];
            part[4 * l + i] = run; ppart[4 * l + i] = prun;
            run += v; prun += pv;
        }
        if (l == 63) { tot2[0] = run; tot2[1] = prun; }
    }
    __syncthreads();
}

// cursor[key] = exclusive scan of cnt over the B * nsh keys (the fill's write cursors); bin_ptr = the same per bin;
// pad_ptr (optional) = per bin with every bin's count rounded up to a multiple of 32 (the padded layout of the shortlist
// stage).  One block of 256 threads.  Also: the batch's segment plan and bin-size statistics (see SegPlan).
__global__ __launch_bounds__(256) void scan_kernel(int *cnt, int B, int nsh, int *bin_ptr, int *cursor, int *pad_ptr,
                                                    int *zero_me, SegPlan seg, int *stats, Gate gate)
{
    CHB_GATE(gate);
    if (zero_me != nullptr && threadIdx.x == 0) *zero_me = 0;   // (the fallback list's counter: saves a launch)
    __shared__ int part[256], ppart[256], tot2[2];
    __shared__ int s_max_tiles, s_ng, s_ni;
    if (threadIdx.x == 0) { s_max_tiles = 0; s_ng = 0; s_ni = 0; }
    // ---- phase 1: the keys
    const int BK = B * nsh;
    {
        const int per = (BK + 255) / 256;
        const int k0 = min(BK, (int)threadIdx.x * per), k1 = min(BK, k0 + per);
        int s = 0;
        for (int k = k0; k < k1; ++k) s += cnt[k];
        part[threadIdx.x] = s; ppart[threadIdx.x] = 0;
        block_scan256(part, ppart, tot2);
        int run = part[threadIdx.x];
        for (int k = k0; k < k1; ++k) {
            cursor[k] = run;
            run += cnt[k];
            cnt[k] = 0;   // left clean for the next count (launch_bucket_base needs no separate fill)
        }
    }
    const int total = tot2[0];
    __syncthreads();   // (cursor is read back below; tot2 is reused)
    // ---- phase 2: the bins
    const int per = (B + 255) / 256;
    const int b0 = min(B, (int)threadIdx.x * per), b1 = min(B, b0 + per);
    auto count_of = [&](int b) { return (b + 1 < B ? cursor[(b + 1) * nsh] : total) - cursor[b * nsh]; };
    int sp = 0;
    for (int b = b0; b < b1; ++b) sp += (count_of(b) + 31) / 32 * 32;
    part[threadIdx.x] = 0; ppart[threadIdx.x] = sp;
    block_scan256(part, ppart, tot2);
    if (threadIdx.x == 0) {
        bin_ptr[B] = total;
        if (pad_ptr) pad_ptr[B] = tot2[1];
    }
    int prun = ppart[threadIdx.x];
    const long long tot_tiles = tot2[1] / 32;
    for (int b = b0; b < b1; ++b) {
        const int c = count_of(b);
        bin_ptr[b] = cursor[b * nsh];
        if (pad_ptr) pad_ptr[b] = prun;
        const int ntile = (c + 31) / 32;
        if (stats != nullptr) atomicMax(&s_max_tiles, ntile);
        if (seg.gflag != nullptr) {
            // the batch's segment plan: a bin far larger than the rest becomes up to 16 work items of its own
            int g = -1;
            if (seg.launch && ntile > kSegMinTiles && (long long)ntile * B > 4 * tot_tiles) {
                const int len = max(kSegLenTiles, (ntile + 15) / 16);
                const int ns = (ntile + len - 1) / len;
                g = atomicAdd(&s_ng, 1);
                if (g < seg.gcap) {
                    const int i0 = atomicAdd(&s_ni, ns);
                    for (int sgi = 0; sgi < ns; ++sgi)
                        seg.items[i0 + sgi] = make_int4(b, sgi * len, min(ntile, (sgi + 1) * len),
                                                        (g << 8) | (sgi << 4) | (ns - 1));
                } else {
                    g = -1;
                }
            }
            seg.gflag[b] = g;
        }
        prun += (c + 31) / 32 * 32;
    }
    if (seg.gflag != nullptr || stats != nullptr) {
        __syncthreads();
        if (threadIdx.x == 0) {
            if (seg.nseg != nullptr) *seg.nseg = s_ni;
            // (stats[2], stats[3]: wave-tiles skipped / seen, accumulated by the coming shortlist launch)
            if (stats != nullptr) { stats[0] = s_max_tiles; stats[1] = (int)tot_tiles; stats[2] = 0; stats[3] = 0; stats[4] = 0; stats[6] = 0; stats[7] = 0; }
        }
    }
}

// Block-aggregated fill: one global atomic per (block, key) instead of one per sample.
__global__ __launch_bounds__(256) void fill_base_kernel(const int *labels, const int *inb, int N, int B,
                                                        int *cursor, int *memb_id, const float4 *ms,
                                                        const float *shell_inv, int nsh, Gate gate)
{
    CHB_GATE(gate);
    extern __shared__ int sh[];      // [B * nsh] local counts, then [B * nsh] block base offsets
    const int BK = B * nsh;
    int *cnt = sh, *base = sh + BK;
    for (int b = threadIdx.x; b < BK; b += blockDim.x) cnt[b] = 0;
    __syncthreads();
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    int l = -1, my = 0;
    if (p < N) {
        l = labels[p];
        if (!(l >= 0 && l < B && inb[p] < 0)) l = -1;
        else l = member_key(l, p, ms, shell_inv, nsh);
    }
    if (l >= 0) my = atomicAdd(&cnt[l], 1);
    __syncthreads();
    for (int b = threadIdx.x; b < BK; b += blockDim.x)
        base[b] = cnt[b] ? atomicAdd(&cursor[b], cnt[b]) : 0;
    __syncthreads();
    if (l >= 0) memb_id[base[l] + my] = p;
}

// ---- the batch's own members.  Position i appears (a) in bin lab_prev[i] for every LATER query
// (code i+1: eligible iff query pos > i) -- its speculative new label -- and (b) in bin lab_old[i]
// for every EARLIER query (code -(i+1): eligible iff query pos < i) -- it has not been visited yet
// when those queries run (algorithm.py:46-60 visits in permutation order).
// lab_old == nullptr selects the "everyone but myself" form used by chb_topm_per_bin:
// code -(1<<30) - i, eligible iff query pos != i.

// CSR of the batch's own entries in ONE workgroup (K is a few thousand): LDS histogram, scan,
// LDS cursors.  Entry codes: see chb_internal.h (TopmArgs::memb_code).
__global__ __launch_bounds__(1024) void bucket_batch_kernel(const int *lab_prev, const int *lab_old, const int *bq,
                                                            int K, int B, int *bin_ptr, int *pad_ptr, int *memb_id,
                                                            int *memb_code, int *first_change, int *n_slow, int *nflag,
                                                            float4 *bb_zero, Gate gate)
{
    CHB_GATE(gate);
    extern __shared__ int sh[];   // [B] counts -> cursors, [1024] scan partials x 2
    int *cnt = sh, *part = sh + B, *ppart = part + 1024;
    const int tid = threadIdx.x;
    if (bb_zero != nullptr)   // (the pack kernel of the batch's own entries accumulates the per-bin bounds into it)
        for (int b = tid; b < B; b += 1024) bb_zero[b] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid == 0) {   // the round's scalars
        if (first_change) *first_change = K;
        if (n_slow) *n_slow = 0;
        if (nflag) *nflag = 0;
    }
    for (int b = tid; b < B; b += 1024) cnt[b] = 0;
    __syncthreads();
    // (the first kReg entries of a thread stay in registers between the two passes: K <= 8192 at the default batch)
    constexpr int kReg = 8;
    int ra[kReg], rb[kReg];
#pragma unroll
    for (int t = 0; t < kReg; ++t) {
        const int i = tid + 1024 * t;
        ra[t] = i < K ? lab_prev[i] : -1;
        rb[t] = (lab_old && i < K) ? lab_old[i] : -1;
    }
#pragma unroll
    for (int t = 0; t < kReg; ++t) {
        if (ra[t] >= 0 && ra[t] < B) atomicAdd(&cnt[ra[t]], 1);
        if (rb[t] >= 0 && rb[t] < B) atomicAdd(&cnt[rb[t]], 1);
    }
    for (int i = tid + 1024 * kReg; i < K; i += 1024) {
        const int a = lab_prev[i];
        if (a >= 0 && a < B) atomicAdd(&cnt[a], 1);
        if (lab_old) {
            const int b = lab_old[i];
            if (b >= 0 && b < B) atomicAdd(&cnt[b], 1);
        }
    }
    __syncthreads();
    const int per = (B + 1023) / 1024;
    const int b0 = min(B, tid * per), b1 = min(B, b0 + per);
    int s = 0, sp = 0;
    for (int b = b0; b < b1; ++b) { s += cnt[b]; sp += (cnt[b] + 31) / 32 * 32; }
    part[tid] = s; ppart[tid] = sp;
    __syncthreads();
    if (tid < 64) {
        // exclusive scan of the partials by one wavefront: lane l owns parts 16 l .. 16 l + 15
        // (threads beyond ceil(B / per) own no bin and contribute zeros)
        int loc = 0, ploc = 0;
        for (int i = 0; i < 16; ++i) { loc += part[16 * tid + i]; ploc += ppart[16 * tid + i]; }
        int inc = loc, pinc = ploc;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(inc, off, 64), po = __shfl_up(pinc, off, 64);
            if (tid >= off) { inc += o; pinc += po; }
        }
        int run = inc - loc, prun = pinc - ploc;
        for (int i = 0; i < 16; ++i) {
            const int v = part[16 * tid + i], pv = ppart[16 * tid + i];
            part[16 * tid + i] = run; ppart[16 * tid + i] = prun;
            run += v; prun += pv;
        }
        if (tid == 63) {
            bin_ptr[B] = run;   // (the totals)
            if (pad_ptr) pad_ptr[B] = prun;
        }
    }
    __syncthreads();
    int run = part[tid], prun = ppart[tid];
    for (int b = b0; b < b1; ++b) {
        const int c = cnt[b];
        bin_ptr[b] = run;
        if (pad_ptr) pad_ptr[b] = prun;
        cnt[b] = run;   // becomes the bin's cursor
        run += c; prun += (c + 31) / 32 * 32;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < kReg; ++t) {
        const int i = tid + 1024 * t;
        if (ra[t] >= 0 && ra[t] < B) {
            const int e = atomicAdd(&cnt[ra[t]], 1);
            memb_id[e] = bq[i];
            memb_code[e] = lab_old ? (i + 1) : (-(1 << 30) - i);
        }
        if (rb[t] >= 0 && rb[t] < B) {
            const int e = atomicAdd(&cnt[rb[t]], 1);
            memb_id[e] = bq[i];
            memb_code[e] = -(i + 1);
        }
    }
    for (int i = tid + 1024 * kReg; i < K; i += 1024) {
        const int a = lab_prev[i];
        if (a >= 0 && a < B) {
            const int e = atomicAdd(&cnt[a], 1);
            memb_id[e] = bq[i];
            memb_code[e] = lab_old ? (i + 1) : (-(1 << 30) - i);
        }
        if (lab_old) {
            const int b = lab_old[i];
            if (b >= 0 && b < B) {
                const int e = atomicAdd(&cnt[b], 1);
                memb_id[e] = bq[i];
                memb_code[e] = -(i + 1);
            }
        }
    }
}

// algorithm.py:47-60: min_distance = inf, min_cluster = current label; strict '>' so the lowest
// bin wins ties and NaN never wins.
// Eight lanes per position (bins c = j, j + 8, ... on lane j: 64 contiguous bytes per step), combined in
// (distance, bin) order -- the result of the sequential loop over c: the lowest bin among equal minima.
__global__ __launch_bounds__(256) void argmin_kernel(const double *dist, const int *lab_old, int *lab_prev,
                              int pos_begin, int pos_end, int B, int *lab_new, double *mind, double *second,
                              int *first_change, int in_place, Gate gate)
{
    CHB_GATE(gate);
    const int j = threadIdx.x & 7;
    const int pos = pos_begin + (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 3);
    const bool valid = pos < pos_end;
    double best = kInf, runner = kInf;   // runner: smallest distance of any OTHER bin (for the margin report)
    int bc = -1;                         // -1: no bin beat +inf (the label stays)
    if (valid) {
        const double *row = dist + (size_t)pos * B;
        for (int c = j; c < B; c += 8) {
            const double d = row[c];
            if (best > d) { runner = best; best = d; bc = c; }
            else if (runner > d) runner = d;
        }
    }
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) {
        const double ob = __shfl_xor(best, off, 64), orr = __shfl_xor(runner, off, 64);
        const int oc = __shfl_xor(bc, off, 64);
        // the other side wins on a smaller distance, or on the same (finite) distance with the lower bin
        const bool take = ob < best || (ob == best && oc >= 0 && (bc < 0 || oc < bc));
        const double lose = take ? best : ob;
        runner = fmin(fmin(runner, orr), lose);
        if (take) { best = ob; bc = oc; }
    }
    if (!valid || j != 0) return;
    if (bc < 0) bc = lab_old[pos];
    lab_new[pos] = bc;
    mind[pos] = best;
    if (second != nullptr) second[pos] = runner;
    if (bc != lab_prev[pos]) atomicMin(first_change, pos);
    if (in_place) lab_prev[pos] = bc;   // (each position is read and written by its own thread only)
}

__global__ void first_change_kernel(const int *lab_new, const int *lab_prev, int p0, int K, int *first_change, Gate gate)
{
    CHB_GATE(gate);
    const int pos = p0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (pos < K && lab_new[pos] != lab_prev[pos]) atomicMin(first_change, pos);
}

__global__ void copy_i32_kernel(int *dst, const int *src, int n, Gate gate)
{
    CHB_GATE(gate);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// ---- framed exchange of the sharded loop (chb_api.hip: open_batch / finish_round).  What a rank sends in a round's
// all-gather is a FRAME: kXchgHdr header words -- {tag = exchange number of the fit << 4 | kind, wave-tiles skipped / seen /
// never loaded by its base shortlist launch, fill mark of its persistent pack's arena, 0, 0, 0} -- followed by the C labels
// of its slice.  Everything that steers the host loop (and with it the ORDER of the collectives) is then derived from what
// ALL ranks sent: the statistics are summed / maximised over the frames before they travel home with the verdict, so every
// rank takes the same decisions in the same batch; and a rank that is out of step (its tag differs) is noticed on the
// device and fails the fit at the sweep's end instead of silently mixing label buffers.
// (preset: slot[0] = K, the start value of the unpack kernel's first-change minimum -- this launch is ordered before the
//  all-gather and the unpack launch on the stream)
__global__ void xchg_pack_kernel(int *frames, int rank, int C, const int *src, int tag, int *slot, int with_stats,
                                 int with_mark, int preset, int K, Gate gate)
{
    CHB_GATE(gate);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int *f = frames + (size_t)rank * (C + kXchgHdr);
    if (i < C) f[kXchgHdr + i] = src[(size_t)rank * C + i];
    if (i == 0) {
        f[0] = tag;
        f[1] = with_stats ? slot[3] : 0; f[2] = with_stats ? slot[4] : 0; f[3] = with_stats ? slot[5] : 0;
        f[4] = with_mark ? slot[6] : 0;
        f[5] = with_stats ? slot[7] : 0; f[6] = with_stats ? slot[8] : 0; f[7] = 0;
        if (preset) slot[0] = K;
    }
}

// labels of every rank's frame -> dst[pos]; round kind (lab_prev != nullptr): positions >= active are compared with
// lab_prev (first change -> slot[0], preset to K by the pack launch) and then written to it; thread r < world checks rank
// r's tag; with_stats: slot[3..5] = sums over the ranks, slot[6] = largest fill mark
__global__ void xchg_unpack_kernel(const int *frames, int world, int C, int K, int tag, int *dst, int *lab_prev, int active,
                                   int *slot, int with_stats, int *xerr, Gate gate)
{
    CHB_GATE(gate);
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int F = C + kXchgHdr;
    if (idx < world * C) {
        const int r = idx / C, i = idx - r * C, pos = r * C + i;
        if (pos < K) {
            const int v = frames[(size_t)r * F + kXchgHdr + i];
            if (lab_prev == nullptr) dst[pos] = v;
            else if (pos >= active) {
                if (v != lab_prev[pos]) atomicMin(slot, pos);
                lab_prev[pos] = v;
                dst[pos] = v;
            }
        }
    }
    if (idx < world) {
        const int theirs = frames[(size_t)idx * F];
        if (theirs != tag && atomicCAS(&xerr[0], 0, 1) == 0) { xerr[1] = tag; xerr[2] = theirs; xerr[3] = idx; }
    }
    if (idx == 0 && with_stats) {
        int a = 0, b = 0, c = 0, mk = 0, pc = 0, pp = 0;
        for (int r = 0; r < world; ++r) {
            const int *f = frames + (size_t)r * F;
            // (saturating: the sums only feed ratio tests)
            a = (int)min(0x7fffffffll, (long long)a + f[1]); b = (int)min(0x7fffffffll, (long long)b + f[2]);
            c = (int)min(0x7fffffffll, (long long)c + f[3]); mk = max(mk, f[4]);
            pc = (int)min(0x7fffffffll, (long long)pc + f[5]); pp = (int)min(0x7fffffffll, (long long)pp + f[6]);
        }
        slot[3] = a; slot[4] = b; slot[5] = c; slot[6] = mk; slot[7] = pc; slot[8] = pp;
    }
}

// First-round label guess for batch members that carry no label yet (sweep 1): the bin of the
// single nearest outside member.  Only a guess -- the rounds converge to the exact sequential
// labels from any starting point; a good guess just saves a round.
// 16 lanes per position, each over every 16th bin; ties go to the lower bin as in a serial scan.
__global__ __launch_bounds__(256) void guess_kernel(const double *list_d, const int *list_cnt, const int *lab_old,
                                                    int p0, int K, int B, int m, int Kcap, int *lab_prev)
{
    const int l16 = threadIdx.x & 15;
    const int pos = p0 + blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool valid = pos < K;
    int g = valid ? lab_old[pos] : 0;
    if (__ballot(valid && g < 0) != 0ull) {
        double best = kInf;
        int bc = -1;
        if (valid && g < 0)
            for (int c = l16; c < B; c += 16) {
                const size_t slot = (size_t)c * Kcap + pos;
                if (list_cnt[slot] > 0) {
                    const double d = list_d[slot * m];
                    if (d < best) { best = d; bc = c; }
                }
            }
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) {
            const double od = __shfl_xor(best, off, 16);
            const int oc = __shfl_xor(bc, off, 16);
            if (oc >= 0 && (od < best || (od == best && (bc < 0 || oc < bc)))) { best = od; bc = oc; }
        }
        if (g < 0) g = bc;
    }
    if (valid && l16 == 0) lab_prev[pos] = g;
}

// the same kind of guess from the shortlist stage's bounds near[bin][Kcap] of the m-th nearest distance:
// the bin whose m-th nearest member is closest
__global__ __launch_bounds__(256) void guess_near_kernel(const float *near, const int *lab_old, int p0, int K, int B,
                                                         int Kcap, int *lab_prev, Gate gate)
{
    CHB_GATE(gate);
    const int l16 = threadIdx.x & 15;
    const int pos = p0 + blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool valid = pos < K;
    int g = valid ? lab_old[pos] : 0;
    if (__ballot(valid && g < 0) != 0ull) {
        float best = INFINITY;
        int bc = -1;
        if (valid && g < 0)
            for (int c = l16; c < B; c += 16) {
                const float d = near[(size_t)c * Kcap + pos];
                if (d < best) { best = d; bc = c; }
            }
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) {
            const float od = __shfl_xor(best, off, 16);
            const int oc = __shfl_xor(bc, off, 16);
            if (oc >= 0 && (od < best || (od == best && (bc < 0 || oc < bc)))) { best = od; bc = oc; }
        }
        if (g < 0) g = bc;
    }
    if (valid && l16 == 0) lab_prev[pos] = g;
}

// distance_matrix.py:47-62 on a caller-supplied distance row: m rounds of a block-wide
// lexicographic (distance, index) argmin above the previously selected key.
__global__ __launch_bounds__(256) void select_row_kernel(const int *labels, const double *row, int N,
                                                         int c, int m, int *out_idx, int *out_cnt)
{
    __shared__ double sd[256];
    __shared__ int si[256];
    double last_d = -kInf;
    int last_i = -1, cnt = 0;
    for (int r = 0; r < m; ++r) {
        double bd = kInf;
        int bi = 0x7fffffff;
        for (int p = threadIdx.x; p < N; p += 256) {
            if (labels[p] != c) continue;
            const double d = row[p];
            if (!(d == d)) continue;  // NaN never selected
            const bool above = d > last_d || (d == last_d && p > last_i);
            if (above && (d < bd || (d == bd && p < bi))) { bd = d; bi = p; }
        }
        sd[threadIdx.x] = bd; si[threadIdx.x] = bi;
        __syncthreads();
        for (int off = 128; off >= 1; off >>= 1) {
            if ((int)threadIdx.x < off) {
                const double od = sd[threadIdx.x + off];
                const int oi = si[threadIdx.x + off];
                if (od < sd[threadIdx.x] || (od == sd[threadIdx.x] && oi < si[threadIdx.x])) {
                    sd[threadIdx.x] = od; si[threadIdx.x] = oi;
                }
            }
            __syncthreads();
        }
        const double gd = sd[0];
        const int gi = si[0];
        __syncthreads();
        if (gi == 0x7fffffff) break;
        if (threadIdx.x == 0) out_idx[r] = gi;
        last_d = gd; last_i = gi; ++cnt;
    }
    if (threadIdx.x == 0) {
        for (int r = cnt; r < m; ++r) out_idx[r] = -1;
        *out_cnt = cnt;
    }
}

// ---- compaction of the (position, bin) pairs with a non-empty shortlist (update rounds).
// Two small passes instead of an atomic append: every wavefront of the shortlist kernel bumping ONE
// device-scope counter costs ~0.1 us apiece on 8 XCDs -- 1.6 ms per sweep, measured.
constexpr int kActChunk = 4096;   // pairs per block
__global__ __launch_bounds__(256) void active_count_kernel(const int *cand_cnt, int pos_begin, int npos, int B,
                                                           int Kcap, int *blk_cnt)
{
    __shared__ int red[256];
    const long long total = (long long)npos * B;
    const long long p0 = (long long)blockIdx.x * kActChunk;
    int n = 0;
    for (int i = threadIdx.x; i < kActChunk; i += 256) {
        const long long p = p0 + i;   // bin-major pair order: coalesced reads of cand_cnt
        if (p < total) {
            const int c = (int)(p / npos), q = (int)(p - (long long)c * npos);
            n += cand_cnt[(size_t)c * Kcap + pos_begin + q] > 0;
        }
    }
    red[threadIdx.x] = n;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) blk_cnt[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void active_write_kernel(const int *cand_cnt, int pos_begin, int npos, int B,
                                                           int Kcap, const int *blk_cnt, int *active, int *n_active)
{
    __shared__ int red[256];
    __shared__ int sbase;
    // my output offset = sum of the counts of the blocks before me
    int pre = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256) pre += blk_cnt[b];
    red[threadIdx.x] = pre;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) sbase = red[0];
    __syncthreads();
    int base = sbase;
    const long long total = (long long)npos * B;
    const long long p0 = (long long)blockIdx.x * kActChunk;
    for (int i0 = 0; i0 < kActChunk; i0 += 256) {
        const long long p = p0 + i0 + threadIdx.x;
        int c = 0, q = 0;
        bool on = false;
        if (p < total) {
            c = (int)(p / npos); q = (int)(p - (long long)c * npos);
            on = cand_cnt[(size_t)c * Kcap + pos_begin + q] > 0;
        }
        // ordered compaction of the 256 flags: ballots per wavefront + 4 wavefront counts
        const unsigned long long bal = __ballot(on);
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        __syncthreads();
        if (lane == 0) red[w] = __popcll(bal);
        __syncthreads();
        int off = 0;
        for (int k = 0; k < w; ++k) off += red[k];
        const int tot = red[0] + red[1] + red[2] + red[3];
        off += __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
        if (on) active[base + off] = q * B + c;
        base += tot;
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *n_active = base;
}

}  // namespace

void launch_guess(const double *list_d, const int *list_cnt, const int *lab_old, int p0, int p1,
                  int B, int m, int Kcap, int *lab_prev, hipStream_t s)
{
    if (p1 > p0)
        hipLaunchKernelGGL(guess_kernel, dim3((p1 - p0 + 15) / 16), dim3(256), 0, s, list_d, list_cnt, lab_old, p0, p1, B, m, Kcap, lab_prev);
}

void launch_guess_near(const float *near, const int *lab_old, int p0, int p1, int B, int Kcap, int *lab_prev,
                       hipStream_t s)
{
    if (p1 > p0)
        hipLaunchKernelGGL(guess_near_kernel, dim3((p1 - p0 + 15) / 16), dim3(256), 0, s, near, lab_old, p0, p1, B, Kcap, lab_prev, g_gate);
}

void launch_first_change(const int *lab_new, const int *lab_prev, int p0, int K, int *first_change,
                         hipStream_t s)
{
    if (K > p0)
        hipLaunchKernelGGL(first_change_kernel, dim3((K - p0 + 255) / 256), dim3(256), 0, s, lab_new, lab_prev, p0, K, first_change,
                           g_gate);
}

void launch_copy_i32(int *dst, const int *src, int n, hipStream_t s)
{
    if (n > 0) hipLaunchKernelGGL(copy_i32_kernel, dim3((n + 255) / 256), dim3(256), 0, s, dst, src, n, g_gate);
}

void launch_xchg_pack(int *frames, int rank, int C, const int *src, int tag, int *slot, bool with_stats, bool with_mark,
                      bool preset, int K, hipStream_t s)
{
    hipLaunchKernelGGL(xchg_pack_kernel, dim3((std::max(C, 1) + 255) / 256), dim3(256), 0, s, frames, rank, C, src, tag, slot,
                       with_stats ? 1 : 0, with_mark ? 1 : 0, preset ? 1 : 0, K, g_gate);
}

void launch_xchg_unpack(const int *frames, int world, int C, int K, int tag, int *dst, int *lab_prev, int active, int *slot,
                        bool with_stats, int *xerr, hipStream_t s)
{
    const int n = std::max(world * C, world);
    hipLaunchKernelGGL(xchg_unpack_kernel, dim3((n + 255) / 256), dim3(256), 0, s, frames, world, C, K, tag, dst, lab_prev,
                       active, slot, with_stats ? 1 : 0, xerr, g_gate);
}

void launch_compact_active(const int *cand_cnt, int pos_begin, int pos_end, int B, int Kcap, int *blk_cnt,
                           int *active, int *n_active, hipStream_t s)
{
    const long long total = (long long)(pos_end - pos_begin) * B;
    if (total <= 0) { launch_fill_i32(n_active, 0, 1, s); return; }
    const int nblk = (int)((total + kActChunk - 1) / kActChunk);
    hipLaunchKernelGGL(active_count_kernel, dim3(nblk), dim3(256), 0, s, cand_cnt, pos_begin, pos_end - pos_begin, B,
                       Kcap, blk_cnt);
    hipLaunchKernelGGL(active_write_kernel, dim3(nblk), dim3(256), 0, s, cand_cnt, pos_begin, pos_end - pos_begin, B,
                       Kcap, blk_cnt, active, n_active);
}

void launch_select_row(const int *labels, const double *row, int N, int c, int m, int *out_idx,
                       int *out_cnt, hipStream_t s)
{
    hipLaunchKernelGGL(select_row_kernel, dim3(1), dim3(256), 0, s, labels, row, N, c, m, out_idx, out_cnt);
}

#ifdef CHB_DEV_KNOBS
// developer builds (CHB_SL_VALIDATE=1): the shortlist stage's output and the index arrays behind it, checked before the
// hull kernels consume them: err = {code, bin, position, value}; 1 count out of range, 2 fewer candidates than min(m, bin
// size), 3 candidate id out of range, 4 member id out of range, 5 seating order out of range, 6 a position seated twice
// (err[4 ..]: one counter per position, zeroed by the caller)
__global__ void validate_batch_kernel(const int *cand, const int *cand_cnt, int B, int Kcap, int pos_begin, int pos_end,
                                      int cap, int N, const int *bin_ptr, const int *memb_id, const int *qord, int m,
                                      int *err, Gate gate)
{
    CHB_GATE(gate);   // (a speculated batch whose launches all returned at once has nothing to check)
    const int nq = pos_end - pos_begin;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    auto report = [&](int code, int c, int pos, int val) {
        if (atomicCAS(&err[0], 0, code) == 0) { err[1] = c; err[2] = pos; err[3] = val; }
    };
    if (idx < (long long)B * nq) {
        const int c = (int)(idx / nq), pos = pos_begin + (int)(idx % nq);
        const size_t slot = (size_t)c * Kcap + pos;
        const int cnt = cand_cnt[slot], nb = bin_ptr[c + 1] - bin_ptr[c];
        if (cnt < 0 || cnt > cap) report(1, c, pos, cnt);
        else {
            if (cnt < (m < nb ? m : nb)) report(2, c, pos, cnt);
            for (int i = 0; i < cnt; ++i) {
                const int id = cand[slot * cap + i];
                if (id < 0 || id >= N) { report(3, c, pos, id); break; }
            }
        }
    }
    if (idx < bin_ptr[B]) { const int id = memb_id[idx]; if (id < 0 || id >= N) report(4, -1, (int)idx, id); }
    if (qord != nullptr && idx < nq) {
        const int q = qord[idx];
        if (q < pos_begin || q >= pos_end) report(5, -1, (int)idx, q);
        else if (atomicAdd(&err[4 + q - pos_begin], 1) != 0) report(6, -1, (int)idx, q);   // seated twice
    }
}

// fault injection for the test of the product build's shortlist check (FusedArgs::short_cnt): the first pair of the batch
// whose bin has a member loses candidates until it holds one fewer than min(m, bin size)
__global__ void inject_short_kernel(int *cand_cnt, int B, int Kcap, int pos, const int *bin_ptr, int m, Gate gate)
{
    CHB_GATE(gate);
    for (int c = 0; c < B; ++c) {
        const int sz = bin_ptr[c + 1] - bin_ptr[c];
        if (sz > 0) { cand_cnt[(size_t)c * Kcap + pos] = min(m, sz) - 1; return; }
    }
}
void launch_inject_short(int *cand_cnt, int B, int Kcap, int pos, const int *bin_ptr, int m, hipStream_t s)
{
    hipLaunchKernelGGL(inject_short_kernel, dim3(1), dim3(1), 0, s, cand_cnt, B, Kcap, pos, bin_ptr, m, g_gate);
}

void launch_validate_batch(const int *cand, const int *cand_cnt, int B, int Kcap, int pos_begin, int pos_end, int cap, int N,
                           const int *bin_ptr, const int *memb_id, const int *qord, int m, int *err, hipStream_t s)
{
    const long long n = std::max<long long>((long long)B * (pos_end - pos_begin), N);
    hipLaunchKernelGGL(validate_batch_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, cand, cand_cnt, B, Kcap,
                       pos_begin, pos_end, cap, N, bin_ptr, memb_id, qord, m, err, g_gate);
}
#endif

void launch_fill_i32(int *p, int v, int n, hipStream_t s)
{
    if (n > 0) hipLaunchKernelGGL(fill_i32_kernel, dim3((n + 255) / 256), dim3(256), 0, s, p, v, n, g_gate);
}
void launch_batch_close(int *labels, int *inb, const int *bq, const int *lab, int K, hipStream_t s)
{
    if (K > 0) hipLaunchKernelGGL(batch_close_kernel, dim3((K + 255) / 256), dim3(256), 0, s, labels, inb, bq, lab, K);
}

void launch_bucket_base(const int *labels, int *inb, int N, int B, int *cnt, int *bin_ptr,
                        int *cursor, int *memb_id, int *pad_ptr, int *zero_me, hipStream_t s, const int *open_bq,
                        int open_K, int *open_lab_old, const SegPlan *seg, int *stats, const void *ms,
                        const float *shell_inv, int nsh)
{
    // (cnt is all zero here: allocated zeroed, and scan_kernel clears what it has read; cnt / cursor hold B * nsh keys)
    if (ms == nullptr || shell_inv == nullptr || nsh < 1) nsh = 1;
    const float4 *ms4 = reinterpret_cast<const float4 *>(ms);
    int blocks = (N + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(count_base_kernel, dim3(blocks), dim3(256), B * nsh * sizeof(int), s, labels, inb, N, B, cnt, open_bq,
                       open_K, open_lab_old, ms4, shell_inv, nsh, g_gate);
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(256), 0, s, cnt, B, nsh, bin_ptr, cursor, pad_ptr, zero_me,
                       seg ? *seg : SegPlan{}, stats, g_gate);
    hipLaunchKernelGGL(fill_base_kernel, dim3((N + 255) / 256), dim3(256), 2 * B * nsh * sizeof(int), s, labels, inb, N, B,
                       cursor, memb_id, ms4, shell_inv, nsh, g_gate);
}

void launch_bucket_batch(const int *lab_prev, const int *lab_old, const int *bq, int K, int B,
                         int *cnt, int *bin_ptr, int *cursor, int *memb_id, int *memb_code, int *pad_ptr,
                         int *first_change, int *n_slow, int *nflag, hipStream_t s, void *bb_zero)
{
    (void)cnt; (void)cursor;   // (scratch of the former three-kernel form)
    hipLaunchKernelGGL(bucket_batch_kernel, dim3(1), dim3(1024), sizeof(int) * ((size_t)B + 2048), s, lab_prev,
                       lab_old, bq, K, B, bin_ptr, pad_ptr, memb_id, memb_code, first_change, n_slow, nflag,
                       reinterpret_cast<float4 *>(bb_zero), g_gate);
}

void launch_argmin(const double *dist, const int *lab_old, int *lab_prev, int pos_begin,
                   int pos_end, int B, int *lab_new, double *mind, double *second, int *first_change, bool in_place,
                   hipStream_t s)
{
    const int n = pos_end - pos_begin;
    if (n > 0)
        hipLaunchKernelGGL(argmin_kernel, dim3((n + 31) / 32), dim3(256), 0, s, dist, lab_old, lab_prev, pos_begin, pos_end, B,
                           lab_new, mind, second, first_change, in_place ? 1 : 0, g_gate);   // 8 lanes per position
}

}  // namespace chb
