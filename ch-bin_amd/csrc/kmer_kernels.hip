// Canonical k-mer frequency vectors of a set of contigs.
//
// Replaces what the reference obtains from the external `seq2vec` tool (kmer_count.py:65-107; the
// deprecated kmer-counter path, kmer_count.py:18-62, normalises the same way at :57-59): one row per
// contig, one column per canonical k-mer (a k-mer and its reverse complement count as one), entries
// = count / total count of the contig.  k = 4 gives the 136 columns of the reference's feature
// table (config/default.ini:10).  seq2vec's source is not part of the reference tree, so two
// conventions are OURS and unpinned: columns are ordered by the smaller of the 2-bit codes
// (A < C < G < T) of a k-mer and its reverse complement; a window containing anything but
// A/C/G/T (either case) is skipped.  Neither changes a Euclidean distance between rows beyond a
// permutation of columns.
//
// One workgroup per 4096-base chunk of a contig: bytes -> 2-bit codes staged in LDS, one window
// per thread, canonical index from a 4^k-entry table, histogram in LDS (one private copy per
// wavefront), then one global atomic per non-zero counter.  HBM traffic is one byte per base; the
// kernel is bound by the LDS atomic rate.
#include "chb_internal.h"

#include <algorithm>
#include <vector>

namespace chb {
namespace {

constexpr int kKmerChunk = 4096;   // windows per workgroup

__device__ __forceinline__ int base_code(unsigned char ch)
{
    switch (ch) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return -1;
    }
}

// work item -> (contig, chunk): chunk_ptr[i] = first work item of contig i
__global__ __launch_bounds__(256) void kmer_count_kernel(const unsigned char *seq, const long long *offsets,
                                                         const int *chunk_ptr, int n_contigs, int k, int ncanon, int copies,
                                                         const unsigned short *canon, unsigned int *counts)
{
    extern __shared__ unsigned int hist[];   // [copies][ncanon] private histograms, then the staged codes
    signed char *codes = reinterpret_cast<signed char *>(hist + copies * ncanon);   // [kKmerChunk + 8]
    const int item = blockIdx.x;
    // contig of this work item: binary search in chunk_ptr
    int lo = 0, hi = n_contigs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (chunk_ptr[mid] <= item) lo = mid; else hi = mid - 1;
    }
    const int ci = lo;
    const long long s0 = offsets[ci], s1 = offsets[ci + 1];
    const long long w0 = (long long)(item - chunk_ptr[ci]) * kKmerChunk;   // first window of the chunk
    const long long nwin = s1 - s0 - k + 1;                                // windows of the contig
    for (int i = threadIdx.x; i < copies * ncanon; i += 256) hist[i] = 0u;
    const int nb = (int)min((long long)kKmerChunk + k - 1, s1 - s0 - w0);  // bases this chunk touches
    for (int i = threadIdx.x; i < nb; i += 256) codes[i] = (signed char)base_code(seq[s0 + w0 + i]);
    __syncthreads();
    unsigned int *mine = hist + ((threadIdx.x >> 6) % copies) * ncanon;
    for (int i = threadIdx.x; i < kKmerChunk && w0 + i < nwin; i += 256) {
        int code = 0;
        bool ok = true;
        for (int j = 0; j < k; ++j) {
            const int b = codes[i + j];
            ok = ok && b >= 0;
            code = (code << 2) | (b & 3);
        }
        if (ok) atomicAdd(&mine[canon[code]], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ncanon; i += 256) {
        unsigned int v = 0;
        for (int c = 0; c < copies; ++c) v += hist[c * ncanon + i];
        if (v) atomicAdd(&counts[(size_t)ci * ncanon + i], v);
    }
}

// freq[i][j] = counts[i][j] / sum_j counts[i][j]  (kmer_count.py:57-59); a contig without a single
// valid window gives a row of zeros.  One wavefront per contig.
__global__ __launch_bounds__(256) void kmer_normalise_kernel(const unsigned int *counts, int n_contigs, int ncanon,
                                                             double *freq)
{
    const int lane = threadIdx.x & 63;
    const int ci = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ci >= n_contigs) return;
    const unsigned int *row = counts + (size_t)ci * ncanon;
    unsigned long long tot = 0;
    for (int j = lane; j < ncanon; j += 64) tot += row[j];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) tot += __shfl_xor(tot, off, 64);
    const double t = (double)tot;
    for (int j = lane; j < ncanon; j += 64) freq[(size_t)ci * ncanon + j] = tot ? (double)row[j] / t : 0.0;
}

}  // namespace

int kmer_canonical_table(int k, std::vector<unsigned short> *table)
{
    if (k < 1 || k > 7) return -1;
    const int n = 1 << (2 * k);
    std::vector<int> rank((size_t)n, -1);
    int next = 0;
    for (int code = 0; code < n; ++code) {
        int rc = 0, c = code;
        for (int j = 0; j < k; ++j) { rc = (rc << 2) | (3 - (c & 3)); c >>= 2; }
        if (code <= rc) rank[(size_t)code] = next++;   // codes ascend, so canonical ranks ascend too
    }
    if (table) {
        table->resize((size_t)n);
        for (int code = 0; code < n; ++code) {
            int rc = 0, c = code;
            for (int j = 0; j < k; ++j) { rc = (rc << 2) | (3 - (c & 3)); c >>= 2; }
            (*table)[(size_t)code] = (unsigned short)rank[(size_t)std::min(code, rc)];
        }
    }
    return next;
}

void launch_kmer_count(const unsigned char *seq, const long long *offsets, const int *chunk_ptr, int n_contigs,
                       int n_items, int k, int ncanon, const unsigned short *canon, unsigned int *counts,
                       double *freq, hipStream_t s)
{
    if (n_contigs <= 0) return;
    (void)hipMemsetAsync(counts, 0, sizeof(unsigned int) * (size_t)n_contigs * ncanon, s);
    const int copies = ncanon <= 2560 ? 4 : 1;   // one private histogram per wavefront while it fits 48 KB
    const size_t lds = sizeof(unsigned int) * copies * (size_t)ncanon + kKmerChunk + 16;
    if (n_items > 0)
        hipLaunchKernelGGL(kmer_count_kernel, dim3(n_items), dim3(256), lds, s, seq, offsets, chunk_ptr, n_contigs,
                           k, ncanon, copies, canon, counts);
    hipLaunchKernelGGL(kmer_normalise_kernel, dim3((n_contigs + 3) / 4), dim3(256), 0, s, counts, n_contigs, ncanon,
                       freq);
}

int kmer_chunk_windows() { return kKmerChunk; }

}  // namespace chb
