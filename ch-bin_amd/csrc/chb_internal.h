// Internal declarations shared by the HIP translation units of libchbin_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace chb {

constexpr int kMaxM = 16;       // CHB_MAX_NEIGHBORS
constexpr int kQTile = 64;      // queries per workgroup tile
constexpr int kPTile = 64;      // bin members per workgroup tile
constexpr int kKChunk = 8;      // feature columns staged per pipeline step
constexpr int kLdsStride = 66;  // doubles per staged k-row (64 + pad: conflict-free transposed writes)

// Per-(bin, batch position) nearest-member lists, layout [bin][Kcap][m].
struct Lists {
    double *d;   // distance, ascending by (distance, index); +inf padding
    int *idx;    // sample index; INT_MAX padding
    int *cnt;    // [bin][Kcap]
};

struct TopmArgs {
    const double *X;     // [N][Dp] row-major, zero padded to Dp % kKChunk == 0
    int Dp;
    const int *bq;       // [K] sample index of each batch position
    int pos_begin, pos_end;
    const int *bin_ptr;  // [B+1] CSR over member entries
    const int *memb_id;  // sample index of each entry
    const int *memb_code;  // nullptr: always eligible. code>0: eligible iff query pos > code-1,
                           // -(1<<30) < code < 0: eligible iff query pos < -code-1,
                           // code <= -(1<<30): eligible iff query pos != -(1<<30)-code
    int B, m, Kcap;
    Lists in;            // in.d == nullptr: start from empty lists
    Lists out;
};

void launch_topm(const TopmArgs &a, hipStream_t s);
// same, but only work items (bin, query tile of 64) whose flag is set run; the rest exit at once
void launch_topm_flagged(const TopmArgs &a, const int *flags64, hipStream_t s);

// ---- two-stage exact selection (prefilter_kernels.hip + rescore in topm_kernels.hip)
constexpr int kCandCap = 128;   // shortlist capacity per (bin, batch position)

// Low-precision shadow rows (one per CSR entry): features minus the bin's centre rounded to bf16,
// the exact squared norm of the rounded vector and the exact rounding distance rho = ||zhat - z||.
struct Shadow {
    const unsigned short *Z;  // [rows][Dz] bf16, Dz % 16 == 0, zero padded
    const float *nrm;         // [rows] ||zhat||^2 (rounded up)
    const float *rho;         // [rows] ||zhat - z|| (rounded up)
    int Dz;
};
void launch_bin_centers(const double *X, int D, int Dp, const int *memb_id, const int *bin_ptr, int B,
                        double *centers, hipStream_t s);
void launch_pack_centered(const double *X, int D, int Dp, const int *memb_id, const int *bin_ptr, int B,
                          int rows_hint, const double *centers, unsigned short *Zp, int Dz, float *nrm_p,
                          float *rho_p, hipStream_t s);
void launch_query_shadow(const double *X, int D, int Dp, const int *bq, int pos_begin, int pos_end, int B,
                         int Kcap, const double *centers, unsigned short *Zq, int Dz, void *qs,
                         hipStream_t s);
void launch_bin_bounds(const float *rho_p, const float *nrm_p, const int *bin_ptr, int B, float *rho_out,
                       float *sn_out, hipStream_t s);
void launch_sample_shadow(const double *X, int D, int Dp, const int *ids, int n, const int *labels,
                          int B, const double *centers, unsigned short *Zs, int Dz, float *nrm_s,
                          float *rho_s, hipStream_t s);
void launch_pack_rows(const Shadow &src, const int *memb_id, const int *bin_ptr, int B, int n_max,
                      unsigned short *Zp, float *nrm_p, float *rho_p, hipStream_t s);

struct PrefilterArgs {
    const unsigned short *Zq; // [B][Kcap][Dz] query shadow rows bf16(x_j - mu_c) (query_shadow_kernel)
    const float4 *qs;         // [B][Kcap] {||zh||^2, rho, ||z||^2 up, ||z||^2 down} of that row
    Shadow shm;             // member rows relative to their bin's centre, packed in CSR order
    const float *rho_bound; // [B] largest rho among each bin's packed members
    const float *sn_bound;  // [B] largest ||zh|| among each bin's packed members
    const int *bq;
    int pos_begin, pos_end;
    const int *bin_ptr;
    const int *memb_id;
    const int *memb_code;  // non-null selects the update mode: batch members, fixed tau from `seed`
    float *code_s, *code_b;  // update mode scratch [#members]: memb_code as (s, b), eligible <=> s q + b >= 0
    const float *inf_ptr;    // one float +infinity in device memory (DMA source for rows past a bin's end)
    Lists seed;
    int B, m, Kcap;
    int *cand;       // [B][Kcap][kCandCap] sample indices
    int *cand_cnt;   // [B][Kcap]
    int *active;     // update mode: compacted list of (position, bin) pairs with a non-empty shortlist
    int *n_active;   // [1]
    int *overflow;   // [1] number of (bin, position) pairs whose shortlist overflowed
};
// flags64[bin][ceil(nq/64)] (pre-zeroed): set for (query tile of 64, bin) pairs whose shortlist overflowed
void launch_prefilter(const PrefilterArgs &a, int *flags64, hipStream_t s);
size_t prefilter_lds_bytes(int Dz);

struct RescoreArgs {
    const double *X;
    int Dp;
    const int *bq;
    int pos_begin, pos_end;
    int B, m, Kcap;
    const int *cand;
    const int *cand_cnt;
    const int *active;    // non-null: only these pairs (index = (pos - pos_begin) * B + bin) ...
    const int *n_active;  // ... *n_active of them; everything else in `out` must already hold `in`
    Lists in;    // in.d == nullptr: start from empty lists
    Lists out;
};
void launch_rescore(const RescoreArgs &a, hipStream_t s);

// rows [r0,r1) of the full Euclidean distance matrix, out[(r-r0)*N + j]
void launch_pairwise(const double *X, int N, int Dp, int r0, int r1, double *out, hipStream_t s);

struct QpArgs {
    const double *X;
    int D, Dp;
    const int *bq;
    int pos_begin, pos_end;
    int B, m, Kcap;
    Lists lists;
    Lists prev;    // lists of the previous round of this batch (idx == nullptr: none): unchanged
                   // vertex lists keep their distance
    double *dist;  // [Kcap][B]
    int metric;    // 0 convex hull (hull_distance.py:7-35), 1 affine hull (hull_distance.py:38-87)
};
void launch_hull_qp(const QpArgs &a, hipStream_t s);

// explicit problems: query sample q[p], hull_idx[p][m_max] compacted, hull_cnt[p] vertices
void launch_hull_qp_indexed(const double *X, int D, int Dp, const int *q, const int *hull_idx,
                            const int *hull_cnt, int P, int m_max, int metric, double *dist,
                            double *alpha, hipStream_t s);

// label / bucket helpers
void launch_fill_i32(int *p, int v, int n, hipStream_t s);
void launch_mark_batch(int *inb, const int *bq, int K, int set, hipStream_t s);
void launch_gather_labels(const int *labels, const int *bq, int K, int *out, hipStream_t s);
void launch_scatter_labels(int *labels, const int *bq, const int *lab, int K, hipStream_t s);
// CSR of all labelled samples outside the batch
void launch_bucket_base(const int *labels, const int *inb, int N, int B, int *cnt, int *bin_ptr,
                        int *cursor, int *memb_id, hipStream_t s);
// CSR of the batch's own members: earlier positions under lab_prev, later positions under lab_old
void launch_bucket_batch(const int *lab_prev, const int *lab_old, const int *bq, int K, int B,
                         int *cnt, int *bin_ptr, int *cursor, int *memb_id, int *memb_code,
                         hipStream_t s);
// first position in [p0,K) whose label changed (atomicMin into *first_change)
void launch_first_change(const int *lab_new, const int *lab_prev, int p0, int K, int *first_change,
                         hipStream_t s);
// round-0 label guess: lab_old where >= 0, else bin of the nearest outside member
void launch_guess(const double *list_d, const int *list_cnt, const int *lab_old, int p0, int p1,
                  int B, int m, int Kcap, int *lab_prev, hipStream_t s);
// select up to m smallest (row[p], p) among labels[p] == c; one workgroup
void launch_select_row(const int *labels, const double *row, int N, int c, int m, int *out_idx,
                       int *out_cnt, hipStream_t s);
// strict-'>' argmin over bins (algorithm.py:57), first change position
void launch_argmin(const double *dist, const int *lab_old, const int *lab_prev, int pos_begin,
                   int pos_end, int B, int *lab_new, double *mind, int *first_change,
                   hipStream_t s);

}  // namespace chb
