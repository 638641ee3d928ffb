// Internal declarations shared by the HIP translation units of libchbin_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

namespace chb {

// Speculative continuation (chb_api.hip: chb_fit_cluster): the kernels of the next batch are enqueued
// before the host knows whether the current batch's round has converged; they carry a gate and return at
// once unless *flag >= need (flag = the current batch's first-changed position, need = its size).
struct Gate {
    const int *flag = nullptr;
    int need = 0;
};
extern thread_local Gate g_gate;   // what every launch_* helper passes to its kernels
#define CHB_GATE(g) do { if ((g).flag != nullptr && *(g).flag < (g).need) return; } while (0)

constexpr int kMaxM = 16;       // largest num_neighbors of the tuned kernels
constexpr int kMaxMGeneric = 64; // CHB_MAX_NEIGHBORS: beyond kMaxM the plain one-wavefront-per-problem kernels run
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
    const int *memb_id;  // sample index of each entry (negative: no member -- a hole of the persistent base pack)
    const int *bin_cnt = nullptr;   // optional [B]: entries of each bin (nullptr: bin_ptr[c + 1] - bin_ptr[c])
    const int *memb_code;  // nullptr: always eligible. code>0: eligible iff query pos > code-1,
                           // -(1<<30) < code < 0: eligible iff query pos < -code-1,
                           // code <= -(1<<30): eligible iff query pos != -(1<<30)-code
    int B, m, Kcap;
    Lists in;            // in.d == nullptr: start from empty lists
    Lists out;           // out.d == nullptr: no list output
    // optional (fused selection path): the exact top-m as a candidate list, [slot][cand_cap] / [slot]
    int *cand_out = nullptr, *cand_cnt_out = nullptr;
    int cand_cap = 0;
    float *tau_out = nullptr;    // [slot] exact m-th distance x S, rounded up (+inf: fewer than m members)
    double S = 1.0;
};

void launch_topm(const TopmArgs &a, hipStream_t s);
// the same for 16 < m <= kMaxMGeneric: one wavefront per (bin, query), no tiling (slow; see topm_kernels.hip)
void launch_topm_generic(const TopmArgs &a, hipStream_t s);
// same, but only work items (bin, query tile of 64) whose flag is set run; the rest exit at once
// same for the listed work items only (flaglist[0 .. *nflag): indices bin * ceil(nq / 64) + query tile, written by
// the shortlist kernel; flags64 de-duplicates the list and is cleared here)
void launch_topm_flagged(const TopmArgs &a, int *flags64, const int *flaglist, const int *nflag, hipStream_t s);

// ---- two-stage exact selection (prefilter_kernels.hip + rescore in topm_kernels.hip)
constexpr int kCandCap = 128;   // shortlist capacity per (bin, batch position)
constexpr int kCandCapU = 32;   // same for the batch's own entries in the fused selection path (m <= 16)

// fp16 shadow data of the shortlist stage (prefilter_kernels.hip explains the quantities).
// Members of all bins, grouped by bin, every bin padded to a multiple of 32 rows:
struct MemberPack {
    unsigned short *Z;        // [rows][Dz] fp16 bits of (x_p - mu_c) S; base pack: columns D .. D + 2 carry the three
                              // fp16 pieces of -bias / 2^15 (first piece -inf in the padding rows); padding: zero rows
    // per-row columns of the BATCH-ENTRY pack only (update mode; the base pack's tile loop reads nothing but Z):
    float *bias;              // [rows] ||zh||^2 + 2 <(mu_c - mu_g) S, zh>; +inf in the padding
    float *sn;                // [rows] ||zh|| (rounded up)
    float *cs, *cb;           // [rows] entry eligible for position q <=> cs q + cb >= 0
    float *tsn;               // [rows / 32] base pack only: largest ||zh|| (rounded up) of each 32-row tile
    int *pad_ptr;             // [B+1] first row of each bin (multiples of 32)
    float4 *bb;               // [B] {largest rho, largest ||zh||, largest amax, largest residual of the bias pieces}
                              // (the batch-entry pack accumulates the largest ||zh||^2 in .y; its .w is unused)
    const int *nt = nullptr;  // optional [B]: tiles of each bin (nullptr: (pad_ptr[c + 1] - pad_ptr[c]) / 32).  The persistent
                              // base pack keeps every bin in a region of its own with room to grow: pad_ptr[c] = the region's
                              // first row, nt[c] = the tiles in use
};
int shortlist_list_len(int m);  // entries per (query, segment) list of SegPlan::lists for num_neighbors = m
int shadow_row_elems(int D);   // Dz: 144 or 160, at least three spare columns (0: D too large for the shortlist stage)
// mu_g = column means (deterministic two-pass sum), *rmax = float bits of max |x - mu_g|
void launch_global_center(const double *X, int N, int D, int Dp, double *part, int part_blocks, double *mu_g,
                          unsigned int *rmax, hipStream_t s);
// query-side rows of all samples: Gs[N][Dz], gq[N] = float2 {||qh||^2, rho}
void launch_global_shadow(const double *X, int N, int D, int Dp, const double *mu_g, double S,
                          unsigned short *Gs, int Dz, void *gq, hipStream_t s);
void launch_bin_centers(const double *X, int D, int Dp, const int *memb_id, const int *bin_ptr, int B,
                        double *centers, hipStream_t s);
// shell_inv[c] = nsh / (1.25 x largest ||zh|| of the CSR's members of bin c) (0: empty bin): the unit of the shell key
void launch_shell_scale(const void *ms, const int *memb_id, const int *bin_ptr, int B, int nsh, float *shell_inv,
                        hipStream_t s);
// member-side row of each listed sample relative to the centre of its current bin:
// Zs[N][Dz], ms[N] = float4 {bias, rho, ||zh||^2, amax}
// (commit form: ids = the batch, new_lab[i] = final label of ids[i]: the kernel also writes it to
// labels[] and clears the batch mark inb[] -- scatter, unmark and shadow refresh in one launch)
void launch_sample_shadow(const double *X, int D, int Dp, const int *ids, int n, int *labels,
                          int B, const double *centers, const double *mu_g, double S, unsigned short *Zs,
                          int Dz, void *ms, const int *new_lab, int *inb, hipStream_t s);
// the batch's own entries (CSR + eligibility codes; P.pad_ptr from launch_bucket_batch) -> padded pack
void launch_pack_centered(const double *X, int D, int Dp, const int *memb_id, const int *memb_code,
                          const int *bin_ptr, int B, int rows_hint, const double *centers, const double *mu_g,
                          double S, int Dz, const MemberPack &P, hipStream_t s);
// Two independent pieces of a batch start in ONE launch (each too small to fill the chip on its own): the base
// members' shadow rows (CSR; P.pad_ptr from launch_bucket_base) gathered into the padded pack, and the per-bin bounds
// P.bb (shells: the CSR is in shell order -- the tile-skipping build)
void launch_pack_build(const unsigned short *Zs, const void *ms, int D, int Dz, const int *memb_id, const int *bin_ptr,
                       int B, int rows_hint, const MemberPack &P, bool shells, hipStream_t s);
// Once per fit (the bin centres are fixed for the fit): qn[N][B] = float2 {||(x_j - mu_c) S||^2 rounded up, rounded down}
// of EVERY sample against every bin centre; ckey (optional, [N]) = {N_jc bits, bin} of every sample's nearest bin centre
void launch_query_norms(const double *X, int D, int Dp, int N, int B, const double *centers, double S, void *qn,
                        unsigned long long *ckey, hipStream_t s);
// qord[0 .. pos_end - pos_begin) = the positions sorted by the bin of their sample's key (samples without a key last)
void launch_query_order(const unsigned long long *ckey, const int *bq, int pos_begin, int pos_end, int B, int *qord, int *home,
                        hipStream_t s);

// the same for all batches of a sweep in one launch: geo[b] = int4 {first permutation index of batch b, first / end position of
// this rank's slice, -}; qord_all[geo[b].x + geo[b].y ..) = the slice's positions in seating order, home_all[b * B + bin]
void launch_query_order_sweep(const unsigned long long *ckey, const int *perm, const void *geo, int nbatch, int B, int *qord_all,
                              int *home_all, hipStream_t s);

// Plan of the bins that are cut into segments for the shortlist stage (see shortlist_kernel, SEG): made on the device
// by the CSR scan of the batch start, consumed by the three shortlist launches of the batch.
struct SegPlan {
    int *nseg = nullptr;     // [1] segment items of this batch
    int4 *items = nullptr;   // [cap] {bin, first tile, end tile, giant slot << 8 | segment << 4 | (segments - 1)}
    int *gflag = nullptr;    // [B] giant slot of a segmented bin, -1 otherwise
    float *lists = nullptr;  // [giant slots][16][Kcap][ML] phase 1 -> phase 2: the m best accumulators per segment
    int cap = 0;             // capacity of items (16 per giant slot)
    int gcap = 0;            // giant slots
    bool launch = false;     // host: this batch runs the two segment launches (else no bin is marked)
};
constexpr int kSegMinTiles = 256;   // a bin is segmented if it has more tiles than this AND more than 4x the average
constexpr int kSegLenTiles = 128;   // shortest segment (a bin has at most 16)

// ---- threshold POOLS of the shortlist stage (prefilter_kernels.hip, "threshold pools"; round 5).  For every ordered pair
// (bin c, home bin h) the kPoolRows base members of c nearest to the CENTRE of h, kept as one 32-row tile of shadow rows.
// A query whose nearest centre is h takes tau(j, c) = the m-th smallest upper bound over the tile (c, h) -- any m base
// members of c bound the m-th nearest distance from above -- and the base shortlist launch then streams the bin ONCE
// (the admission sweep) instead of twice (threshold sweep + admission sweep).
constexpr int kPoolRows = 32;
constexpr int kPoolMaxHomes = 8;   // a workgroup whose queries span more home bins than this keeps the two-sweep form
struct PoolState {
    unsigned short *Z;   // [B * B * 32][Dz] tile (c, h) at rows (c * B + h) * 32; empty slots / holes: first bias piece -inf
    int *id;             // [B * B * 32] sample of each slot (-1: empty)
    float *key;          // [B * B * 32] ||(x_p - mu_h) S||^2 (rounded up) of the slot's sample (+inf: empty)
    float *sn;           // [B * B * 32] ||zh_p|| (rounded up; 0: empty)
    int *hole;           // [B * B * 32] 1: the slot's sample is in the open batch (not a base member right now)
    float *tsn;          // [B * B + 64] largest ||zh|| of the tile's slots (the tile's query-rounding term)
    int *ok;             // [B * B] usable rows of the tile while the current batch is open (written by every batch open)
};
struct ShortlistArgs {
    const unsigned short *Gs;  // [N][Dz] query-side rows
    const float2 *gq;          // [N]
    const float2 *qn;          // [N][B] (per fit: launch_query_norms)
    MemberPack P;
    int Dz;
    double S;
    const int *bq;
    int pos_begin, pos_end;
    const int *qord;           // optional: the order in which the queries are seated ([pos_end - pos_begin] positions sorted
                               // by nearest bin centre); nullptr: by position
    const int *home;           // optional (with qord): [B] the query tile where the positions nearest to each bin start
    int skip;                  // base mode: 1 = skip member tiles by the norm bound (needs shell-ordered members to pay)
    int *skip_stat;            // optional: [3] wave-tiles skipped / seen / never loaded, reported by about 64 workgroups spread over the launch
    unsigned long long *dbg;   // developer builds: per workgroup {start, end (100 MHz clock), tiles computed, hardware id}
    const int *bin_ptr;        // (unpadded) CSR the pack was built from
    const int *memb_id;
    bool update;               // update mode: the batch's own entries, fixed tau from `seed`
    Lists seed;
    int B, m, Kcap;
    int *cand;       // [B][Kcap][cand_cap] sample indices
    int *cand_cnt;   // [B][Kcap]
    int cand_cap;    // kCandCap or kCandCapU
    // base mode, optional output per (bin, position): an upper bound of the m-th smallest distance in
    // shadow units (+inf: fewer than m members)
    float *tau_out;
    const float *tau_in;   // update mode: tau per (bin, position) instead of `seed` (nullptr: use seed)
    int *overflow;   // [1] number of (bin, position) pairs whose shortlist overflowed
    int *flaglist;   // work items (query tile of 64, bin) whose shortlist overflowed, for launch_topm_flagged ...
    int *nflag;      // ... and their number (zeroed by the caller before the launch)
    SegPlan seg;         // base mode: segmented bins (seg.gflag == nullptr: none)
    // base mode, optional: threshold pools (a non-null pool.Z selects the one-sweep builds; needs qord / ckey)
    PoolState pool{};
    const unsigned long long *ckey = nullptr;   // [N] {norm bits, nearest bin centre} of every sample (launch_query_norms)
    int *pool_stat = nullptr;   // optional [2]: candidates admitted / (query, bin) pairs, reported by about 64 workgroups
    const int *worklist = nullptr;   // work-list form (launch_shortlist_worklist): the items (bin * ceil(nq / 64) + query tile of 64)
    const int *nwork = nullptr;      // ... and their number, both on the device
    int *viol = nullptr;        // developer builds (CHB_SL_BOUNDS=1): [8] first out-of-range access of the launch {code, ...} -- the
                                // access is then skipped instead of faulting
    long long viol_rows = 0, viol_pool_rows = 0, viol_members = 0;   // ... and the limits: pack rows, pool rows, CSR entries
    float gamma;         // accumulation error factor g (set by launch_shortlist)
    int tile_best_min;   // bins with at least this many tiles learn tau from per-tile bests (ditto)
    int tile_k2;         // ... or, where the top m crowd into tile halves (4 tiles < m^2), from the tile_k2 best per tile half
};
// ---- the persistent base pack (prefilter_kernels.hip, "persistent base pack"): the member pack of the shortlist stage
// kept across the batches of a fit instead of being rebuilt from the labels at every batch start
struct PackState {
    int *start;    // [B] first row of each bin's region (multiple of 32)
    int *cap;      // [B] rows of the region (multiple of 32)
    int *fill;     // [B] rows in use (members and holes); the rest of the region is padding
    int *live;     // [B] members (rows in use minus holes)
    int *nt;       // [B] tiles in use, ceil(fill / 32): written at every batch start
    int *memb;     // [arena rows] sample of each row, -1: hole / padding
    int *row;      // [N] row of each sample, -1: not in the pack
    int *ctl;      // [4] {rows of the arena handed out, overflowed appends of the commit in flight, error flag, -}
    int *ovf;      // [K] batch positions whose append found its bin's region full (served by the fix kernel)
    int *dest;     // [K] where each committed sample's row goes: 2 * row + (1: a new row of its bin), -1: nowhere / overflowed
    int arena_rows;
};
// from the compact CSR of all labelled samples (launch_bucket_base without shells): regions with room to grow, rows,
// bounds; every sample's row; the arena's fill mark
void launch_pack_state_build(const PackState &ps, const MemberPack &P, const unsigned short *Zs, const void *ms, int D, int Dz,
                             const int *memb_id, const int *bin_ptr, int B, int N, int grow, hipStream_t s);
// (grow: rows every region gets at least -- what a bin is expected to hold once the unlabelled contigs are in)
// batch start: the batch is opened (labels remembered, members marked, their rows turned into holes), the tiles per bin,
// bin-size statistics and segment plan are written: one launch
void launch_pack_state_start(const PackState &ps, const MemberPack &P, int D, int Dz, const int *labels, int *inb,
                             const int *open_bq, int open_K, int *open_lab_old, int B, const SegPlan *seg, int *stats,
                             int *zero_me, hipStream_t s);
// batch commit: launch_sample_shadow's commit form, which also puts every committed sample's row back into the pack (in
// place when its label is the one it was removed under, else appended to its new bin), then the regions that ran full are
// moved to larger ones
void launch_pack_state_commit(const PackState &ps, const MemberPack &P, const double *X, int D, int Dp, const int *ids, int n,
                              int *labels, int B, const double *centers, const double *mu_g, double S, unsigned short *Zs,
                              int Dz, void *ms, const int *new_lab, const int *lab_old, int *inb, hipStream_t s);

// ---- threshold pools: maintenance (PoolState above ShortlistArgs)
// all pools from the CSR of the labelled samples (fit start)
void launch_pool_build(const PoolState &ps, const unsigned short *Zs, const void *ms, const void *qn, int D, int Dz,
                       const int *memb_id, const int *bin_ptr, int B, hipStream_t s);
// batch open (behind the kernel that marks the batch's samples in inb): the slots of batch members become holes,
// ok[c * B + h] = usable rows
// (zero_me, optional: one int reset here -- the second-chance launch's overflow counter)
void launch_pool_open(const PoolState &ps, const int *inb, int D, int Dz, int B, int *zero_me, hipStream_t s);
// batch commit (behind the kernel that writes the final labels and refreshes the samples' shadow rows): holes whose sample
// stayed in the bin are usable again, the others are emptied; the batch's ARRIVALS of every bin (final label c, label at
// the batch's start another one) compete for the pools (c, *), replacing the slot with the largest key
// (holes: the batch may have held labelled samples -- false in a fit's first sweep, whose batches are all unlabelled)
void launch_pool_commit(const PoolState &ps, const unsigned short *Zs, const void *ms, const void *qn, int D, int Dz,
                        const int *ids, int n, const int *new_lab, const int *lab_old, const int *labels, int B, bool holes,
                        hipStream_t s);

// flags64[bin][ceil(nq/64)] (pre-zeroed): set for (query tile of 64, bin) pairs whose shortlist overflowed
// (and listed once in a.flaglist)
void launch_shortlist(const ShortlistArgs &a, int *flags64, hipStream_t s);
// the exact two-sweep selection for the listed work items only (a.worklist / a.nwork = the overflow list of a launch that
// took its thresholds from the pools); what overflows here as well goes on a.flaglist / a.nflag for the brute-force kernel
void launch_shortlist_worklist(const ShortlistArgs &a, int *flags64, hipStream_t s);

struct RescoreArgs {
    const double *X;
    int Dp;
    const int *bq;
    int pos_begin, pos_end;
    int B, m, Kcap;
    const int *cand;
    const int *cand_cnt;
    int cand_cap;         // row length of cand
    const int *cand2;     // optional second candidate source per pair (the batch's own entries) ...
    const int *cand2_cnt;
    int cand2_cap;
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
    const int *active;    // non-null: only these pairs (index = (pos - pos_begin) * B + bin), ...
    const int *n_active;  // ... *n_active of them (device side); `prev` is then ignored
};
void launch_hull_qp(const QpArgs &a, hipStream_t s);

// Fused selection + hull distance (m <= 16): per (position, bin) pair the candidates of the shortlist
// stage (base members + this round's batch entries) are gathered ONCE; their full shifted Gram gives
// both the squared distances (diagonal) that pick the m nearest and the m x m Gram of the hull QP.
// Pairs with more than the kernel's candidate capacity, or whose m-th / (m+1)-th candidates are too
// close to be ordered without cdist's exact rounding, are appended to `slow` for the exact path
// (rescore_kernel + hull_qp_kernel on that list).
struct FusedArgs {
    const double *X;
    long long n_samples;   // rows of X (the m <= 5 kernel addresses rows by 32-bit byte offsets below 4 GiB)
    int D, Dp;
    const int *bq;
    int pos_begin, pos_end;
    int B, m, Kcap;
    const int *cand;  const int *cand_cnt;    // base candidates [slot][kCandCap]
    const int *candu; const int *candu_cnt;   // this round's batch-entry candidates [slot][kCandCapU]
    const int *candp; const int *candp_cnt;   // previous round's (nullptr: none): a pair whose batch-entry
                                              // candidate set did not change keeps its distance
    double *dist;     // [Kcap][B]
    int metric;
    int *slow;        // pair indices (pos - pos_begin) * B + bin
    int *n_slow;      // zeroed by the caller
    int stripe = 0;   // set by the launcher: work striped over the XCDs by bin (see fused_pair_of)
    // the shortlist invariant, checked in the product build: every (position, bin) base shortlist holds at least
    // min(m, members of the bin outside the batch) candidates, all of them sample indices.  bin_ptr = the base CSR;
    // *short_cnt counts the pairs that violate it (the host turns a non-zero count into an error at the sweep's end)
    const int *bin_ptr = nullptr;
    const int *bin_size = nullptr;   // (instead of bin_ptr: the members of each bin, [B] -- the persistent base pack's live counts)
    int *short_cnt = nullptr;
};
// false: not supported by the fused kernels (caller uses the list-based path): m <= 16, padded rows of at most
// kFusedMaxDp doubles (the 16-lane kernel stages the query row in LDS)
constexpr int kFusedMaxDp = 288;
bool fused_supported(int m, int Dp);
void launch_hull_select_qp(const FusedArgs &a, hipStream_t s);

// explicit problems: query sample q[p], hull_idx[p][m_max] compacted, hull_cnt[p] vertices
// 16 < m <= kMaxMGeneric: one wavefront per problem, Gram and the solver's inverse in LDS (slow, see
// qp_kernels.hip); same argument meaning as launch_hull_qp / launch_hull_qp_indexed
// false: the current device does not grant the ~68 KB of dynamic LDS per workgroup these kernels need
bool hull_generic_supported();
void launch_hull_generic(const QpArgs &a, hipStream_t s);
void launch_hull_generic_indexed(const double *X, int D, int Dp, const int *q, const int *hull_idx,
                                 const int *hull_cnt, int P, int m_max, int metric, double *dist,
                                 double *alpha, hipStream_t s);
void launch_hull_qp_indexed(const double *X, int D, int Dp, const int *q, const int *hull_idx,
                            const int *hull_cnt, int P, int m_max, int metric, double *dist,
                            double *alpha, hipStream_t s);

// label / bucket helpers
void launch_fill_i32(int *p, int v, int n, hipStream_t s);
void launch_copy_i32(int *dst, const int *src, int n, hipStream_t s);   // (carries the look-ahead gate, like the fill)
#ifdef CHB_DEV_KNOBS
void launch_validate_batch(const int *cand, const int *cand_cnt, int B, int Kcap, int pos_begin, int pos_end, int cap, int N,
                           const int *bin_ptr, const int *memb_id, const int *qord, int m, int *err, hipStream_t s);
void launch_inject_short(int *cand_cnt, int B, int Kcap, int pos, const int *bin_ptr, int m, hipStream_t s);
#endif
// batch end: labels[bq[i]] = lab[i], inb[bq[i]] = -1  (batch start -- lab_old[i] = labels[bq[i]], inb[bq[i]] = i --
// rides in launch_bucket_base)
void launch_batch_close(int *labels, int *inb, const int *bq, const int *lab, int K, hipStream_t s);
// CSR of all labelled samples outside the batch
// (pad_ptr, optional: the same CSR with every bin padded to a multiple of 32 rows -> MemberPack)
// (zero_me, optional: one int the scan also resets -- the fallback list's counter of the coming shortlist launch)
// (open_bq, optional: the batch is opened in the count's launch: lab_old[i] = labels[bq[i]], inb[bq[i]] = i)
// (seg + stats, optional: the scan also makes the batch's segment plan -- only if seg->launch -- and leaves
//  stats[0] = tiles of the largest bin, stats[1] = tiles of all bins, for the host's decision about the NEXT batches)
void launch_bucket_base(const int *labels, int *inb, int N, int B, int *cnt, int *bin_ptr,
                        int *cursor, int *memb_id, int *pad_ptr, int *zero_me, hipStream_t s,
                        const int *open_bq = nullptr, int open_K = 0, int *open_lab_old = nullptr,
                        const SegPlan *seg = nullptr, int *stats = nullptr, const void *ms = nullptr,
                        const float *shell_inv = nullptr, int nsh = 1);
// (stats: [4] ints -- tiles of the largest bin, tiles of all bins, and two counters zeroed here for the shortlist launch)
// (ms + shell_inv + nsh > 1, optional: the members of a bin are grouped by SHELLS of their distance from the bin's centre,
//  outermost first -- cnt / cursor then hold B * nsh entries; see member_key in aux_kernels.hip)
constexpr int kShells = 32;       // shells per bin (fewer when B * kShells would exceed kMaxKeys)
constexpr int kMaxKeys = 8192;    // CSR keys the count / fill kernels keep in LDS
// CSR of the batch's own members: earlier positions under lab_prev, later positions under lab_old
// (also starts the round's scalars: *first_change = K, *n_slow = 0, *nflag = 0 where the pointers are non-null)
void launch_bucket_batch(const int *lab_prev, const int *lab_old, const int *bq, int K, int B,
                         int *cnt, int *bin_ptr, int *cursor, int *memb_id, int *memb_code, int *pad_ptr,
                         int *first_change, int *n_slow, int *nflag, hipStream_t s, void *bb_zero = nullptr);
// (bb_zero, optional: float4[B] per-bin bounds of the batch-entry pack, reset here for launch_pack_centered)
// first position in [p0,K) whose label changed (atomicMin into *first_change)
void launch_first_change(const int *lab_new, const int *lab_prev, int p0, int K, int *first_change,
                         hipStream_t s);
// round-0 label guess: lab_old where >= 0, else bin of the nearest outside member
void launch_guess(const double *list_d, const int *list_cnt, const int *lab_old, int p0, int p1,
                  int B, int m, int Kcap, int *lab_prev, hipStream_t s);
// same from the shortlist stage's bounds tau[bin][Kcap] of the m-th nearest distance (+inf: fewer than m
// members): the bin with the smallest one
void launch_guess_near(const float *tau, const int *lab_old, int p0, int p1, int B, int Kcap, int *lab_prev,
                       hipStream_t s);
// ---- framed exchange of the sharded loop (aux_kernels.hip): a rank's slice of a round's all-gather = kXchgHdr header
// words {tag, skipped, seen, unloaded, arena mark, 0, 0, 0} + the C labels of its positions
constexpr int kXchgHdr = 8;
// frames[rank] <- header + src[rank * C .. + C); slot = the batch's verdict slot (statistics in slot[3..6]); preset: slot[0] = K
void launch_xchg_pack(int *frames, int rank, int C, const int *src, int tag, int *slot, bool with_stats, bool with_mark,
                      bool preset, int K, hipStream_t s);
// dst[pos] <- every frame's labels (pos < K); lab_prev != nullptr: round form -- positions >= active compared with lab_prev
// (first change -> slot[0]) and written to it; tags checked against `tag` (xerr[0..3] = {1, mine, theirs, rank} on the first
// mismatch); with_stats: slot[3..5] = sums over the ranks' headers, slot[6] = their largest mark
void launch_xchg_unpack(const int *frames, int world, int C, int K, int tag, int *dst, int *lab_prev, int active, int *slot,
                        bool with_stats, int *xerr, hipStream_t s);
// active[] = the pairs (pos - pos_begin) * B + bin whose cand_cnt is positive, *n_active their number
// (blk_cnt: scratch, one int per 4096 pairs)
void launch_compact_active(const int *cand_cnt, int pos_begin, int pos_end, int B, int Kcap, int *blk_cnt,
                           int *active, int *n_active, hipStream_t s);
// select up to m smallest (row[p], p) among labels[p] == c; one workgroup
void launch_select_row(const int *labels, const double *row, int N, int c, int m, int *out_idx,
                       int *out_cnt, hipStream_t s);
// strict-'>' argmin over bins (algorithm.py:57), first change position
// (in_place: also lab_prev[pos] = lab_new[pos], after the comparison)
// (second, optional: the smallest hull distance among the OTHER bins, i.e. the runner-up)
void launch_argmin(const double *dist, const int *lab_old, int *lab_prev, int pos_begin,
                   int pos_end, int B, int *lab_new, double *mind, double *second, int *first_change, bool in_place,
                   hipStream_t s);


// ---- canonical k-mer frequency vectors (kmer_kernels.hip)
// number of canonical k-mers (136 for k = 4); table[code] = column of the 2-bit k-mer code; -1: k unsupported
int kmer_canonical_table(int k, std::vector<unsigned short> *table);
int kmer_chunk_windows();
// counts[n][ncanon] (zeroed here) and freq[n][ncanon] = counts / row sum; chunk_ptr[i] = first work
// item of contig i (one item per kmer_chunk_windows() windows), n_items = their total
void launch_kmer_count(const unsigned char *seq, const long long *offsets, const int *chunk_ptr, int n_contigs,
                       int n_items, int k, int ncanon, const unsigned short *canon, unsigned int *counts,
                       double *freq, hipStream_t s);

}  // namespace chb
