/*
 * chbin_hip.h -- C ABI of libchbin_hip.so, the MI355X (gfx950) implementation of CH-Bin's
 * convex-hull binning hot path (ch_bin/core/clustering, AlgoDistanceMetric=convex).
 *
 * The reference (kdsuneraavinash/CH-Bin) is pure Python and has no FFI of its own; its seams are
 * plain Python functions.  Each entry point below names the reference function it replaces
 * (file:line into the reference tree).  The Python host side in ch-bin_amd/ binds these with
 * ctypes and re-exposes the reference's own signatures; INTEGRATION.md shows the stub a CH-Bin
 * maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, a negative CHB_E* code on failure; chb_last_error()
 *     returns a thread-local human-readable message.  No exceptions cross the ABI.
 *   - pointers are caller-owned HOST pointers unless the name says `_device`; C-contiguous;
 *     nothing is retained after return except by chb_set_samples_device (borrowed, see below).
 *   - labels / indices are int64 at the ABI (numpy's default, what pandas hands the reference:
 *     cli/clustering.py:52); -1 = unassigned.  Features are float64 (cli/clustering.py:53).
 *   - calls are blocking; a context is not thread-safe (the reference caller is single-threaded).
 *   - there is NO CPU fallback: without a gfx950 device every compute call fails with
 *     CHB_ENODEVICE.
 *   - environment switches read once by chb_create (none is a tuning knob, none touches an error bound: each
 *     selects a slower, independently written formulation of the same exact result, for A/B tests):
 *       CHB_PREFILTER=0     brute-force fp64 selection instead of the fp16 shortlist stage
 *       CHB_FUSED=0         list-based path (exact rescoring + hull kernel) instead of the fused kernels
 *       CHB_FUSED_PTR64=1   64-bit row pointers in the m <= 5 fused kernel (what a matrix >= 4 GiB gets)
 *       CHB_SPECULATE=0     no look-ahead across batches in chb_fit_cluster
 *       CHB_FORCE_GATHER=1  exchange path of the sharded loop even with one rank
 *       CHB_SEGMENTS=0      bins far larger than the rest are never cut into segments for the shortlist stage
 *       CHB_FUSED_STRIPE=0  position-major work order in the fused kernels (default: striped over the XCDs by bin)
 *       CHB_PACK_INCR=0     CSR and member pack of the shortlist stage rebuilt from the labels at every batch start (default:
 *                           kept across the batches of a fit and updated by each commit, where tiles are not skipped)
 *       CHB_POOL_TAU=0      the base shortlist launch always streams a bin twice (threshold sweep + admission sweep); default:
 *                           the threshold comes from a per-(bin, home bin) pool tile where one exists, and the bin is streamed once
 *                           (fits of m <= 8 neighbours, rows of up to 157 columns and at least 512 contigs per bin on average;
 *                           CHB_POOL_TAU=2: whatever the fit's size)
 *       CHB_TILE_SKIP=0     the shortlist stage never skips member tiles (default: on for fits whose first batches
 *                           show that tiles can be skipped -- data with several coverage columns)
 */
#ifndef CHBIN_HIP_H
#define CHBIN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CHB_OK 0
#define CHB_EINVAL (-1)    /* bad argument */
#define CHB_ENODEVICE (-2) /* no usable HIP device */
#define CHB_EHIP (-3)      /* a HIP runtime call failed */
#define CHB_ESTATE (-4)    /* call sequence violated (e.g. no samples set) */
#define CHB_EUNSUPPORTED (-5)

/* AlgoNumNeighbors supported (default.ini:16 -> 5, function default 15: algorithm.py:17).  The tuned kernels
 * cover 1..16; 17..64 run on plain one-wavefront-per-problem kernels (much slower, same results). */
#define CHB_MAX_NEIGHBORS 64

typedef struct chb_ctx chb_ctx;

const char *chb_last_error(void);
int chb_version(void);
/* number of visible HIP devices (0 when there is none); never fails */
int chb_device_count(void);

/* one context per process per GPU */
int chb_create(int device_id, chb_ctx **out);
int chb_destroy(chb_ctx *h);

/* hull_distance.py:90-108 calculate_distance's metric dispatch (AlgoDistanceMetric): selects what
 * every later hull-distance evaluation of this context computes (chb_fit_cluster,
 * chb_hull_distance_*).  CONVEX = distance to the convex hull ("convex", default.ini:18);
 * AFFINE = distance to the affine hull ("affine" :69-87 and "affine-qp" :38-66, the same quantity).
 * The nearest-member selection is identical for both. */
#define CHB_METRIC_CONVEX 0
#define CHB_METRIC_AFFINE 1
int chb_set_metric(chb_ctx *h, int metric);

/* Feature matrix `samples` of fit_cluster (algorithm.py:13): copied to HBM once and kept resident. */
int chb_set_samples(chb_ctx *h, const double *X, int64_t N, int64_t D);
/* same, from a device buffer (e.g. a torch tensor's data_ptr); copied device-to-device */
int chb_set_samples_device(chb_ctx *h, const double *X_device, int64_t N, int64_t D);

/* distance_matrix.py:33-44 create_in_mem_distance_matrix / :12-30 create_distance_matrix:
 * rows [row_begin,row_end) of the N x N Euclidean matrix, bit-identical to scipy cdist
 * (sqrt of the k-sequential, unfused sum of squared differences).  out: (row_end-row_begin) x N. */
int chb_pairwise_distance(chb_ctx *h, int64_t row_begin, int64_t row_end, double *out);

/* distance_matrix.py:47-62 find_nearest_from_cluster, batched over queries and ALL bins:
 * for query contig query_idx[q] and bin c, the (up to) m members of bin c nearest to the query,
 * ordered by (distance, index); the query itself is never a member (algorithm.py:50).
 * nbr_idx: Q*B*m (-1 padded), nbr_dist: Q*B*m (+inf padded, may be NULL), nbr_cnt: Q*B. */
int chb_topm_per_bin(chb_ctx *h, const int64_t *labels, int64_t B, int m, const int64_t *query_idx,
                     int64_t Q, int64_t *nbr_idx, double *nbr_dist, int32_t *nbr_cnt);

/* distance_matrix.py:47-62 find_nearest_from_cluster with the reference's exact signature: the
 * caller supplies one row of a distance matrix (any provenance) and the current labels; selects
 * among {p : labels[p] == c} the (up to) m smallest by (row[p], p).  out_idx[m] (-1 padded). */
int chb_find_nearest_from_row(chb_ctx *h, int64_t c, const int64_t *labels, const double *row,
                              int64_t N, int m, int64_t *out_idx, int32_t *out_cnt);

/* hull_distance.py:7-35 convex_hull_distance (+ solve_qp.py:96-132), batched:
 * problem p = distance from sample query_idx[p] to conv{ samples[hull_idx[p*m_max + a]] }, entries
 * < 0 are padding; an empty hull gives +inf.  alpha (P*m_max, may be NULL) receives the convex
 * weights in hull_idx order (0 at padding). */
int chb_hull_distance_batch(chb_ctx *h, const int64_t *query_idx, int64_t P, const int64_t *hull_idx,
                            int m_max, double *dist, double *alpha);

/* hull_distance.py:90-108 calculate_distance(x, mat_p, qp_solver, "convex") for explicit points:
 * x[D], pts[m][D].  Does not touch the resident samples. */
int chb_hull_distance_points(chb_ctx *h, const double *x, const double *pts, int m, int64_t D,
                             double *dist, double *alpha);

/* algorithm.py:12-76 fit_cluster(samples, num_clusters, initial_bins, distance_matrix,
 * num_neighbors, max_iterations, "convex", qp_solver): the whole reassignment loop with the
 * reference's sequential (Gauss-Seidel) semantics reproduced exactly by speculative batches.
 *   initial_bins[N]      -1 = movable (algorithm.py:38), others are fixed seeds
 *   perms[max_iter*n_move] the permutations algorithm.py:45 would draw, pre-drawn by the caller
 *                        from the legacy numpy RNG so the MT19937 stream matches ch_bin.py:22
 *   batch               speculative batch size (0 = default)
 *   labels_out[N], *iters_run, changed_per_iter[max_iter] (algorithm.py:63-68 counts); on an error return the
 *                        contents of labels_out are undefined (it may hold an earlier sweep's labels),
 *   min_dist_out[N] (may be NULL): winning hull distance of each movable contig's last visit */
int chb_fit_cluster(chb_ctx *h, int64_t B, const int64_t *initial_bins, const int64_t *perms,
                    int64_t n_move, int m, int max_iter, int batch, int64_t *labels_out,
                    int *iters_run, int64_t *changed_per_iter, double *min_dist_out);

/* Same, additionally margin_out[N] (may be NULL; needs min_dist_out; single GPU): the runner-up bin's hull
 * distance minus the winner's at each movable contig's last visit -- how far the argmin of
 * algorithm.py:57 is from flipping (+inf when no other bin has a member, NaN for seeds). */
int chb_fit_cluster_ex(chb_ctx *h, int64_t B, const int64_t *initial_bins, const int64_t *perms,
                       int64_t n_move, int m, int max_iter, int batch, int64_t *labels_out,
                       int *iters_run, int64_t *changed_per_iter, double *min_dist_out, double *margin_out);

/* ---- stepwise form of the same loop (one process per GPU; the host side exchanges labels
 * between ranks with RCCL/gloo between rounds).  Query slice [q_lo,q_hi) of each batch is the
 * part this rank evaluates; labels stay replicated on every rank. */
int chb_fit_begin(chb_ctx *h, int64_t B, const int64_t *initial_bins, int m);
/* open a batch: perm_slice[K] are the contigs visited, in order */
int chb_batch_begin(chb_ctx *h, const int64_t *perm_slice, int64_t K, int64_t q_lo, int64_t q_hi);
/* optional: starting labels for the rounds of this batch, positions [q_lo,q_hi) of guess[K] --
 * the contig's current label, or for a still unlabelled contig the bin of its nearest member
 * outside the batch.  Any start gives the same final labels; a good one saves rounds. */
int chb_batch_guess(chb_ctx *h, int64_t *guess);
/* one speculative round: lab_prev[K] in; for positions [max(active,q_lo), q_hi) writes
 * lab_new[pos] and min_dist[pos] (arrays of length K, other entries untouched) */
int chb_batch_round(chb_ctx *h, const int64_t *lab_prev, int64_t active, int64_t *lab_new,
                    double *min_dist);
/* close the batch: labels[perm_slice[i]] = final[i] */
int chb_batch_commit(chb_ctx *h, const int64_t *final_labels);
int chb_fit_labels(chb_ctx *h, int64_t *labels_out);

/* ---- multi-GPU inside chb_fit_cluster: one process (and one context) per GPU.  Rank 0 obtains a
 * 128-byte RCCL unique id, the host side broadcasts it (torch.distributed / MPI / files), every
 * rank calls chb_comm_init.  chb_fit_cluster then shards each speculative batch's positions over
 * the ranks and exchanges the label slices with RCCL all-gathers over xGMI; every rank must make
 * the same call with the same arguments and receives the same, complete result.  Before the first batch the ranks
 * all-gather {B, m, n_move, max_iter, batch size, N, D, metric, formulation switches, hashes of perms and initial_bins}:
 * a rank that was given something else makes the call fail with CHB_EINVAL on EVERY rank (nobody is left inside a
 * collective); the switches that may differ per context (CHB_SPECULATE, CHB_TILE_SKIP, CHB_PACK_INCR, the tile-skipping
 * memo of earlier fits) take their most conservative value of all ranks for that fit.  Every later exchange carries a
 * {sequence number, kind} tag and the statistics that steer the loop, so that all ranks take the same decisions; ranks
 * found out of step make the fit fail with CHB_ESTATE on every rank at the sweep's end. */
int chb_comm_unique_id(char *out128);
int chb_comm_init(chb_ctx *h, const char *id128, int rank, int world);
int chb_comm_destroy(chb_ctx *h);
/* The same sharded loop with the exchange done by the caller: fn(user, send, recv, bytes) must deliver the
 * `bytes` of every rank's `send` into recv[rank * bytes ..] on every rank (an all-gather on HOST buffers;
 * return 0 on success).  For transports other than RCCL (MPI, gloo, pipes) -- and the way two ranks can
 * share one GPU, which RCCL refuses.  Called from inside chb_fit_cluster, between device synchronisations. */
typedef int (*chb_allgather_fn)(void *user, const void *send, void *recv, size_t bytes);
int chb_comm_init_hook(chb_ctx *h, int rank, int world, chb_allgather_fn fn, void *user);
/* chb_set_samples for every rank of the communicator with ONE crossing of the host boundary: rank `root` passes its host
 * matrix X[N][D] (cli/clustering.py:53), the other ranks pass NULL; the matrix is uploaded on `root`, broadcast to the
 * other GPUs by RCCL over xGMI (109 MB at N = 100k, 1.17 GB at N = 1M) and every rank builds its own resident copy and
 * shadow rows.  N, D and root must agree on all ranks: before the broadcast the ranks exchange {own status, N, D, root}
 * (one small all-gather), and a rank that failed beforehand (no matrix on the root, an allocation) or disagrees makes the
 * call fail on EVERY rank instead of leaving the others blocked in the collective.  Needs chb_comm_init (with the hook
 * transport every rank simply calls chb_set_samples). */
int chb_bcast_samples(chb_ctx *h, const double *X, int64_t N, int64_t D, int root);
/* what the context's communicator really is: *rank / *world as given to chb_comm_init*, *comm_ranks = the rank count
 * RCCL itself reports for the communicator (ncclCommCount; 0 without an RCCL communicator), *transport = 0 none,
 * 1 RCCL, 2 host-staged hook.  Lets a benchmark line carry evidence of the exchange it ran over. */
int chb_comm_info(chb_ctx *h, int *rank, int *world, int *comm_ranks, int *transport);

/* ---- feature assembly: canonical k-mer frequency vectors.
 * Replaces the external seq2vec run of ch_bin/core/features/kmer_count.py:65-107 (and the
 * normalisation of the deprecated kmer-counter path, kmer_count.py:57-59): row i = counts of the
 * canonical k-mers (a k-mer and its reverse complement are one column) of contig i divided by
 * their sum; k = 4 gives the 136 k-mer columns of features.csv (config/default.ini:10).
 * Columns are ordered by the smaller 2-bit code (A<C<G<T) of the k-mer and its reverse complement;
 * windows containing anything but A/C/G/T (either case) are skipped; a contig without a valid
 * window yields a zero row.  seq2vec itself is not part of the reference tree: these two
 * conventions are unpinned (they permute / do not change distances between rows). */
/* number of columns for k (1 <= k <= 7), or a negative error code */
int chb_kmer_dim(int k);
/* seq: the contigs' bases back to back (host), offsets[n+1] their bounds; freq_out[n * dim] doubles;
 * counts_out (optional) [n * dim] raw counts.  Needs a context only for its device and stream. */
int chb_kmer_frequencies(chb_ctx *h, const unsigned char *seq, const int64_t *offsets, int64_t n, int k,
                         double *freq_out, uint32_t *counts_out);

/* ---- measurement: HIP-event timing of kernel launches on the context's stream.
 * on = 0 off, 1 every kernel, 2 only "prefilter" and "hull_qp" (the two that dominate a sweep: four
 * event records per batch, cheap enough to leave on inside a timed region) */
int chb_profile_enable(chb_ctx *h, int on);
int chb_profile_reset(chb_ctx *h);
/* kernel: "prefilter" | "prefilter_update" | "rescore" | "rescore_update" | "query_norms" (once per fit) |
 * "fit_start" (bin centres + every labelled sample's shadow row, once per fit) |
 * "topm_fallback" | "topm_base" | "topm_update" | "hull_qp" | "slow_path" | "argmin" | "bucket" |
 * "pool" (upkeep of the shortlist stage's threshold pools: build once per fit, open + commit per batch) |
 * "prefilter_retry" (the exact two-sweep selection for the work items a pool batch's launch left on its overflow list) |
 * "pairwise" | "kmer_count".  For m <= 16 "hull_qp" is the fused selection + hull-distance kernel and
 * "slow_path" the exact path for what it leaves over; "rescore*" then only appear for m > 16 or CHB_FUSED=0. */
int chb_profile_get(chb_ctx *h, const char *kernel, double *total_ms, int64_t *launches,
                    double *work_units);
/* counters of the last chb_fit_cluster call: [0]=batches [1]=rounds [2]=hull distances evaluated
 * (incl. speculative re-evaluation) [3]=hull distances the sequential loop needs (sweeps*n_move*B) */
int chb_fit_stats(chb_ctx *h, int64_t *out4);
/* diagnostic counters: "prefilter_enabled" (1 when the fp16 shortlist stage is active: feature vectors of up to 573
 * columns -- 157 on the narrow builds, two to four 144-column slices beyond, which covers KmerK = 5; wider samples, or the
 * environment variable CHB_PREFILTER=0, select the brute-force selection kernel instead),
 * "prefilter_overflow" (shortlists that overflowed and were recomputed by brute force since the
 * last chb_fit_begin), "fused_enabled" (1 when the fused selection + hull kernels serve the fit), "segment_batches"
 * (batches of the last fit that cut a giant bin into segments), "batch_size" (speculative batch size of the last fit),
 * "tile_skip_state" (tile skipping of the last fit: 0 undecided, 1 kept on, -1 turned off because next to nothing could be
 * skipped), "tile_skipped" / "tile_seen" / "tile_unloaded" (wave-tiles whose compute was skipped / that were met / that
 * were never loaded, as sampled by about 64 workgroups of each base shortlist launch, a batch's launch counted once),
 * "shortlist_short" ((position, bin) pairs of the last fit whose base shortlist reached the hull kernels with fewer than
 * min(num_neighbors, members of the bin) candidates or with a wild index: always 0, or chb_fit_cluster has returned
 * CHB_ESTATE at the end of that sweep -- the product build checks the shortlist stage's contract in every fused hull
 * launch), "lookahead_batches" (batches of the last fit whose successor was enqueued ahead of their convergence verdict
 * and kept: on one GPU and, since round 4, under the RCCL exchange), "lookahead_failed" (... and discarded because the
 * batch needed further rounds), "pool_batches" (batches of the last fit whose base shortlist launch took its thresholds
 * from the pools), "pool_state" (0 undecided = on, 1 kept on, -1 turned off because the shortlists came out long),
 * "pool_candidates" / "pool_pairs" (sampled shortlist lengths behind that decision), "exchanges" (framed all-gathers of the last fit under an exchange: one per batch for the
 * label guess, one per round) */
int chb_counter(chb_ctx *h, const char *name, int64_t *out);

#ifdef __cplusplus
}
#endif
#endif /* CHBIN_HIP_H */
