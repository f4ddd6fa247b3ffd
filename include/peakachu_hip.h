/*
 * peakachu_hip.h -- C ABI of the MI355X-native Peakachu scoring hot path.
 *
 * The reference (tariks/peakachu v2.3) is pure Python and has no FFI of its
 * own; this header is the boundary a maintainer would bind with ctypes from
 * peakachu/scoreUtils.py.  Every entry point names the
 * reference code it replaces (paths relative to the reference repository).
 *
 * Conventions
 *  - plain C types only; no C++ / torch types cross this boundary.
 *  - functions returning int: 0 = success, negative = error (PK_E_*); the
 *    message is available from pk_last_error() (thread-local).
 *  - functions returning a handle: NULL = error, message in pk_last_error().
 *  - host pointers are BORROWED for the duration of the call; outputs are
 *    caller-allocated; device memory lives in the opaque handles.
 *  - threading: one HIP stream pair and one lock PER DEVICE.  Every entry
 *    point takes the lock of the device its handle lives on, so calls on one
 *    device are serialised by the library and calls on different devices run
 *    side by side (one thread per device in one process works; the
 *    deployment peakachu_amd/dist.py sets up is still one process per GPU).
 *    Process-wide state -- the option DEFAULTS, the Gaussian taps, the
 *    kernel timers -- has its own small locks; errors are thread-local.
 *  - there is no CPU fallback: without a gfx950 device every compute call
 *    fails with PK_E_NODEVICE.
 */
#ifndef PEAKACHU_HIP_H
#define PEAKACHU_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PK_ABI_VERSION 1

#define PK_OK 0
#define PK_E_INVALID (-1)   /* bad argument */
#define PK_E_NODEVICE (-2)  /* no usable HIP device */
#define PK_E_HIP (-3)       /* HIP runtime error */
#define PK_E_NOMEM (-4)     /* allocation failure */
#define PK_E_UNSUPPORTED (-5)
#define PK_E_COMM (-6)      /* RCCL error */

typedef struct pk_forest pk_forest;
typedef struct pk_matrix pk_matrix;
typedef struct pk_cands pk_cands;
typedef struct pk_comm pk_comm;
typedef struct pk_csr pk_csr;

/* ---- library / device --------------------------------------------------- */
int pk_abi_version(void);
const char *pk_last_error(void);
/* number of visible HIP devices; 0 when there is none (never negative) */
int pk_device_count(void);
int pk_device_name(int device, char *buf, int buflen);
int pk_device_synchronize(int device);
/* which HIP this process really bound: hipRuntimeGetVersion / hipDriverGetVersion (no device
 * needed; -1 where a query fails).  A process that had mapped another ROCm before loading this
 * library -- a framework's bundled copy under the same sonames -- reports THAT copy here; bench.py
 * prints both next to the paths found in /proc/self/maps. */
int pk_runtime_versions(int *hip_runtime, int *hip_driver);

/* ---- forest: the model object used at peakachu/scoreUtils.py:109 --------
 * (model.predict_proba; sklearn RandomForestClassifier trained at
 * peakachu/trainUtils.py:46-63).  Arrays are sklearn's per-tree node arrays
 * laid end to end: tree t owns nodes [tree_off[t], tree_off[t+1]); left/right
 * are child indices relative to tree_off[t] (-1 at a leaf); feat in [0,F);
 * thr = split threshold (x <= thr goes left); miss_left = where NaN goes;
 * p1 = value[node,0,1], the class-1 fraction returned at a leaf. */
pk_forest *pk_forest_create(int device, int T, int F, const int32_t *tree_off,
                            const int32_t *left, const int32_t *right,
                            const int32_t *feat, const double *thr,
                            const uint8_t *miss_left, const double *p1);
void pk_forest_destroy(pk_forest *);
int pk_forest_info(const pk_forest *, int *T, int *F, int64_t *n_nodes, int *max_depth);

/* ---- matrix: Chromosome.M / exp_arr (peakachu/scoreUtils.py:16-33) -------
 * Canonical CSR (sorted column indices, no duplicates) of the already
 * band-filtered contact matrix plus the expected-by-distance vector.  The
 * library re-lays it out in HBM as a diagonal-major dense band covering
 * col-row in [dlo, dhi]; CSR entries outside that range are ignored (they
 * read as 0, like absent cells at peakachu/scoreUtils.py:81). */
pk_matrix *pk_matrix_create(int device, int32_t n, const int32_t *indptr,
                            const int32_t *indices, const double *data,
                            const double *exp_arr, int32_t exp_len,
                            int32_t dlo, int32_t dhi);
void pk_matrix_destroy(pk_matrix *);

/* ---- the per-chromosome preparation of Chromosome.__init__ on the device ----------
 * (peakachu/scoreUtils.py:13-38, peakachu/utils.py:139-170): the contact matrix is
 * uploaded ONCE as the canonical CSR the driver holds; the band filter, the validity
 * flags and the diagonal means of calculate_expected and the facts get_candidate needs
 * about the counts are computed from that copy on the device.
 * pk_csr_info: info[0] finite non-zero entries, [1] non-finite entries, [2] finite entries
 * that are not non-negative integers, [3] finite negative entries; *vmax largest finite value. */
pk_csr *pk_csr_upload(int device, int32_t n, const int32_t *indptr, const int32_t *indices,
                      const double *data);
/* The same from a chromosome AS A CONTACT-MAP FILE STORES IT: the upper triangle, pixels sorted
 * by (bin1, bin2) -- a .cool's pixel table, which the reference has cooler mirror and balance on
 * the host first (peakachu/score_genome.py:55-57: matrix(balance=name, sparse=True).fetch, then
 * utils.tocsr, peakachu/utils.py:10-15).  indptr[n+1]: first pixel of each bin1 (relative to the
 * chromosome's first pixel), cols: bin2 relative to the chromosome's first bin (entries with
 * cols >= n are pixels of other chromosomes and are ignored), counts: int32, or float64 when
 * counts_are_f64.  The pk_csr stands for the MIRRORED matrix; with bias != NULL its values are
 * (bias[row] * bias[col]) * count -- cooler's api.matrix multiplies the two biases first -- NaN
 * where a bias is NaN.  A table that is not in that order (a column left of the diagonal, columns
 * not strictly ascending inside a row) is refused with PK_E_INVALID's message. */
pk_csr *pk_csr_upload_upper(int device, int32_t n, const int32_t *indptr, const int32_t *cols,
                            const void *counts, int counts_are_f64, const double *bias);
/* another view of the same stored entries with other biases (NULL: the plain values): nothing is
 * uploaded again.  score_genome's balanced mode needs two matrices per chromosome, the balanced
 * one for the windows and the raw one for the Poisson candidates (peakachu/score_genome.py:55-58). */
pk_csr *pk_csr_view(pk_csr *, const double *bias);
void pk_csr_destroy(pk_csr *);
int pk_csr_info(const pk_csr *, int64_t info[4], double *vmax);
/* band of col-row in [dlo, dhi] holding the finite non-zero entries (keep_nan = 0: the
 * filter of peakachu/scoreUtils.py:30-33) or every non-zero entry, NaN included
 * (keep_nan = 1: what utils.calculate_expected keeps in balanced mode, utils.py:156).
 * The expected curve is attached later with pk_matrix_set_expected. */
pk_matrix *pk_matrix_from_csr(pk_csr *, int32_t dlo, int32_t dhi, int keep_nan);
int pk_matrix_set_expected(pk_matrix *, const double *exp_arr, int32_t exp_len);
/* means of diagonals first..top of `band` over the valid bins, as pk_expected_means, with
 * the validity flags taken from the uploaded matrix: mode 0 (raw, utils.py:145-148) a bin is
 * valid when its column holds a positive finite entry (= column sum > 0 for non-negative
 * counts; PK_E_UNSUPPORTED when negative entries exist), mode 1 (utils.py:150-155) when it
 * is the row or column of a finite non-zero entry.  means[i - first]. */
int pk_csr_expected_means(pk_csr *, pk_matrix *band, int first, int top, int mode, double *means);

/* ---- Chromosome.getwindow (peakachu/scoreUtils.py:70-93) and the body of
 * trainUtils.buildmatrix (peakachu/trainUtils.py:31-42): gather the
 * (2w+1)^2 window, distance_normalize (peakachu/utils.py:211-237), gaussian
 * blur (sigma=1), image_normalize (peakachu/utils.py:204-209), ravel.
 * keep[i] = index into the input of survivor i (input order).
 * Coordinates are whatever the caller has, as in the reference: those with x-w < 0 or
 * y+w+1 > n are dropped; a lower-triangle coordinate (x > y) that passes is served from
 * the stored diagonals (cells with col-row <= -2w read 0, scoreUtils.py:30-33; a column
 * y-w+j < 0 is column n + (y-w+j), scipy's negative index); where the reference's gather
 * raises IndexError (row x+w >= n, column y-w < -n) the call fails with PK_E_INVALID.
 * fea64 / fea32: [n_keep, F] row-major, either may be NULL.  A feature that is
 * NaN in the reference is NaN here; the sign bit / payload of a NaN is not
 * reproduced (IEEE 754 leaves it open and x86 and gfx950 differ). */
int pk_extract(pk_matrix *, int w, int64_t N, const int32_t *x, const int32_t *y,
               double *fea64, float *fea32, int64_t *keep, int64_t *n_keep);

/* ---- model.predict_proba(fea)[:, 1] (peakachu/scoreUtils.py:109) -------- */
int pk_predict(pk_forest *, int64_t N, const float *fea32, double *p1);

/* ---- Chromosome.score (peakachu/scoreUtils.py:95-125), device-resident ---
 * pk_cands holds the candidate list (Chromosome.ridx/cidx,
 * peakachu/scoreUtils.py:68) and all per-candidate outputs in HBM. */
pk_cands *pk_cands_create(int device, int64_t N, const int32_t *x, const int32_t *y);
void pk_cands_destroy(pk_cands *);
/* ALLOW exact early termination for the runs of THIS candidate list (what Chromosome.score
 * needs: only the pixels with p > thre, peakachu/scoreUtils.py:110-113): a candidate may stop
 * walking the forest once its sum can no longer exceed thre * T (every remaining tree adds at most
 * 1.0).  The scored pixels are identical; pk_score_fetch_all reports 0 for a candidate that was
 * stopped.  The library uses the permission where it pays:
 *   - launches of at least 2^18 candidates on the rank kernels (forest_qr_kernel, forest_q_kernel,
 *     forest_q2_kernel) are CUT IN TWO at a tree-group boundary -- the head walks the groups in front
 *     of the cut over every candidate and parks the ones still open (partial sum, rank codes), the
 *     tail walks the rest over the parked ones only, sums continued in tree order.  From the default
 *     threshold 0.5 on (config 2: the forest's time x 0.75 at 0.5, x 0.32 at 0.9); the cut moves
 *     later, or is given up, when the calls show that too many candidates stay open (forest options
 *     forest_split, forest_split_at, forest_split_frac, forest_split_min; read-only
 *     stat_split_group / _trees / _parked / _shift);
 *   - elsewhere the in-kernel exit (a whole 256-candidate tile stops) for thre >= 0.55 on lists of
 *     >= 2^19 candidates; option "early_exit" = 1 on the list forces it.
 * pk_score (host buffers, scored pixels only) always gives the permission.
 * Off by default on a pk_cands (every candidate gets its full probability). */
int pk_cands_set_prune(pk_cands *, int on);
/* extract -> predict -> (p > thre) -> compact, with the reference's batch
 * rule (a batch of `batch` candidates with fewer than two surviving windows
 * contributes nothing; peakachu/scoreUtils.py:104-108).  Results stay on the
 * device, in candidate order.  Returns after the stream has drained. */
int pk_score_run(pk_matrix *, pk_forest *, pk_cands *, int w, double thre,
                 int64_t batch, int64_t *n_out);
/* copy the n_out scored pixels of the last pk_score_run to the host:
 * (row, col, probability, signal = M[row, col]; peakachu/scoreUtils.py:118-121) */
int pk_score_fetch(pk_cands *, int32_t *ox, int32_t *oy, double *op, double *osignal);
/* per-candidate view of the last run (tests, diagnostics): status[i] = 1 if
 * the window survived the filters, prob[i] = its probability (else 0) */
int pk_score_fetch_all(pk_cands *, uint8_t *status, double *prob);
/* host-buffer convenience: create + run + fetch + destroy */
int pk_score(pk_matrix *, pk_forest *, int w, double thre, int64_t batch,
             int64_t N, const int32_t *x, const int32_t *y, int32_t *ox,
             int32_t *oy, double *op, double *osignal, int64_t *n_out);

/* ---- the diagonal means of utils.calculate_expected (peakachu/utils.py:157-170) ---
 * means[i] = mean of M.diagonal(i) over the bins with valid[r] && valid[r+i] when
 * more than 10 such pixels exist, else 0, for i = 0..top; summed in numpy's own
 * order (8192-element buffers, pairwise inside) so the values are bit-identical to
 * ndarray.mean().  `band0` is a pk_matrix built with dlo = 0 and dhi >= top from the
 * entries calculate_expected keeps (finite ones in raw mode).  The isotonic fit that
 * follows stays on the host (sklearn). */
int pk_expected_means(pk_matrix *band0, int top, const uint8_t *valid, double *means);

/* ---- Chromosome.get_candidate (peakachu/scoreUtils.py:40-68) ------------------
 * Scan the raw-count band for pixels on diagonals lower..upper whose Poisson
 * survival p-value against the expected count is < 0.01 and build the candidate
 * list on the device, in the reference's order (diagonal ascending, row
 * ascending).  The p-value test itself is delegated to tables the host makes
 * with scipy, so the decision is scipy's:
 *   raw mode (weights == NULL):  candidate  <=>  count > 0 and floor(count) >= kstar[d]
 *   balanced mode:               mu = bg[d] / (w[r] * w[r+d]);
 *                                candidate  <=>  mu < mustar[floor(count)]
 *     pixels with |mu / mustar - 1| < 1e-9 are not decided here: they are counted
 *     in *n_ambiguous and the caller must fall back to scipy for this chromosome.
 * raw: a pk_matrix holding the RAW counts with dlo <= lower, dhi >= upper.
 * Returns a pk_cands (NULL on error); *n_cand = its length. */
pk_cands *pk_candidates_create(pk_matrix *raw, int lower, int upper, const int64_t *kstar,
                               const double *bg, const double *weights, const double *mustar,
                               int64_t n_mustar, int64_t *n_cand, int64_t *n_ambiguous);
/* copy the candidate coordinates of a pk_cands back to the host (N entries each) */
int pk_cands_fetch(pk_cands *, int32_t *x, int32_t *y);

/* ---- the Gaussian taps ----------------------------------------------------- */
/* The window blur of getwindow is scipy.ndimage.gaussian_filter(sigma=1) (peakachu/utils.py:
 * 211-237), whose 9 taps are exp(-x*x/2)/sum evaluated by the caller's numpy; two of them
 * differ in the last bit between numpy releases (1.26 vs 2.2).  taps5 = centre tap and the
 * four on one side, as the HOST's numpy computes them: the kernels then reproduce the
 * reference of that very environment.  Default: the values numpy 2.2 / scipy 1.15 give.
 * Applies to every device, now and later; must be finite, positive and decreasing. */
int pk_set_gauss_taps(const double *taps5);
int pk_get_gauss_taps(double *taps5);

/* ---- tuning / measurement ------------------------------------------------ */
/* named integer knobs ("chunk", "forest_ilp", "forest_lds", ...); PK_E_INVALID for an unknown
 * name or a value out of range.  Every handle carries its OWN set: pk_set_option sets the defaults
 * that handles created afterwards start from (existing handles are not touched);
 * pk_<handle>_set_option changes one handle's copy -- forest: which kernel family walks it and
 * how (forest_*); matrix: which extractor runs (extract_*) and, for the calls that have no
 * candidate handle (pk_score, pk_extract), the pipeline options; candidates: the pipeline
 * options of pk_score_run (chunk, overlap, sub_chunk, early_exit).  Results never depend on
 * them; the route does.  (The reference has no such knobs: one code path, peakachu/scoreUtils.py:95-125.)
 * Read-only names report what ran: "stat_extract_clean" / "stat_extract_general" (process: launches of either
 * extractor), "stat_extract_strip" (launches of the clean extractor's LDS-staged variant, option extract_strip),
 * and per forest, once its first call has planned it, "stat_q_mode" (rank image: 0 narrow word, 1 wide,
 * 2 the 12-bit rank word; -1 no rank image), "stat_q_rows" (rows of a rank tile), "stat_q_shape" (64-candidate blocks
 * per workgroup), "stat_q_trees" (trees of the image: the model's, or more when trees were cut into pieces). */
int pk_set_option(const char *name, int64_t value);
int64_t pk_get_option(const char *name);
int pk_forest_set_option(pk_forest *, const char *name, int64_t value);
int64_t pk_forest_get_option(pk_forest *, const char *name);
int pk_matrix_set_option(pk_matrix *, const char *name, int64_t value);
int64_t pk_matrix_get_option(pk_matrix *, const char *name);
int pk_cands_set_option(pk_cands *, const char *name, int64_t value);
int64_t pk_cands_get_option(pk_cands *, const char *name);
/* HIP-event timing of the library's own kernels on its own stream */
int pk_prof_enable(int on);
int pk_prof_reset(void);
/* accumulated device time (ms) and launch count of kernel class `name`
 * ("extract", "forest", "compact", "band") since the last reset */
int pk_prof_get(const char *name, double *ms_total, int64_t *launches);
/* diagnostic, needs no device: 1 if a thread can take the lock of device_b while the caller holds
 * the lock of device_a (different devices do not wait for each other), 0 if it cannot (same device:
 * calls are serialised), PK_E_INVALID for indices outside 0..63 */
int pk_debug_lock_probe(int device_a, int device_b);
/* diagnostic builds only (option "forest_dbg" bit 4): in-kernel cycle stamps of
 * one workgroup, written to a buffer no kernel reads; n <= 65536 entries */
int pk_debug_read(int device, int64_t *out, int64_t n);
/* diagnostic, needs no device: the "LDS image" the forest kernel walks for
 * model.predict_proba (peakachu/scoreUtils.py:109), built for `slots` tree slots,
 * so that tests can walk it on the CPU.  Forest arrays as in pk_forest_create.
 * layout8 = {half-tile bytes, region A bytes, region B offset, value area offset,
 * flag area offset, image capacity, slots, F}; words = 8-byte words of all groups;
 * gtab = 4 ints per group (first tree, trees, offset and size in 16-byte units);
 * troot / tdepth = per tree the word a walk starts from and its number of levels. */
int pk_debug_forest_image(int T, int F, const int32_t *tree_off, const int32_t *left,
                          const int32_t *right, const int32_t *feat, const double *thr,
                          const uint8_t *miss_left, const double *p1, int slots,
                          int32_t *layout8, int64_t cap_words, uint64_t *words,
                          int64_t *n_words, int64_t cap_groups, int32_t *gtab,
                          int32_t *n_groups, uint64_t *troot, int32_t *tdepth);

/* ---- host-side helper of the contact-map reader (no device involved) ------------------
 * The reference reads a .cool through cooler -> h5py -> libhdf5, whose C filter pipeline inflates
 * and un-shuffles the chunks of the pixel table (peakachu/score_genome.py:55-57).  The package's
 * own reader parses the container in Python and hands the chunk pipeline of a ranged read to this
 * call: chunk i = src[i] (src_len[i] bytes as stored); inflate to chunk_bytes when `deflate`
 * (zlib, looked up at run time); un-shuffle with element size shuffle_es when > 1; bytes
 * [skip[i], skip[i] + take[i]) of the result go to dst[i].  Chunks run side by side on `threads`
 * host threads.  dst[i] needs NO alignment (a slice may start anywhere in the caller's array);
 * PK_E_INVALID for a slice outside the chunk, a missing pointer, a stored chunk shorter than
 * chunk_bytes (no deflate) or one that does not inflate to exactly chunk_bytes (truncated or
 * corrupt stream: nothing is written beyond dst[i] + take[i] in any case); PK_E_UNSUPPORTED when
 * no zlib can be found (the caller then inflates itself). */
int pk_host_unfilter_chunks(int n_chunks, const void *const *src, const int64_t *src_len, int deflate,
                            int shuffle_es, int64_t chunk_bytes, const int64_t *skip, const int64_t *take,
                            void *const *dst, int threads);

/* ---- multi-GPU: one process per GPU, one gather of the scored pixels -----
 * Chromosomes / candidate blocks are scored independently per rank
 * (peakachu/score_genome.py:46-84 shares nothing between iterations); the
 * only exchange is a gather-v of (row, col, prob, signal) to rank 0 over
 * RCCL.  The 128-byte unique id is created on rank 0 and handed to the other
 * ranks by the host's own rendezvous. */
int pk_comm_unique_id(uint8_t id[128]);
pk_comm *pk_comm_create(int device, int nranks, int rank, const uint8_t id[128]);
void pk_comm_destroy(pk_comm *);
/* ranks RCCL itself counts in the communicator (ncclCommCount), or PK_E_COMM */
int pk_comm_ranks(pk_comm *);
/* ncclGetVersion of the RCCL this process bound (no device, no communicator needed) */
int pk_comm_version(int *rccl);
/* No gather waits without bound: a rank polls its stream for PK_COMM_TIMEOUT seconds (environment,
 * default 120); when a peer went away between its "ready" and its send the wait runs out, the
 * communicator is aborted (ncclCommAbort), the call returns PK_E_COMM and so does every later call
 * on that communicator -- the process should report and exit non-zero (a retry is a fresh process). */
/* every rank passes its pk_cands after pk_score_run; on rank 0 `counts`
 * (nranks entries) and the concatenated outputs (capacity `cap` pixels) are
 * filled in rank order; other ranks may pass NULL outputs. */
int pk_comm_gather_scored(pk_comm *, pk_cands *, int64_t *counts, int64_t cap,
                          int32_t *ox, int32_t *oy, double *op, double *osignal);

/* generic gather-v of host bytes to rank 0 (score_genome: every rank sends the
 * packed records of the chromosomes it scored; peakachu/score_genome.py:83-84
 * appends them to one file).  counts (nranks entries, bytes) is filled on
 * every rank; recv (capacity cap bytes) only on rank 0, in rank order. */
int pk_comm_gatherv_bytes(pk_comm *, const void *send, int64_t nbytes, int64_t *counts,
                          void *recv, int64_t cap);

/* diagnostic, needs no device: how the library classifies a candidate list when it is made (the route of the
 * extractor, never a result): bit 0 = consecutive candidates are rarely neighbours on a diagonal (get_candidate's
 * lists: loads in the order of the window's diagonals), bit 1 = the batches of 32 consecutive candidates are runs
 * on one diagonal ("every non-zero pixel of the band": the wave's strip of the band is staged in LDS). */
int pk_debug_classify_coords(int64_t N, const int32_t *x, const int32_t *y);

/* diagnostic, needs no device: the bound the forest kernels' early exit / cut tests `acc + remaining`
 * against for threshold `thre`, a forest of T trees and at most `additions` terms still to add --
 * thre * T less a proven rounding margin (csrc/pk_common.h: pk_prune_bound; tests/test_prune_bound.py). */
double pk_debug_prune_bound(double thre, int T, int64_t additions);

/* diagnostic, needs no device: the rank tables and the RANK image (4-byte nodes over
 * 16-bit rank codes) the default forest kernel walks, for `slots` tree slots and `ch`
 * walks per lane, so that tests can quantize and walk on the CPU.
 * `ch` | 0x100 asks for fixed tree slots (the early-staging mode of the kernel).
 * layout8 (32 ints) = {half-tile bytes, walks per lane | second tile's offset << 8, flag
 * area offset, value area offset, image offset, image capacity, slots, F, total bytes of the
 * fixed tree slots or 0, then 17 slot offsets, then [26] = R, the ROWS of a rank tile: the F
 * features plus one virtual feature per further 2 047 distinct thresholds of a feature}; qsrc = per
 * row the float feature it is quantized from (rows 0..F-1: themselves); qoff = R+1 offsets into
 * qthr (per row its sorted distinct float32 thresholds); qlut = [R][4096] lookup cells (thresholds
 * in lower cells | thresholds in the cell << 16), qpar = [R][2] (lower end, cells per unit) -- the
 * caller's qoff / qlut / qpar / qsrc hold cap_rows rows; pairs = the trees' 8-byte child pairs;
 * gtab = 4 ints per group (first tree, trees, offset and size in 16-byte units); ttab = 4 ints per
 * tree (byte offset inside its group, levels to walk | wide word: the tree's split << 16, root word,
 * 16-byte units). */
int pk_debug_forest_qimage(int T, int F, const int32_t *tree_off, const int32_t *left,
                           const int32_t *right, const int32_t *feat, const double *thr,
                           const uint8_t *miss_left, const double *p1, int slots, int ch,
                           int32_t *layout8, int32_t *qoff, int64_t cap_thr, float *qthr,
                           uint32_t *qlut, float *qpar, int64_t cap_pairs, uint64_t *pairs,
                           int64_t *n_pairs, int64_t cap_groups, int32_t *gtab,
                           int32_t *n_groups, int32_t *ttab, int32_t cap_rows, int32_t *qsrc);

/* diagnostic, needs no device: where the library would cut a forest in two (pk_cands_set_prune above) and
 * what it learns from its calls.  trees_in_front[g] = trees in front of tree group g (n_groups + 1 entries,
 * the last one = trees of the image); split_sum = thre * T; frac_permille = option forest_split_frac.
 * Simulates n_calls scoring calls of 1e6 candidates each: call i is cut in front of group cuts[i] (0: not
 * cut) and leaves open_frac[i] of its candidates open, which the next call's cut takes into account. */
int pk_debug_cut_policy(const int32_t *trees_in_front, int n_groups, double split_sum, int frac_permille,
                        const double *open_frac, int n_calls, int32_t *cuts);

#ifdef __cplusplus
}
#endif
#endif /* PEAKACHU_HIP_H */
