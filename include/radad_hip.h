/*
 * radad_hip.h -- C ABI of libradad_hip.so: the MI355X (gfx950) implementation of RADAD's
 * segment -> embed -> retrieve hot path.
 *
 * Conventions (all entry points):
 *   - plain C types only; no torch / C++ types cross this boundary.
 *   - pointers named *_dev are DEVICE pointers (e.g. torch.Tensor.data_ptr()); *_host are host pointers.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Kernels are enqueued on it
 *     and the call returns without synchronising unless the comment says otherwise.
 *   - every function returns 0 on success and a negative RADAD_E* code on failure; the message for the
 *     calling thread is available from radad_last_error().
 *   - handles are opaque.  A handle may be searched/reconstructed from several threads (host calls are serialised by a
 *     mutex, the device work of consecutive searches by stream order -- a search on ANOTHER stream than the previous one first
 *     records an event behind that stream's work and waits for it: they share one workspace; a stream a handle was last
 *     searched on should therefore outlive the handle's next search, else that search synchronises the device); add/load/destroy
 *     must not race with anything else.
 *   - "without synchronising" has one exception per store: the FIRST large-batch search after the store was created, grown
 *     past its capacity, or re-decided (radad_knn_plane_rebuilds) builds the f16 plane inside the call -- hipDeviceSynchronize,
 *     hipMalloc of 2 bytes per element, one small read-back -- i.e. it stalls every stream of the device once (0.84 ms of
 *     kernel time per million rows of 512).  Build-then-search callers never notice; a caller that interleaves adds and
 *     searches on several streams should radad_knn_reserve up front and run one warm-up search after the last add.
 *
 * Each group cites the reference interface (file:line under the RADAD repository) it replaces.
 */
#ifndef RADAD_HIP_H
#define RADAD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RADAD_ABI_VERSION 1

/* error codes */
#define RADAD_OK 0
#define RADAD_EINVAL (-1)   /* bad argument (maps to ValueError in the Python mirror)      */
#define RADAD_EHIP (-2)     /* a HIP runtime call failed                                    */
#define RADAD_ENOMEM (-3)   /* device or host allocation failed                             */
#define RADAD_EIO (-4)      /* save/load failed                                             */
#define RADAD_ESTATE (-5)   /* handle in the wrong state (e.g. search on an empty index)   */

/* metric: what faiss.IndexFlatL2 / IndexFlatIP / IndexFlatIP+normalise compute
 * (vector_database.py:61-64, :97, :100-105) */
#define RADAD_METRIC_L2 0      /* squared Euclidean distance, ascending                    */
#define RADAD_METRIC_IP 1      /* inner product, descending                                */
#define RADAD_METRIC_COSINE 2  /* rows and queries L2-normalised (x/(|x|+1e-12)), then IP  */

/* pooling mode (config.tpp_pooling_type, pooling.py:74-80) */
#define RADAD_POOL_MAX 0
#define RADAD_POOL_AVG 1

int radad_abi_version(void);
const char* radad_last_error(void);
/* number of visible HIP devices, or a negative error code */
int radad_device_count(void);

/* ------------------------------------------------------------------------------------------------
 * Vector store + brute-force kNN.
 * Replaces the faiss index object that vector_database.py keeps in `self.index`:
 *   create   -> faiss.IndexFlatL2/IP(dimension)                 vector_database.py:56-97
 *   add      -> index.add(batch) after _maybe_normalize          vector_database.py:100-105,118,138
 *   ntotal   -> index.ntotal                                     vector_database.py:151,169; pipeline.py:465
 *   search   -> index.search(q, k)                               vector_database.py:181
 *   reconstruct -> index.reconstruct(i)                          pipeline.py:503
 *   save/load -> faiss.write_index / read_index                  vector_database.py:203,230
 * ---------------------------------------------------------------------------------------------- */
typedef struct radad_knn_s* radad_knn_t;

/* dim must be a positive multiple of 4.  id_base is added to every returned index (row-shard offset;
 * 0 for an unsharded store).  Rows are stored as fp32 in HBM in insertion order. */
int radad_knn_create(int dim, int metric, int device, int64_t id_base, radad_knn_t* out);
/* same with a choice of storage type.  RADAD_STORE_F16 is the reference's `use_float16` knob (faiss
 * GpuIndexFlatConfig.useFloat16, vector_database.py:80): rows are rounded to IEEE fp16 on the way in (after the cosine
 * normalisation), the scan runs on v_mfma_f32_32x32x16_f16 (fp16 operands, fp32 accumulate; dim % 64 == 0 and k <= 26,
 * other shapes decode to fp32 while staging), the float64 re-rank uses the fp32 queries against the decoded rows,
 * reconstruct returns the decoded rows. */
#define RADAD_STORE_F32 0
#define RADAD_STORE_F16 1
int radad_knn_create_ex(int dim, int metric, int store_dtype, int device, int64_t id_base, radad_knn_t* out);
/* Kernel choices of a handle, for experiments and A/B measurements (every value returns the same results; only speed differs).
 * Set them right after creation: HI_PLANE and CENTRE are refused once the f16 plane of the store has been built.
 *   RADAD_KNN_OPT_HI_PLANE   1 (default) / 0: large batches scan the f16 "hi plane" of an fp32 store (certified scan + float64 re-rank)
 *                            / every batch scans on the fp32 kernels (~8x slower 1024-query scans)
 *   RADAD_KNN_OPT_CENTRE     -1 (default: decided per store from |mean|^2 / mean |y|^2) / 0 never / 1 always subtract the rows' column
 *                            mean before rounding the plane
 *   RADAD_KNN_OPT_SMALLQ_HI  1 (default) / 0: batches of <= 16 queries stream the f16 plane / the fp32 rows (twice the bytes)
 *   RADAD_KNN_OPT_WIDE_MIN_Q smallest batch that takes the 256-query tile scan instead of the streaming kernels (default 17)
 *   RADAD_KNN_OPT_DENSE      1 (default) / 0: an fp32 store of <= 6144 rows (<= 16384 for <= 16 queries) is searched by writing out every
 *                            score and selecting on them / by the register-list kernels
 *   RADAD_KNN_OPT_LIVE_FLOOR 0 (default) / 1: the certified tile scan covers the store in one launch per phase with a floor kernel between
 *                            them / in ONE launch that raises its admission floors inside it (built and measured in round 5: 1-4 % slower)
 * The environment variables RADAD_KNN_HI, RADAD_KNN_CENTRE, RADAD_KNN_SMALLQ_HI, RADAD_WIDE_MIN_Q override the DEFAULTS of handles
 * created while they are set (announced once per process on stderr); the product path never needs them. */
#define RADAD_KNN_OPT_HI_PLANE 0
#define RADAD_KNN_OPT_CENTRE 1
#define RADAD_KNN_OPT_SMALLQ_HI 2
#define RADAD_KNN_OPT_WIDE_MIN_Q 3
#define RADAD_KNN_OPT_DENSE 4
#define RADAD_KNN_OPT_LIVE_FLOOR 5
int radad_knn_set_option(radad_knn_t h, int option, int value);
int radad_knn_destroy(radad_knn_t h);
int radad_knn_dim(radad_knn_t h, int* dim);
int radad_knn_metric(radad_knn_t h, int* metric);
int radad_knn_ntotal(radad_knn_t h, int64_t* n);
/* reserve HBM for `capacity` rows up front (optional; add grows geometrically otherwise) */
int radad_knn_reserve(radad_knn_t h, int64_t capacity);
/* append n rows [n, dim] fp32 (device pointer).  Cosine: rows are normalised on the way in, so
 * reconstruct returns the normalised row exactly as faiss would return what was added. */
int radad_knn_add(radad_knn_t h, const float* rows_dev, int64_t n, void* stream);
/* same, from a host buffer (what index.add(np.ndarray) does); synchronous */
int radad_knn_add_host(radad_knn_t h, const float* rows_host, int64_t n);
/* top-k of every query against the whole store.
 *   q_dev        [nq, dim] fp32
 *   out_dist_dev [nq, k] fp32  L2: squared distances ascending; IP/cosine: inner products descending
 *   out_idx_dev  [nq, k] int64 id_base + row; ties are broken towards the LOWER index, always
 *   slots beyond ntotal are filled with index -1 and distance +inf (L2) / -inf (IP), as faiss does.
 * k must be in [1, RADAD_KNN_MAX_K]. */
#define RADAD_KNN_MAX_K 1024
int radad_knn_search(radad_knn_t h, const float* q_dev, int64_t nq, int k, float* out_dist_dev,
                     int64_t* out_idx_dev, void* stream);
/* same search, additionally returning the float64 distances the final ranking was made on
 * (out_key_dev [nq, k] double, may be NULL).  The MFMA scan is a filter with a measured error bound eps(q): every listed
 * row whose scan score is within 2 eps of the k-th best is re-scored in float64 from the stored rows (L2 as sum (q-y)^2)
 * and ordered by (float64 distance, id); a per-query certificate checks that no unlisted row can reach that threshold, and
 * the queries it rejects are searched again by an exact float64 kernel (driven from the device, no host round trip).  For
 * k <= 128 ids and order are therefore those of an exact float64 brute force, ties to the lower id, and out_dist is the
 * correctly rounded distance; for larger k the k + 6 best scan candidates are re-ranked without a certificate.  Sharded
 * searches merge on these keys (radad_topk_merge_f64) so that no cross-shard pair is decided by fp32 rounding.
 * A handle may be searched from several threads and streams: calls are serialised by a mutex and a search enqueued on another
 * stream than the previous one waits (event) for that one's device work, since they share the handle's workspace. */
int radad_knn_search_f64(radad_knn_t h, const float* q_dev, int64_t nq, int k, float* out_dist_dev,
                         int64_t* out_idx_dev, double* out_key_dev, void* stream);
/* same with a choice of query type: RADAD_Q_BF16 = bfloat16 queries [nq, dim] (BASELINE config 5: bf16 embeddings,
 * the reference's autocast knob feature_extractor.py:84,140) -- decoded exactly to fp32 on the device, then the same
 * path (cosine normalisation in fp32, as vector_database.py:166 does on whatever it is handed). */
#define RADAD_Q_F32 0
#define RADAD_Q_BF16 1
int radad_knn_search_ex(radad_knn_t h, const void* q_dev, int q_dtype, int64_t nq, int k, float* out_dist_dev,
                        int64_t* out_idx_dev, double* out_key_dev, void* stream);
/* The same search in two halves, for a store that is ROW-SHARDED over several GPUs (the reference is single-GPU,
 * vector_database.py:23; sharding is this build's, north_star).  _begin prepares the queries and scans this shard; it writes to
 * topk_lower_bounds_dev [nq, k] lower bounds of the exact scores of this shard's k best rows per query, unordered (scores: q.y
 * for inner product / cosine, -|q - y|^2 for L2; -inf where the shard has fewer rows or the scan that ran offers none).  The
 * caller gathers them from all shards (one all-gather of 4 nq k bytes per shard) and takes, per query, the k-th LARGEST of the
 * G k values: a lower bound of the exact k-th best score of the whole store.  _finish, given that [nq] vector, re-ranks in
 * float64 only the candidates that can still be among the GLOBAL k best -- instead of every shard certifying its own top k,
 * G x the work of one GPU on G shards.  With a bound the rows a shard returns are those of its rows that can be in the global
 * top k (fewer than k is normal; the rest is -1 filled); merged over the shards (radad_topk_merge_f64) the result is exactly the
 * unsharded search's.  global_lower_bound_dev == NULL: _finish returns this shard's own top k, i.e. _begin + _finish ==
 * radad_knn_search_ex.  One begun search per handle at a time; other searches on the handle fail until it is finished. */
int radad_knn_search_begin(radad_knn_t h, const void* q_dev, int q_dtype, int64_t nq, int k, float* topk_lower_bounds_dev, void* stream);
int radad_knn_search_finish(radad_knn_t h, const float* global_lower_bound_dev, float* out_dist_dev, int64_t* out_idx_dev,
                            double* out_key_dev, void* stream);
/* Gives up a begun search (e.g. the exchange of the bounds between the shards failed): the handle accepts searches again.  No-op when
 * nothing was begun.  While a search is begun, add / reserve / load / load_range fail with RADAD_EINVAL (its second half reads the
 * rows live); _finish may run on another stream than _begin (it waits for _begin's device work through an event). */
int radad_knn_search_abort(radad_knn_t h);
/* host-buffer variant (what index.search(np.ndarray, k) does); synchronous */
int radad_knn_search_host(radad_knn_t h, const float* q_host, int64_t nq, int k, float* out_dist_host,
                          int64_t* out_idx_host);
/* batched index.reconstruct: out[i,:] = row (idx[i] - id_base); idx < 0 or out of range -> zeros
 * (the zero padding pipeline.py:511-512 applies) */
int radad_knn_reconstruct(radad_knn_t h, const int64_t* idx_dev, int64_t n, float* out_dev, void* stream);
int radad_knn_reconstruct_host(radad_knn_t h, const int64_t* idx_host, int64_t n, float* out_host);
/* binary snapshot of the store ("RADADKNN" header + rows as stored, fp32 or fp16); load replaces the contents of h
 * (VectorDatabase.save / .load, vector_database.py:190-242).  Both directions stream through two pinned staging
 * buffers on a private stream (file I/O overlaps the DMA); load maps the file read-only, so
 * radad_knn_load_range -- rows [row0, row0 + n_rows) of the snapshot, n_rows < 0 = through the end -- touches only
 * its own byte range: G ranks load one snapshot as G row shards (create each handle with id_base = row0). */
int radad_knn_save(radad_knn_t h, const char* path);
int radad_knn_load(radad_knn_t h, const char* path);
int radad_knn_load_range(radad_knn_t h, const char* path, int64_t row0, int64_t n_rows);
/* header of a snapshot without a handle (any out pointer may be NULL) */
int radad_knn_snapshot_info(const char* path, int* dim_out, int* metric_out, int* store_dtype_out, int64_t* ntotal_out);
/* seconds spent in the most recent search's kernels are NOT measured here; use HIP events on `stream`.
 * Query the launch geometry of the last search (for roofline accounting in bench.py). */
int radad_knn_last_launch(radad_knn_t h, int* n_query_tiles, int* n_db_splits, int* block_threads);
/* which scan kernel the last search ran: what it streams decides the bytes a roofline is quoted on */
#define RADAD_SCAN_F32_TILE 0     /* fp32-MFMA tile kernels (fp32 rows; also the generic kernel) */
#define RADAD_SCAN_HI_TILE 1      /* certified f16-MFMA tile scan over the f16 plane / fp16 store (more than 16 queries) */
#define RADAD_SCAN_F32_SMALLQ 2   /* <= 16 queries, fp32 rows streamed (4 bytes per element) */
#define RADAD_SCAN_HI_SMALLQ 3    /* <= 16 queries, f16 plane / fp16 store streamed (2 bytes per element), certified */
#define RADAD_SCAN_F16_TILE 4     /* fp16 store on the fp16-MFMA tile kernel without the certificate path */
#define RADAD_SCAN_F32_DENSE 5    /* small fp32 store: every (row, query) score written out (fp32 MFMA), selection on the scores */
int radad_knn_last_scan_kind(radad_knn_t h, int* kind);
/* scan-kernel launches of the last search (the certified tile scan covers a large store in two: the first eighth with the
 * sample's admission floor, the rest with the floor the first eighth's candidates give); radad_knn_profile_read has one entry each */
/* How ONE launch of the certified tile scan over n_rows rows and n_queries queries is laid out (pure host arithmetic, no device): 256-query
 * tiles, `chunks` row chunks (a multiple of 8) of rows_per_chunk rows (a multiple of the 256-row tile) -- query_tiles x chunks workgroups.
 * Chosen by what the launch costs: (rounds of 256 workgroups) x (tiles per chunk), both rounded up. */
int radad_knn_scan_geometry(int64_t n_rows, int64_t n_queries, int* query_tiles, int* chunks, int64_t* rows_per_chunk);
int radad_knn_last_scan_launches(radad_knn_t h, int* n_launches);
/* phases of the last search's tile scan: 1 + the number of times the admission floors were raised from the candidates emitted so far --
 * between launches (the default: 2 up to ~1.2 M rows, 3 up to ~9.5 M, 4 beyond) or inside the one launch (RADAD_KNN_OPT_LIVE_FLOOR 1) */
int radad_knn_last_scan_phases(radad_knn_t h, int* n_phases);
/* Diagnostics of the last search when it was a certified tile scan of nq queries (synchronises with it): per query the number of
 * candidates the scan emitted (more than the candidate buffer holds -- 1024, 4096 after the handle widened them -- rejects the query)
 * and, optionally, the admission floor the scan ended with.  Host pointers. */
int radad_knn_last_emitted(radad_knn_t h, int* counts_host, float* floors_host, int64_t nq);
/* the f16 plane the certified scans read, once a search has built it: built (0/1); centred = the common component (column mean)
 * of the rows is subtracted before rounding (stores of embeddings that share most of their mean); one_scale = one power-of-two
 * scale for all rows (rows of one magnitude: the scan applies no per-score arithmetic) */
int radad_knn_plane_info(radad_knn_t h, int* built, int* centred, int* one_scale);
/* How often the plane of this store was RE-decided (dropped and rebuilt with a new centre and scale): it is decided from the rows the
 * store holds when it is first built; it is decided again -- 0.84 ms per million rows, one synchronisation, inside the first
 * large-batch search that notices -- when the store has doubled since, when appended rows measure 8x beyond what the decision saw
 * (largest element or rounding residual), or when a batch was mostly rejected by the certificate and rows were appended since
 * (vector_database.py:134-138 appends 10 000 rows at a time; a store that drifts keeps the certified scan instead of the 8x slower
 * fp32 fallback).  Results never depend on it. */
int radad_knn_plane_rebuilds(radad_knn_t h, int* n_out);
/* The handle's self-tuning state, read WITHOUT synchronising (advisory; results never depend on it).  Every certified search leaves
 * a report (queries rejected by the certificate, batch size) in pinned host memory when its device work ends; every later search
 * acts on the reports that have arrived since -- however far the host has run ahead.  A batch of >= 64 queries rejected by more
 * than a quarter first re-decides the plane (if rows were appended since), then widens the candidate buffers (cap_boost 1 -> 4),
 * then moves to the fp32 kernels for 8 searches (fp32_searches_left).  reports_consumed counts the reports looked at so far.
 * (The reference has no counterpart: faiss's flat search has one code path, vector_database.py:181.) */
int radad_knn_tuning_info(radad_knn_t h, int* cap_boost, int* fp32_searches_left, int64_t* reports_consumed);
/* Certificate of the most recent search (see radad_knn_search_f64): number of queries the float64 re-rank could NOT certify
 * and that were therefore searched again by the exact float64 kernel (results are exact either way).  Synchronises with
 * that search.  radad_knn_last_certificate additionally returns the batch size and stats6 = {rejected queries, sum over
 * the batch of candidates re-scored in float64, rejections because: more than 512 candidates within the error bound, the scan's
 * candidate buffer (or a chunk's list, on the fp32 kernels) overflowed, the scan's admission floor was above the threshold, the
 * scan dropped a candidate}. */
int radad_knn_last_recheck(radad_knn_t h, int* n_queries);
int radad_knn_last_certificate(radad_knn_t h, int64_t* n_queries, int* stats6);

/* HIP-event timing of the scan kernel alone, on the stream each search is enqueued on:
 * enable = 1 -> every search records an event pair around each scan-kernel launch (ring of 64); enable = n > 1 -> every n-th search
 * does (an event record between dependent kernels delays the next one by ~6 us: four per certified search of a large store);
 * 0 -> off.  read synchronises on the recorded events and returns the kernel durations in ms, oldest first.  For bench.py's
 * roofline line. */
int radad_knn_profile(radad_knn_t h, int enable);
int radad_knn_profile_read(radad_knn_t h, float* ms_out, int cap, int* n_out);

/* merge P partial top-k lists per query into one (used for the multi-GPU all-gather merge and
 * internally by search):  in_dist/in_idx are [P, nq, k]; metric decides the order; (dist, idx)
 * lexicographic, idx -1 entries sort last. */
int radad_topk_merge(int metric, const float* in_dist_dev, const int64_t* in_idx_dev, int n_parts, int64_t nq,
                     int k, float* out_dist_dev, int64_t* out_idx_dev, int device, void* stream);

int radad_topk_merge_f64(int metric, const double* in_key_dev, const int64_t* in_idx_dev, int n_parts, int64_t nq,
                         int k, float* out_dist_dev, int64_t* out_idx_dev, double* out_key_dev, int device,
                         void* stream);

/* Device-side form of the exclusion loop of retrieve_similar_vectors (pipeline.py:491-515): for every query row keep,
 * in order, the first k_keep of its k_in search hits whose row tag is NOT in the exclusion set; pad with id -1 and
 * distance NaN.  row_tags_dev [ntotal] int64 (e.g. a hash of os.path.basename(path), one per stored row, indexed by
 * id - id_base); excl_sorted_dev [n_excl] int64 ascending (may be NULL when n_excl == 0); hits with id < 0 are skipped. */
int radad_filter_topk(const float* in_dist_dev, const int64_t* in_idx_dev, int64_t nq, int k_in, int k_keep,
                      const int64_t* row_tags_dev, int64_t ntotal, int64_t id_base, const int64_t* excl_sorted_dev,
                      int64_t n_excl, float* out_dist_dev, int64_t* out_idx_dev, int device, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Inverted-file flat index: faiss.IndexIVFFlat(IndexFlatL2 quantiser, d, nlist, METRIC_L2), the reference's optional
 * `vector_db_index_type == "IVF"` (vector_database.py:65-70 create with nlist = max(64, ivf_nlist), :124-128 train on the
 * first <= 50 000 rows, :174-179 nprobe = config.vector_db_nprobe).  dim must be a multiple of 32, k <= 128: up to k = 26 the
 * probed lists are scanned; 27..128 is answered by the exact certified scan of the same rows (recall 1.0).
 * A search returns the exact top-k (squared L2, ties to the lower id) among the rows of the nprobe lists whose
 * centroids are nearest to the query; -1 / +inf where those lists hold fewer than k rows.
 * ---------------------------------------------------------------------------------------------- */
typedef struct radad_ivf_s* radad_ivf_t;
int radad_ivf_create(int dim, int nlist, int device, radad_ivf_t* out);
int radad_ivf_destroy(radad_ivf_t h);
int radad_ivf_is_trained(radad_ivf_t h, int* trained);
int radad_ivf_ntotal(radad_ivf_t h, int64_t* n);
int radad_ivf_nlist(radad_ivf_t h, int* nlist);
/* k-means (Lloyd, `niter` iterations; faiss trains its IVF quantiser with 10) on n training rows; synchronous */
int radad_ivf_train(radad_ivf_t h, const float* rows_dev, int64_t n, int niter, void* stream);
/* install centroids computed elsewhere ([nlist, dim] fp32) instead of training; only on an empty index */
int radad_ivf_set_centroids(radad_ivf_t h, const float* centroids_dev, void* stream);
int radad_ivf_centroids(radad_ivf_t h, float* out_dev, void* stream);
/* list of every stored row, in insertion order (host int32 [ntotal]) */
int radad_ivf_assignments_host(radad_ivf_t h, int32_t* out_host, int64_t cap);
int radad_ivf_add(radad_ivf_t h, const float* rows_dev, int64_t n, void* stream);               /* synchronous */
int radad_ivf_search(radad_ivf_t h, const float* q_dev, int64_t nq, int k, int nprobe, float* out_dist_dev,
                     int64_t* out_idx_dev, void* stream);      /* asynchronous on `stream` (the (query, probe) pairs are grouped by list on
                                                                    the device); only the first search after an add rebuilds the list layout
                                                                    synchronously */
/* 1 if the most recent radad_ivf_search was answered by the exact scan of the index's flat store (26 < k <= 128): every row an IVF
 * search could return AND the ones its probing would have missed -- a superset of faiss.IndexIVFFlat's answer (which holds only rows
 * of the nprobe lists, vector_database.py:174-179), at the flat scan's cost, nprobe ignored.  0: the nprobe lists were scanned. */
int radad_ivf_last_search_exact(radad_ivf_t h, int* exact_out);
/* which list scan answered the most recent radad_ivf_search, and how many of its queries the f16 scan's certificate handed to the fp32
 * pass (synchronises with that search):
 *   RADAD_IVF_SCAN_F32         fp32 rows, v_mfma_f32_16x16x4_f32, k + 6 candidates per (query, list) re-ranked in float64 (dim % 64 != 0,
 *                              no f16 plane, more probed lists than the re-rank stages, RADAD_IVF_OPT_HI_SCAN 0)
 *   RADAD_IVF_SCAN_HI          the flat store's f16 plane gathered list-major, certified per query as the flat scan is; rejected queries
 *                              are answered by the fp32 list scan in the same call
 *   RADAD_IVF_SCAN_EXACT_FLAT  k > 26: the exact scan of the flat store (radad_ivf_last_search_exact) */
#define RADAD_IVF_SCAN_F32 0
#define RADAD_IVF_SCAN_HI 1
#define RADAD_IVF_SCAN_EXACT_FLAT 2
int radad_ivf_last_search_info(radad_ivf_t h, int* kind_out, int* rejected_out);
/* RADAD_IVF_OPT_HI_SCAN 1 (default) / 0: list scans over the f16 plane / over the fp32 rows (A/B measurements; same results);
 * 2 (tests): the f16 scan runs and every query is then treated as rejected by its certificate, i.e. answered by the fp32 pass */
#define RADAD_IVF_OPT_HI_SCAN 0
int radad_ivf_set_option(radad_ivf_t h, int option, int value);
int radad_ivf_reconstruct(radad_ivf_t h, const int64_t* idx_dev, int64_t n, float* out_dev, void* stream);

/* out[r] = the k-th largest of the groups x per_group values of row r, value (g, i) at in[(g n + r) per_group + i] -- the layout
 * [groups][n][per_group] an all-gather of the shards' [n][k] bound blocks has (groups = 1: plain [n][m]); groups x per_group <= 1280,
 * 1 <= k <= groups x per_group; NaN ranks lowest.  For the sharded search: the G k lower bounds per query gathered from G
 * shards' radad_knn_search_begin -> the bound radad_knn_search_finish takes. */
int radad_kth_largest(const float* in_dev, int64_t n, int groups, int per_group, int k, float* out_dev, int device, void* stream);
/* row L2 normalisation x / (|x| + 1e-12)  (vector_database.py:100-105); in-place allowed */
int radad_rownorm(const float* in_dev, float* out_dev, int64_t n, int dim, int device, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Embedding: segment -> front-end -> frame projection -> temporal pyramid pooling -> segment mean.
 *   segment plan          segmenter.py:8-39 (n_seg = max(1,(N-L)//hop+1); zero pad only when N < L)
 *   zero-mean/unit-var    feature_extractor.py:25-30 (HF Wav2Vec2FeatureExtractor, eps 1e-7)
 *   log-mel               feature_extractor.py:94-97 (HF WhisperFeatureExtractor: hann 400 / hop 160,
 *                         |X|^2, drop last frame, slaney mel 201->80, log10 clamp 1e-10, max-8, (x+4)/4)
 *   frame projection      stands where the pretrained encoder forward sits (feature_extractor.py:33,110,167)
 *   pooling               pooling.py:66-103 (adaptive max/avg bins, bin-major layout, levels concatenated)
 *   segment mean          pipeline.py:411
 * ---------------------------------------------------------------------------------------------- */
#define RADAD_MAX_LEVELS 8
typedef struct radad_embed_cfg {
    int32_t segment_length;    /* samples per segment (config.segment_length*sample_rate = 32000)   */
    int32_t hop_length;        /* int(segment_length*(1-overlap)) = 16000                            */
    int32_t normalize;         /* 1: per-segment zero-mean/unit-variance before the spectrogram      */
    int32_t n_fft;             /* must be 400                                                        */
    int32_t fft_hop;           /* must be 160                                                        */
    int32_t n_mels;            /* must be 80                                                         */
    int32_t padded_samples;    /* 0: spectrogram of the segment itself (segment_length/160 frames);
                                  otherwise zero-pad to this many samples first (HF pads to 480000)  */
    int32_t feat_dim;          /* F, multiple of 32                                                  */
    int32_t n_levels;          /* pyramid levels (config.tpp_levels)                                 */
    int32_t levels[RADAD_MAX_LEVELS];
    int32_t pool_mode;         /* RADAD_POOL_MAX / RADAD_POOL_AVG                                    */
} radad_embed_cfg;

typedef struct radad_embed_s* radad_embed_t;

/* mel_filters_host [201, 80] fp32 (bins x mels); proj_w_host [80, F] fp32; proj_b_host [F] fp32 */
int radad_embed_create(const radad_embed_cfg* cfg, const float* mel_filters_host, const float* proj_w_host,
                       const float* proj_b_host, int device, radad_embed_t* out);
/* same with a choice of log-mel kernel (all give feature_extraction_whisper.py:135-168 per segment within the 1e-4 bar; for A/B
 * measurements and parity tests): flags = 0 is radad_embed_create.
 *   RADAD_EMBED_NO_SHARED_FRAMES  one transform per (segment, frame) even where overlapping segments share frames (k_logmel_h)
 *   RADAD_EMBED_LOGMEL_F32        the folded DFT on the fp32 matrix pipe (k_logmel, ~2x slower than k_logmel_h)
 *   RADAD_EMBED_LOGMEL_DFT_GEMM   shared frames as a DFT-as-GEMM on the f16 matrix pipe (k_logmel_h_clip) instead of the radix FFT on the
 *                                 vector ALU (k_logmel_fft_clip, the default where the configuration allows sharing)
 * The environment variables RADAD_LOGMEL_SHARED=0, RADAD_LOGMEL_F32=1, RADAD_LOGMEL_FFT=0 set the same bits for extractors created
 * while they are set (announced once per process on stderr). */
#define RADAD_EMBED_NO_SHARED_FRAMES 1
#define RADAD_EMBED_LOGMEL_F32 2
#define RADAD_EMBED_LOGMEL_DFT_GEMM 4
int radad_embed_create_ex(const radad_embed_cfg* cfg, int flags, const float* mel_filters_host, const float* proj_w_host,
                          const float* proj_b_host, int device, radad_embed_t* out);
int radad_embed_destroy(radad_embed_t h);
int radad_embed_output_dim(radad_embed_t h, int* dim);     /* sum(levels)*F  (pooling.py:119-122)    */
int radad_embed_num_frames(radad_embed_t h, int* frames);   /* frames per segment                      */
/* number of segments the plan yields for a clip of n samples (segmenter.py:25) */
int64_t radad_segment_count(int64_t n_samples, int32_t segment_length, int32_t hop_length);
/* clip embeddings for a batch: clip b is wave_dev[clip_offsets_host[b] .. clip_offsets_host[b+1]).
 * out_dev [n_clips, output_dim] fp32.  (process_audio_batch, pipeline.py:392-414, minus file loading) */
int radad_embed_forward(radad_embed_t h, const float* wave_dev, const int64_t* clip_offsets_host, int64_t n_clips,
                        float* out_dev, void* stream);

/* same with a choice of output type: RADAD_OUT_BF16 emits the clip embeddings as bfloat16 (rounded to nearest even in
 * the epilogue of the pooling kernel; BASELINE config 5, the reference's reduced-precision knob feature_extractor.py:84,140) */
#define RADAD_OUT_F32 0
#define RADAD_OUT_BF16 1
int radad_embed_forward_ex(radad_embed_t h, const float* wave_dev, const int64_t* clip_offsets_host, int64_t n_clips,
                           void* out_dev, int out_dtype, void* stream);
/* clip offsets resident on the DEVICE (int64 [n_clips + 1]): the segment plan (segmenter.py:25-39: n_seg per clip,
 * exclusive scan, start / valid samples of every segment) is built by a kernel, nothing synchronises with the host.
 * n_samples_total (= clip_offsets[n_clips], known to the caller from the wave buffer's size) only bounds the number of
 * segments for buffer and grid sizes. */
int radad_embed_forward_dev(radad_embed_t h, const float* wave_dev, const int64_t* clip_offsets_dev, int64_t n_clips,
                            int64_t n_samples_total, void* out_dev, int out_dtype, void* stream);
/* 16-bit PCM (what audio files hold) -> the float32 samples the reference's loader produces (pipeline.py / dataset.py: librosa.load,
 * i.e. sample / 32768, exact in fp32), on the device: upload the PCM, convert here, hand the result to radad_embed_forward* --
 * half the bytes over PCIe.  out_dev may not alias pcm_dev. */
int radad_pcm16_to_f32(const int16_t* pcm_dev, float* out_dev, int64_t n, int device, void* stream);
/* The device-resident offsets cannot be validated by the host without a synchronisation (the host path of
 * radad_embed_forward rejects bad offsets up front, as segmenter.py:18-19 raises on bad input).  The plan kernel therefore
 * REPAIRS them -- every clip is clamped into [0, n_samples_total], a clip that ends before it starts becomes empty, the
 * segment count is capped -- so no kernel reads outside the wave buffer, and records what it repaired.  This call waits for the
 * most recent radad_embed_forward_dev batch and returns (once) those flags: 0 = offsets were fine; bit 0 an offset outside
 * [0, n_samples_total]; bit 1 non-monotone offsets; bit 2 more segments than n_samples_total / hop + n_clips.  A non-zero
 * value means the embeddings of that batch are NOT those of the intended clips. */
int radad_embed_plan_flags(radad_embed_t h, int* flags_out);
/* The same without waiting: the flags of the device-offset batches that have COMPLETED since the last report (OR-ed; each batch is
 * reported once, by whichever of the two calls sees it first), and how many batches are still in flight.  A caller that queues batches
 * back to back polls before each one (it never blocks behind the previous batch) and calls radad_embed_plan_flags once after the
 * last. */
int radad_embed_plan_flags_poll(radad_embed_t h, int* flags_out, int* batches_pending_out);
/* Which log-mel kernel the most recent embedding call took: 0 = one transform per (segment, frame) (k_logmel_h, or k_logmel under
 * RADAD_LOGMEL_F32); 1, 2 = the frames overlapping segments share were transformed once per CLIP (chosen by the configuration --
 * segment hop a multiple of 160 samples and smaller than the segment, Slaney filter bank -- for calls that hand over clips;
 * RADAD_LOGMEL_SHARED=0 turns it off): 2 = as a radix FFT on the vector ALU (k_logmel_fft_clip, the default), 1 = as a DFT-as-GEMM on
 * the f16 matrix pipe (k_logmel_h_clip: RADAD_LOGMEL_FFT=0, or a filter bank that is not triangular).  All give
 * feature_extraction_whisper.py:135-168 per zero-mean/unit-variance segment. */
int radad_embed_last_logmel_kind(radad_embed_t h, int* kind_out);
/* How k_logmel_h_clip cuts a clip of n_segments segments (frames_per_segment frames each, segment hop = hop_frames frames) into
 * workgroup chunks -- host arithmetic only, no device needed (the kernel, the plan kernel and the host share it): out5 = { full
 * chunks of 104 interior frames, interior frames of the tail chunk, edge frames the tail chunk carries, edge-only chunks of <= 32
 * edge frames, total chunks }.  The clip has (n_segments - 1) hop_frames + frames_per_segment - 3 interior frames and
 * 3 n_segments edge frames (a segment's frames 0, 1 and T - 1). */
int radad_embed_clip_chunks(int n_segments, int frames_per_segment, int hop_frames, int32_t* out5);
/* The same for k_logmel_fft_clip (the shared-frame work list as a radix FFT on the vector ALU, csrc/logmel_fft.inc): chunks of 64
 * frame slots; out5 = { full chunks of 64 interior frames, interior frames of the tail chunk, edge frames the tail chunk carries,
 * edge-only chunks of <= 26 edge frames, total chunks }.  Host arithmetic only. */
int radad_embed_fft_clip_chunks(int n_segments, int frames_per_segment, int hop_frames, int32_t* out5);
/* The per-lane constant tables of k_logmel_fft_clip for a mel filter bank [201, 80] (host arithmetic only, no device needed;
 * tests/test_fft_tables.py drives a numpy restatement of the kernel's data flow with them and compares it with numpy.fft.rfft and the
 * oracle's log-mel).  tab_out must hold `cap` >= 2224 floats: 25 groups (index k1, or the sample group m for the window) of
 * [hann(16 m + 2 r), hann(16 m + 2 r + 1)] [cos, sin of W200^(r k1)] [cos, sin of W400^k, k = 25 k2 + k1] [fb[k][b] / 4,
 * fb[k][b + 1] / 4, byte offset of band b's slot in a row of the mel tile (int32 bits), 1 if the next bin starts at band b + 1
 * (int32 bits)], each for the 8 lanes of an octet (lane p holds the residue class r = p < 4 ? p : 11 - p of the packed frame and
 * ends with the bin block k2 = bitrev3(p)); then per lane the constants of the three exchange stages [g1, c1, s1, g2, c2, s2, g3, 0]:
 * own <- (own + g partner)(c + i s); then per band the byte offsets (int32) of the two row slots that add up to it.  info4 =
 * { 1 if the bank has the shape the sparse mel step needs (every bin feeds at most two adjacent bands, the band index never falls and
 * rises by at most one per bin, bin 200 weightless; otherwise the matrix-pipe kernels run), floats written, floats per row of the mel
 * tile, float offset of the band table }. */
int radad_embed_fft_tables(const float* mel_filters_host, float* tab_out, int cap, int32_t* info4);

/* same HIP-event timing for the two embedding kernels (k_logmel, k_proj_pool) of radad_embed_forward */
int radad_embed_profile(radad_embed_t h, int enable);
int radad_embed_profile_read(radad_embed_t h, float* logmel_ms_out, float* projpool_ms_out, int cap, int* n_out);

/* stage entry points, for parity with the reference's per-stage functions ------------------------ */
/* zero-mean/unit-var of S segments: seg s = wave_dev[seg_start_host[s] .. +seg_valid_host[s]) zero
 * padded to segment_length; out_dev [S, segment_length] */
int radad_embed_normalize(radad_embed_t h, const float* wave_dev, const int64_t* seg_start_host,
                          const int32_t* seg_valid_host, int64_t n_seg, float* out_dev, void* stream);
/* log-mel of S segments -> out_dev [S, frames, 80] (frame-major; HF returns [80, frames]) */
int radad_embed_logmel(radad_embed_t h, const float* wave_dev, const int64_t* seg_start_host,
                       const int32_t* seg_valid_host, int64_t n_seg, float* out_dev, void* stream);
/* frame features of S segments -> out_dev [S, frames, F]  (extract_features protocol) */
int radad_embed_frame_features(radad_embed_t h, const float* wave_dev, const int64_t* seg_start_host,
                               const int32_t* seg_valid_host, int64_t n_seg, float* out_dev, void* stream);

/* stand-alone temporal pyramid pooling of one or more [T_i, F] feature blocks stored back to back:
 * item i = feats_dev[row_offsets_host[i]*F .. row_offsets_host[i+1]*F); out_dev [n_items, sum(levels)*F]
 * (TemporalPyramidPooling.pool_features / pool_features_batch, pooling.py:88-117) */
int radad_tpp_forward(const float* feats_dev, const int64_t* row_offsets_host, int64_t n_items, int feat_dim,
                      const int32_t* levels, int n_levels, int pool_mode, float* out_dev, int device,
                      void* stream);
/* mean over groups of rows: out[g,:] = mean(in[group_offsets[g]..group_offsets[g+1], :])  (pipeline.py:411) */
int radad_group_mean(const float* in_dev, const int64_t* group_offsets_host, int64_t n_groups, int dim,
                     float* out_dev, int device, void* stream);

/* ------------------------------------------------------------------------------------------------
 * ProjectionLayer inference forward (projection.py:68-106), fused:
 *   s = W2 tanh(W1 x + b1) + b2; a = softmax_K(s); c = W4 relu(W3 x + b3) + b4; u = sum_K a c;
 *   out = W6 LN(W5 u + b5; eps 1e-6) + b6
 * Weights are given in torch nn.Linear layout ([out, in] row-major), device pointers.
 * One pass over x: a split-K GEMM against [W1;W3], then one block per batch row (csrc/proj.hip).
 * w54t / b54 hold the inference-time fold  W5 (W4 h + b4) + b5 = (W5 W4) h + (W5 b4 + b5): build them once
 * per set of weights with radad_projection_fold and keep them; when NULL the forward re-folds into its
 * workspace on every call (correct, slower).
 * ---------------------------------------------------------------------------------------------- */
typedef struct radad_proj_weights {
    const float *w1, *b1;       /* attention_score   [H, D], [H]   */
    const float *w2, *b2;       /* attention_final   [1, H], [1]   */
    const float *w3, *b3;       /* cst_hidden        [H, D], [H]   */
    const float *w4, *b4;       /* cst_output        [D, H], [D]   */
    const float *w5, *b5;       /* weight_sum        [H, D], [H]   */
    const float *ln_g, *ln_b;   /* normalization     [H], [H]      */
    const float *w6, *b6;       /* unified_embedding [O, H], [O]   */
    const float *w54t, *b54;    /* optional fold: (W5 W4)^T [H(in), H(out)], W5 b4 + b5 [H]; NULL = fold per call */
} radad_proj_weights;
int radad_projection_forward(const radad_proj_weights* w, const float* x_dev /*[B,K,D]*/, int64_t batch, int k,
                             int dim, int hidden, int out_dim, float* out_dev /*[B,O]*/, float* workspace_dev,
                             int64_t workspace_bytes, int device, void* stream);
int64_t radad_projection_workspace_bytes(int64_t batch, int k, int dim, int hidden, int out_dim);
/* w54t_out_dev [H,H], b54_out_dev [H] from w->w4,b4,w5,b5 (float64 accumulation, rounded once) */
int radad_projection_fold(const radad_proj_weights* w, int dim, int hidden, float* w54t_out_dev, float* b54_out_dev,
                          int device, void* stream);

/* out[r, :] = act(W x[r, :] + bias)   nn.Linear forward, W [out_features, in_features] with row stride ldw;
 * act 0 none / 1 tanh / 2 relu.  Split-K MFMA GEMM (few rows x wide in_features still fills the chip). */
int radad_linear_forward(const float* x_dev, int64_t ldx, const float* w_dev, int64_t ldw, const float* bias_dev,
                         int act, int64_t rows, int out_features, int in_features, float* out_dev, int64_t ldo,
                         float* workspace_dev, int64_t workspace_bytes, int device, void* stream);
int64_t radad_linear_workspace_bytes(int64_t rows, int out_features, int in_features);

/* ------------------------------------------------------------------------------------------------
 * RADADModel.forward after the projection (radad_model.py:38-41), inference:
 *   fused  = Wf cat[tpp, proj] + bf                      (:39; the concatenation is never built)
 *   logits = DetectionModel(fused)                       (:40; detection_model.py:41-72 in eval mode:
 *            per hidden layer Linear -> BatchNorm1d (running stats, given as scale/shift) -> ReLU;
 *            dropout = identity; last layer Linear only)
 * n_layers == 0 stops after the fuse Linear.  Either output pointer may be NULL when not wanted.
 * ---------------------------------------------------------------------------------------------- */
#define RADAD_HEAD_MAX_LAYERS 6
typedef struct radad_head_weights {
    const float *wf, *bf;                           /* fuse [P, D+P], [P]                                  */
    int32_t n_layers;                               /* Linear layers of the detection MLP                  */
    int32_t dims[RADAD_HEAD_MAX_LAYERS + 1];        /* dims[0] = P, dims[i+1] = out width of layer i       */
    const float* lw[RADAD_HEAD_MAX_LAYERS];         /* [dims[i+1], dims[i]]                                */
    const float* lb[RADAD_HEAD_MAX_LAYERS];         /* [dims[i+1]]                                         */
    const float* bn_scale[RADAD_HEAD_MAX_LAYERS];   /* gamma / sqrt(running_var + eps), or NULL            */
    const float* bn_shift[RADAD_HEAD_MAX_LAYERS];   /* beta - running_mean * scale, or NULL                */
} radad_head_weights;
int radad_fuse_head_forward(const radad_head_weights* w, const float* tpp_dev /*[B,D]*/, const float* proj_dev /*[B,P]*/,
                            int64_t batch, int dim, int proj_dim, float* fused_out_dev /*[B,P] or NULL*/,
                            float* logits_out_dev /*[B, dims[n_layers]]*/, float* workspace_dev,
                            int64_t workspace_bytes, int device, void* stream);
int64_t radad_fuse_head_workspace_bytes(int64_t batch, int dim, int proj_dim);

/* ------------------------------------------------------------------------------------------------
 * Synthetic inputs (bench / tests only): a stateless integer hash so host and device produce the
 * SAME bits for element (seed, row, col).  See oracle/synth.py for the host twin.
 * ---------------------------------------------------------------------------------------------- */
int radad_synth_rows(float* out_dev, int64_t row0, int64_t n_rows, int dim, uint64_t seed, int device, void* stream);
int radad_synth_audio(float* out_dev, int64_t clip0, int64_t n_clips, int64_t samples_per_clip, uint64_t seed,
                      int device, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RADAD_HIP_H */
