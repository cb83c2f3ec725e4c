/*
 * dge.h — C ABI of libdge.so: the MI355X-native random-walk + SGNS engine.
 *
 * This is the drop-in boundary for the hot path named by BASELINE.json:north_star.  The reference
 * (thekingofkings/embedding) is plain Java with no FFI; the seam is introduced UNDER its public
 * classes.  Each entry point cites the reference interface it replaces, with
 * J/ = embedding/src/main/java/embedding/.  The JNI / ctypes bindings a maintainer adds on the
 * reference side are shown in INTEGRATION.md.
 *
 * Conventions
 *   - extern "C", opaque handles, plain pointers and sizes; every function returns a status
 *     (DGE_OK == 0).  dge_last_error() returns a thread-local message for the last failure.
 *   - No CPU compute path exists in this library: every call needs a visible gfx950 device and fails
 *     with DGE_ERR_DEVICE otherwise.
 *   - Vertex ids are the caller's insertion ordinals (J/LayeredGraph.java:160,166): name <-> id
 *     interning ("h-regionId" strings) stays on the host-language side.
 *   - "host" pointers are ordinary process memory; "d_" pointers are device memory on the handle's
 *     device (e.g. a torch tensor's data_ptr()).  Handles work on their own non-blocking HIP stream: entry points that
 *     READ a caller's device buffer first wait for the device (hipDeviceSynchronize), entry points that WRITE one return
 *     after their stream has drained, so no stream handshake is needed on the caller's side.
 *   - Handles are not thread-safe; the reference's walk API is single-threaded
 *     (J/CrossTimeGraph.java:134-140) and the trainer owns its workers (J/DeepWalk.java:75).
 */
#ifndef DGE_H
#define DGE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DGE_VERSION 106   /* 106: dge_selftest_atomics_wave_block; the block kernels' accumulator banks (DGE_TUNE_ACC_ROWS / ACC_DRAIN apply to both tables of a block); k_sgns_train_small and DGE_TUNE_SMALL_ROWS.
                             105: stream-ordered partition copies (dge_model_export/import_partition_async, dge_model_stream), dge_host_sync_count, the lock kernels' watchdog,
                             forced schedules refused where they would diverge or spin (DGE_ERR_ARG), DGE_TUNE_ALLOW_UNSAFE / WATCHDOG_MS / HS_COPIES (additions only) */

enum {
    DGE_OK = 0,
    DGE_ERR_ARG = 1,      /* null / negative / inconsistent argument */
    DGE_ERR_RANGE = 2,    /* vertex id outside the graph */
    DGE_ERR_TOPK = 3,     /* keep_top_k: a vertex has fewer than k edges (Java: IndexOutOfBoundsException) */
    DGE_ERR_CAP = 4,      /* caller buffer too small */
    DGE_ERR_STATE = 5,    /* call order violated (e.g. walks before alias tables) */
    DGE_ERR_DEVICE = 6,   /* no usable gfx950 device / HIP failure */
    DGE_ERR_IO = 7
};

typedef struct dge_graph dge_graph;   /* per-timeslice edge store + alias tables, resident in HBM   */
typedef struct dge_walks dge_walks;   /* walk corpus int32 [n_walks x max_len], pad -1, resident in HBM */
typedef struct dge_model dge_model;   /* vocabulary + syn0/syn1neg tables, resident in HBM           */

const char* dge_last_error(void);
int  dge_version(void);
/* "kernels=<hash> sorted=<hash>": 12 hex digits of the SHA-1 of the trainer kernels' sources this library was built from (sgns_kernels.h + dge_algos.h + sgns.hip — the kernels and the host file that picks their launch geometry and policy;
   the same + sgns_sorted.hip).  The committed counter profiles (profiles/traffic.json) carry the stamp of the build they were collected with; bench.py
   quotes a profile's bytes per pair only when the stamp matches the loaded library. */
const char* dge_build_stamp(void);
int  dge_device_count(int* n);

/* ------------------------------------------------------------------------------------------------
 * Edge store — replaces LayeredGraph's HashMap/ArrayList store (J/LayeredGraph.java:142-148).
 * ---------------------------------------------------------------------------------------------- */
/* new LayeredGraph()  J/LayeredGraph.java:150-155.  device >= 0 (no CPU path). */
int  dge_graph_create(dge_graph** out, int device);
void dge_graph_free(dge_graph* g);
/* run this handle's work on an existing hipStream_t (default: a stream the handle owns) */
int  dge_graph_set_stream(dge_graph* g, void* hip_stream);
/* bulk addEdge(fn, tn, weight)  J/LayeredGraph.java:157-174.  Appends in call order; duplicates are
 * kept; per-vertex edge order = insertion order; outDegree = running sum (J/LayeredGraph.java:46-49). */
int  dge_graph_add_edges(dge_graph* g, const int32_t* src, const int32_t* dst, const double* w, int64_t n);
/* same, COO already in device memory (synthetic generators, device-side OD ingest) */
int  dge_graph_add_edges_device(dge_graph* g, const int32_t* d_src, const int32_t* d_dst, const double* d_w, int64_t n);
/* bulk addSourceVertex(vn)  J/LayeredGraph.java:180-189 (after all edges).  stream_sum = 0: sourceWeightSum
 * is the running += of addSourceVertex; 1: DoubleStream.sum() as in J/SpatialGraph.java:56-57,82-83. */
int  dge_graph_set_sources(dge_graph* g, const int32_t* v, int64_t n, int stream_sum);
/* Vertex ids [0, n) exist even when no edge names them (isolated vertices; the unregistered source vertices that
 * addSourceVertex creates for unknown names, J/LayeredGraph.java:182-183).  Before set_sources / build_alias. */
int  dge_graph_reserve_vertices(dge_graph* g, int32_t n);
/* Vertex.outDegree is a PUBLIC FIELD of the reference (J/LayeredGraph.java:35; assigned by J/SpatialGraph.java:33): a host that
 * keeps that field hands over its values (host double[n], n = vertex count) instead of the recomputed running sums.
 * After all edges; adding edges afterwards recomputes the sums. */
int  dge_graph_set_out_degree(dge_graph* g, const double* out_degree, int32_t n);
/* LayeredGraph.sourceWeightSum is a protected field that subclasses assign (J/SpatialGraph.java:57,83): fix it to the
 * host's value.  After dge_graph_set_sources (which recomputes it). */
int  dge_graph_set_source_weight_sum(dge_graph* g, double sum);
/* SpatialGraph.keepNearestKVertices(k)  J/SpatialGraph.java:29-35: stable sort by weight descending,
 * keep the first k, outDegree recomputed.  DGE_ERR_TOPK if some vertex has fewer than k edges. */
int  dge_graph_keep_top_k(dge_graph* g, int32_t k);
/* initiateAliasTables()  J/LayeredGraph.java:195-226 (+ Vertex.initiateAliasTable :54-82).
 * exact_reference_order = 1: the reference's own pairing order (bit-identical prob/alias arrays);
 * 0: Vose O(k) pairing — same sampling distribution, different alias indices (scalable form). */
int  dge_graph_build_alias(dge_graph* g, int exact_reference_order);
int  dge_graph_num_vertices(const dge_graph* g, int32_t* n);
int  dge_graph_num_edges(const dge_graph* g, int64_t* n);
/* Vertex.{probTable,aliasTable,edgesOut,outDegree} read-back for one vertex (J/LayeredGraph.java:31-37);
 * any output pointer may be null.  *k receives the degree even when cap is too small (DGE_ERR_CAP). */
int  dge_graph_get_alias(const dge_graph* g, int32_t v, double* prob, int32_t* alias, int32_t* nbr,
                         double* weight, int32_t cap, int32_t* k, double* out_degree);
/* the same for ALL vertices at once, in CSR order (edges of vertex v at [row_ptr[v], row_ptr[v+1]), insertion order kept):
 * what a host needs to fill every Vertex's probTable/aliasTable after initiateAliasTables(), or to rebuild edgesOut /
 * outDegree after keepNearestKVertices.  Host buffers; any output may be null; row_ptr has cap_vertices + 1 entries. */
int  dge_graph_get_csr(const dge_graph* g, int64_t* row_ptr, int32_t* nbr, double* weight, double* prob, int32_t* alias,
                       double* out_degree, int32_t cap_vertices, int64_t cap_edges);
/* LayeredGraph.{probTable,aliasTable,sourceVertices,sourceWeightSum}  J/LayeredGraph.java:145-148 */
int  dge_graph_get_source_alias(const dge_graph* g, double* prob, int32_t* alias, int32_t* src,
                                int32_t cap, int32_t* k, double* weight_sum);
/* Vertex.sampleNextVertex(double x)  J/LayeredGraph.java:123-132 (test overload); *next = -1 when the
 * vertex has no out-edges.  Runs the device sampler for one explicit x. */
int  dge_graph_sample_next(const dge_graph* g, int32_t v, double x, int32_t* next);

/* ------------------------------------------------------------------------------------------------
 * Walk sampler — replaces sampleVertexSequence() J/LayeredGraph.java:232-252 and the writer loops
 * J/CrossTimeGraph.java:134-140, J/SpatialGraph.java:103-113.
 *   rng_mode 0 ("java-sequential"): one java.util.Random(seed) stream consumed walk after walk, one
 *            nextDouble() per decision — what the reference produces after
 *            `LayeredGraph.rnd = new Random(seed)`.  first_index = draws already consumed.
 *   rng_mode 1 ("strided"): walk i owns draws [i*max_len, (i+1)*max_len) of that same stream.
 *   draws_consumed: mode 0 = draws this call took from the stream (add it to first_index for the next call);
 *            mode 1 = n_walks*max_len, the span of the stream the call owns.
 *            first_index = global index of the first walk (shards / batches).
 *            Both modes give identical walks on graphs where no walk dead-ends.
 * A dead end yields a shorter walk (pad -1), never an error (J/LayeredGraph.java:247-248).
 * ---------------------------------------------------------------------------------------------- */
int  dge_sample_walks(const dge_graph* g, int64_t n_walks, int32_t max_len, int64_t seed, int rng_mode,
                      int64_t first_index, int32_t* out /* host [n_walks*max_len] */, int64_t* draws_consumed);
int  dge_sample_walks_device(const dge_graph* g, int64_t n_walks, int32_t max_len, int64_t seed, int rng_mode,
                             int64_t first_index, dge_walks** out, int64_t* draws_consumed);
/* re-sample into an existing corpus: rows [row0, row0+n_walks) (strided mode only) */
int  dge_sample_walks_into(const dge_graph* g, dge_walks* w, int64_t row0, int64_t n_walks, int64_t seed,
                           int64_t first_index);
int  dge_walks_from_host(int device, const int32_t* walks, int64_t n_walks, int32_t max_len, dge_walks** out);
int  dge_walks_to_host(const dge_walks* w, int32_t* out, int64_t cap_elems);
/* d_ptr: the corpus in device memory, READ-ONLY for the caller: a trainer keeps what it derived from a corpus the library has not
 * written since (vocabulary rows, word offsets) */
int  dge_walks_info(const dge_walks* w, int64_t* n_walks, int32_t* max_len, const int32_t** d_ptr);
/* SpatialGraph's position prefix "j-name" (J/SpatialGraph.java:105-108): token j of each walk becomes
 * j*region_count + id, i.e. lands in layer j of the cross-time id space. */
int  dge_walks_add_position_prefix(dge_walks* w, int32_t region_count);
void dge_walks_free(dge_walks* w);

/* ------------------------------------------------------------------------------------------------
 * SGNS trainer — replaces new Word2Vec.Builder()...build(); w2v.fit()  J/DeepWalk.java:73-79
 * (DL4J-NLP 0.7.2 SkipGram + ND4J AggregateSkipGram; behaviour restated).
 * ---------------------------------------------------------------------------------------------- */
typedef struct dge_train_config {
    int32_t dim;             /* .layerSize(n)          J/DeepWalk.java:62-66,74 */
    int32_t window;          /* .windowSize(n)         J/DeepWalk.java:74 (= LayeredGraph.numLayer) */
    int32_t negative;        /* .negativeSample(5)     J/DeepWalk.java:75 */
    int32_t min_count;       /* .minWordFrequency(2)   J/DeepWalk.java:73 */
    int32_t epochs;          /* .iterations(1) x epochs(1)  J/DeepWalk.java:74 */
    int32_t workers;         /* .workers(8) J/DeepWalk.java:75.  1 = one in-order worker (deterministic);
                                0 = fill the device (Hogwild); n>1 = exactly n concurrent walk workers */
    float   alpha;           /* DL4J default learningRate 0.025 */
    float   min_alpha;       /* DL4J default minLearningRate 1e-4 */
    uint64_t seed;
    int64_t table_size;      /* unigram^0.75 table length; 0 -> 100000000 (word2vec.c) */
    int32_t n_vertices;      /* vertex-id space of the corpus */
    int32_t update_policy;   /* how concurrent workers update the tables (MI355X has 8 L2s that are not coherent):
                                0 = auto: 5 when the vocabulary has >= 131072 rows and its negative-sampling
                                    distribution is flat enough for lock attempts to succeed (expected failure
                                    rate < 0.4) and no single row is busy enough to serialise its pairs behind its lock
                                    (workers x p_row <= 0.5: such rows alone leave the locks, 7); 7 when a head of at most V/4 rows carries
                                    the skew; otherwise (a small vocabulary, a head too large, a block of a schedule of
                                    >= 2 ranks on a flat vocabulary) 8 when the tables are below 4 GiB and a synchronous mini-batch of >= 1e6 items
                                    (5e5 on rows of > 128 floats) keeps the busiest row below 4096 terms; else 2.  A vocabulary whose busiest row
                                    caps the workers below a quarter of the device (48 / its share of the tokens < 4096) runs under 2; the workers of any Hogwild launch are capped so that at most 96 of one row's updates are in flight (profiles/r05_hot_row_inflight.txt).
                                    The constants come from scripts/policy_sweep.py (profiles/r04_policy_sweep.txt);
                                1 = agent-scope row read-modify-write, write-through (last writer of a row wins);
                                2 = agent-scope loads + memory-side float atomics (no update is lost);
                                3 = plain cached accesses (debug only: every XCD trains a private stale copy);
                                5 = every row update under that row's commit lock, 16-byte write-through rows, relaxed
                                    commit (a re-lock can overtake the write-through: measured loss <= 4e-7 of the row
                                    updates at >= 65k rows; fastest);
                                6 = as 5 with strict commit (one returning atomic per stored 128-B line: no update
                                    is ever lost);
                                7 = as 5, but the head of the vocabulary (the rows many workers want at once; how many
                                    is derived from the counts) stays out of the lock protocol and takes atomics as in 2.
                                8 = owner-computes: every (context, target, label) term of a mini-batch becomes an item; items
                                    sorted by target row are applied by the row's owner in order, then sorted by context row and
                                    summed — no locks, no atomics, no lost update; the result is a deterministic function of the
                                    batch whatever the worker count (bit-exact against the oracle at full concurrency).  Within a
                                    mini-batch (~100 items per live row) the other table is read as it stood before the
                                    mini-batch, and the whole mini-batch trains at the learning rate of its first walk.  Tables below 4 GiB, no hierarchical softmax.
                                workers == 1 with policy 0/3/8 is the in-order schedule with plain accesses. */
    int32_t use_hs;          /* .useHierarchicSoftmax(b): 0 = negative sampling only (the north-star path);
                                1 = the hierarchical-softmax term as well, before the negatives of each pair — what
                                DL4J's builder leaves on when J/DeepWalk.java:73-76 does not call it.  Huffman codes over
                                the vocabulary counts (word2vec.c CreateBinaryTree), inner-node table syn1 [V-1 x dim].
                                Policies 0/2/3 only (in-order, or memory-side atomics). */
    int32_t reserved;        /* 0 */
} dge_train_config;

typedef struct dge_train_stats {
    int64_t pairs;           /* (center, context) pairs trained = "edges" of BASELINE.json's metric */
    int64_t words;           /* in-vocabulary tokens consumed */
    double  kernel_ms;       /* HIP-event time of the SGNS kernel launches, summed */
    double  walk_kernel_ms;  /* HIP-event time of the walk kernel launches issued through this model */
    int64_t launches;        /* SGNS kernel launches */
} dge_train_stats;

/* vocabulary pass (DL4J VocabConstructor.buildJointVocabulary): d_counts[v] += occurrences of v.
 * d_counts is device int64[n_vertices]; the caller may all-reduce it across ranks before
 * dge_model_create. */
int  dge_count_tokens(const dge_walks* w, int64_t row0, int64_t n_rows, int32_t n_vertices, int64_t* d_counts);
/* build vocabulary (count >= min_count, ordered by count desc, id asc), unigram table, sigmoid LUT and
 * initial weights (InMemoryLookupTable.resetWeights) */
int  dge_model_create(int device, const dge_train_config* cfg, const int64_t* d_counts, dge_model** out);
int  dge_model_set_stream(dge_model* m, void* hip_stream);
/* one pass of the trainer over corpus rows [row0, row0+n_rows).
 *   walk_index_base : global index of row0 within the epoch (RNG streams are keyed on it)
 *   epoch           : 0-based epoch number
 *   words_before    : in-vocab tokens trained before row0 in this epoch (learning-rate schedule)
 *   words_scale     : 1.0, or the number of ranks when ranks advance through the epoch in parallel */
int  dge_model_train(dge_model* m, const dge_walks* w, int64_t row0, int64_t n_rows, int64_t walk_index_base,
                     int32_t epoch, int64_t words_before, double words_scale, int64_t total_walks_per_epoch);
/* one-shot forms of w2v.fit(): vocabulary + all epochs */
int  dge_train_sgns(int device, const int32_t* walks /* host */, int64_t n_walks, int32_t max_len,
                    const dge_train_config* cfg, dge_model** out);
int  dge_train_sgns_device(const dge_walks* w, const dge_train_config* cfg, dge_model** out);
/* fused step used by bench.py: sample rows [row0,row0+n) of the corpus again from the graph (strided RNG)
 * and train on them, without leaving the device */
int  dge_model_walk_and_train(dge_model* m, const dge_graph* g, dge_walks* w, int64_t row0, int64_t n_rows,
                              int64_t walk_seed, int64_t walk_index_base, int32_t epoch, int64_t words_before,
                              double words_scale, int64_t total_walks_per_epoch);
/* results (w2v.lookupTable): host copies [V x dim], borrowed until the next call on m / dge_model_free */
int  dge_model_vectors(dge_model* m, const float** syn0, const int32_t** vocab_ids, int64_t* V, int32_t* dim);
int  dge_model_syn1neg(dge_model* m, const float** syn1neg);
/* use_hs: inner-node table [max(V-1,0) x dim]; and the Huffman paths of the vocabulary rows in CSR form —
 * offsets[V+1], points (inner-node rows, root first), codes (bit d of codes[r] = branch taken at points[offsets[r]+d]) */
int  dge_model_syn1(dge_model* m, const float** syn1, int64_t* rows);
int  dge_model_huffman(dge_model* m, const int64_t** offsets, const int32_t** points, const uint64_t** codes);
int  dge_model_counts(dge_model* m, const int64_t** counts);
int  dge_model_table(dge_model* m, const int32_t** table, int64_t* table_size);
int  dge_model_stats(const dge_model* m, dge_train_stats* out);
/* Diagnostic: how fast this model's memory answers the four things the lock kernels do to it, measured on the model's stream (tables below
   4 GiB; the rewrite leaves every value as it was; any output may be NULL): GB/s of rows read at random, GB/s (read + written) of rows read and
   stored back write-through, exchanges per second on random lock words, look-ups per second in the unigram table.  Consecutive processes on one box — and two models of one process —
   differ by up to 15 % in training speed with the device's copy rate unchanged (profiles/r02_box_drift.txt); the difference follows the
   allocation and shows here without training anything. */
int  dge_model_row_rates(dge_model* m, double* read_gb_per_s, double* rewrite_gb_per_s, double* lock_exchanges_per_s, double* table_lookups_per_s);
int  dge_model_reset_stats(dge_model* m);
/* Where the tables lie.  Allocations of hundreds of megabytes fall into discrete classes of memory, up to 15 % apart in how fast random rows can be read
   and written back in them (profiles/r03_placement.txt, last block); dge_model_create therefore takes every table of 64 MB ... 2 GiB (syn1neg first) from
   the best of up to 32 candidates — hipMalloc and virtual-memory allocations in turn, all held until the choice — under a 2-ms read-modify-write probe, and
   stops as soon as one candidate is 14 % above the slowest seen.  This reports, for table 0 (syn0), 1 (syn1neg) or 2 (syn1), how many candidates were
   probed and the best (= the one kept) and worst probe rate in GB/s (0 candidates: the table was small or of 2 GiB and more, or the probe could not run). */
int  dge_model_table_placement(const dge_model* m, int32_t table, int32_t* candidates, double* best_gb_per_s, double* worst_gb_per_s);
/* The negative-sampling table's run form.  Rows are ordered by count; the rows of equal count form runs (up to 2 046 of them, counted from the vocabulary's tail; the head rows in front keep the table), the
   slot -> row map of word2vec's unigram^0.75 table is that many straight segments, and the lock kernels compute a negative's row from the run arrays in
   LDS instead of reading the table (5 requests to memory a pair less; profiles/r03_shape_sweep.txt).  dge_model_create compares the closed form with
   the table slot by slot and keeps it only if at most 64 slots differ (those are listed and looked up): the rows drawn are the table's, bit for bit.
   n_runs = 0: this model has no run form (a skewed vocabulary, or too many exceptions). */
int  dge_model_table_runs(const dge_model* m, int32_t* n_runs, int32_t* n_exceptions);
/* Placement search (profiles/r03_placement.txt): which physical memory the allocator handed each of the model's large arrays decides a launch's
   duration by up to 15 %, array by array, under no rule that could be asked for.  Trains rows [row0, row0 + n_rows) of w as a probe; then for
   the negative-sampling table, the lock words, syn1neg and syn0 in turn up to candidates - 1 copies in fresh memory are tried and the faster
   placement kept.  Tables, counters and statistics are saved before and restored after: the model trains exactly as an untuned one.  The arrays are gone
   through in passes until a pass moves nothing (at most six): 2 + 4 (candidates - 1) probe launches per pass, transiently 2 x the tables' memory.  ms_before / ms_after: probe launch before and after (may be NULL). */
int  dge_model_tune_placement(dge_model* m, const dge_walks* w, int64_t row0, int64_t n_rows, int32_t candidates, double* ms_before, double* ms_after,
                              int32_t* arrays_moved);
/* Whether the search has run on this model (the one-shot dge_train_sgns_device runs it only where it can pay: when a tenth of the projected training,
   epochs x walks, exceeds one pass of probes) and its latest report: probe launch before / after (ms), arrays moved. */
int  dge_model_placement_search(const dge_model* m, int32_t* runs, double* ms_before, double* ms_after, int32_t* arrays_moved);
/* what the latest training launch resolved `update_policy` 0 / `workers` 0 to: the policy that ran (0 = in-order plain),
 * the concurrent workers, and for policy 7 the head rows kept out of the lock protocol */
int  dge_model_schedule(const dge_model* m, int32_t* update_policy, int64_t* workers, int32_t* hot_rows);
/* ... and the trainer kernel (name and form) that launch ran, as text: e.g. "k_sgns_train_locked<relaxed>", "k_sgns_train_hsw<negatives under commit locks, 7 waves> (...)",
 * "k_sorted_phase (owner-computes: ...)".  The policy number alone does not say it (hierarchical softmax has three kernels). */
int  dge_model_kernel(const dge_model* m, char* buf, int32_t cap);
/* One block of the multi-GPU schedule under the lock kernels (k_sgns_train_locked<.., PART>), since the last dge_model_reset_stats: pairs that were put back because
 * their context row's lock was taken, lock rounds that left some wanted row unwon, and lock rounds in all — what a block's speed runs against (a block's live rows are
 * V / N per table: locks collide N times as often as on one GPU).  The one-GPU kernels do not count (they have no register to spare). */
int  dge_model_lock_stats(const dge_model* m, int64_t* pairs_put_back, int64_t* rounds_short, int64_t* rounds);
/* Blocking waits the library has made in this process so far — stream / device / event synchronisations and blocking copies, counted at every call site.  An episode
 * of the multi-GPU block schedule makes none once its buffers exist (tests/test_gpu_distributed.py counts them); a global batch makes two (the item store's sizes). */
int  dge_host_sync_count(int64_t* n);
/* WordVectorSerializer.writeWordVectors(w2v, path)  J/DeepWalk.java:82: "name v1 .. vD\n" per vocabulary
 * row, no header (header != 0 writes the LINE-style "V D" first line of miscs/taxi_all.txt:1).
 * names[v] is the string of vertex id v; null -> the decimal id. */
int  dge_write_vec(dge_model* m, const char* const* names, const char* path, int header);
void dge_model_free(dge_model* m);

/* ------------------------------------------------------------------------------------------------
 * Multi-GPU block schedule (new; the reference is single-host).  Rows are split by row % n_parts.  With
 * a partition set, dge_model_train / dge_model_walk_and_train only train the pairs whose context row
 * (syn0) lies in ctx_part and whose centre row (syn1neg) lies in tgt_part, and move each drawn negative
 * to the row of tgt_part nearest below it (rows are ordered by count: the frequency rank is kept).  Rank g of
 * N runs episodes e = 0..N-1 with (ctx_part, tgt_part) = (g, (g+e) % N) over the same global batch of
 * walks: the blocks of one episode are row-disjoint, after N episodes every pair was trained once.
 * After an episode a rank hands the syn1neg partition it trained to rank g-1, which trains it next (export -> point-to-point
 * transfer -> import: a ring); syn0 partitions never leave their rank until the final gather.  n_parts <= 1 switches the filter off.
 * Policies under a partition: 0 (auto), 2, 3, 5, 8, and 7 = row locks on syn1neg's tail only, the pair's syn0 row by atomics.  Auto keeps the head / tail
 * split of policy 7 inside a block on skewed vocabularies (the block's own head, derived from the counts with the block's collision rate).
 * With use_hs (policies 0 / 2 / 3): inner-node rows are split by node % n_parts too; a block visits EVERY centre for the inner nodes of its path that lie in
 * tgt_part, the negative-sampling terms of a pair stay with the block of its centre's partition; syn1 partition p travels with syn1neg partition p. */
int  dge_model_set_partition(dge_model* m, int32_t n_parts, int32_t ctx_part, int32_t tgt_part);
/* floats of one packed partition buffer: ceil(V / n_parts) rows x row stride (same for every partition) */
int  dge_model_partition_floats(const dge_model* m, int32_t n_parts, int64_t* n_floats);
/* table: 0 = syn0, 1 = syn1neg, 2 = syn1 (use_hs); d_buf is device memory of dge_model_partition_floats floats */
int  dge_model_export_partition(dge_model* m, int table, int32_t n_parts, int32_t part, float* d_buf);
int  dge_model_import_partition(dge_model* m, int table, int32_t n_parts, int32_t part, const float* d_buf);
/* The same, STREAM-ORDERED: the host never waits.  export: the pack kernel runs on the model's stream and `consumer_stream` (a hipStream_t; NULL = the legacy default
 * stream) is made to wait for it with an event — whatever the caller enqueues there afterwards (ncclSend, a copy) sees the packed buffer.  import: the model's stream
 * is made to wait for everything `producer_stream` holds at the time of the call (the transfer that fills d_buf), then unpacks.  Passing the model's own stream
 * (dge_model_stream) skips the handshake: a host that enqueues its RCCL calls on that stream needs no event at all (dge_model_ring_pass does exactly that).
 * d_buf must stay untouched by the caller until the model's stream has passed the copy. */
int  dge_model_export_partition_async(dge_model* m, int table, int32_t n_parts, int32_t part, float* d_buf, void* consumer_stream);
int  dge_model_import_partition_async(dge_model* m, int table, int32_t n_parts, int32_t part, const float* d_buf, void* producer_stream);
int  dge_model_stream(const dge_model* m, void** hip_stream);

/* ------------------------------------------------------------------------------------------------
 * Multi-GPU exchange at epoch boundaries (new; the reference is single-host).  Each rank trains its
 * walk shard from a common snapshot; delta = current - snapshot is summed across ranks (RCCL
 * all-reduce on the caller's communicator, e.g. torch.distributed) and applied with a scale.
 * ---------------------------------------------------------------------------------------------- */
int  dge_model_sync_size(const dge_model* m, int64_t* n_floats);          /* 2 * V * row_stride */
int  dge_model_snapshot(dge_model* m);                                     /* snapshot = current (start of a shard) */
int  dge_model_export_delta(dge_model* m, float* d_buf);                  /* d_buf = current - snapshot */
int  dge_model_import_delta(dge_model* m, const float* d_buf, float scale); /* current = snapshot + scale*d_buf; re-snapshot */

/* The same exchange with RCCL called by the library itself (hosts without torch.distributed, e.g. the JNI form):
 * rank 0 obtains an id, the host application hands it to the other ranks (any channel: file, socket, MPI), every rank
 * creates its communicator, then dge_model_allreduce_deltas replaces the export/all-reduce/import triple.
 * librccl is loaded lazily (dlopen) on the first of these calls. */
typedef struct dge_comm dge_comm;
typedef struct dge_unique_id { char bytes[128]; } dge_unique_id;          /* = ncclUniqueId */
int  dge_comm_unique_id(dge_unique_id* out);
int  dge_comm_create(dge_comm** out, const dge_unique_id* id, int rank, int nranks, int device);
void dge_comm_free(dge_comm* c);
int  dge_model_allreduce_deltas(dge_model* m, dge_comm* c);
/* block schedule with RCCL called from the library (N = nranks).  dge_model_ring_pass: after episode `episode` rank g hands the
 * syn1neg partition it just trained, (g + episode) % N, to rank g-1 and takes (g + 1 + episode) % N — the one it trains next — from
 * rank g+1 (ncclSend / ncclRecv; with use_hs the syn1 partition of the same number in the same transfer).  dge_model_gather_table: every rank publishes partition `rank` of `table` (0 = syn0, 1 = syn1neg, 2 = syn1)
 * and takes the others (all-gather): the end of training.  NOT YET RUN ON MORE THAN ONE GPU (README.md: verification status). */
int  dge_model_ring_pass(dge_model* m, dge_comm* c, int32_t episode);
int  dge_model_gather_table(dge_model* m, dge_comm* c, int table);

/* ------------------------------------------------------------------------------------------------
 * Quality metric ("next" row of the scope table): pairwiseEstimator of P/embeddingEvaluation_tract.py:169-196 — for every
 * row of features [n x dim] the k other rows nearest in cosine distance (1 - cos; a zero vector is at distance 2 from
 * everything), ascending, smaller index first among equals.  Exact-f32 MFMA tiles fused with the top-k selection.
 * dim <= 256, k <= 64.  Host buffers.
 * ---------------------------------------------------------------------------------------------- */
int  dge_knn_cosine(int device, const float* features, int32_t n, int32_t dim, int32_t k, int32_t* out_idx, float* out_dist,
                    double* ms_kernel);
/* ndcg_atK of P/embeddingEvaluation_tract.py:249-260 wholly on the device: the k nearest neighbours of every region in `features`
 * [n x dim] are scored with relevance 1 - (cosine distance in gnd_features [n x gnd_dim]), DCG = sum_i relv_i / log2(i + 1), normalised
 * by the DCG of the ground features' own k nearest; *ndcg = mean over the n regions (rows of the two arrays are the same regions). */
int  dge_ndcg_at_k(int device, const float* features, int32_t dim, const float* gnd_features, int32_t gnd_dim, int32_t n, int32_t k,
                   double* ndcg, double* ms_kernels);

/* ------------------------------------------------------------------------------------------------
 * Ablation / test knobs of the trainer (process-wide relaxed atomics; nothing in a normal run sets them).  value < 0 puts
 * a knob back to the library's own rule.
 * ---------------------------------------------------------------------------------------------- */
enum {
    DGE_TUNE_HOT_ROWS = 0,        /* update_policy 7: head rows [0, value) stay out of the lock protocol instead of the count-derived head */
    DGE_TUNE_HS_DRAIN = 1,        /* hierarchical softmax: additions between drains of an LDS accumulator (default 64) */
    DGE_TUNE_FORCE_SEGMENTS = 2,  /* > 0: address the tables through per-segment descriptors as tables of >= 4 GiB are (parity tests) */
    DGE_TUNE_SEGMENT_SHIFT = 3,   /* rows per descriptor segment = 2^value (with FORCE_SEGMENTS: many segments on a small table) */
    DGE_TUNE_SORTED_CHUNK = 4,    /* update_policy 8: items per work unit (default 128); a row's item list longer than what is left of a unit is split */
    DGE_TUNE_SORTED_WALKS = 5,    /* update_policy 8: walks per synchronous mini-batch (default: as many as the item buffers hold) */
    DGE_TUNE_WORKERS = 6,         /* workers = 0 (fill the device): this many concurrent walks instead of the count the library derives */
    DGE_TUNE_STATIC_WALKS = 7,    /* > 0: the lock kernels' worker w trains walks w, w + workers, ... instead of taking them from a launch-wide counter */
    DGE_TUNE_HS_COLD = 8,         /* hierarchical softmax under atomics: inner nodes [0, value) take plain read-modify-write (default: those on < 2e-5 of the paths — derived from the COUNTS the model was created with: they must describe the corpus that is trained, or updates of nodes that are busier than their count says are lost; 0 = every node by atomics) */
    DGE_TUNE_HS_WAVE = 9,         /* hierarchical softmax under atomics: 0 = the workers issue their atomics themselves, 1 = through the workgroup's atomics wave (default: the wave from 65 536 rows on) */
    DGE_TUNE_ACC_ROWS = 10,       /* update_policy 7: the hottest rows [0, value) add their syn1neg updates up in LDS (one set of atomics per DGE_TUNE_ACC_DRAIN updates; the kernel caps the value at what its LDS holds, 8 .. 16); one GPU: default 0 = none (measured: 2-4 %, and from 16 updates a flush on it shifts the trained scores).
                                     One block of the multi-GPU schedule (v106): the partition's hottest `value` rows of BOTH tables (a bank each; slot = the row's rank inside the partition); default 16 where the busiest row's chain of atomics is long against the block, else 0; 0 = off, > 0 = on whatever the chain */
    DGE_TUNE_ACC_DRAIN = 11,      /* updates of such a row between two flushes (one GPU: default 16; a block: default 2048 / workgroups of the launch = 4 — 16 diverges there, measured) */
    DGE_TUNE_TABLE_RUNS = 12,     /* the negative-sampling table's run form (dge_model_table_runs): 0 = not built / not used (the lock kernels read the table), N > 0 = built from at most N runs of the vocabulary's tail (tests: the head rows in front stay on the table); default: up to 2 046 runs */
    DGE_TUNE_BLOCK_SYN0_FREE = 13, /* block schedule, mixed lock kernel: 1 = the pair's syn0 row is never locked (agent-scope read, atomics), 0 = it is locked unless it is a head row; default: the library's rule */
    DGE_TUNE_HS_CENTRE = 14,      /* hierarchical softmax under atomics: 0 = pair by pair (k_sgns_train), 1 = a wave per centre wherever it applies (rows of up to 128 floats, walks of up to 64 tokens; k_sgns_train_hsw); 2 = that kernel with the pair's negatives and the centre's gathered syn1neg update under the rows' commit locks instead of atomics, 3 = the same in workgroups of seven training waves (one a compute unit) that share their LDS accumulators; default: a wave per centre from 65 536 vocabulary rows on — form 3 where update_policy 0 would pick the commit locks for the negative-sampling kernels, form 1 elsewhere */
    DGE_TUNE_HS_HOT_KB = 15,      /* k_sgns_train_hsw: > 0 = keep the inner nodes next to the root in that many KB of LDS accumulators per workgroup (drained every DGE_TUNE_HS_DRAIN additions; round 4's first form: faster by a tenth, staler) instead of the default, the busiest nodes in copies (nothing parked; the root's number of copies: DGE_TUNE_HS_COPIES).  Capped at 100 (seven-wave workgroups) / 30 (three-wave) */
    DGE_TUNE_ALLOW_UNSAFE = 16,   /* > 0: a FORCED update_policy runs even where the library would refuse it (8 on a vocabulary whose busiest row would take > 8192 terms of one mini-batch; 5 / 6 where workers x the busiest row's share > 2) — tests of the watchdog, reproductions of the failure */
    DGE_TUNE_WATCHDOG_MS = 17,    /* the lock kernels' watchdog: a worker still waiting for a row lock after this many milliseconds of the launch gives up (dge_model_stats then returns DGE_ERR_STATE); 0 = no watchdog; default: 5 s + 100 x the launch's bytes at the 8 TB/s roofline */
    DGE_TUNE_HS_COPIES = 18,      /* k_sgns_train_hsw, copies form: the root's number of copies (a node's copies = ceil(its share of the paths x value), at most 16); default 16 */
    DGE_TUNE_SMALL_ROWS = 19,     /* rows of 17 .. 32 floats under the atomics policy: 1 = k_sgns_train_small (32 lanes a worker, a row = one request), 0 = k_sgns_train's 16-lane groups; default: the small-row kernel wherever it applies (v106) */
    DGE_TUNE_COUNT = 20
};
int  dge_set_tuning(int32_t knob, int64_t value);
int  dge_get_tuning(int32_t knob, int64_t* value);   /* -1 = the library's own rule */

/* ------------------------------------------------------------------------------------------------
 * Device self-test of the commit-lock protocol of update_policy 5 (new; no reference counterpart): n_workers groups
 * each do `iters` rounds of "lock 5 pseudo-random rows of an n_rows x 128 table, add 1.0 to every element, unlock".
 * Returns the number of row increments performed and the largest |element - increments of its row| (0 when no
 * update was lost).
 * ---------------------------------------------------------------------------------------------- */
int  dge_selftest_locked_rows(int device, int32_t n_rows, int64_t n_workers, int32_t iters, uint64_t seed,
                              int32_t commit /* 0 relaxed (policy 5), 1 strict (policy 6), 2 agent release fence */,
                              int64_t* total_increments, double* max_abs_error);
/* The atomics wave of the mixed kernels in isolation, LDS accumulators of the n_acc hottest rows included: `blocks` workgroups x 12 workers x `iters`
   messages of 5 rows each, half of them among the 8 hottest; every element of a row must end at the number of times the row was posted. */
int  dge_selftest_atomics_wave(int device, int32_t n_rows, int32_t n_acc, int32_t drain, int32_t blocks, int32_t iters, uint64_t seed,
                               int64_t* total_updates, double* max_abs_error);
/* the same with the rows of ONE BLOCK of a div-rank schedule (v106): messages alternate between the two tables' accumulator banks, a row's slot is its rank
   inside its partition (row / div); div = 1 is the call above */
int  dge_selftest_atomics_wave_block(int device, int32_t n_rows, int32_t n_acc, int32_t drain, int32_t div, int32_t blocks, int32_t iters, uint64_t seed,
                                     int64_t* total_updates, double* max_abs_error);
/* the LDS combining of the hierarchical-softmax updates near the root (hot_add) in isolation: n_workers workers add 1.0
 * to skewed pseudo-random rows `iters` times with the given drain period; max_abs_error = worst |row element - additions
 * that row received| (0 when no addition is lost or doubled). */
/* host code only: the vector file's number formatter (csrc/fmt_g9.h: printf's "%.9g" by integer arithmetic) against snprintf on n pseudo-random floats;
 * *fast_path = how many took the formatter (the rest are outside its range and go through std::to_chars in dge_write_vec), *mismatches must come back 0 */
int  dge_selftest_fmt_g9(int64_t n, uint64_t seed, int64_t* fast_path, int64_t* mismatches);
int  dge_selftest_hot_add(int device, int32_t n_hot, int64_t n_workers, int32_t iters, int32_t drain, uint64_t seed,
                          int64_t* total_additions, double* max_abs_error);

#ifdef __cplusplus
}
#endif
#endif
