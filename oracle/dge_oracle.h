/*
 * dge_oracle.h — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference's random-walk + SGNS hot path.
 * It is NOT part of the product: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The product (libdge.so) never links,
 * loads or calls anything in oracle/.
 *
 * Parity status:
 *   - walk half  : PINNED by T/LayeredGraphTest.java:12-44 (alias build + 5 draws)
 *                  and by the public java.util.Random spec KATs.
 *   - SGNS half  : PARITY UNPINNED.  The arithmetic lives in the un-vendored
 *                  org.deeplearning4j:deeplearning4j-nlp:0.7.2 / org.nd4j:nd4j-native:0.7.2
 *                  (embedding/pom.xml:14-16,42-51); no reference test holds a vector for it.
 *                  What is restated is the published word2vec skip-gram negative-sampling
 *                  update with DL4J's pair enumeration.
 *
 * Reference citations use J/ = embedding/src/main/java/embedding/.
 */
#ifndef DGE_ORACLE_H
#define DGE_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------- java.util.Random (public spec; J/LayeredGraph.java:14 uses it) ---------- */
typedef struct { uint64_t s; } orc_jrand;
void     orc_jrand_seed(orc_jrand* r, int64_t seed);
int32_t  orc_jrand_next(orc_jrand* r, int bits);
int32_t  orc_jrand_next_int(orc_jrand* r);
double   orc_jrand_next_double(orc_jrand* r);
void     orc_jrand_jump(orc_jrand* r, uint64_t n_lcg_steps);

/* ---------- edge store + alias tables + walks (J/LayeredGraph.java) ---------- */
typedef struct orc_graph orc_graph;
orc_graph* orc_graph_create(void);
void  orc_graph_free(orc_graph* g);
int   orc_graph_add_edges(orc_graph* g, const int32_t* src, const int32_t* dst, const double* w, int64_t n);
int   orc_graph_set_sources(orc_graph* g, const int32_t* v, int64_t n, int stream_sum);
int   orc_graph_keep_top_k(orc_graph* g, int32_t k);
/* hosts that keep the reference's public fields hand them over: vertex count incl. isolated vertices
 * (J/LayeredGraph.java:182-183), Vertex.outDegree (:35), sourceWeightSum (:146; J/SpatialGraph.java:57,83) */
int   orc_graph_reserve_vertices(orc_graph* g, int32_t n);
int   orc_graph_set_out_degree(orc_graph* g, const double* od, int32_t n);
int   orc_graph_set_source_weight_sum(orc_graph* g, double s);
int   orc_graph_get_csr(const orc_graph* g, int64_t* row_ptr, int32_t* nbr, double* weight, double* prob, int32_t* alias, double* out_degree);
int   orc_graph_build_alias(orc_graph* g, int exact_reference_order);
int32_t orc_graph_num_vertices(const orc_graph* g);
int64_t orc_graph_num_edges(const orc_graph* g);
int   orc_graph_get_alias(const orc_graph* g, int32_t v, double* prob, int32_t* alias, int32_t* nbr,
                          double* weight, int32_t cap, int32_t* k, double* out_degree);
int   orc_graph_get_source_alias(const orc_graph* g, double* prob, int32_t* alias, int32_t* src,
                                 int32_t cap, int32_t* k, double* weight_sum);
int   orc_graph_sample_next(const orc_graph* g, int32_t v, double x, int32_t* next);
/* rng_mode 0: one java.util.Random(seed) stream consumed walk after walk (first_index = draws
 *             already consumed); rng_mode 1: walk i owns draws [i*max_len,(i+1)*max_len). */
int   orc_sample_walks(const orc_graph* g, int64_t n_walks, int32_t max_len, int64_t seed, int rng_mode,
                       int64_t first_index, int32_t* out, int64_t* draws_consumed);

/* ---------- SGNS trainer (behaviour of DL4J Word2Vec.fit(), J/DeepWalk.java:73-79) ---------- */
typedef struct orc_train_config {
    int32_t dim;             /* layerSize            J/DeepWalk.java:62-66,74 */
    int32_t window;          /* windowSize           J/DeepWalk.java:74       */
    int32_t negative;        /* negativeSample       J/DeepWalk.java:75       */
    int32_t min_count;       /* minWordFrequency     J/DeepWalk.java:73       */
    int32_t epochs;          /* epochs x iterations  J/DeepWalk.java:74       */
    int32_t threads;         /* workers (1 = sequential, deterministic)  J/DeepWalk.java:75 */
    float   alpha;           /* DL4J default learningRate 0.025 */
    float   min_alpha;       /* DL4J default minLearningRate 1e-4 */
    uint64_t seed;
    int64_t table_size;      /* unigram^0.75 table length (word2vec.c: 1e8) */
    int32_t arith;           /* 0 = word2vec.c order (sequential dot, unfused mul+add)
                                1 = HIP lane order (lane j owns elements 64c+16m+j; fmaf; xor tree)   */
    int32_t n_vertices;      /* vertex-id space of the walks (ids in [0,n_vertices)) */
    int64_t walk_index_base; /* global index of walks[0] (multi-rank sharding) */
    int64_t total_walks;     /* global number of walks per epoch (0 = n_walks) */
    int64_t total_words;     /* global in-vocab tokens per epoch (0 = count locally) */
    int64_t words_before;    /* in-vocab tokens of walks before walk_index_base */
    int32_t use_hs;          /* hierarchical-softmax term as well (DL4J's default when the builder does not disable it,
                                J/DeepWalk.java:73-76; word2vec.c -hs 1): Huffman codes over the counts, table syn1 */
    int32_t part_n;          /* > 1: the multi-GPU block schedule (include/dge.h, dge_model_set_partition) run sequentially —
                                per epoch, episodes e = 0..N-1, ranks g = 0..N-1: the block (contexts in partition g,
                                centres and negatives in partition (g+e) % N) over all walks */
    int32_t sorted_chunk;    /* > 0: the OWNER-COMPUTES schedule of update_policy 8 (embedding_amd/csrc/sgns_sorted.hip) restated
                                sequentially: per mini-batch every (context, target, label) term becomes an item; items sorted by target
                                row, a row's items applied in order by chunks of sorted_chunk items (a row that straddles chunk
                                borders: independent segments from the same starting row, deltas added in chunk order), the step g
                                of every item kept; items sorted by context row, every context row takes the sum of g * target row.
                                Lane order of the 16-byte-per-lane row layout.  A mini-batch trains at the learning rate of its first walk.
                                Deterministic whatever the worker count. */
    int32_t sorted_walks;    /* walks per synchronous mini-batch (0 = all walks of the launch) */
} orc_train_config;

typedef struct orc_model orc_model;
int   orc_train_sgns(const int32_t* walks, int64_t n_walks, int32_t max_len,
                     const orc_train_config* cfg, orc_model** out);
/* same, continuing from a given state: counts [n_vertices] of the whole corpus (vocabulary, unigram table, total_words) and the
 * tables reached so far, rows in the vocabulary's order (count desc, vertex id asc); any of the three may be NULL */
int   orc_train_sgns_from(const int32_t* walks, int64_t n_walks, int32_t max_len, const orc_train_config* cfg,
                          const int64_t* counts, const float* syn0_init, const float* syn1neg_init, orc_model** out);
int   orc_train_sgns_from_hs(const int32_t* walks, int64_t n_walks, int32_t max_len, const orc_train_config* cfg,
                             const int64_t* counts, const float* syn0_init, const float* syn1neg_init, const float* syn1_init, orc_model** out);
int64_t orc_model_vocab_size(const orc_model* m);
int32_t orc_model_dim(const orc_model* m);
const float*   orc_model_syn0(const orc_model* m);      /* [V x dim] */
const float*   orc_model_syn1neg(const orc_model* m);   /* [V x dim] */
const float*   orc_model_syn1(const orc_model* m);      /* [V-1 x dim] inner nodes of the Huffman tree (use_hs), else null */
/* word2vec.c CreateBinaryTree over counts sorted descending: codelen[V], points/codes [V x 40] */
int orc_huffman(const int64_t* counts, int64_t V, int32_t* codelen, int32_t* points, uint8_t* codes);
int32_t orc_model_code(const orc_model* m, int64_t row, int32_t* points, uint8_t* codes, int32_t cap);   /* code length of a word */
const int32_t* orc_model_vocab_ids(const orc_model* m); /* vertex id of row r */
const int64_t* orc_model_counts(const orc_model* m);    /* token count of row r */
const int32_t* orc_model_table(const orc_model* m);     /* [table_size] */
int64_t orc_model_pairs(const orc_model* m);
int64_t orc_model_total_words(const orc_model* m);
double  orc_model_seconds(const orc_model* m);          /* wall time of the training loop only */
void  orc_model_free(orc_model* m);
float orc_exp_table(int i);                              /* sigmoid LUT entry (i in [0,1000)) */
uint64_t orc_mix64(uint64_t x);
/* 1: the trainer runs the plain word2vec.c-shaped loop (train_walk, the definition); 0 (default): the same floating-point operations with the dot products of
 * a pair's distinct rows taken side by side (train_walk_ilp) — bit-identical tables, ~3x the speed; tests/test_oracle_kats.py compares the two */
void  orc_set_plain(int on);

#ifdef __cplusplus
}
#endif
#endif
