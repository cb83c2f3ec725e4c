/*
 * dge_oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE ONLY (see dge_oracle.h header).
 *
 * Walk half  : follows J/LayeredGraph.java line by line (citations at each function).
 *              PINNED by T/LayeredGraphTest.java:12-44 and java.util.Random spec KATs.
 * SGNS half  : PARITY UNPINNED — third-party DL4J-NLP 0.7.2 / ND4J-native 0.7.2 arithmetic is not
 *              under /root/reference; this restates the published word2vec.c skip-gram
 *              negative-sampling update with DL4J's pair enumeration.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -mavx2 -mfma -fopenmp).
 * The SGNS pair loop is here twice: train_walk (word2vec.c's shape: the definition) and train_walk_ilp (the same operations, a pair's dot products side by side:
 * what runs by default; orc_set_plain(1) selects the former; tests/test_oracle_kats.py holds the two bit-identical).
 */
#include "dge_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* =====================================================================================
 * java.util.Random — public specification (used at J/LayeredGraph.java:14,108,234)
 * ===================================================================================== */
#define JR_MULT 0x5DEECE66DULL
#define JR_ADD  0xBULL
#define JR_MASK ((1ULL << 48) - 1)

void orc_jrand_seed(orc_jrand* r, int64_t seed) { r->s = ((uint64_t)seed ^ JR_MULT) & JR_MASK; }

int32_t orc_jrand_next(orc_jrand* r, int bits) {
    r->s = (r->s * JR_MULT + JR_ADD) & JR_MASK;
    return (int32_t)((int64_t)r->s >> (48 - bits));   /* state < 2^48, so plain shift == Java's >>> */
}
int32_t orc_jrand_next_int(orc_jrand* r) { return orc_jrand_next(r, 32); }

double orc_jrand_next_double(orc_jrand* r) {
    int64_t hi = (int64_t)orc_jrand_next(r, 26);
    int64_t lo = (int64_t)orc_jrand_next(r, 27);
    return (double)((hi << 27) + lo) * 0x1.0p-53;
}

/* advance the LCG by n steps in O(log n): affine-map composition mod 2^48 */
void orc_jrand_jump(orc_jrand* r, uint64_t n) {
    uint64_t acc_m = 1, acc_p = 0, cur_m = JR_MULT, cur_p = JR_ADD;
    while (n) {
        if (n & 1) { acc_m = acc_m * cur_m; acc_p = acc_p * cur_m + cur_p; }
        cur_p = (cur_m + 1) * cur_p;
        cur_m = cur_m * cur_m;
        n >>= 1;
    }
    r->s = (acc_m * r->s + acc_p) & JR_MASK;
}

/* =====================================================================================
 * Edge store (J/LayeredGraph.java:17-49,142-189)
 * Vertex ids are the caller's insertion ordinals (J/LayeredGraph.java:160,166); name<->id
 * interning stays on the host side of the boundary.
 * ===================================================================================== */
struct orc_graph {
    /* COO in insertion order (allEdges, :142) */
    int64_t n_edges, cap_edges;
    int32_t *src, *dst; double* w;
    int32_t n_vertices;
    /* CSR, edges of a vertex kept in insertion order (edgesOut, :34) */
    int built;
    int64_t* row_ptr; int32_t* nbr; double* wt;
    double* out_degree;            /* running sum of weights in insertion order (:46-49) */
    double* prob; int32_t* alias;  /* per-vertex alias tables, concatenated by row_ptr */
    /* sources (:145-148) */
    int64_t n_src; int32_t* srcv; double src_weight_sum; double* src_prob; int32_t* src_alias;
    int src_stream_sum, src_sum_fixed;
    int alias_built;
};

orc_graph* orc_graph_create(void) { return (orc_graph*)calloc(1, sizeof(orc_graph)); }

void orc_graph_free(orc_graph* g) {
    if (!g) return;
    free(g->src); free(g->dst); free(g->w); free(g->row_ptr); free(g->nbr); free(g->wt);
    free(g->out_degree); free(g->prob); free(g->alias); free(g->srcv); free(g->src_prob); free(g->src_alias);
    free(g);
}

/* bulk form of addEdge (J/LayeredGraph.java:157-174); duplicates are NOT merged */
int orc_graph_add_edges(orc_graph* g, const int32_t* src, const int32_t* dst, const double* w, int64_t n) {
    if (!g || n < 0) return 1;
    if (g->n_edges + n > g->cap_edges) {
        int64_t nc = (g->n_edges + n) * 2 + 16;
        g->src = (int32_t*)realloc(g->src, nc * sizeof(int32_t));
        g->dst = (int32_t*)realloc(g->dst, nc * sizeof(int32_t));
        g->w   = (double*)realloc(g->w, nc * sizeof(double));
        g->cap_edges = nc;
    }
    for (int64_t i = 0; i < n; i++) {
        if (src[i] < 0 || dst[i] < 0) return 2;
        g->src[g->n_edges] = src[i]; g->dst[g->n_edges] = dst[i]; g->w[g->n_edges] = w[i];
        g->n_edges++;
        if (src[i] + 1 > g->n_vertices) g->n_vertices = src[i] + 1;
        if (dst[i] + 1 > g->n_vertices) g->n_vertices = dst[i] + 1;
    }
    g->built = 0; g->alias_built = 0;
    return 0;
}

static void build_csr(orc_graph* g) {
    if (g->built) return;
    int32_t V = g->n_vertices; int64_t E = g->n_edges;
    free(g->row_ptr); free(g->nbr); free(g->wt); free(g->out_degree);
    g->row_ptr = (int64_t*)calloc((size_t)V + 1, sizeof(int64_t));
    g->nbr = (int32_t*)malloc((size_t)(E ? E : 1) * sizeof(int32_t));
    g->wt  = (double*)malloc((size_t)(E ? E : 1) * sizeof(double));
    g->out_degree = (double*)calloc((size_t)(V ? V : 1), sizeof(double));
    for (int64_t e = 0; e < E; e++) g->row_ptr[g->src[e] + 1]++;
    for (int32_t v = 0; v < V; v++) g->row_ptr[v + 1] += g->row_ptr[v];
    int64_t* fill = (int64_t*)malloc((size_t)(V ? V : 1) * sizeof(int64_t));
    for (int32_t v = 0; v < V; v++) fill[v] = g->row_ptr[v];
    for (int64_t e = 0; e < E; e++) {             /* stable: insertion order inside a vertex */
        int64_t p = fill[g->src[e]]++;
        g->nbr[p] = g->dst[e]; g->wt[p] = g->w[e];
        g->out_degree[g->src[e]] += g->w[e];      /* addOutEdge :46-49 */
    }
    free(fill);
    g->built = 1;
}

/* java.util.stream.DoubleStream.sum() as shipped in JDK 8 (Collectors.sumWithCompensation +
 * computeFinalSum: Kahan running sum, final = sum + compensation [the JDK-8 form], with the
 * simple-sum fallback for NaN/inf).  Used by keepNearestKVertices (J/SpatialGraph.java:33) and
 * sourceWeightSum of the spatial graph (J/SpatialGraph.java:57,83).  Recalled from the public JDK
 * source, not checkable in this container. */
static double java8_stream_sum(const double* x, int64_t n) {
    double sum = 0.0, comp = 0.0, simple = 0.0;
    for (int64_t i = 0; i < n; i++) {
        double tmp = x[i] - comp;
        double velvel = sum + tmp;
        comp = (velvel - sum) - tmp;
        sum = velvel;
        simple += x[i];
    }
    double tmp = sum + comp;
    if (isnan(tmp) && isinf(simple)) return simple;
    return tmp;
}

/* bulk addSourceVertex (J/LayeredGraph.java:180-189): sourceWeightSum is a running += in call
 * order; stream_sum=1 is the SpatialGraph form (J/SpatialGraph.java:56-57). */
int orc_graph_set_sources(orc_graph* g, const int32_t* v, int64_t n, int stream_sum) {
    if (!g || n < 0) return 1;
    build_csr(g);
    free(g->srcv);
    g->srcv = (int32_t*)malloc((size_t)(n ? n : 1) * sizeof(int32_t));
    g->n_src = n; g->src_weight_sum = 0.0;
    double* od = (double*)malloc((size_t)(n ? n : 1) * sizeof(double));
    for (int64_t i = 0; i < n; i++) {
        if (v[i] < 0 || v[i] >= g->n_vertices) { free(od); return 2; }
        g->srcv[i] = v[i];
        od[i] = g->out_degree[v[i]];
        g->src_weight_sum += od[i];
    }
    if (stream_sum) g->src_weight_sum = java8_stream_sum(od, n);
    free(od);
    g->src_stream_sum = stream_sum ? 1 : 0; g->src_sum_fixed = 0;
    g->alias_built = 0;
    return 0;
}

int orc_graph_reserve_vertices(orc_graph* g, int32_t n) {
    if (!g || n < 0) return 1;
    if (n > g->n_vertices) { g->n_vertices = n; g->built = 0; g->alias_built = 0; }
    return 0;
}
/* Vertex.outDegree (J/LayeredGraph.java:35) is a public field: prob = k*w/outDegree (:62) and the source weights (:208)
 * read whatever it holds */
int orc_graph_set_out_degree(orc_graph* g, const double* od, int32_t n) {
    if (!g || !od) return 1;
    build_csr(g);
    if (n != g->n_vertices) return 1;
    for (int32_t v = 0; v < n; v++) g->out_degree[v] = od[v];
    g->alias_built = 0;
    if (g->n_src > 0 && !g->src_sum_fixed) {        /* the sum of the source weights follows, in the form set_sources used */
        double* sw = (double*)malloc((size_t)g->n_src * sizeof(double));
        double s = 0.0;
        for (int64_t i = 0; i < g->n_src; i++) { sw[i] = g->out_degree[g->srcv[i]]; s += sw[i]; }
        g->src_weight_sum = g->src_stream_sum ? java8_stream_sum(sw, g->n_src) : s;
        free(sw);
    }
    return 0;
}
int orc_graph_set_source_weight_sum(orc_graph* g, double s) {
    if (!g) return 1;
    g->src_weight_sum = s; g->src_sum_fixed = 1; g->alias_built = 0;
    return 0;
}
int orc_graph_get_csr(const orc_graph* gc, int64_t* row_ptr, int32_t* nbr, double* weight, double* prob, int32_t* alias, double* out_degree) {
    orc_graph* g = (orc_graph*)gc;
    if (!g) return 1;
    build_csr(g);
    int32_t V = g->n_vertices; int64_t E = g->row_ptr[V];
    if ((prob || alias) && !g->alias_built) return 5;
    if (row_ptr) memcpy(row_ptr, g->row_ptr, ((size_t)V + 1) * sizeof(int64_t));
    if (out_degree) memcpy(out_degree, g->out_degree, (size_t)V * sizeof(double));
    if (nbr) memcpy(nbr, g->nbr, (size_t)E * sizeof(int32_t));
    if (weight) memcpy(weight, g->wt, (size_t)E * sizeof(double));
    if (prob) memcpy(prob, g->prob, (size_t)E * sizeof(double));
    if (alias) memcpy(alias, g->alias, (size_t)E * sizeof(int32_t));
    return 0;
}

/* keepNearestKVertices (J/SpatialGraph.java:29-35): stable sort by weight descending
 * (List.sort is a stable merge sort; comparator -Double.compare), keep the first k,
 * outDegree = DoubleStream.sum().  subList(0,k) throws if a vertex has fewer than k edges. */
typedef struct { double w; int32_t nbr; int64_t ord; } kt_item;
static int kt_cmp(const void* a, const void* b) {
    const kt_item* x = (const kt_item*)a; const kt_item* y = (const kt_item*)b;
    if (x->w > y->w) return -1;
    if (x->w < y->w) return 1;
    return (x->ord > y->ord) - (x->ord < y->ord);
}
int orc_graph_keep_top_k(orc_graph* g, int32_t k) {
    if (!g || k < 0) return 1;
    build_csr(g);
    int32_t V = g->n_vertices;
    for (int32_t v = 0; v < V; v++)
        if (g->row_ptr[v + 1] - g->row_ptr[v] < k) return 3;   /* IndexOutOfBoundsException */
    int64_t* nrp = (int64_t*)malloc(((size_t)V + 1) * sizeof(int64_t));
    int32_t* nn = (int32_t*)malloc((size_t)((int64_t)V * k + 1) * sizeof(int32_t));
    double*  nw = (double*)malloc((size_t)((int64_t)V * k + 1) * sizeof(double));
    int64_t maxdeg = 0;
    for (int32_t v = 0; v < V; v++) { int64_t d = g->row_ptr[v + 1] - g->row_ptr[v]; if (d > maxdeg) maxdeg = d; }
    kt_item* it = (kt_item*)malloc((size_t)(maxdeg + 1) * sizeof(kt_item));
    for (int32_t v = 0; v < V; v++) {
        int64_t b = g->row_ptr[v], d = g->row_ptr[v + 1] - b;
        for (int64_t j = 0; j < d; j++) { it[j].w = g->wt[b + j]; it[j].nbr = g->nbr[b + j]; it[j].ord = j; }
        qsort(it, (size_t)d, sizeof(kt_item), kt_cmp);
        nrp[v] = (int64_t)v * k;
        for (int32_t j = 0; j < k; j++) { nn[nrp[v] + j] = it[j].nbr; nw[nrp[v] + j] = it[j].w; }
        g->out_degree[v] = java8_stream_sum(nw + nrp[v], k);
    }
    nrp[V] = (int64_t)V * k;
    free(it); free(g->row_ptr); free(g->nbr); free(g->wt);
    g->row_ptr = nrp; g->nbr = nn; g->wt = nw;
    g->n_edges = (int64_t)V * k;   /* COO no longer mirrors the CSR; CSR is authoritative now */
    g->alias_built = 0;
    return 0;
}

/* Vertex.initiateAliasTable (J/LayeredGraph.java:54-82) — the reference's own O(k^2) pairing;
 * the golden test pins the resulting alias INDICES, which depend on this order. */
static void alias_reference(const double* w, int64_t k, double total, double* prob, int32_t* alias) {
    for (int64_t i = 0; i < k; i++) { alias[i] = -1; prob[i] = (double)k * w[i] / total; }   /* :58-63 */
    for (int64_t l1 = 0; l1 < k; l1++) {                                                      /* :65 */
        if (prob[l1] != 1.0 && alias[l1] == -1) {
            for (int64_t l2 = 0; l2 < k; l2++) {
                if (l2 != l1 && alias[l2] == -1) {
                    if (prob[l1] > 1.0 && prob[l2] < 1.0) {
                        alias[l2] = (int32_t)l1;
                        prob[l1] -= 1 - prob[l2];
                    } else if (prob[l1] < 1.0 && prob[l2] > 1.0) {
                        alias[l1] = (int32_t)l2;
                        prob[l2] -= 1 - prob[l1];
                        break;                                                                 /* :76 */
                    }
                }
            }
        }
    }
}

/* Vose O(k) pairing (NOT in the reference: the scalable form for hubs, SURVEY.md §7 hard parts).
 * Same prob[i] = k*w/total initialisation; small stack grows up from scratch[0], large stack grows
 * down from scratch[k-1]; left-overs get prob 1, alias -1.  libdge's device kernel runs this exact
 * sequence per vertex, so the tables are bit-identical. */
static void alias_vose(const double* w, int64_t k, double total, double* prob, int32_t* alias, int32_t* scratch) {
    int64_t ns = 0, nl = 0;
    for (int64_t i = 0; i < k; i++) {
        alias[i] = -1; prob[i] = (double)k * w[i] / total;
        if (prob[i] < 1.0) scratch[ns++] = (int32_t)i; else scratch[k - 1 - (nl++)] = (int32_t)i;
    }
    while (ns > 0 && nl > 0) {
        int32_t s = scratch[--ns];
        int32_t l = scratch[k - nl]; nl--;
        alias[s] = l;
        prob[l] = (prob[l] + prob[s]) - 1.0;
        if (prob[l] < 1.0) scratch[ns++] = l; else scratch[k - 1 - (nl++)] = l;
    }
    while (ns > 0) { int32_t s = scratch[--ns]; prob[s] = 1.0; }
    while (nl > 0) { int32_t l = scratch[k - nl]; nl--; prob[l] = 1.0; }
}

/* initiateAliasTables (J/LayeredGraph.java:195-226): per-vertex tables, then the same algorithm
 * over the source vertices with weight = outDegree. */
int orc_graph_build_alias(orc_graph* g, int exact) {
    if (!g) return 1;
    build_csr(g);
    int32_t V = g->n_vertices; int64_t E = g->row_ptr[V];
    free(g->prob); free(g->alias); free(g->src_prob); free(g->src_alias);
    g->prob = (double*)malloc((size_t)(E ? E : 1) * sizeof(double));
    g->alias = (int32_t*)malloc((size_t)(E ? E : 1) * sizeof(int32_t));
    int64_t maxk = g->n_src;
    for (int32_t v = 0; v < V; v++) { int64_t d = g->row_ptr[v + 1] - g->row_ptr[v]; if (d > maxk) maxk = d; }
    int32_t* scratch = (int32_t*)malloc((size_t)(maxk + 1) * sizeof(int32_t));
    for (int32_t v = 0; v < V; v++) {
        int64_t b = g->row_ptr[v], k = g->row_ptr[v + 1] - b;
        if (exact) alias_reference(g->wt + b, k, g->out_degree[v], g->prob + b, g->alias + b);
        else       alias_vose(g->wt + b, k, g->out_degree[v], g->prob + b, g->alias + b, scratch);
    }
    int64_t S = g->n_src;
    g->src_prob = (double*)malloc((size_t)(S ? S : 1) * sizeof(double));
    g->src_alias = (int32_t*)malloc((size_t)(S ? S : 1) * sizeof(int32_t));
    double* sw = (double*)malloc((size_t)(S ? S : 1) * sizeof(double));
    for (int64_t i = 0; i < S; i++) sw[i] = g->out_degree[g->srcv[i]];
    if (exact) alias_reference(sw, S, g->src_weight_sum, g->src_prob, g->src_alias);
    else       alias_vose(sw, S, g->src_weight_sum, g->src_prob, g->src_alias, scratch);
    free(sw); free(scratch);
    g->alias_built = 1;
    return 0;
}

int32_t orc_graph_num_vertices(const orc_graph* g) { return g ? g->n_vertices : 0; }
int64_t orc_graph_num_edges(const orc_graph* g) { return g ? (g->built ? g->row_ptr[g->n_vertices] : g->n_edges) : 0; }

int orc_graph_get_alias(const orc_graph* g, int32_t v, double* prob, int32_t* alias, int32_t* nbr,
                        double* weight, int32_t cap, int32_t* k, double* out_degree) {
    if (!g || !g->built || v < 0 || v >= g->n_vertices) return 1;
    int64_t b = g->row_ptr[v], d = g->row_ptr[v + 1] - b;
    if (k) *k = (int32_t)d;
    if (out_degree) *out_degree = g->out_degree[v];
    if (d > cap) return 4;
    for (int64_t j = 0; j < d; j++) {
        if (prob && g->alias_built) prob[j] = g->prob[b + j];
        if (alias && g->alias_built) alias[j] = g->alias[b + j];
        if (nbr) nbr[j] = g->nbr[b + j];
        if (weight) weight[j] = g->wt[b + j];
    }
    return 0;
}

int orc_graph_get_source_alias(const orc_graph* g, double* prob, int32_t* alias, int32_t* src,
                               int32_t cap, int32_t* k, double* weight_sum) {
    if (!g) return 1;
    if (k) *k = (int32_t)g->n_src;
    if (weight_sum) *weight_sum = g->src_weight_sum;
    if (g->n_src > cap) return 4;
    for (int64_t j = 0; j < g->n_src; j++) {
        if (prob && g->alias_built) prob[j] = g->src_prob[j];
        if (alias && g->alias_built) alias[j] = g->src_alias[j];
        if (src) src[j] = g->srcv[j];
    }
    return 0;
}

/* one alias draw (J/LayeredGraph.java:104-116,123-132,234-242): i=(int)(x*k); y=x*k-i;
 * y<prob[i] ? slot i : slot alias[i].  alias==-1 ("no alias") is read as "stay in slot i"
 * (the reference would index -1 there with probability <= 2e-15: a latent bug, not a behaviour). */
static inline int64_t alias_pick(const double* prob, const int32_t* alias, int64_t k, double x) {
    int64_t i = (int64_t)(x * (double)k);
    if (i > k - 1) i = k - 1;
    double y = x * (double)k - (double)i;
    if (y < prob[i]) return i;
    int32_t a = alias[i];
    return a < 0 ? i : a;
}

/* test overload sampleNextVertex(double x) (J/LayeredGraph.java:123-132) */
int orc_graph_sample_next(const orc_graph* g, int32_t v, double x, int32_t* next) {
    if (!g || !g->alias_built || v < 0 || v >= g->n_vertices) return 1;
    int64_t b = g->row_ptr[v], k = g->row_ptr[v + 1] - b;
    if (k == 0) { *next = -1; return 0; }
    *next = g->nbr[b + alias_pick(g->prob + b, g->alias + b, k, x)];
    return 0;
}

/* sampleVertexSequence (J/LayeredGraph.java:232-252): one draw for the source (:234-242), then up to
 * max_len-1 steps; a vertex with no out-edges ends the walk BEFORE drawing (:106-107,247-248).
 * Output row: vertex ids, padded with -1. */
static int64_t one_walk(const orc_graph* g, orc_jrand* r, int32_t max_len, int32_t* out) {
    int64_t draws = 0;
    for (int32_t j = 0; j < max_len; j++) out[j] = -1;
    if (g->n_src == 0 || max_len <= 0) return 0;
    double x = orc_jrand_next_double(r); draws++;
    int32_t v = g->srcv[alias_pick(g->src_prob, g->src_alias, g->n_src, x)];
    out[0] = v;
    for (int32_t len = 1; len < max_len; len++) {
        int64_t b = g->row_ptr[v], k = g->row_ptr[v + 1] - b;
        if (k == 0) break;
        x = orc_jrand_next_double(r); draws++;
        v = g->nbr[b + alias_pick(g->prob + b, g->alias + b, k, x)];
        out[len] = v;
    }
    return draws;
}

int orc_sample_walks(const orc_graph* g, int64_t n_walks, int32_t max_len, int64_t seed, int rng_mode,
                     int64_t first_index, int32_t* out, int64_t* draws_consumed) {
    if (!g || !g->alias_built || n_walks < 0 || max_len < 0) return 1;
    int64_t total = 0;
    if (rng_mode == 0) {            /* the reference's shared sequential stream */
        orc_jrand r; orc_jrand_seed(&r, seed);
        orc_jrand_jump(&r, 2ULL * (uint64_t)first_index);
        for (int64_t i = 0; i < n_walks; i++) total += one_walk(g, &r, max_len, out + i * max_len);
    } else {                        /* strided: walk i owns draws [i*L,(i+1)*L) of the same stream */
        for (int64_t i = 0; i < n_walks; i++) {
            orc_jrand r; orc_jrand_seed(&r, seed);
            orc_jrand_jump(&r, 2ULL * (uint64_t)(first_index + i) * (uint64_t)max_len);
            total += one_walk(g, &r, max_len, out + i * max_len);
        }
    }
    if (draws_consumed) *draws_consumed = total;
    return 0;
}

/* =====================================================================================
 * SGNS trainer — PARITY UNPINNED (third-party DL4J 0.7.2; see header).
 * Call site and hyper-parameters: J/DeepWalk.java:62-79.
 * ===================================================================================== */
#define EXP_TABLE_SIZE 1000
#define MAX_EXP 6
#define W2V_MULT 25214903917ULL

static float g_exp_table[EXP_TABLE_SIZE];
static int g_exp_ready = 0;
static void init_exp_table(void) {
    if (g_exp_ready) return;
    for (int i = 0; i < EXP_TABLE_SIZE; i++) {       /* word2vec.c: precomputed sigmoid */
        float e = (float)exp((i / (float)EXP_TABLE_SIZE * 2 - 1) * MAX_EXP);
        g_exp_table[i] = e / (e + 1);
    }
    g_exp_ready = 1;
}
float orc_exp_table(int i) { init_exp_table(); return (i >= 0 && i < EXP_TABLE_SIZE) ? g_exp_table[i] : 0.0f; }

uint64_t orc_mix64(uint64_t x) {                     /* splitmix64 output function */
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

#define MAX_CODE_LENGTH 40
struct orc_model {
    int64_t V; int32_t dim;
    float *syn0, *syn1neg, *syn1;
    int32_t* points; uint8_t* codes; int32_t* codelen;     /* Huffman paths, [V x MAX_CODE_LENGTH] */
    int32_t* vocab_ids; int64_t* counts;
    int32_t* table; int64_t table_size;
    int64_t pairs, total_words;
    double seconds;
};

int64_t orc_model_vocab_size(const orc_model* m) { return m->V; }
int32_t orc_model_dim(const orc_model* m) { return m->dim; }
const float* orc_model_syn0(const orc_model* m) { return m->syn0; }
const float* orc_model_syn1neg(const orc_model* m) { return m->syn1neg; }
const float* orc_model_syn1(const orc_model* m) { return m->syn1; }
int32_t orc_model_code(const orc_model* m, int64_t row, int32_t* points, uint8_t* codes, int32_t cap) {
    if (!m->codelen || row < 0 || row >= m->V) return -1;
    int32_t n = m->codelen[row];
    for (int32_t i = 0; i < n && i < cap; i++) { points[i] = m->points[row * MAX_CODE_LENGTH + i]; codes[i] = m->codes[row * MAX_CODE_LENGTH + i]; }
    return n;
}
const int32_t* orc_model_vocab_ids(const orc_model* m) { return m->vocab_ids; }
const int64_t* orc_model_counts(const orc_model* m) { return m->counts; }
const int32_t* orc_model_table(const orc_model* m) { return m->table; }
int64_t orc_model_pairs(const orc_model* m) { return m->pairs; }
int64_t orc_model_total_words(const orc_model* m) { return m->total_words; }
double orc_model_seconds(const orc_model* m) { return m->seconds; }
void orc_model_free(orc_model* m) {
    if (!m) return;
    free(m->syn0); free(m->syn1neg); free(m->syn1); free(m->points); free(m->codes); free(m->codelen);
    free(m->vocab_ids); free(m->counts); free(m->table); free(m);
}

typedef struct { int64_t cnt; int32_t id; } vc_item;
static int vc_cmp(const void* a, const void* b) {     /* count desc, vertex id asc */
    const vc_item* x = (const vc_item*)a; const vc_item* y = (const vc_item*)b;
    if (x->cnt != y->cnt) return (x->cnt < y->cnt) - (x->cnt > y->cnt);
    return (x->id > y->id) - (x->id < y->id);
}

/* dot products.  arith 0: word2vec.c loop.  arith 1: the HIP kernel's lane order — lane j of a
 * 16-lane group owns elements {64c+16m+j : m=0..3}, accumulates with fmaf in increasing index order,
 * then the 16 partials are combined by an xor-butterfly (1,2,4,8). */
static inline float dot_seq(const float* a, const float* b, int D) {
    float f = 0;
    for (int c = 0; c < D; c++) f += a[c] * b[c];
    return f;
}
static inline float dot_lane16(const float* a, const float* b, int D) {
    float p[16];
    for (int j = 0; j < 16; j++) {
        float acc = 0.0f;
        for (int c = 0; c * 64 < D; c++)
            for (int e = 0; e < 4; e++) {
                int idx = c * 64 + 16 * e + j;
                if (idx < D) acc = fmaf(a[idx], b[idx], acc);
            }
        p[j] = acc;
    }
    for (int s = 1; s < 16; s <<= 1) {
        float q[16];
        for (int j = 0; j < 16; j++) q[j] = p[j] + p[j ^ s];
        memcpy(p, q, sizeof(p));
    }
    return p[0];
}

/* lane order of the kernels that move rows as 16 bytes per lane (k_sgns_train_locked, k_sorted_phase): lane j owns elements
 * {64c + 4j + e : e = 0..3} of chunk c */
static inline float dot_lane16A(const float* a, const float* b, int D) {
    float p[16];
    for (int j = 0; j < 16; j++) {
        float acc = 0.0f;
        for (int c = 0; c * 64 < D; c++)
            for (int e = 0; e < 4; e++) {
                int idx = c * 64 + 4 * j + e;
                if (idx < D) acc = fmaf(a[idx], b[idx], acc);
            }
        p[j] = acc;
    }
    for (int s = 1; s < 16; s <<= 1) {
        float q[16];
        for (int j = 0; j < 16; j++) q[j] = p[j] + p[j ^ s];
        memcpy(p, q, sizeof(p));
    }
    return p[0];
}

static inline float alpha_for(const orc_train_config* cfg, int64_t words_done, int64_t total_words_all) {
    /* word2vec.c: alpha = starting_alpha * (1 - word_count_actual / (iter*train_words + 1)), floored.
     * DL4J floors at minLearningRate (absolute). Evaluated per walk from the exact count of in-vocab
     * tokens that precede it. */
    double a = (double)cfg->alpha * (1.0 - (double)words_done / (double)(total_words_all + 1));
    float af = (float)a;
    if (af < cfg->min_alpha) af = cfg->min_alpha;
    return af;
}

/* word2vec.c CreateBinaryTree: Huffman tree over the vocabulary counts (rows are sorted by count descending);
 * point[] = inner nodes from the root, code[] = branch taken at each of them. */
static void create_binary_tree(orc_model* m) {
    const int64_t V = m->V;
    m->points = (int32_t*)calloc((size_t)(V ? V : 1) * MAX_CODE_LENGTH, sizeof(int32_t));
    m->codes = (uint8_t*)calloc((size_t)(V ? V : 1) * MAX_CODE_LENGTH, 1);
    m->codelen = (int32_t*)calloc((size_t)(V ? V : 1), sizeof(int32_t));
    if (V < 2) return;
    int64_t* count = (int64_t*)calloc((size_t)V * 2 + 1, sizeof(int64_t));
    int64_t* binary = (int64_t*)calloc((size_t)V * 2 + 1, sizeof(int64_t));
    int64_t* parent = (int64_t*)calloc((size_t)V * 2 + 1, sizeof(int64_t));
    for (int64_t a = 0; a < V; a++) count[a] = m->counts[a];
    for (int64_t a = V; a < V * 2; a++) count[a] = (int64_t)1e15;
    int64_t pos1 = V - 1, pos2 = V, min1i, min2i;
    for (int64_t a = 0; a < V - 1; a++) {
        if (pos1 >= 0) { if (count[pos1] < count[pos2]) { min1i = pos1; pos1--; } else { min1i = pos2; pos2++; } }
        else { min1i = pos2; pos2++; }
        if (pos1 >= 0) { if (count[pos1] < count[pos2]) { min2i = pos1; pos1--; } else { min2i = pos2; pos2++; } }
        else { min2i = pos2; pos2++; }
        count[V + a] = count[min1i] + count[min2i];
        parent[min1i] = V + a; parent[min2i] = V + a;
        binary[min2i] = 1;
    }
    for (int64_t a = 0; a < V; a++) {
        int64_t b = a, i = 0; uint8_t code[MAX_CODE_LENGTH]; int64_t point[MAX_CODE_LENGTH];
        for (;;) {
            code[i] = (uint8_t)binary[b]; point[i] = b; i++;
            b = parent[b];
            if (b == V * 2 - 2 || i >= MAX_CODE_LENGTH - 1) break;
        }
        m->codelen[a] = (int32_t)i;
        m->points[a * MAX_CODE_LENGTH] = (int32_t)(V - 2);
        for (b = 0; b < i; b++) {
            m->codes[a * MAX_CODE_LENGTH + i - b - 1] = code[b];
            if (i - b < MAX_CODE_LENGTH) m->points[a * MAX_CODE_LENGTH + i - b] = (int32_t)(point[b] - V);
        }
    }
    free(count); free(binary); free(parent);
}

/* the tree alone, for tests of the product's path builder: counts must be sorted descending */
int orc_huffman(const int64_t* counts, int64_t V, int32_t* codelen, int32_t* points, uint8_t* codes) {
    if (!counts || V < 0) return -1;
    orc_model m; memset(&m, 0, sizeof m);
    m.V = V; m.counts = (int64_t*)counts;
    create_binary_tree(&m);
    for (int64_t r = 0; r < V; r++) {
        codelen[r] = m.codelen[r];
        memcpy(points + r * MAX_CODE_LENGTH, m.points + r * MAX_CODE_LENGTH, MAX_CODE_LENGTH * sizeof(int32_t));
        memcpy(codes + r * MAX_CODE_LENGTH, m.codes + r * MAX_CODE_LENGTH, MAX_CODE_LENGTH);
    }
    free(m.points); free(m.codes); free(m.codelen);
    return 0;
}

static int64_t train_walk(const orc_train_config* cfg, orc_model* m, const int32_t* sen, int len,
                          int64_t gidx_base, float alpha, float* neu1e, int part_ctx, int part_tgt) {
    const int D = cfg->dim, W = cfg->window, K = cfg->negative;
    const int64_t V = m->V, T = m->table_size;
    const int PN = cfg->part_n > 1 ? cfg->part_n : 1;
    int64_t pairs = 0;
    for (int i = 0; i < len; i++) {
        int32_t word = sen[i];
        if (word < 0) continue;
        /* block schedule: the negative-sampling terms of a pair belong to the block of its centre's partition; with the hierarchical softmax every
           centre is visited in every block for the inner nodes of ITS path that lie in partition part_tgt (inner-node rows are split by node % n like
           the vocabulary rows, and the syn1 partition travels the ring with the syn1neg partition of the same number) */
        const int is_tgt = PN <= 1 || word % PN == part_tgt;
        if (!is_tgt && !cfg->use_hs) continue;               /* another block's centre */
        uint64_t s = orc_mix64(cfg->seed + (uint64_t)(gidx_base + i));
        s = s * W2V_MULT + 11;
        int b = (int)(s % (uint64_t)W);
        const uint64_t s_centre = s;
        for (int a = b; a < W * 2 + 1 - b; a++) {
            if (a == W) continue;
            int c = i - W + a;
            if (c < 0 || c >= len) continue;
            int32_t last = sen[c];
            if (last < 0) continue;
            if (PN > 1) {                                      /* block schedule: every pair draws from its own stream */
                if (last % PN != part_ctx) continue;           /* another block's context */
                s = orc_mix64(s_centre + (uint64_t)c);
            }
            float* l1 = m->syn0 + (int64_t)last * D;
            for (int k = 0; k < D; k++) neu1e[k] = 0;
            if (cfg->use_hs)                       /* word2vec.c: HIERARCHICAL SOFTMAX, before the negative sampling */
                for (int d = 0; d < m->codelen[word]; d++) {
                    if (PN > 1 && m->points[(int64_t)word * MAX_CODE_LENGTH + d] % PN != part_tgt) continue;      /* another block's inner node */
                    float* l2 = m->syn1 + (int64_t)m->points[(int64_t)word * MAX_CODE_LENGTH + d] * D;
                    float f = cfg->arith ? dot_lane16(l1, l2, D) : dot_seq(l1, l2, D);
                    if (f <= -MAX_EXP) continue;
                    else if (f >= MAX_EXP) continue;
                    f = g_exp_table[(int)((f + MAX_EXP) * (EXP_TABLE_SIZE / MAX_EXP / 2))];
                    float g = (1 - m->codes[(int64_t)word * MAX_CODE_LENGTH + d] - f) * alpha;
                    if (cfg->arith) {
                        for (int k = 0; k < D; k++) neu1e[k] = fmaf(g, l2[k], neu1e[k]);
                        for (int k = 0; k < D; k++) l2[k] = fmaf(g, l1[k], l2[k]);
                    } else {
                        for (int k = 0; k < D; k++) neu1e[k] += g * l2[k];
                        for (int k = 0; k < D; k++) l2[k] += g * l1[k];
                    }
                }
            for (int d = 0; d < K + 1 && is_tgt; d++) {
                int64_t target; float label;
                if (d == 0) { target = word; label = 1; }
                else {
                    s = s * W2V_MULT + 11;
                    target = m->table[(s >> 16) % (uint64_t)T];
                    if (target == 0 && V > 1) target = (int64_t)(s % (uint64_t)(V - 1)) + 1;
                    if (PN > 1) { target = target / PN * PN + part_tgt; if (target >= V) target -= PN; }
                    if (target == word) continue;
                    label = 0;
                }
                float* l2 = m->syn1neg + target * D;
                float f = cfg->arith ? dot_lane16(l1, l2, D) : dot_seq(l1, l2, D);
                float g;
                if (f > MAX_EXP) g = (label - 1) * alpha;
                else if (f < -MAX_EXP) g = (label - 0) * alpha;
                else g = (label - g_exp_table[(int)((f + MAX_EXP) * (EXP_TABLE_SIZE / MAX_EXP / 2))]) * alpha;
                if (cfg->arith) {
                    for (int k = 0; k < D; k++) neu1e[k] = fmaf(g, l2[k], neu1e[k]);
                    for (int k = 0; k < D; k++) l2[k] = fmaf(g, l1[k], l2[k]);
                } else {
                    for (int k = 0; k < D; k++) neu1e[k] += g * l2[k];
                    for (int k = 0; k < D; k++) l2[k] += g * l1[k];
                }
            }
            for (int k = 0; k < D; k++) l1[k] += neu1e[k];
            pairs += is_tgt;
        }
    }
    return pairs;
}

/* ---- train_walk once more with the same arithmetic and the processor's pipelines kept busy (what the test-suite and the CPU baseline run; the loop
 * above stays the definition and orc_set_plain(1) runs it).
 * Within one pair nothing a dot product reads is written before the pair's last line: l1 (a syn0 row) moves only in `l1[k] += neu1e[k]`, an inner node's row
 * and a negative-sampling row only in their own term.  So where the rows of a pair's terms are pairwise distinct (a Huffman path's nodes always are; the
 * positive target and the K draws unless a draw repeats) the terms' dot products are taken side by side — each one's additions in word2vec.c's order (arith 0)
 * or the kernels' lane order (arith 1) — and the terms are then applied one after the other as above: the same floating-point operations on the same values,
 * bit for bit the tables train_walk leaves.  A pair with a repeated row goes term by term.  The random draws do not depend on the arithmetic (K per pair).
 * tests/test_oracle_kats.py compares both forms on random configurations, vocabularies of a handful of rows included. ---- */
#define ILP_MAX 64
static int g_orc_plain = 0;
void orc_set_plain(int on) { g_orc_plain = on ? 1 : 0; }

static void dots_seq(const float* a, float* const* rows, int n, int D, float* f) {          /* n x dot_seq, eight at a time */
    for (int d = 0; d < n; d += 8) {
        if (n - d == 1) { f[d] = dot_seq(a, rows[d], D); return; }
        const float* b[8];
        for (int j = 0; j < 8; j++) b[j] = d + j < n ? rows[d + j] : a;                      /* (a spare chain reads the vector itself; its sum is dropped) */
        float f0 = 0, f1 = 0, f2 = 0, f3 = 0, f4 = 0, f5 = 0, f6 = 0, f7 = 0;
        for (int c = 0; c < D; c++) {
            const float x = a[c];
            f0 += x * b[0][c]; f1 += x * b[1][c]; f2 += x * b[2][c]; f3 += x * b[3][c];
            f4 += x * b[4][c]; f5 += x * b[5][c]; f6 += x * b[6][c]; f7 += x * b[7][c];
        }
        const float r[8] = {f0, f1, f2, f3, f4, f5, f6, f7};
        for (int j = 0; j < 8 && d + j < n; j++) f[d + j] = r[j];
    }
}
/* dot_lane16 with the sixteen lanes of a step next to each other (lane j still adds its elements 64c + 16e + j in increasing order, then the butterfly's lane 0) */
static inline float dot_lane16_v(const float* a, const float* b, int D) {
    float p[16];
    for (int j = 0; j < 16; j++) p[j] = 0.0f;
    int base = 0;
    for (; base + 16 <= D; base += 16) {
#pragma omp simd
        for (int j = 0; j < 16; j++) p[j] = fmaf(a[base + j], b[base + j], p[j]);
    }
    for (int j = 0; base + j < D; j++) p[j] = fmaf(a[base + j], b[base + j], p[j]);
    /* lane 0 of the butterfly: ((p0 + p1) + (p2 + p3)) + ... — the fifteen additions its value comes from, none of the other lanes' */
    for (int s = 1; s < 16; s <<= 1)
        for (int j = 0; j < 16; j += 2 * s) p[j] = p[j] + p[j + s];
    return p[0];
}
static inline void apply_term(int arith, float g, const float* l1, float* l2, float* neu1e, int D) {
    if (arith) {
#pragma omp simd
        for (int k = 0; k < D; k++) neu1e[k] = fmaf(g, l2[k], neu1e[k]);
#pragma omp simd
        for (int k = 0; k < D; k++) l2[k] = fmaf(g, l1[k], l2[k]);
    } else {
#pragma omp simd
        for (int k = 0; k < D; k++) neu1e[k] += g * l2[k];
#pragma omp simd
        for (int k = 0; k < D; k++) l2[k] += g * l1[k];
    }
}
static inline float ns_step(float f, float label, float alpha) {
    if (f > MAX_EXP) return (label - 1) * alpha;
    if (f < -MAX_EXP) return (label - 0) * alpha;
    return (label - g_exp_table[(int)((f + MAX_EXP) * (EXP_TABLE_SIZE / MAX_EXP / 2))]) * alpha;
}

static int64_t train_walk_ilp(const orc_train_config* cfg, orc_model* m, const int32_t* sen, int len,
                              int64_t gidx_base, float alpha, float* neu1e, int part_ctx, int part_tgt) {
    const int D = cfg->dim, W = cfg->window, K = cfg->negative;
    const int64_t V = m->V, T = m->table_size;
    const int PN = cfg->part_n > 1 ? cfg->part_n : 1;
    int64_t pairs = 0;
    float* rows[ILP_MAX]; float lab[ILP_MAX], f[ILP_MAX]; int64_t tg[ILP_MAX];
    for (int i = 0; i < len; i++) {
        int32_t word = sen[i];
        if (word < 0) continue;
        const int is_tgt = PN <= 1 || word % PN == part_tgt;
        if (!is_tgt && !cfg->use_hs) continue;
        uint64_t s = orc_mix64(cfg->seed + (uint64_t)(gidx_base + i));
        s = s * W2V_MULT + 11;
        int b = (int)(s % (uint64_t)W);
        const uint64_t s_centre = s;
        for (int a = b; a < W * 2 + 1 - b; a++) {
            if (a == W) continue;
            int c = i - W + a;
            if (c < 0 || c >= len) continue;
            int32_t last = sen[c];
            if (last < 0) continue;
            if (PN > 1) {
                if (last % PN != part_ctx) continue;
                s = orc_mix64(s_centre + (uint64_t)c);
            }
            float* l1 = m->syn0 + (int64_t)last * D;
            for (int k = 0; k < D; k++) neu1e[k] = 0;
            if (cfg->use_hs) {                      /* the path's inner nodes: pairwise distinct rows of syn1 */
                int n = 0;
                for (int d = 0; d < m->codelen[word]; d++) {
                    const int32_t node = m->points[(int64_t)word * MAX_CODE_LENGTH + d];
                    if (PN > 1 && node % PN != part_tgt) continue;
                    rows[n] = m->syn1 + (int64_t)node * D;
                    lab[n] = (float)(1 - m->codes[(int64_t)word * MAX_CODE_LENGTH + d]);
                    n++;
                }
                if (cfg->arith) for (int j = 0; j < n; j++) f[j] = dot_lane16_v(l1, rows[j], D);
                else dots_seq(l1, rows, n, D, f);
                for (int j = 0; j < n; j++) {
                    float ff = f[j];
                    if (ff <= -MAX_EXP) continue;
                    else if (ff >= MAX_EXP) continue;
                    ff = g_exp_table[(int)((ff + MAX_EXP) * (EXP_TABLE_SIZE / MAX_EXP / 2))];
                    apply_term(cfg->arith, (lab[j] - ff) * alpha, l1, rows[j], neu1e, D);
                }
            }
            if (is_tgt) {
                int n = 0, distinct = 1;
                tg[n] = word; lab[n] = 1; n++;
                for (int d = 1; d < K + 1; d++) {
                    s = s * W2V_MULT + 11;
                    int64_t target = m->table[(s >> 16) % (uint64_t)T];
                    if (target == 0 && V > 1) target = (int64_t)(s % (uint64_t)(V - 1)) + 1;
                    if (PN > 1) { target = target / PN * PN + part_tgt; if (target >= V) target -= PN; }
                    if (target == word) continue;
                    for (int j = 1; j < n; j++) if (tg[j] == target) distinct = 0;
                    tg[n] = target; lab[n] = 0; n++;
                }
                for (int j = 0; j < n; j++) rows[j] = m->syn1neg + tg[j] * D;
                if (distinct) {
                    if (cfg->arith) for (int j = 0; j < n; j++) f[j] = dot_lane16_v(l1, rows[j], D);
                    else dots_seq(l1, rows, n, D, f);
                    for (int j = 0; j < n; j++) apply_term(cfg->arith, ns_step(f[j], lab[j], alpha), l1, rows[j], neu1e, D);
                } else
                    for (int j = 0; j < n; j++) {       /* a draw repeats a row of this pair: its second term reads what the first wrote */
                        const float ff = cfg->arith ? dot_lane16_v(l1, rows[j], D) : dot_seq(l1, rows[j], D);
                        apply_term(cfg->arith, ns_step(ff, lab[j], alpha), l1, rows[j], neu1e, D);
                    }
            }
            for (int k = 0; k < D; k++) l1[k] += neu1e[k];
            pairs += is_tgt;
        }
    }
    return pairs;
}

/* ---- update_policy 8: the owner-computes schedule (embedding_amd/csrc/sgns_sorted.hip), one block over walks [0, n_walks) ---- */
typedef struct { int32_t key, other; float x; } so_item;      /* key = owning row, other = the row of the other table, x = signed alpha / g */

static void so_stable_sort(const so_item* in, so_item* out, int64_t n, int64_t V, int64_t* seg /* [V+1] */) {
    for (int64_t r = 0; r <= V; r++) seg[r] = 0;
    for (int64_t i = 0; i < n; i++) seg[in[i].key + 1]++;
    for (int64_t r = 0; r < V; r++) seg[r + 1] += seg[r];
    int64_t* fill = (int64_t*)malloc((size_t)(V + 1) * sizeof(int64_t));
    memcpy(fill, seg, (size_t)(V + 1) * sizeof(int64_t));
    for (int64_t i = 0; i < n; i++) out[fill[in[i].key]++] = in[i];
    free(fill);
}

/* one phase over sorted items: own = the table whose rows are owned (phase A: syn1neg, phase B: syn0), oth = the other one */
static void so_phase(const orc_train_config* cfg, int phase_b, float* own, float* own_out, const float* oth, so_item* it, int64_t n, const int64_t* seg, int D) {
    const int64_t C = cfg->sorted_chunk;
    const int64_t n_chunks = (n + C - 1) / C;
    float* scratch = (float*)calloc((size_t)(2 * n_chunks + 2) * D, sizeof(float));
    float* h = (float*)malloc((size_t)D * sizeof(float)); float* d = (float*)malloc((size_t)D * sizeof(float));
    for (int64_t c = 0; c < n_chunks; c++) {
        const int64_t start = c * C, stop = start + C < n ? start + C : n;
        int64_t s0 = start;
        while (s0 < stop) {
            const int32_t row = it[s0].key;
            int64_t s1 = s0;
            while (s1 < stop && it[s1].key == row) s1++;
            float* r = own + (int64_t)row * D;
            for (int k = 0; k < D; k++) { h[k] = r[k]; d[k] = 0.0f; }
            for (int64_t i = s0; i < s1; i++) {
                const float* o = oth + (int64_t)it[i].other * D;
                if (phase_b) {
                    for (int k = 0; k < D; k++) d[k] = fmaf(it[i].x, o[k], d[k]);
                } else {
                    const float a = fabsf(it[i].x), label = signbit(it[i].x) ? 0.0f : 1.0f;
                    const float f = dot_lane16A(h, o, D);
                    float g;
                    if (f > MAX_EXP) g = (label - 1) * a;
                    else if (f < -MAX_EXP) g = (label - 0) * a;
                    else {
                        int idx = (int)((f + MAX_EXP) * (EXP_TABLE_SIZE / MAX_EXP / 2));
                        if (idx < 0) idx = 0;
                        if (idx > EXP_TABLE_SIZE - 1) idx = EXP_TABLE_SIZE - 1;
                        g = (label - g_exp_table[idx]) * a;
                    }
                    for (int k = 0; k < D; k++) { h[k] = fmaf(g, o[k], h[k]); d[k] = fmaf(g, o[k], d[k]); }
                    it[i].x = g;                                   /* kept for phase B */
                }
            }
            const int whole = s0 == seg[row] && s1 == seg[row + 1];
            if (whole) { float* ro = own_out + (int64_t)row * D; for (int k = 0; k < D; k++) ro[k] = phase_b ? r[k] + d[k] : h[k]; }
            else memcpy(scratch + (2 * c + (s0 == start ? 0 : 1)) * D, d, (size_t)D * sizeof(float));
            s0 = s1;
        }
    }
    /* rows shared by several chunks: deltas added in chunk order to the row as it stood before the phase */
    for (int64_t c = 0; c < n_chunks; c++) {
        const int64_t start = c * C, end = start + C < n ? start + C : n;
        const int32_t row = it[end - 1].key;
        const int64_t r0 = seg[row], r1 = seg[row + 1];
        if (r1 <= end || r0 < start) continue;
        const float* r = own + (int64_t)row * D; float* ro = own_out + (int64_t)row * D;
        for (int k = 0; k < D; k++) h[k] = r[k];
        for (int64_t cc = c; cc <= (r1 - 1) / C; cc++) {
            const float* dd = scratch + (2 * cc + (r0 <= cc * C ? 0 : 1)) * D;
            for (int k = 0; k < D; k++) h[k] += dd[k];
        }
        for (int k = 0; k < D; k++) ro[k] = h[k];
    }
    free(scratch); free(h); free(d);
}

static int64_t train_block_sorted(const orc_train_config* cfg, orc_model* m, const int32_t* sen, const int64_t* wb, int64_t n_walks,
                                  int32_t max_len, int ep, int64_t total_walks, int64_t all_words, int part_ctx, int part_tgt) {
    const int D = cfg->dim, W = cfg->window, K = cfg->negative;
    const int64_t V = m->V, T = m->table_size;
    const int PN = cfg->part_n > 1 ? cfg->part_n : 1;
    const int64_t per = cfg->sorted_walks > 0 ? cfg->sorted_walks : n_walks;
    int64_t pairs = 0;
    int64_t* seg = (int64_t*)malloc((size_t)(V + 2) * sizeof(int64_t));
    for (int64_t w0 = 0; w0 < n_walks; w0 += per) {
        const int64_t w1 = w0 + per < n_walks ? w0 + per : n_walks;
        int64_t cap = 1024, n = 0;
        so_item* a = (so_item*)malloc((size_t)cap * sizeof(so_item));
        /* a synchronous mini-batch trains at ONE learning rate, that of its first walk (round 4: the device's items are single 64-bit words without a rate) */
        const float alpha = alpha_for(cfg, (int64_t)ep * m->total_words + cfg->words_before + wb[w0], all_words);
        for (int64_t w = w0; w < w1; w++) {
            int32_t buf[4096]; int len = 0;
            for (int j = 0; j < max_len && len < 4096; j++) { int32_t r = sen[w * max_len + j]; if (r >= 0) buf[len++] = r; }
            const int64_t gbase = (((int64_t)ep * total_walks) + cfg->walk_index_base + w) * (int64_t)max_len;
            for (int i = 0; i < len; i++) {
                const int32_t word = buf[i];
                if (PN > 1 && word % PN != part_tgt) continue;
                uint64_t s = orc_mix64(cfg->seed + (uint64_t)(gbase + i));
                s = s * W2V_MULT + 11;
                const int b = (int)(s % (uint64_t)W);
                const uint64_t s_centre = s;
                for (int aa = b; aa < W * 2 + 1 - b; aa++) {
                    if (aa == W) continue;
                    const int c = i - W + aa;
                    if (c < 0 || c >= len) continue;
                    const int32_t last = buf[c];
                    if (PN > 1) { if (last % PN != part_ctx) continue; s = orc_mix64(s_centre + (uint64_t)c); }
                    if (n + K + 1 > cap) { cap = cap * 2 + K + 1; a = (so_item*)realloc(a, (size_t)cap * sizeof(so_item)); }
                    a[n].key = word; a[n].other = last; a[n].x = alpha; n++;
                    for (int dd = 0; dd < K; dd++) {
                        s = s * W2V_MULT + 11;
                        int64_t target = m->table[(s >> 16) % (uint64_t)T];
                        if (target == 0 && V > 1) target = (int64_t)(s % (uint64_t)(V - 1)) + 1;
                        if (PN > 1) { target = target / PN * PN + part_tgt; if (target >= V) target -= PN; }
                        if (target == word) continue;
                        a[n].key = (int32_t)target; a[n].other = last; a[n].x = -alpha; n++;
                    }
                    pairs++;
                }
            }
        }
        if (n > 0) {
            so_item* b2 = (so_item*)malloc((size_t)n * sizeof(so_item));
            so_stable_sort(a, b2, n, V, seg);                                    /* by target row */
            /* phase A leaves the moved target rows in a shadow table: phase B reads the rows as they stood before the mini-batch */
            float* shadow = (float*)malloc((size_t)(V * D + 1) * sizeof(float));
            memcpy(shadow, m->syn1neg, (size_t)(V * D) * sizeof(float));
            so_phase(cfg, 0, m->syn1neg, shadow, m->syn0, b2, n, seg, D);
            for (int64_t i = 0; i < n; i++) { int32_t t = b2[i].key; b2[i].key = b2[i].other; b2[i].other = t; }     /* (context, target, g) in phase-A order */
            so_stable_sort(b2, a, n, V, seg);                                    /* by context row */
            so_phase(cfg, 1, m->syn0, m->syn0, m->syn1neg, a, n, seg, D);
            memcpy(m->syn1neg, shadow, (size_t)(V * D) * sizeof(float));         /* commit */
            free(shadow); free(b2);
        }
        free(a);
    }
    free(seg);
    return pairs;
}

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int orc_train_sgns(const int32_t* walks, int64_t n_walks, int32_t max_len,
                   const orc_train_config* cfg, orc_model** out) {
    return orc_train_sgns_from(walks, n_walks, max_len, cfg, NULL, NULL, NULL, out);
}

/* The same trainer continuing from a given state: `counts` ([n_vertices] token counts of the WHOLE corpus the vocabulary, the
 * unigram table and total_words come from, instead of the counts of `walks`) and the tables a run has reached so far (rows in the
 * vocabulary's order).  Used by the statistical parity tests: the device trains a long corpus, hands its tables over, and both
 * sides then train the same further slice of walks — sequentially here, at full concurrency there. */
int orc_train_sgns_from(const int32_t* walks, int64_t n_walks, int32_t max_len, const orc_train_config* cfg,
                        const int64_t* counts, const float* syn0_init, const float* syn1neg_init, orc_model** out) {
    return orc_train_sgns_from_hs(walks, n_walks, max_len, cfg, counts, syn0_init, syn1neg_init, NULL, out);
}
/* ... and the inner-node rows of the Huffman tree as well (use_hs; [V-1 x dim], syn1_init may be NULL: zeros, word2vec.c's start) */
int orc_train_sgns_from_hs(const int32_t* walks, int64_t n_walks, int32_t max_len, const orc_train_config* cfg,
                           const int64_t* counts, const float* syn0_init, const float* syn1neg_init, const float* syn1_init, orc_model** out) {
    if (!walks || !cfg || !out || cfg->dim <= 0 || cfg->window <= 0 || cfg->negative < 0 ||
        cfg->n_vertices <= 0 || cfg->table_size <= 0 || max_len <= 0) return 1;
    init_exp_table();
    const int D = cfg->dim; const int32_t NV = cfg->n_vertices;
    orc_model* m = (orc_model*)calloc(1, sizeof(orc_model));
    m->dim = D;

    /* --- vocabulary: count, drop < minWordFrequency, order by (count desc, vertex id asc) --- */
    int64_t* cnt = (int64_t*)calloc((size_t)NV, sizeof(int64_t));
    for (int64_t i = 0; i < n_walks * max_len; i++) {
        int32_t t = walks[i];
        if (t >= NV) { free(cnt); free(m); return 2; }
        if (t >= 0) cnt[t]++;
    }
    if (counts) memcpy(cnt, counts, (size_t)NV * sizeof(int64_t));
    int64_t V = 0;
    for (int32_t v = 0; v < NV; v++) if (cnt[v] >= cfg->min_count && cnt[v] > 0) V++;
    vc_item* items = (vc_item*)malloc((size_t)(V ? V : 1) * sizeof(vc_item));
    int64_t q = 0;
    for (int32_t v = 0; v < NV; v++) if (cnt[v] >= cfg->min_count && cnt[v] > 0) { items[q].cnt = cnt[v]; items[q].id = v; q++; }
    qsort(items, (size_t)V, sizeof(vc_item), vc_cmp);
    m->V = V;
    m->vocab_ids = (int32_t*)malloc((size_t)(V ? V : 1) * sizeof(int32_t));
    m->counts = (int64_t*)malloc((size_t)(V ? V : 1) * sizeof(int64_t));
    int32_t* remap = (int32_t*)malloc((size_t)NV * sizeof(int32_t));
    for (int32_t v = 0; v < NV; v++) remap[v] = -1;
    int64_t local_words = 0;
    for (int64_t r = 0; r < V; r++) {
        m->vocab_ids[r] = items[r].id; m->counts[r] = items[r].cnt; remap[items[r].id] = (int32_t)r;
        local_words += items[r].cnt;
    }
    free(items); free(cnt);
    m->total_words = cfg->total_words > 0 ? cfg->total_words : local_words;

    /* --- weights: word2vec.c InitNet — syn0 = ((lcg & 0xFFFF)/65536 - 0.5)/D, syn1neg = 0 --- */
    m->syn0 = (float*)malloc((size_t)(V * D + 1) * sizeof(float));
    m->syn1neg = (float*)calloc((size_t)(V * D + 1), sizeof(float));
    if (cfg->use_hs) { m->syn1 = (float*)calloc((size_t)(V * D + 1), sizeof(float)); create_binary_tree(m); }
    {
        uint64_t r = cfg->seed;
        for (int64_t a = 0; a < V; a++)
            for (int b = 0; b < D; b++) {
                r = r * W2V_MULT + 11;
                m->syn0[a * D + b] = (((r & 0xFFFF) / (float)65536) - 0.5f) / D;
            }
    }
    if (syn0_init) memcpy(m->syn0, syn0_init, (size_t)(V * D) * sizeof(float));
    if (syn1neg_init) memcpy(m->syn1neg, syn1neg_init, (size_t)(V * D) * sizeof(float));
    if (syn1_init && m->syn1 && V > 1) memcpy(m->syn1, syn1_init, (size_t)((V - 1) * D) * sizeof(float));

    /* --- unigram^0.75 table: word2vec.c InitUnigramTable --- */
    const int64_t T = cfg->table_size;
    m->table_size = T;
    m->table = (int32_t*)malloc((size_t)T * sizeof(int32_t));
    if (V > 0) {
        double train_words_pow = 0; const double power = 0.75;
        for (int64_t a = 0; a < V; a++) train_words_pow += pow((double)m->counts[a], power);
        int64_t i = 0;
        double d1 = pow((double)m->counts[0], power) / train_words_pow;
        for (int64_t a = 0; a < T; a++) {
            m->table[a] = (int32_t)i;
            if (a / (double)T > d1) {
                i++;
                d1 += (i < V ? pow((double)m->counts[i], power) : 0.0) / train_words_pow;
            }
            if (i >= V) i = V - 1;
        }
    } else memset(m->table, 0, (size_t)T * sizeof(int32_t));

    /* --- remap walks to vocab rows; per-walk exclusive prefix of in-vocab tokens --- */
    int32_t* sen = (int32_t*)malloc((size_t)(n_walks * max_len + 1) * sizeof(int32_t));
    int64_t* wb = (int64_t*)malloc((size_t)(n_walks + 1) * sizeof(int64_t));
    int64_t acc = 0;
    for (int64_t w = 0; w < n_walks; w++) {
        wb[w] = acc;
        for (int j = 0; j < max_len; j++) {
            int32_t t = walks[w * max_len + j];
            int32_t r = t >= 0 ? remap[t] : -1;
            sen[w * max_len + j] = r;
            if (r >= 0) acc++;
        }
    }
    free(remap);

    /* --- training (sequential when threads<=1; Hogwild blocks of walks otherwise) --- */
    const int64_t total_walks = cfg->total_walks > 0 ? cfg->total_walks : n_walks;
    const int64_t all_words = (int64_t)cfg->epochs * m->total_words;
    int64_t pairs = 0;
    double t0 = now_s();
    const int PN = cfg->part_n > 1 ? cfg->part_n : 1;
    if (V > 0)
    for (int ep = 0; ep < cfg->epochs; ep++)
    for (int blk = 0; blk < PN * PN; blk++) {
        const int part_ctx = blk % PN, part_tgt = (blk % PN + blk / PN) % PN;     /* episode blk / PN, rank blk % PN */
        if (cfg->sorted_chunk > 0) {
            pairs += train_block_sorted(cfg, m, sen, wb, n_walks, max_len, ep, total_walks, all_words, part_ctx, part_tgt);
            continue;
        }
        int nt = cfg->threads > 1 ? cfg->threads : 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(nt) reduction(+ : pairs)
#endif
        {
            float* neu1e = (float*)malloc((size_t)D * sizeof(float));
            int tid = 0, nthr = 1;
#ifdef _OPENMP
            tid = omp_get_thread_num(); nthr = omp_get_num_threads();
#endif
            int64_t lo = n_walks * tid / nthr, hi = n_walks * (tid + 1) / nthr;
            for (int64_t w = lo; w < hi; w++) {
                /* compact the sentence: DL4J/word2vec drop out-of-vocabulary tokens before windowing */
                int32_t buf[4096]; int len = 0;
                for (int j = 0; j < max_len && len < 4096; j++) { int32_t r = sen[w * max_len + j]; if (r >= 0) buf[len++] = r; }
                int64_t done = (int64_t)ep * m->total_words + cfg->words_before + wb[w];
                float alpha = alpha_for(cfg, done, all_words);
                int64_t gbase = (((int64_t)ep * total_walks) + cfg->walk_index_base + w) * (int64_t)max_len;
                pairs += (g_orc_plain || cfg->negative + 1 > ILP_MAX ? train_walk : train_walk_ilp)(cfg, m, buf, len, gbase, alpha, neu1e, part_ctx, part_tgt);
            }
            free(neu1e);
        }
    }
    m->seconds = now_s() - t0;
    m->pairs = pairs;
    free(sen); free(wb);
    *out = m;
    return 0;
}
