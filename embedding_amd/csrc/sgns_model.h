// sgns_model.h — the trainer's handle (vocabulary + tables + per-call work buffers), shared by sgns.hip (host side of the C ABI)
// and sgns_sorted.hip (the owner-computes schedule).
#pragma once
#include <algorithm>
#include <atomic>
#include <string>
#include <vector>

#include "dge_internal.h"

struct EventPair { hipEvent_t a, b; int kind; };

// ablation / test knobs (dge_set_tuning, include/dge.h): -1 = the library's own rule
extern std::atomic<int64_t> g_dge_tuning[DGE_TUNE_COUNT];      // (set from the host's thread, read by whichever thread launches: relaxed atomics, no ordering implied)

struct dge_sorted_work;       // buffers of the owner-computes schedule (sgns_sorted.hip)

struct dge_model {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    dge_train_config cfg{};
    int64_t V = 0;
    int32_t D = 0, stride = 0, NV = 0;
    int64_t T = 0;
    int64_t total_words = 0;
    double row_share_max = 1.0;                 // largest share one row has of the tokens / of the negative draws
    double neg_collision = 1.0;                 // sum of squared negative-sampling probabilities: P(two draws hit one row)
    int64_t hot_rows_auto = 0;                  // head rows that policy 7 keeps out of the lock protocol (see dge_model_create)
    int64_t hot_rows_serial = 0;                // head rows whose own pairs, serialised by the row's lock, would outlast a launch
    double neg_norm = 1.0;                      // sum of count^0.75 over the vocabulary (the unigram table's normaliser)
    // the same head for one block of an n-rank block schedule (computed on first use, kept per n: block_head in sgns.hip)
    int32_t block_head_n = 0; int64_t block_head_workers = 0; int64_t block_head_rows = 0;
    int n_cus = 256;
    float *d_syn0 = nullptr, *d_syn1neg = nullptr, *d_snap = nullptr;
    int placed_seen[3] = {0, 0, 0}; double placed_best[3] = {0, 0, 0}, placed_worst[3] = {0, 0, 0};   // table_alloc's report for syn0, syn1neg, syn1: candidates probed, their best and worst rate (GB/s)
    // hierarchical softmax (cfg.use_hs): inner-node table and the Huffman paths in CSR form
    float* d_syn1 = nullptr;
    int64_t* d_hs_off = nullptr; int32_t* d_hs_points = nullptr; uint64_t* d_hs_codes = nullptr;
    std::vector<int64_t> h_hs_off; std::vector<int32_t> h_hs_points; std::vector<uint64_t> h_hs_codes;
    std::vector<float> h_syn1;

    int32_t* d_vocab_ids = nullptr;
    int64_t* d_counts = nullptr;
    int32_t* d_remap = nullptr;
    uint4* d_ctab = nullptr; int64_t ctab_blocks = 0;   // word2vec's unigram table in rank-block form (neg_table_row)
    // ... and in run form (neg_row_by_runs) when the vocabulary has at most DGE_RUN_MAX distinct adjacent counts and the closed form matches the table
    // everywhere but in at most DGE_RUN_EXC slots; n_runs == 0: not available
    double* d_run_base = nullptr; uint32_t* d_run_row = nullptr; uint32_t* d_exc_slot = nullptr; int32_t* d_exc_row = nullptr;
    int32_t n_runs = 0, n_exc = 0;
    int32_t hs_rep_auto = 0;                    // hierarchical softmax: the inner nodes [V-1 - hs_rep_auto, V-1) are each on a tenth of all paths and more (copies in k_sgns_train_hsw)
    int32_t hs_rep_thr32[32] = {0};             // node >= hs_rep_thr32[k]: on more than k/32 of all paths
    int32_t hs_cold_auto = 0;                   // hierarchical softmax: inner nodes [0, hs_cold_auto) are each on fewer than 2e-5 of the paths
    float* d_exp = nullptr;
    // per-call work buffers
    int64_t cap_rows = 0; int32_t cap_L = 0;
    int32_t* d_sen = nullptr; int64_t* d_len = nullptr; int64_t* d_wb = nullptr;
    void* d_scan_tmp = nullptr; size_t scan_tmp_bytes = 0;
    unsigned long long* d_counters = nullptr;   // [0]=pairs [1]=words
    int* d_locks = nullptr;                     // commit-lock word per syn1neg row
    // host mirrors for the read-back API
    std::vector<float> h_syn0, h_syn1neg;
    std::vector<int32_t> h_vocab_ids, h_table;
    std::vector<int64_t> h_counts;
    // stats
    std::vector<EventPair> pending;
    double kernel_ms = 0, walk_ms = 0;
    int64_t launches = 0;
    int last_policy = -1; int64_t last_workers = 0; int32_t last_hot_rows = 0;   // what the latest launch ran with
    std::string last_kernel;                                                     // ... and the trainer kernel's name and form (dge_model_kernel)
    int32_t search_runs = 0, search_moved = 0; double search_ms_before = 0, search_ms_after = 0;      // dge_model_tune_placement's latest report
    int32_t part_n = 1, part_ctx = 0, part_tgt = 0;                              // block schedule (dge_model_set_partition)
    const int32_t* seen_rows = nullptr; int64_t seen_n = 0; int32_t seen_L = 0; uint64_t seen_gen = 0;   // what d_sen/d_len/d_wb were derived from
    hipEvent_t ev_peer = nullptr;                                                // stream-ordered partition copies: the handshake with the caller's stream
    dge_sorted_work* sorted = nullptr;                                           // update_policy 8 (allocated on first use)
};

// pairs of a full-length walk under DL4J's window (radius uniform in 1 .. W): what a launch's size is estimated from without reading anything back
static inline double dge_expected_pairs_per_walk(int L, int W) {
    double e = 0.0;
    for (int i = 0; i < L; i++)
        for (int r = 1; r <= W; r++) e += (double)(std::min(L - 1, i + r) - std::max(0, i - r)) / (double)W;
    return e;
}

// update_policy 8 (sgns_sorted.hip): one pass of the owner-computes schedule over compacted walks [0, n_rows) of m->d_sen
struct TrainParams;
int dge_sorted_train(dge_model* m, const TrainParams& p);
// items of one synchronous mini-batch of that schedule for this model (0: its vocabulary is too skewed for the schedule)
int64_t dge_sorted_batch_items(const dge_model* m, int part_n);
void dge_sorted_release(dge_model* m);
