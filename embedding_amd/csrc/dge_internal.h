// dge_internal.h — handle layouts and error plumbing shared by graph.hip and sgns.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include <string>
#include <vector>

#include "../../include/dge.h"

void dge_set_error(const char* fmt, ...);

#define DGE_FAIL(code, ...)            \
    do {                               \
        dge_set_error(__VA_ARGS__);    \
        return (code);                 \
    } while (0)

#define DGE_HIP(expr)                                                                         \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess) {                                                              \
            dge_set_error("HIP error %s at %s:%d: %s", hipGetErrorName(e__), __FILE__, __LINE__, #expr); \
            return DGE_ERR_DEVICE;                                                            \
        }                                                                                     \
    } while (0)

// Host synchronisations the library performs (blocking stream / device / event waits and blocking copies), counted process-wide: a multi-GPU episode must not
// contain any in steady state (tests count them: dge_host_sync_count, include/dge.h).  Every call site below goes through these wrappers.
extern std::atomic<int64_t> g_dge_host_syncs;
static inline hipError_t dge_counted_stream_sync(hipStream_t s) { g_dge_host_syncs.fetch_add(1, std::memory_order_relaxed); return hipStreamSynchronize(s); }
static inline hipError_t dge_counted_device_sync() { g_dge_host_syncs.fetch_add(1, std::memory_order_relaxed); return hipDeviceSynchronize(); }
static inline hipError_t dge_counted_event_sync(hipEvent_t e) { g_dge_host_syncs.fetch_add(1, std::memory_order_relaxed); return hipEventSynchronize(e); }
static inline hipError_t dge_counted_memcpy(void* d, const void* s, size_t n, hipMemcpyKind k) { g_dge_host_syncs.fetch_add(1, std::memory_order_relaxed); return hipMemcpy(d, s, n, k); }
#define hipStreamSynchronize(s) dge_counted_stream_sync(s)
#define hipDeviceSynchronize() dge_counted_device_sync()
#define hipEventSynchronize(e) dge_counted_event_sync(e)
#define hipMemcpy(d, s, n, k) dge_counted_memcpy((d), (s), (n), (k))

int dge_require_device(int device);   // DGE_OK when `device` is a usable gfx950 device, sets it current

template <typename T>
static inline int dge_dev_alloc(T** p, size_t n) {
    *p = nullptr;
    if (n == 0) n = 1;
    DGE_HIP(hipMalloc((void**)p, n * sizeof(T)));
    return DGE_OK;
}
static inline void dge_dev_free(void* p) { if (p) (void)hipFree(p); }

// scratch device buffer of one call: freed on every exit path
template <typename T>
struct dge_tmp {
    T* p = nullptr;
    dge_tmp() = default;
    dge_tmp(const dge_tmp&) = delete;
    dge_tmp& operator=(const dge_tmp&) = delete;
    ~dge_tmp() { if (p) (void)hipFree(p); }
    int alloc(size_t n) { if (p) { (void)hipFree(p); p = nullptr; } return dge_dev_alloc(&p, n); }
    T* release() { T* q = p; p = nullptr; return q; }
    operator T*() const { return p; }
};

// One alias slot as the walk kernel reads it: 32 bytes (two 16-byte loads of one aligned sector), the ONLY memory access of a walk
// step.  nbr_alias already resolves alias[i] to the neighbour it points at ("alias == -1" -> nbr itself), and the slot carries the
// row bounds of both candidates, so the next step needs no row_ptr look-up: one dependent round trip per step instead of two.
struct __attribute__((aligned(32))) dge_slot {
    double prob;
    int32_t nbr;
    int32_t nbr_alias;
    uint32_t base, k;               // edges of nbr:       [base, base + k)   (the store holds fewer than 2^32 edges)
    uint32_t base_alias, k_alias;   // edges of nbr_alias
};

struct dge_graph {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // COO staging in insertion order (allEdges, J/LayeredGraph.java:142)
    int64_t n_coo = 0, cap_coo = 0;
    int32_t *d_coo_src = nullptr, *d_coo_dst = nullptr;
    double* d_coo_w = nullptr;
    int32_t max_id = -1;
    // CSR
    bool csr_built = false, alias_built = false;
    int32_t V = 0;
    int64_t E = 0;
    int64_t* d_row_ptr = nullptr;
    int32_t* d_nbr = nullptr;
    double* d_w = nullptr;
    double* d_outdeg = nullptr;
    double* d_prob = nullptr;
    int32_t* d_alias = nullptr;
    dge_slot* d_slots = nullptr;
    // sources (J/LayeredGraph.java:145-148)
    int64_t S = 0;
    int32_t* d_srcv = nullptr;
    double src_weight_sum = 0.0;
    int src_stream_sum = 0;          // how set_sources summed (0 running +=, 1 DoubleStream.sum())
    bool src_sum_fixed = false;      // the host assigned sourceWeightSum itself (dge_graph_set_source_weight_sum)
    double* d_src_w = nullptr;
    double* d_src_prob = nullptr;
    int32_t* d_src_alias = nullptr;
    dge_slot* d_src_slots = nullptr;
};

struct dge_walks {
    int device = 0;
    int64_t n = 0;
    int32_t L = 0;
    int32_t* d = nullptr;
    uint64_t gen = 0;          // changes whenever the library writes the corpus (a trainer may keep what it derived from an unchanged one)
};
uint64_t dge_next_generation();

int dge_graph_ensure_csr(dge_graph* g);
// launches the strided walk kernel on `stream`; rows [row0,row0+n) of out (row length L)
int dge_launch_walks_strided(const dge_graph* g, hipStream_t stream, int32_t* d_out, int64_t n, int32_t L,
                             int64_t seed, int64_t first_index, int32_t* d_deadend_count);
