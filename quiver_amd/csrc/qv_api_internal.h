// qv_api_internal.h — shared by the C-ABI translation units (qv_api.cpp: indexes; qv_graph_api.cpp: HNSW graphs;
// qv_sharded_api.cpp: multi-GPU shards).  Not installed.
#pragma once
#include "../../include/qv.h"
#include "qv_device.h"
#include "qv_coalesce.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <chrono>
#include <mutex>
#include <new>
#include <vector>

int qv_fail(int code, const char* fmt, ...);      // sets the thread-local message, returns code
#define fail qv_fail

#define HIPCHK(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(e_ == hipErrorOutOfMemory ? QV_ERR_OOM : QV_ERR_DEVICE, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

struct Buf {            // growable device buffer
    void* p = nullptr; size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return QV_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = std::max(bytes, (size_t)4096);
        HIPCHK(hipMalloc(&p, want));
        cap = want; return QV_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};
struct PinBuf {         // growable pinned host buffer
    void* p = nullptr; size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return QV_OK;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        size_t want = std::max(bytes, (size_t)4096);
        // coherent (fine-grained): a kernel's stores are visible to the host while the kernel is still running — the single-launch
        // small scan writes its results and then a sequence number here, and the host polls that instead of waiting for the stream
        HIPCHK(hipHostMalloc(&p, want, hipHostMallocCoherent));
        cap = want; return QV_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

struct Workspace { Buf ws; Buf tickets; std::mutex mu; };      // tickets: 64 zeroed words of the single-launch small scan (k_flat_scan_small)

// qv_api.cpp, for qv_sharded_api.cpp: exact scan over the rows of a device-resident candidate bitmap (see the definition)
int qv_internal_search_candidates_device(qv_index* idx, const float* d_queries, uint32_t nq, uint32_t k_stride, const uint64_t* d_candidates,
                                         uint64_t matching, uint32_t* d_rows_out, float* d_dist_out, void* stream);

// qv_api.cpp, for qv_sharded_api.cpp: the exact scan for the flagged queries of a batch, decided on the device (see the definition)
int qv_internal_redo_flagged_device(qv_index* idx, const float* d_queries, uint32_t nq, uint32_t k, const uint32_t* d_flags,
                                    uint32_t* d_rows_out, float* d_dist_out, void* stream);

// qv_api.cpp, for qv_sharded_api.cpp: forget (and free) the workspace the *_device entry points keep for `stream` — called when a
// call context of the sharded handle, and with it the stream, goes away
void qv_internal_drop_stream_workspace(qv_index* idx, void* stream);

struct SearchCtx {
    hipStream_t stream = nullptr;
    Buf d_q, d_rows, d_dist, d_ids, d_mask, ws, tickets;
    PinBuf h_q, h_rows, h_dist, h_ids, h_mask, h_flag;
    uint32_t flag_seq = 0;                   // sequence number the small scan writes into h_flag when its results are in h_rows / h_dist
    void release() {
        d_q.release(); d_rows.release(); d_dist.release(); d_ids.release(); d_mask.release(); ws.release(); tickets.release();
        h_q.release(); h_rows.release(); h_dist.release(); h_ids.release(); h_mask.release(); h_flag.release();
        if (stream) (void)hipStreamDestroy(stream);
        stream = nullptr;
    }
};


struct qv_index {
    int device = 0;
    int cus = 256;
    uint32_t dim = 0, dim4 = 0;
    int metric = QV_COSINE;
    uint64_t flags = 0;
    int filter = 0;                            // qv_index_set_filter
    uint32_t n_rows = 0, n_live = 0;
    uint64_t row_writes = 0;                   // bumped by every call that writes row CONTENTS (add / update): copies kept elsewhere (a graph's hubs) compare it
    uint64_t cap_tiles = 0;
    float* d_tiles = nullptr;
    double* d_rnorm = nullptr;
    uint64_t* d_alive = nullptr;
    float* d_rres = nullptr;                   // |r - bf16(r)| per row: refreshed by every call that writes rows
    float* d_rowmaj = nullptr;
    uint16_t* d_bf16 = nullptr;                // QV_FLAG_BF16_ROWS: refreshed by every call that writes rows
    std::vector<uint64_t> alive_host;          // mirror of d_alive, for size bookkeeping and validation
    Buf mut_stage;                             // grow-only staging buffer of the mutating calls (add / remove / update run under the
                                               // caller's exclusion, so one buffer serves them all: no hipMalloc per single-row Insert)

    std::mutex ctx_mu;
    std::vector<SearchCtx*> free_ctx;
    std::vector<SearchCtx*> all_ctx;
    uint64_t batched_redo = 0;                 // queries the MFMA path handed back to the exact scan
    bool profiling = false;                    // qv_index_profile: event pairs around scan kernels
    std::mutex prof_mu;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
    std::mutex ws_mu;
    std::map<hipStream_t, Workspace*> stream_ws;   // workspaces of the *_device entry points, one per caller stream
    // concurrent single-query callers (all the reference's host ever produces: collection.go:647, db.go:805-828) ride the next
    // pass together (qv_coalesce.h): one pass in flight — a flat scan is HBM-bound, a second one beside it only halves both —
    // and up to 256 queries per group, the size the matrix-core filter walks the corpus once for
    qvco::Front front{1, 256, 4};

    qv::IndexView view() const {
        qv::IndexView v;
        v.tiles = d_tiles; v.rnorm = d_rnorm; v.alive = d_alive; v.rres = d_rres; v.rowmaj = d_rowmaj; v.bf16 = d_bf16;
        v.dim = dim; v.dim4 = dim4; v.n_rows = n_rows; v.n_tiles = (n_rows + 63) / 64; v.metric = metric; v.filter = filter;
        return v;
    }
    size_t tile_bytes() const { return (size_t)dim4 * 64 * 16; }
    size_t bf16_tile_bytes() const { return (size_t)(((dim4 + 1) / 2 + 1) / 2) * 2 * 64 * 16; }   // whole 16-dim steps (k_bf16_plane's layout)
};

