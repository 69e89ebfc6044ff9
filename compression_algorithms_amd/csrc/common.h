// common.h — internal declarations shared by the HIP translation units of libmi_codec.so.
// gfx950 (CDNA4) only: 64-wide wavefronts are assumed everywhere.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/mi_codec.h"

#define MI_WAVE 64

// kernel timing records: plain C arrays (the host side of this library uses no C++ containers or strings)
#define MI_PROF_KERNELS 48
#define MI_PROF_PENDING 8192
struct mi_prof_entry {
    const char *name;                   // a string literal of the launch site
    double      ms;
    uint64_t    launches, bytes;
};
struct mi_prof_pending { hipEvent_t a, b; uint64_t bytes; int idx; };

struct mi_ctx {
    int         device = 0;
    hipStream_t stream = nullptr;       // the context's own stream (host-buffer entry points; created on first use: mi_host_stream)
    hipStream_t side = nullptr;         // second stream: overlaps the replay/parse of batch i with the find of batch i+1
    hipStream_t fb = nullptr;           // low priority: the (normally empty) fallback chain must not hold LDS-hungry launches in front of real work
    hipStream_t fb2 = nullptr;          // a second one, alive only while the input lives in the fallback (created / released at the start of an
                                        // encode call by the hint of the calls before it, lz_emit.hip): the chains of consecutive batches then run
                                        // side by side.  Not permanent: one more stream in the process — even idle, whatever the creation order or
                                        // GPU_MAX_HW_QUEUES — cost the text pipeline 7 % (20.4 -> 19.1 GB/s)
    hipStream_t parse = nullptr;        // third stage: parse / emit / concatenate
#define MI_SETS 3
    hipEvent_t  ev_find[MI_SETS] = {}, ev_done[MI_SETS] = {}, ev_part[MI_SETS] = {}, ev_fb[MI_SETS] = {}, ev_replay[MI_SETS] = {}, ev_wide[MI_SETS] = {};
    hipEvent_t  ev_fork = nullptr;
    int         last_hip = 0;
    int         profiling = 0;
    int         num_cu = 256;
    int         lds_rank_ok = 0;       // LDS returning atomics serve the lanes of one instruction in lane order (self-check at creation)
    int         test_break_rank = 0;   // MI_LZ_TEST_BREAK_RANK=1: the scatter mis-ranks on purpose (tests of the order check)
    int         test_force_fb = 0;     // MI_LZ_TEST_FORCE_FALLBACK=1 (lib_test only): every block goes to the fallback pipeline
    uint64_t   *d_stats = nullptr;     // device counters behind mi_lz_path_stats: [0] fallback blocks, [1] wide parts
    uint32_t   *h_order = nullptr;     // pinned, device-visible: set by a kernel that found a sort out of order (lz_common.h)
    uint32_t   *d_order = nullptr;     // the same word as the device addresses it
    uint32_t    order_violations = 0;  // violations noticed so far (mi_order_poll)
    uint32_t    order_reported = 0;    // ... and already returned through mi_sync
    // workspace (device), grown on demand, never inside a captured region
    void       *ws = nullptr;
    size_t      ws_bytes = 0;
    // small pinned host staging area for info structs
    void       *h_pinned = nullptr;
    size_t      h_pinned_bytes = 0;
    // decoder status words: every decode call takes its own (round robin), so calls on different streams of one context
    // never share one (they used to share ws[0])
    uint64_t   *lz_dbg = nullptr;      // phase counters of the LZ pipeline (MI_LZ_DEBUG=1), inside the workspace
    uint32_t   *d_err = nullptr;
    uint32_t    err_next = 0;
#define MI_ERR_SLOTS 256
    mi_prof_entry    prof[MI_PROF_KERNELS];
    int              nprof = 0;
    mi_prof_pending *pending = nullptr;     // [MI_PROF_PENDING], allocated when profiling is first switched on
    int              npending = 0;
    hipEvent_t      *event_pool = nullptr;  // [2 * MI_PROF_PENDING]
    int              npool = 0;
};

#define MI_HIP(ctx, call)                                                     \
    do {                                                                      \
        hipError_t e__ = (call);                                              \
        if (e__ != hipSuccess) {                                              \
            (ctx)->last_hip = (int)e__;                                       \
            return MI_ERR_HIP;                                                \
        }                                                                     \
    } while (0)

// has a kernel reported a sort out of order since the last poll?  If so the context ranks with ballots from now on.
// Returns 1 when a NEW violation was seen.  (The word lives in pinned host memory: no synchronisation.)
int mi_order_poll(mi_ctx *ctx);
// grow the context workspace to at least `bytes` (256-byte aligned carve-outs are the caller's job)
mi_status mi_ws_reserve(mi_ctx *ctx, size_t bytes);
// a zeroed-on-stream status word for one decode call (never the workspace: two streams may decode at once)
uint32_t *mi_err_slot(mi_ctx *ctx, hipStream_t s);
// host check of a block table (exclusive prefix in bits): monotonic, aligned, inside the stream
extern "C" mi_status mi_validate_block_table(const uint64_t *h_block_bits, uint64_t nblocks, uint64_t stream_bytes, uint32_t align_bits);
// the context's own stream for the host-buffer entry points, created on first use
hipStream_t mi_host_stream(mi_ctx *ctx);
// the block decoders without their closing status read (lz_emit.hip, defh.hip): host_api.hip launches one per chunk
mi_status mi_lz_decode_launch(mi_ctx *ctx, const mi_lz_params *p, const uint8_t *d_stream, uint64_t stream_bytes,
                              const uint64_t *d_block_bits, uint8_t *d_out, uint64_t n, uint32_t *err, hipStream_t s);
mi_status mi_deflate_h_decode_launch(mi_ctx *ctx, const mi_lz_params *p, const uint8_t *d_stream, uint64_t stream_bytes,
                                     const uint64_t *d_block_bits, uint8_t *d_out, uint64_t n, uint32_t *err, hipStream_t s);

// profiling: bracket a launch with events on the launch stream
struct mi_prof_scope {
    mi_ctx *ctx; int idx; hipStream_t s; hipEvent_t a = nullptr, b = nullptr; uint64_t pbytes = 0;
    mi_prof_scope(mi_ctx *c, const char *name, hipStream_t st, uint64_t bytes);
    ~mi_prof_scope();
};

static inline size_t mi_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// simple carve-out allocator over the workspace
struct mi_carver {
    uint8_t *base; size_t off = 0;
    explicit mi_carver(void *b) : base((uint8_t *)b) {}
    template <typename T> T *take(size_t count) {
        off = mi_align_up(off, 256);
        T *p = (T *)(base + off);
        off += count * sizeof(T);
        return p;
    }
};
