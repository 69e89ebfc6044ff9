// ctx.hip — context, workspace, error strings and kernel timing for libmi_codec.so.
#include "common.h"
#include <stdlib.h>

// Self-check behind LZP_ARANK (lz_common.h, radix_pass): do the lanes of ONE returning LDS add that hit the same address
// receive their values in lane order?  The stable radix scatter relies on it; the ISA does not promise it, so every
// context measures it once (8 waves on their own counters, digit patterns from one address to 256, a few thousand
// instructions) and the ballot ranking is used if a single pair is out of order.  MI_LZ_NO_ARANK=1 forces the ballots.
__global__ __launch_bounds__(512)
static void k_lds_rank_probe(uint32_t rounds, uint32_t *__restrict__ bad)
{
    __shared__ uint32_t cnt[256 * 9];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (uint32_t i = tid; i < 256 * 9; i += 512) cnt[i] = 0;
    __syncthreads();
    uint32_t errs = 0, x = tid * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (uint32_t r = 0; r < rounds; ++r) {
        x = x * 1664525u + 1013904223u;
        const uint32_t ndig = 1u << (r % 9u);                                  // 1, 2, 4, ... 256 distinct addresses per wave
        const uint32_t d = (blockIdx.x & 1u) ? ((x >> 16) % ndig) : (((x >> 16) % ndig) * 32u) % 256u;   // odd blocks: spread; even: one bank
        const uint32_t old = atomicAdd(&cnt[d * 9 + wave], 1u);
        for (uint32_t o = 1; o < 64; ++o) {
            const uint32_t src = (lane + o) & 63u;
            const uint32_t od = __shfl(d, src), oo = __shfl(old, src);
            if (od == d && ((src < lane) != (oo < old))) ++errs;
        }
    }
    if (errs) atomicAdd(bad, errs);
}

static int lds_rank_selfcheck(mi_ctx *c)
{
    if (getenv("MI_LZ_NO_ARANK")) return 0;
    uint32_t h = 1;
    if (hipMemset(c->d_err, 0, 4) != hipSuccess) return 0;
    hipLaunchKernelGGL(k_lds_rank_probe, dim3(64), dim3(512), 0, 0, 72u, c->d_err);
    if (hipGetLastError() != hipSuccess || hipMemcpy(&h, c->d_err, 4, hipMemcpyDeviceToHost) != hipSuccess) return 0;
    (void)hipMemset(c->d_err, 0, 4);
    return h == 0;
}

extern "C" {

const char *mi_version(void) { return "mi_codec 0.1 (gfx950)"; }

const char *mi_status_str(mi_status s)
{
    switch (s) {
    case MI_OK: return "ok";
    case MI_ERR_ARG: return "bad argument";
    case MI_ERR_HIP: return "HIP call failed";
    case MI_ERR_NOMEM: return "out of memory";
    case MI_ERR_CAPACITY: return "output buffer too small";
    case MI_ERR_EMPTY_INPUT: return "empty input (reference: ERROR: Queue is empty)";
    case MI_ERR_SINGLE_SYMBOL: return "single distinct symbol (reference: ERROR: No code for character)";
    case MI_ERR_CODE_TOO_LONG: return "Huffman code longer than 32 bits";
    case MI_ERR_CORRUPT: return "corrupt stream";
    case MI_ERR_NO_DEVICE: return "no HIP device: this library has no CPU fallback";
    case MI_ERR_TRANSPORT: return "multi-device gather: RCCL missing or an RCCL call failed";
    case MI_ERR_UNSTABLE: return "a sort came out unstable (LDS atomics not lane-ordered): encode again, the context now ranks with ballots";
    }
    return "unknown";
}

mi_status mi_ctx_create(mi_ctx **out, int device)
{
    if (!out) return MI_ERR_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count)
        return MI_ERR_NO_DEVICE;
    mi_ctx *c = (mi_ctx *)calloc(1, sizeof(mi_ctx));
    if (!c) return MI_ERR_NOMEM;
    c->num_cu = 256;
    c->device = device;
    if (hipSetDevice(device) != hipSuccess) { free(c); return MI_ERR_NO_DEVICE; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->num_cu = prop.multiProcessorCount;
    // The host-buffer stream is created on first use (mi_host_stream).  Measured on the deflate pipeline, same box: the
    // creation order of these streams does not matter (11.83 .. 11.93 GB/s for three orders); a HIGH-priority stream for
    // the partition + find stage cost 5-7 % even while it sat idle, so there is none; a fifth normal-priority stream
    // (the wave replay beside the lane replays) cost 4 % with the wave replay on it and 7 % idle: the process's streams
    // share a handful of hardware queues, and one more stream re-deals them.
    {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);      // lo = least urgent
        // stage B's stream: MI_SIDE_PRIO=lo creates it at the least urgent priority (A/B: partition + find are the chain the
        // pipeline is bound by, and the replay kernels' small waves fragment the LDS their workgroups wait for)
        const char *sp_ = getenv("MI_SIDE_PRIO");
        bool ok = (sp_ && sp_[0] == 'l') ? hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, lo) == hipSuccess
                                         : hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) == hipSuccess;
        ok = ok && hipStreamCreateWithPriority(&c->fb, hipStreamNonBlocking, lo) == hipSuccess;
        // stage C at raised priority: k_lz_parse_emit wants a whole CU's LDS and only gets one when all three workgroups of
        // k_lz2_find on it have left — first in line it spans 4.5 ms per launch instead of 6.4 and the step is 0.6 % shorter
        // (MI_PARSE_PRIO=0: default priority, for A/B)
        const char *pp = getenv("MI_PARSE_PRIO");
        if (pp && pp[0] == '0') ok = ok && hipStreamCreateWithFlags(&c->parse, hipStreamNonBlocking) == hipSuccess;
        else ok = ok && hipStreamCreateWithPriority(&c->parse, hipStreamNonBlocking, hi) == hipSuccess;
        if (!ok) { mi_ctx_destroy(c); return MI_ERR_HIP; }
    }
    for (int i = 0; i < MI_SETS; ++i) {
        hipEventCreateWithFlags(&c->ev_replay[i], hipEventDisableTiming);
        hipEventCreateWithFlags(&c->ev_wide[i], hipEventDisableTiming);
        hipEventCreateWithFlags(&c->ev_part[i], hipEventDisableTiming);
        hipEventCreateWithFlags(&c->ev_fb[i], hipEventDisableTiming);
        hipEventCreateWithFlags(&c->ev_find[i], hipEventDisableTiming);
        hipEventCreateWithFlags(&c->ev_done[i], hipEventDisableTiming);
    }
    hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    c->h_pinned_bytes = 1 << 16;
    if (hipHostMalloc(&c->h_pinned, c->h_pinned_bytes, hipHostMallocDefault) != hipSuccess) {
        mi_ctx_destroy(c); return MI_ERR_NOMEM;
    }
    if (hipMalloc((void **)&c->d_err, MI_ERR_SLOTS * sizeof(uint32_t)) != hipSuccess) { mi_ctx_destroy(c); return MI_ERR_NOMEM; }
    if (hipHostMalloc((void **)&c->h_order, 64, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **)&c->d_order, c->h_order, 0) != hipSuccess) { mi_ctx_destroy(c); return MI_ERR_NOMEM; }
    c->h_order[0] = 0; c->h_order[1] = 0; c->h_order[2] = 0;   // [0] order flag, [1] blocks the last batch sent to the fallback (a grid-size hint), [2] blocks of that batch
    c->lds_rank_ok = lds_rank_selfcheck(c);
    c->test_break_rank = getenv("MI_LZ_TEST_BREAK_RANK") != nullptr;
    c->test_force_fb = getenv("MI_LZ_TEST_FORCE_FALLBACK") != nullptr;       // (only a -DMI_TEST_HOOKS build looks at the flag)
    if (hipMalloc((void **)&c->d_stats, 64) != hipSuccess || hipMemset(c->d_stats, 0, 64) != hipSuccess) { mi_ctx_destroy(c); return MI_ERR_NOMEM; }
    *out = c;
    return MI_OK;
}

void mi_ctx_destroy(mi_ctx *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    for (int i = 0; i < c->npending; ++i) { hipEventDestroy(c->pending[i].a); hipEventDestroy(c->pending[i].b); }
    for (int i = 0; i < c->npool; ++i) hipEventDestroy(c->event_pool[i]);
    free(c->pending); free(c->event_pool);
    if (c->ws) hipFree(c->ws);
    if (c->d_err) hipFree(c->d_err);
    if (c->d_stats) hipFree(c->d_stats);
    if (c->h_pinned) hipHostFree(c->h_pinned);
    if (c->h_order) hipHostFree(c->h_order);
    if (c->stream) hipStreamDestroy(c->stream);
    if (c->side) hipStreamDestroy(c->side);
    if (c->fb) hipStreamDestroy(c->fb);
    if (c->fb2) hipStreamDestroy(c->fb2);
    if (c->parse) hipStreamDestroy(c->parse);
    for (int i = 0; i < MI_SETS; ++i) {
        if (c->ev_replay[i]) hipEventDestroy(c->ev_replay[i]);
        if (c->ev_wide[i]) hipEventDestroy(c->ev_wide[i]);
        if (c->ev_find[i]) hipEventDestroy(c->ev_find[i]);
        if (c->ev_done[i]) hipEventDestroy(c->ev_done[i]);
        if (c->ev_part[i]) hipEventDestroy(c->ev_part[i]);
        if (c->ev_fb[i]) hipEventDestroy(c->ev_fb[i]);
    }
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    free(c);
}

int mi_last_hip_error(const mi_ctx *c) { return c ? c->last_hip : 0; }

mi_status mi_sync(mi_ctx *c, void *stream)
{
    if (!c) return MI_ERR_ARG;
    MI_HIP(c, hipStreamSynchronize((hipStream_t)stream));
    mi_order_poll(c);
    if (c->order_violations != c->order_reported) { c->order_reported = c->order_violations; return MI_ERR_UNSTABLE; }
    return MI_OK;
}

uint32_t mi_order_violations(mi_ctx *c)
{
    if (!c) return 0;
    mi_order_poll(c);
    return c->order_violations;
}

mi_status mi_lz_path_stats(mi_ctx *c, uint64_t *fallback_blocks, uint64_t *wide_parts)
{
    if (!c || !c->d_stats) return MI_ERR_ARG;
    uint64_t h[2] = {0, 0};
    MI_HIP(c, hipSetDevice(c->device));
    MI_HIP(c, hipDeviceSynchronize());
    MI_HIP(c, hipMemcpy(h, c->d_stats, 16, hipMemcpyDeviceToHost));
    if (fallback_blocks) *fallback_blocks = h[0];
    if (wide_parts) *wide_parts = h[1];
    return MI_OK;
}

mi_status mi_set_profiling(mi_ctx *c, int on)
{
    if (!c) return MI_ERR_ARG;
    if (on && !c->pending) {
        c->pending = (mi_prof_pending *)calloc(MI_PROF_PENDING, sizeof(mi_prof_pending));
        c->event_pool = (hipEvent_t *)calloc(2 * MI_PROF_PENDING, sizeof(hipEvent_t));
        if (!c->pending || !c->event_pool) { free(c->pending); free(c->event_pool); c->pending = nullptr; c->event_pool = nullptr; return MI_ERR_NOMEM; }
    }
    c->profiling = on;
    return MI_OK;
}

int mi_get_kernel_times(mi_ctx *c, mi_kernel_time *out, int cap)
{
    if (!c || !out) return 0;
    for (int i = 0; i < c->npending; ++i) {
        mi_prof_pending &q = c->pending[i];
        float ms = 0.f;
        hipEventSynchronize(q.b);
        if (hipEventElapsedTime(&ms, q.a, q.b) == hipSuccess) {
            mi_prof_entry &p = c->prof[q.idx];
            p.ms += ms; p.launches += 1; p.bytes += q.bytes;
        }
        c->event_pool[c->npool++] = q.a;
        c->event_pool[c->npool++] = q.b;
    }
    c->npending = 0;
    int n = 0;
    for (int i = 0; i < c->nprof; ++i) {
        mi_prof_entry &p = c->prof[i];
        if (!p.launches || n >= cap) continue;
        out[n].name = p.name;
        out[n].ms = p.ms / (double)p.launches;
        out[n].launches = p.launches;
        out[n].bytes = p.bytes;
        ++n;
        p.ms = 0; p.launches = 0; p.bytes = 0;
    }
    return n;
}

}  // extern "C"

uint32_t *mi_err_slot(mi_ctx *c, hipStream_t s)
{
    uint32_t *e = c->d_err + (c->err_next++ % MI_ERR_SLOTS);
    if (hipMemsetAsync(e, 0, 4, s) != hipSuccess) return nullptr;
    return e;
}

// A block table is an exclusive prefix sum of per-block stream lengths in bits.  Decoders index the stream with it, so a
// table taken from a file or a peer is checked before any kernel sees it: non-decreasing, every entry a multiple of
// `align_bits` (1 for the bit-packed lz77 flavour, 8 for deflate tokens, 32 for mode-H / FSE records), the last one inside
// the stream.
extern "C" mi_status mi_validate_block_table(const uint64_t *t, uint64_t nblocks, uint64_t stream_bytes, uint32_t align_bits)
{
    if (!t || !align_bits) return MI_ERR_ARG;
    if (stream_bytes > (UINT64_MAX >> 3)) return MI_ERR_ARG;
    const uint64_t limit = stream_bytes * 8;
    uint64_t prev = t[0];
    if (prev % align_bits) return MI_ERR_CORRUPT;
    for (uint64_t i = 1; i <= nblocks; ++i) {
        const uint64_t v = t[i];
        if (v < prev || (v % align_bits)) return MI_ERR_CORRUPT;
        prev = v;
    }
    return prev <= limit ? MI_OK : MI_ERR_CORRUPT;
}

hipStream_t mi_host_stream(mi_ctx *c)
{
    if (!c->stream && hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) c->stream = nullptr;   // NULL stream then
    return c->stream;
}


int mi_order_poll(mi_ctx *c)
{
    if (!c->h_order) return 0;
    const uint32_t v = __atomic_exchange_n(c->h_order, 0u, __ATOMIC_RELAXED);
    if (!v) return 0;
    ++c->order_violations;
    c->lds_rank_ok = 0;                 // ballots from now on (and the test hook, which needs LZP_ARANK, is off with it)
    return 1;
}

mi_status mi_ws_reserve(mi_ctx *c, size_t bytes)
{
    if (bytes <= c->ws_bytes) return MI_OK;
    // growing frees the old block: only legal while nothing that uses it is in flight
    MI_HIP(c, hipDeviceSynchronize());
    if (c->ws) { MI_HIP(c, hipFree(c->ws)); c->ws = nullptr; c->ws_bytes = 0; }
    size_t want = mi_align_up(bytes + bytes / 8, 1 << 20);
    if (hipMalloc(&c->ws, want) != hipSuccess) {
        // without the slack; a caller that can work in smaller batches halves its request on MI_ERR_NOMEM (lz_emit.hip)
        (void)hipGetLastError();
        want = mi_align_up(bytes, 1 << 20);
        c->ws = nullptr;
        if (hipMalloc(&c->ws, want) != hipSuccess) { (void)hipGetLastError(); c->ws = nullptr; return MI_ERR_NOMEM; }
    }
    c->ws_bytes = want;
    return MI_OK;
}

static hipEvent_t take_event(mi_ctx *c)
{
    if (c->npool > 0) return c->event_pool[--c->npool];
    hipEvent_t e = nullptr;
    hipEventCreate(&e);
    return e;
}

mi_prof_scope::mi_prof_scope(mi_ctx *c, const char *name, hipStream_t st, uint64_t bytes) : ctx(c), idx(-1), s(st)
{
    if (!c->profiling || !c->pending || c->npending >= MI_PROF_PENDING) return;
    for (int i = 0; i < c->nprof; ++i) if (strcmp(c->prof[i].name, name) == 0) { idx = i; break; }
    if (idx < 0) {
        if (c->nprof >= MI_PROF_KERNELS) return;
        idx = c->nprof++;
        c->prof[idx].name = name; c->prof[idx].ms = 0; c->prof[idx].launches = 0; c->prof[idx].bytes = 0;
    }
    a = take_event(c); b = take_event(c);
    pbytes = bytes;
    hipEventRecord(a, s);
}

mi_prof_scope::~mi_prof_scope()
{
    if (idx < 0) return;
    hipEventRecord(b, s);
    mi_prof_pending &q = ctx->pending[ctx->npending++];
    q.a = a; q.b = b; q.bytes = pbytes; q.idx = idx;
}
