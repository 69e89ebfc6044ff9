// multi.hip — several GPUs of one node from ONE process: BASELINE config 5 behind the C boundary (include/mi_codec.h,
// "Several GPUs of one node").
//
// The reference's block loop (algorithms/deflate/deflate.c:47-63) hands lz77_compress one block after the other; blocks
// are independent here (fresh table per block), so device g encodes a contiguous range of them with its own
// context and the unchanged single-GPU pipeline — one host thread per device, no data-path collective — and the ranges
// meet on device 0:
//   sizes      8 bytes per device, on the host (the threads join there anyway)
//   streams    one group of point-to-point transfers into device 0: ncclGroupStart .. ncclSend / ncclRecv x (ndev - 1) ..
//              ncclGroupEnd over RCCL (every peer owns a direct xGMI link into device 0, so the variable-length gather is
//              not ring-bound), or hipMemcpyPeerAsync when a device is listed twice (the one-GPU test shape) / on request
//   placement  a shard that starts on a dword of the final stream is received in place; any other lands in a staging area
//              and k_bits_append shifts it in (bit-contiguous lz77 streams: the seam dword is OR-ed)
//   tables     per-device exclusive prefixes, rebased by their shard's first bit (k_table_rebase)
// librccl.so (0.5 GB) is loaded with dlopen on first use: libmi_codec.so does not link it, and a process that only ever
// uses one GPU never pays for it.
#include "common.h"
#include <rccl/rccl.h>          // types and prototypes only: every call goes through the table below
#include <dlfcn.h>
#include <pthread.h>
#include <stdlib.h>

#define MI_MULTI_MAX 64

struct mi_rccl_api {
    void *handle;
    decltype(&ncclCommInitAll)    CommInitAll;
    decltype(&ncclCommDestroy)    CommDestroy;
    decltype(&ncclGroupStart)     GroupStart;
    decltype(&ncclGroupEnd)       GroupEnd;
    decltype(&ncclSend)           Send;
    decltype(&ncclRecv)           Recv;
    decltype(&ncclGetErrorString) GetErrorString;
};

struct mi_multi_dev {
    int         device;
    mi_ctx     *ctx;
    hipStream_t stream;
    uint8_t    *d_in;   size_t in_cap;       // host-buffer entry points: the shard's bytes
    uint8_t    *d_out;  size_t out_cap;      // the shard's stream (device 0: the assembled stream of the host-buffer entry points)
    uint64_t   *d_bits; size_t bits_cap;     // the shard's block table (u64 entries)
};

struct mi_multi {
    int          ndev;
    mi_multi_dev dev[MI_MULTI_MAX];
    int          use_rccl;
    ncclComm_t   comm[MI_MULTI_MAX];
    int          comm_ok;
    mi_rccl_api  api;
    uint8_t     *d_stage; size_t stage_cap;  // device 0: shards that do not start on a dword, and the peers' block tables
    uint64_t    *h_tot;                      // pinned: bits produced per device
    char         terr[256];
};

// dst bits [bit0, bit0 + nbits) <- src bits [0, nbits) (LSB first, src dword aligned, one readable dword past its end).
// One thread per destination dword; the first one may share its low bits with what is already there (OR), every other is
// written whole — the same seam rule as k_lz_concat (lz_emit.hip) and the Huffman merge.
__global__ __launch_bounds__(256)
static void k_bits_append(uint32_t *__restrict__ dst, uint64_t bit0, const uint32_t *__restrict__ src, uint64_t nbits, uint64_t cap_words)
{
    const uint64_t w0 = bit0 >> 5;
    uint64_t w1 = (bit0 + nbits + 31) >> 5;
    if (w1 > cap_words) w1 = cap_words;
    for (uint64_t j = w0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < w1; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t g0 = j << 5;
        const uint64_t lo = g0 > bit0 ? g0 : bit0, hi = (g0 + 32 < bit0 + nbits) ? g0 + 32 : bit0 + nbits;
        const uint32_t k = (uint32_t)(hi - lo);                                   // 1..32
        const uint64_t sp = lo - bit0;
        const uint64_t two = (uint64_t)src[sp >> 5] | ((uint64_t)src[(sp >> 5) + 1] << 32);
        uint32_t v = (uint32_t)(two >> (sp & 31u));
        if (k < 32) v &= (1u << k) - 1u;
        v <<= (uint32_t)(lo - g0);
        if (j == w0 && (bit0 & 31u)) dst[j] |= v; else dst[j] = v;
    }
}

__global__ __launch_bounds__(256)
static void k_table_rebase(uint64_t *__restrict__ dst, const uint64_t *__restrict__ src, uint64_t count, uint64_t base)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) dst[i] = src[i] + base;
}

__device__ __forceinline__ uint8_t pattern_byte(uint64_t i) { return (uint8_t)(((uint32_t)i * 2654435761u) >> 24) ^ (uint8_t)(i >> 32); }
__global__ static void k_pattern_fill(uint8_t *p, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = pattern_byte(i);
}
__global__ static void k_pattern_check(const uint8_t *p, uint64_t n, uint32_t *bad)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) if (p[i] != pattern_byte(i)) atomicAdd(bad, 1u);
}

static mi_status transport_error(mi_multi *m, const char *what, const char *detail)
{
    snprintf(m->terr, sizeof m->terr, "%s: %s", what, detail ? detail : "");
    return MI_ERR_TRANSPORT;
}

static mi_status rccl_load(mi_multi *m)
{
    mi_rccl_api &a = m->api;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (int i = 0; i < 3 && !a.handle; ++i) a.handle = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!a.handle) return transport_error(m, "dlopen(librccl.so)", dlerror());
#define MI_RCCL_SYM(field, name) do { a.field = (decltype(a.field))dlsym(a.handle, name); if (!a.field) return transport_error(m, "dlsym", name); } while (0)
    MI_RCCL_SYM(CommInitAll, "ncclCommInitAll");
    MI_RCCL_SYM(CommDestroy, "ncclCommDestroy");
    MI_RCCL_SYM(GroupStart, "ncclGroupStart");
    MI_RCCL_SYM(GroupEnd, "ncclGroupEnd");
    MI_RCCL_SYM(Send, "ncclSend");
    MI_RCCL_SYM(Recv, "ncclRecv");
    MI_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef MI_RCCL_SYM
    return MI_OK;
}

template <typename T>
static mi_status grow(mi_ctx *ctx, T **p, size_t *cap, size_t want)          // device buffer of at least `want` elements (current device)
{
    if (want <= *cap && *p) return MI_OK;
    if (*p) { MI_HIP(ctx, hipDeviceSynchronize()); MI_HIP(ctx, hipFree(*p)); *p = nullptr; *cap = 0; }
    const size_t take = want + want / 8 + 64;
    if (hipMalloc((void **)p, take * sizeof(T)) != hipSuccess) { (void)hipGetLastError(); *p = nullptr; return MI_ERR_NOMEM; }
    *cap = take;
    return MI_OK;
}

extern "C" {

void mi_multi_shard(uint64_t nblocks, int g, int ndev, uint64_t *lo, uint64_t *hi)
{
    if (ndev < 1) ndev = 1;
    const uint64_t per = (nblocks + (uint64_t)ndev - 1) / (uint64_t)ndev;
    uint64_t a = (uint64_t)g * per;
    if (a > nblocks) a = nblocks;
    uint64_t b = a + per;
    if (b > nblocks) b = nblocks;
    if (lo) *lo = a;
    if (hi) *hi = b;
}

void mi_multi_destroy(mi_multi *m)
{
    if (!m) return;
    if (m->comm_ok && m->api.CommDestroy) for (int g = 0; g < m->ndev; ++g) if (m->comm[g]) (void)m->api.CommDestroy(m->comm[g]);
    for (int g = 0; g < m->ndev; ++g) {
        mi_multi_dev &d = m->dev[g];
        if (!d.ctx) continue;
        (void)hipSetDevice(d.device);
        (void)hipDeviceSynchronize();
        if (d.d_in) (void)hipFree(d.d_in);
        if (d.d_out) (void)hipFree(d.d_out);
        if (d.d_bits) (void)hipFree(d.d_bits);
        if (g == 0 && m->d_stage) (void)hipFree(m->d_stage);
        if (d.stream) (void)hipStreamDestroy(d.stream);
        mi_ctx_destroy(d.ctx);
    }
    if (m->h_tot) (void)hipHostFree(m->h_tot);
    // (the RCCL handle stays loaded: unloading a library with live device state is not worth its risk)
    free(m);
}

mi_status mi_multi_create(mi_multi **out, const int *devices, int ndev)
{
    if (!out || !devices || ndev < 1 || ndev > MI_MULTI_MAX) return MI_ERR_ARG;
    *out = nullptr;
    mi_multi *m = (mi_multi *)calloc(1, sizeof(mi_multi));
    if (!m) return MI_ERR_NOMEM;
    m->ndev = ndev;
    bool distinct = true;
    for (int g = 0; g < ndev; ++g) for (int k = 0; k < g; ++k) if (devices[k] == devices[g]) distinct = false;
    for (int g = 0; g < ndev; ++g) {
        mi_multi_dev &d = m->dev[g];
        d.device = devices[g];
        const mi_status st = mi_ctx_create(&d.ctx, devices[g]);                     // (leaves devices[g] current)
        if (st) { mi_multi_destroy(m); return st; }
        if (hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking) != hipSuccess) { mi_multi_destroy(m); return MI_ERR_HIP; }
    }
    if (hipHostMalloc((void **)&m->h_tot, MI_MULTI_MAX * sizeof(uint64_t), hipHostMallocDefault) != hipSuccess) { mi_multi_destroy(m); return MI_ERR_NOMEM; }
    const char *e = getenv("MI_MULTI_TRANSPORT");
    const bool want_rccl = e && e[0] == 'r', want_peer = e && e[0] == 'p';
    if (want_rccl && !distinct) { mi_multi_destroy(m); return MI_ERR_TRANSPORT; }      // one communicator rank per device
    m->use_rccl = want_rccl || (!want_peer && distinct && ndev > 1);
    if (m->use_rccl) {
        mi_status st = rccl_load(m);
        if (st == MI_OK) {
            const ncclResult_t r = m->api.CommInitAll(m->comm, ndev, devices);
            if (r != ncclSuccess) st = transport_error(m, "ncclCommInitAll", m->api.GetErrorString(r));
            else m->comm_ok = 1;
        }
        if (st) { fprintf(stderr, "mi_multi_create: %s\n", m->terr); mi_multi_destroy(m); return st; }
    } else {
        // peer copies: let device 0 reach its peers directly where the platform allows it (an error only means a staged copy)
        (void)hipSetDevice(devices[0]);
        for (int g = 1; g < ndev; ++g) if (devices[g] != devices[0]) { (void)hipDeviceEnablePeerAccess(devices[g], 0); (void)hipGetLastError(); }
    }
    *out = m;
    return MI_OK;
}

int         mi_multi_ndev(const mi_multi *m) { return m ? m->ndev : 0; }
mi_ctx     *mi_multi_ctx(mi_multi *m, int g) { return (m && g >= 0 && g < m->ndev) ? m->dev[g].ctx : nullptr; }
const char *mi_multi_transport(const mi_multi *m) { return (m && m->use_rccl) ? "rccl" : "peer-copy"; }
const char *mi_multi_last_transport_error(const mi_multi *m) { return m ? m->terr : ""; }

}  // extern "C"

// ---- the gather's transport: `count` transfers {source device index, source, destination on device 0, bytes}, all in flight
//      together; the destinations are ready for stream 0's next kernel when this returns (ordered on stream 0)
struct mi_xfer { int g; const void *src; void *dst; size_t bytes; };

static mi_status run_transfers(mi_multi *m, const mi_xfer *x, int count)
{
    mi_ctx *c0 = m->dev[0].ctx;
    hipStream_t s0 = m->dev[0].stream;
    if (count == 0) return MI_OK;
    if (!m->use_rccl) {
        MI_HIP(c0, hipSetDevice(m->dev[0].device));
        for (int i = 0; i < count; ++i) {
            if (!x[i].bytes) continue;
            MI_HIP(c0, hipMemcpyPeerAsync(x[i].dst, m->dev[0].device, x[i].src, m->dev[x[i].g].device, x[i].bytes, s0));
        }
        return MI_OK;
    }
    // RCCL: sends on the peers' streams (their encoders have finished: the worker threads synchronised), receives on stream 0
    ncclResult_t r = m->api.GroupStart();
    if (r != ncclSuccess) return transport_error(m, "ncclGroupStart", m->api.GetErrorString(r));
    ncclResult_t bad = ncclSuccess; const char *where = "";
    for (int i = 0; i < count && bad == ncclSuccess; ++i) {
        if (!x[i].bytes) continue;
        const int g = x[i].g;
        (void)hipSetDevice(m->dev[g].device);
        r = m->api.Send(x[i].src, x[i].bytes, ncclUint8, 0, m->comm[g], m->dev[g].stream);
        if (r != ncclSuccess) { bad = r; where = "ncclSend"; break; }
        (void)hipSetDevice(m->dev[0].device);
        r = m->api.Recv(x[i].dst, x[i].bytes, ncclUint8, g, m->comm[0], s0);
        if (r != ncclSuccess) { bad = r; where = "ncclRecv"; }
    }
    r = m->api.GroupEnd();
    (void)hipSetDevice(m->dev[0].device);
    if (bad != ncclSuccess) return transport_error(m, where, m->api.GetErrorString(bad));
    if (r != ncclSuccess) return transport_error(m, "ncclGroupEnd", m->api.GetErrorString(r));
    return MI_OK;
}

static mi_status drain_senders(mi_multi *m)
{
    if (!m->use_rccl) return MI_OK;
    for (int g = 1; g < m->ndev; ++g) {
        MI_HIP(m->dev[g].ctx, hipSetDevice(m->dev[g].device));
        MI_HIP(m->dev[g].ctx, hipStreamSynchronize(m->dev[g].stream));
    }
    (void)hipSetDevice(m->dev[0].device);
    return MI_OK;
}

// ---- per-device encode, one host thread each ---------------------------------------------------------------------------
struct mi_multi_job {
    mi_multi *m; int g; const mi_lz_params *p; int mode_h;
    const uint8_t *src; int src_is_host; uint64_t n_g, nb_g;
    uint8_t *out; uint64_t out_cap; uint64_t *bits;          // device 0: the caller's buffers; peers: NULL (their own)
    mi_status st;
};

static uint64_t lz_bound(const mi_lz_params *p, int mode_h, uint64_t n) { return mode_h ? mi_deflate_h_bound_bytes(n, p) : mi_lz_bound_bytes(n, p); }

static void *multi_worker(void *arg)
{
    mi_multi_job *j = (mi_multi_job *)arg;
    mi_multi *m = j->m;
    mi_multi_dev &d = m->dev[j->g];
    mi_ctx *ctx = d.ctx;
    j->st = MI_OK;
    m->h_tot[j->g] = 0;
    if (hipSetDevice(d.device) != hipSuccess) { j->st = MI_ERR_HIP; return nullptr; }
    if (j->nb_g == 0) return nullptr;
    const uint8_t *in = j->src;
    if (j->src_is_host) {
        if ((j->st = grow(ctx, &d.d_in, &d.in_cap, (size_t)j->n_g + 64))) return nullptr;
        if (hipMemcpyAsync(d.d_in, j->src, j->n_g, hipMemcpyHostToDevice, d.stream) != hipSuccess) { j->st = MI_ERR_HIP; return nullptr; }
        in = d.d_in;
    }
    uint8_t *out = j->out; uint64_t cap = j->out_cap; uint64_t *bits = j->bits;
    if (!out) {
        cap = lz_bound(j->p, j->mode_h, j->n_g) + 64;
        if ((j->st = grow(ctx, &d.d_out, &d.out_cap, (size_t)cap))) return nullptr;
        if ((j->st = grow(ctx, &d.d_bits, &d.bits_cap, (size_t)j->nb_g + 1))) return nullptr;
        out = d.d_out; bits = d.d_bits;
    }
    for (int attempt = 0; attempt < 2; ++attempt) {
        j->st = j->mode_h ? mi_deflate_h_encode_dev(ctx, j->p, in, j->n_g, out, cap, bits, d.stream)
                          : mi_lz_encode_dev(ctx, j->p, in, j->n_g, out, cap, bits, d.stream);
        if (j->st) return nullptr;
        if (hipMemcpyAsync(&m->h_tot[j->g], bits + j->nb_g, 8, hipMemcpyDeviceToHost, d.stream) != hipSuccess) { j->st = MI_ERR_HIP; return nullptr; }
        j->st = mi_sync(ctx, d.stream);
        if (j->st != MI_ERR_UNSTABLE) break;             // a sort came out unstable: the context ranks with ballots now, encode once more
    }
    return nullptr;
}

static mi_status multi_encode(mi_multi *m, const mi_lz_params *p, int mode_h, const uint8_t *const *d_in, const uint8_t *h_in, uint64_t n,
                              uint8_t *d_out0, uint64_t cap_bytes, uint64_t *d_block_bits0)
{
    if (!m || !p || !d_out0 || !d_block_bits0 || (n && !d_in && !h_in) || !p->block) return MI_ERR_ARG;
    // (blocks above 64 KiB — the lz77 flavour's sliced finder — synchronise their stream once per batch: harmless here, every device
    //  has its own host thread)
    if (((uintptr_t)d_out0 & 3u) != 0) return MI_ERR_ARG;
    if (cap_bytes < lz_bound(p, mode_h, n)) return MI_ERR_CAPACITY;
    const uint64_t nblocks = (n + p->block - 1) / p->block;
    const int nd = m->ndev;
    mi_ctx *c0 = m->dev[0].ctx;
    hipStream_t s0 = m->dev[0].stream;
    mi_multi_job job[MI_MULTI_MAX];
    uint64_t lo[MI_MULTI_MAX], hi[MI_MULTI_MAX];
    pthread_t th[MI_MULTI_MAX];
    bool started[MI_MULTI_MAX];
    for (int g = 0; g < nd; ++g) {
        mi_multi_shard(nblocks, g, nd, &lo[g], &hi[g]);
        const uint64_t b0 = lo[g] * p->block, b1 = hi[g] * p->block < n ? hi[g] * p->block : n;
        job[g] = mi_multi_job{m, g, p, mode_h, nullptr, h_in ? 1 : 0, hi[g] > lo[g] ? b1 - b0 : 0, hi[g] - lo[g],
                              g == 0 ? d_out0 : nullptr, g == 0 ? cap_bytes : 0, g == 0 ? d_block_bits0 : nullptr, MI_OK};
        if (job[g].nb_g) job[g].src = h_in ? h_in + b0 : d_in[g];
        if (job[g].nb_g && !job[g].src) return MI_ERR_ARG;
    }
    if (nblocks == 0) {                                    // an empty input: the table's single entry
        MI_HIP(c0, hipSetDevice(m->dev[0].device));
        MI_HIP(c0, hipMemsetAsync(d_block_bits0, 0, 8, s0));
        MI_HIP(c0, hipStreamSynchronize(s0));
        return MI_OK;
    }
    for (int g = 0; g < nd; ++g) started[g] = pthread_create(&th[g], nullptr, multi_worker, &job[g]) == 0;
    for (int g = 0; g < nd; ++g) { if (started[g]) pthread_join(th[g], nullptr); else multi_worker(&job[g]); }
    for (int g = 0; g < nd; ++g) if (job[g].st) return job[g].st;
    MI_HIP(c0, hipSetDevice(m->dev[0].device));
    // where every shard starts in the assembled stream
    uint64_t start[MI_MULTI_MAX + 1];
    start[0] = 0;
    for (int g = 0; g < nd; ++g) start[g + 1] = start[g] + m->h_tot[g];
    if ((start[nd] + 7) / 8 > cap_bytes) return MI_ERR_CAPACITY;
    // staging on device 0: shards that do not start on a dword + every peer's table
    size_t stage_need = 0, soff[MI_MULTI_MAX], toff[MI_MULTI_MAX];
    for (int g = 1; g < nd; ++g) {
        if (!job[g].nb_g) continue;
        const size_t bytes = (size_t)((m->h_tot[g] + 7) / 8);
        soff[g] = stage_need;
        if (start[g] & 31u) stage_need += mi_align_up(bytes + 8, 256);
        toff[g] = stage_need;
        stage_need += mi_align_up((size_t)(job[g].nb_g + 1) * 8, 256);
    }
    if (stage_need) { const mi_status st = grow(c0, &m->d_stage, &m->stage_cap, stage_need); if (st) return st; }
    mi_xfer x[2 * MI_MULTI_MAX]; int nx = 0;
    for (int g = 1; g < nd; ++g) {
        if (!job[g].nb_g) continue;
        const size_t bytes = (size_t)((m->h_tot[g] + 7) / 8);
        void *dst = (start[g] & 31u) ? (void *)(m->d_stage + soff[g]) : (void *)(d_out0 + start[g] / 8);
        x[nx++] = mi_xfer{g, m->dev[g].d_out, dst, bytes};
        x[nx++] = mi_xfer{g, m->dev[g].d_bits, m->d_stage + toff[g], (size_t)(job[g].nb_g + 1) * 8};
    }
    {
        mi_prof_scope pr(c0, "multi_gather", s0, start[nd] / 8);
        mi_status st = run_transfers(m, x, nx);
        if (st) return st;
    }
    // a shard received in place ends on a byte; the rest of its last dword must read as zero, as one GPU leaves it (the next
    // shard ORs its first bits into that dword)
    for (int g = 1; g < nd; ++g) {
        if (!job[g].nb_g || (start[g] & 31u)) continue;
        const uint64_t endb = (start[g + 1] + 7) / 8;
        if ((endb & 3u) && endb + (4 - (endb & 3u)) <= cap_bytes) MI_HIP(c0, hipMemsetAsync(d_out0 + endb, 0, 4 - (endb & 3u), s0));
    }
    for (int g = 1; g < nd; ++g) {
        if (!job[g].nb_g) continue;
        if (start[g] & 31u) {
            const uint64_t words = (m->h_tot[g] + 63) / 32;
            // (the staging area is not cleared: the dword past a shard's last is read, masked away, never written)
            hipLaunchKernelGGL(k_bits_append, dim3((unsigned)((words + 255) / 256 < 65535 ? (words + 255) / 256 : 65535)), dim3(256), 0, s0,
                               reinterpret_cast<uint32_t *>(d_out0), start[g], reinterpret_cast<const uint32_t *>(m->d_stage + soff[g]), m->h_tot[g], cap_bytes / 4);
        }
        const uint64_t cnt = job[g].nb_g + 1;
        hipLaunchKernelGGL(k_table_rebase, dim3((unsigned)((cnt + 255) / 256 < 1024 ? (cnt + 255) / 256 : 1024)), dim3(256), 0, s0,
                           d_block_bits0 + lo[g], reinterpret_cast<const uint64_t *>(m->d_stage + toff[g]), cnt, start[g]);
    }
    MI_HIP(c0, hipGetLastError());
    MI_HIP(c0, hipStreamSynchronize(s0));
    return drain_senders(m);
}

extern "C" {

mi_status mi_lz_encode_multi_dev(mi_multi *m, const mi_lz_params *p, int mode_h, const uint8_t *const *d_in, uint64_t n,
                                 uint8_t *d_out0, uint64_t cap_bytes, uint64_t *d_block_bits0)
{
    if (n && !d_in) return MI_ERR_ARG;
    return multi_encode(m, p, mode_h, d_in, nullptr, n, d_out0, cap_bytes, d_block_bits0);
}

static mi_status multi_encode_host(mi_multi *m, const mi_lz_params *p, int mode_h, const uint8_t *h_in, uint64_t n,
                                   uint8_t *h_out, uint64_t cap_bytes, uint64_t *h_block_bits)
{
    if (!m || !p || !h_out || !h_block_bits || (n && !h_in) || !p->block) return MI_ERR_ARG;
    const uint64_t bound = lz_bound(p, mode_h, n), nblocks = (n + p->block - 1) / p->block;
    if (cap_bytes < bound) return MI_ERR_CAPACITY;
    mi_multi_dev &d0 = m->dev[0];
    MI_HIP(d0.ctx, hipSetDevice(d0.device));
    mi_status st = grow(d0.ctx, &d0.d_out, &d0.out_cap, (size_t)bound + 64);
    if (st == MI_OK) st = grow(d0.ctx, &d0.d_bits, &d0.bits_cap, (size_t)nblocks + 1);
    if (st) return st;
    st = multi_encode(m, p, mode_h, nullptr, h_in, n, d0.d_out, bound + 64, d0.d_bits);
    if (st) return st;
    MI_HIP(d0.ctx, hipSetDevice(d0.device));
    MI_HIP(d0.ctx, hipMemcpy(h_block_bits, d0.d_bits, (nblocks + 1) * 8, hipMemcpyDeviceToHost));
    const uint64_t bytes = (h_block_bits[nblocks] + 7) / 8;
    if (bytes) MI_HIP(d0.ctx, hipMemcpy(h_out, d0.d_out, bytes, hipMemcpyDeviceToHost));
    return MI_OK;
}

mi_status mi_lz_encode_multi(mi_multi *m, const mi_lz_params *p, const uint8_t *h_in, uint64_t n,
                             uint8_t *h_out, uint64_t cap_bytes, uint64_t *h_block_bits)
{
    return multi_encode_host(m, p, 0, h_in, n, h_out, cap_bytes, h_block_bits);
}

mi_status mi_deflate_h_encode_multi(mi_multi *m, const mi_lz_params *p, const uint8_t *h_in, uint64_t n,
                                    uint8_t *h_out, uint64_t cap_bytes, uint64_t *h_block_bits)
{
    return multi_encode_host(m, p, 1, h_in, n, h_out, cap_bytes, h_block_bits);
}

mi_status mi_multi_selftest_transport(mi_multi *m, uint64_t bytes)
{
    if (!m || !bytes) return MI_ERR_ARG;
    const int last = m->ndev - 1;
    mi_multi_dev &dl = m->dev[last], &d0 = m->dev[0];
    uint8_t *src = nullptr;
    MI_HIP(dl.ctx, hipSetDevice(dl.device));
    if (hipMalloc((void **)&src, bytes) != hipSuccess) return MI_ERR_NOMEM;
    hipLaunchKernelGGL(k_pattern_fill, dim3(256), dim3(256), 0, dl.stream, src, bytes);
    mi_status st = hipStreamSynchronize(dl.stream) == hipSuccess ? MI_OK : MI_ERR_HIP;
    uint32_t h_bad = 1;
    if (st == MI_OK) { (void)hipSetDevice(d0.device); st = grow(d0.ctx, &m->d_stage, &m->stage_cap, (size_t)bytes + 256); }
    if (st == MI_OK) {
        uint32_t *err = mi_err_slot(d0.ctx, d0.stream);
        mi_xfer x{last, src, m->d_stage, (size_t)bytes};
        st = err ? run_transfers(m, &x, 1) : MI_ERR_HIP;
        if (st == MI_OK) {
            hipLaunchKernelGGL(k_pattern_check, dim3(256), dim3(256), 0, d0.stream, m->d_stage, bytes, err);
            if (hipMemcpyAsync(&h_bad, err, 4, hipMemcpyDeviceToHost, d0.stream) != hipSuccess || hipStreamSynchronize(d0.stream) != hipSuccess) st = MI_ERR_HIP;
        }
        if (st == MI_OK && m->use_rccl) { (void)hipSetDevice(dl.device); if (hipStreamSynchronize(dl.stream) != hipSuccess) st = MI_ERR_HIP; }
    }
    (void)hipSetDevice(dl.device);
    (void)hipFree(src);
    (void)hipSetDevice(d0.device);
    if (st == MI_OK && h_bad) st = transport_error(m, "selftest", "the bytes that arrived on device 0 are not the ones sent");
    return st;
}

}  // extern "C"
