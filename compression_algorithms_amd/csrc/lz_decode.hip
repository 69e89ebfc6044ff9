// lz_decode.hip — decoders of the two token formats, blocks of any size the encoders accept.
//
//   k_lz_decode_bytes<RING>   deflate flavour: {0, byte} / {1, distance lo, distance hi, length}
//                             (algorithms/deflate/lz77.c write_literal / write_length_distance)
//   k_lz_decode_bits<RING>    lz77 flavour: flag bit, 8-bit literal or wbits distance + lbits length, LSB first
//                             (algorithms/lz77/lz77.c:347-377)
//
// One wave per block.  Round 1's decoder fetched every token with its own global load (two dependent ~1 µs round trips per
// token, 443 ms per 10^9 bytes); here a wave keeps 64 stream dwords in one register per lane, reads them with v_readlane,
// and — for the byte format — classifies 64 two-byte units at once: a ballot marks the units that open a match, a run of
// literal units between two of them is ONE parallel LDS store.  The output passes through a small LDS ring (lz_decode.h: 8 KiB
// by default, far matches read the output buffer), so the wave slots, not the LDS, bound how many blocks share a CU.
#include "lz_decode.h"
#include <stdlib.h>

template <uint32_t RING>
__global__ __launch_bounds__(64)
void k_lz_decode_bytes(const uint8_t *__restrict__ stream, uint64_t stream_bytes, const uint64_t *__restrict__ block_bits, LzP P,
                       uint8_t *__restrict__ out, uint64_t n_total, uint32_t *__restrict__ err)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_ring[RING];
    const uint32_t lane = threadIdx.x;
    const uint64_t b = blockIdx.x;
    const uint64_t off = b * (uint64_t)P.block;
    const uint32_t n = (uint32_t)((n_total - off) < P.block ? (n_total - off) : P.block);
    const uint64_t p0 = block_bits[b], end = block_bits[b + 1];
    // the block's bit range must lie inside the stream (an untrusted table must not steer reads anywhere else); byte
    // tokens are whole bytes
    bool bad = end < p0 || end > stream_bytes * 8ull || (p0 & 7u) || ((end - p0) & 7u);
    if (bad) { if (lane == 0) atomicOr(err, 1u); return; }
    const uint8_t *src = stream + (p0 >> 3);
    // two-byte units; n bytes of output need at most 2 n of them
    const uint64_t units64 = (end - p0) >> 4;
    const uint32_t nunits = units64 > 2ull * n + 2u ? 2u * n + 2u : (uint32_t)units64;
    auto unit = [&](uint32_t u) -> uint32_t { return u < nunits ? (uint32_t)src[2ull * u] | ((uint32_t)src[2ull * u + 1] << 8) : 2u; };   // 2 = stop

    OutRing<RING> ring;
    ring.init(s_ring, out + off, lane);
    uint32_t u0 = 0, s = 0, o = 0;
    uint32_t cur = unit(lane), nxt = unit(64u + lane);
    bool stop = false;
    while (!stop) {
        const uint32_t lowb = cur & 0xFFu;
        const uint64_t m_match = __ballot(lowb == 1u), m_stop = __ballot(lowb > 1u);
        while (s < 64u) {
            const uint64_t rest = (m_match | m_stop) >> s;
            uint32_t r = rest ? (uint32_t)__builtin_ctzll(rest) : 64u - s;         // literal units from s on
            if (r) {
                if (r > n - o) r = n - o;
                if (lane >= s && lane < s + r) s_ring[(o + lane - s) & (RING - 1u)] = (uint8_t)(cur >> 8);
                o += r; s += r;
                __builtin_amdgcn_wave_barrier();
                if (o >= n) { stop = true; break; }
                ring.advance(o);
                continue;
            }
            if ((m_stop >> s) & 1ull) { stop = true; break; }                       // a flag above 1, or the end of the units
            if (u0 + s + 1u >= nunits) { stop = true; break; }                      // half a match token
            const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)cur, (int)s);
            const uint32_t t1 = s < 63u ? (uint32_t)__builtin_amdgcn_readlane((int)cur, (int)(s + 1u))
                                        : (uint32_t)__builtin_amdgcn_readlane((int)nxt, 0);
            const uint32_t d = (t0 >> 8) | ((t1 & 0xFFu) << 8), len = t1 >> 8;
            if (d == 0 || d > o) { bad = true; stop = true; break; }
            const uint32_t take = (o + len <= n) ? len : n - o;
            ring.copy(o, d, take);
            o += take; s += 2u;
            __builtin_amdgcn_wave_barrier();
            if (o >= n) { stop = true; break; }
            ring.advance(o);
        }
        if (stop) break;
        s -= 64u; u0 += 64u;
        cur = nxt; nxt = unit(u0 + 64u + lane);
    }
    if (o != n) bad = true;
    if (bad) { if (lane == 0) atomicOr(err, 1u); return; }
    ring.finish(n);
}

template <uint32_t RING>
__global__ __launch_bounds__(64)
void k_lz_decode_bits(const uint8_t *__restrict__ stream, uint64_t stream_bytes, const uint64_t *__restrict__ block_bits, LzP P,
                      uint8_t *__restrict__ out, uint64_t n_total, uint32_t *__restrict__ err)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_ring[RING];
    const uint32_t lane = threadIdx.x;
    const uint64_t b = blockIdx.x;
    const uint64_t off = b * (uint64_t)P.block;
    const uint32_t n = (uint32_t)((n_total - off) < P.block ? (n_total - off) : P.block);
    const uint64_t p0 = block_bits[b], end = block_bits[b + 1];
    bool bad = end < p0 || end > stream_bytes * 8ull;
    if (bad) { if (lane == 0) atomicOr(err, 1u); return; }
    const uint32_t MB = 1u + P.wbits + P.lbits, W = 1u << P.wbits;
    const uint32_t dmask = W - 1u, lmask = (1u << P.lbits) - 1u;
    // n bytes of output need at most n tokens
    uint64_t total = end - p0;
    const uint64_t most = (uint64_t)n * (MB > 9u ? MB : 9u);
    if (total > most) total = most;
    BitsLsb br;
    br.init(stream, p0, total, lane);
    OutRing<RING> ring;
    ring.init(s_ring, out + off, lane);
    uint32_t o = 0;
    uint64_t used = 0;
    while (o < n && used < total) {
        br.refill();
        const uint32_t flag = br.peek(1);
        const uint32_t need = flag ? MB : 9u;
        if (used + need > total) { bad = true; break; }
        if (!flag) {
            ring.put_literal(o, br.peek(9) >> 1);
            o += 1u;
        } else {
            const uint32_t t = (uint32_t)(br.buf >> 1);
            const uint32_t d = t & dmask, len = (t >> P.wbits) & lmask;
            if (d == 0 || d > o) { bad = true; break; }
            const uint32_t take = (o + len <= n) ? len : n - o;
            ring.copy(o, d, take);
            o += take;
        }
        br.skip(need); used += need;
        __builtin_amdgcn_wave_barrier();
        ring.advance(o);
    }
    if (o != n) bad = true;
    if (bad) { if (lane == 0) atomicOr(err, 1u); return; }
    ring.finish(n);
}

// host side --------------------------------------------------------------------------------------------------------
template <uint32_t RING>
static void launch_ring(const uint8_t *d_stream, uint64_t stream_bytes, const uint64_t *d_block_bits, const LzP &P, uint8_t *d_out,
                        uint64_t n, uint64_t nblocks, uint32_t *err, hipStream_t s)
{
    if (P.deflate) hipLaunchKernelGGL(k_lz_decode_bytes<RING>, dim3((unsigned)nblocks), dim3(64), 0, s, d_stream, stream_bytes, d_block_bits, P, d_out, n, err);
    else           hipLaunchKernelGGL(k_lz_decode_bits<RING>, dim3((unsigned)nblocks), dim3(64), 0, s, d_stream, stream_bytes, d_block_bits, P, d_out, n, err);
}

// An 8 KiB ring whatever the window (lz_decode.h: matches that reach back farther read the output buffer); MI_LZ_DECODE_RING=
// 4096 .. 65536 forces a size (65536: the ring-is-the-window shape) for A/B runs.
void lz_launch_decode(const uint8_t *d_stream, uint64_t stream_bytes, const uint64_t *d_block_bits, const LzP &P, uint8_t *d_out,
                      uint64_t n, uint64_t nblocks, uint32_t *err, hipStream_t s)
{
    const uint32_t W = 1u << P.wbits, need = W < P.block ? W : P.block;
    const char *e = getenv("MI_LZ_DECODE_RING");
    // few blocks (fewer than the waves a ring of the window's size lets the chip hold): more waves are no use, the far
    // reads only cost (381 blocks of 256 KiB: 5.3 GB/s with the window in the ring, 4.4 with 16 KiB)
    const uint32_t want = e ? (uint32_t)atoi(e) : (nblocks < 1024u ? need : 8192u);          // 8 KiB: 36 GB/s for byte tokens (16 KiB: 31, 4 KiB: 27, 2 KiB: 16)
    if (want <= 4096u) launch_ring<4096u>(d_stream, stream_bytes, d_block_bits, P, d_out, n, nblocks, err, s);
    else if (want <= 8192u) launch_ring<8192u>(d_stream, stream_bytes, d_block_bits, P, d_out, n, nblocks, err, s);
    else if (need <= 16384u || want <= 16384u) launch_ring<16384u>(d_stream, stream_bytes, d_block_bits, P, d_out, n, nblocks, err, s);
    else if (need <= 32768u || want <= 32768u) launch_ring<32768u>(d_stream, stream_bytes, d_block_bits, P, d_out, n, nblocks, err, s);
    else launch_ring<65536u>(d_stream, stream_bytes, d_block_bits, P, d_out, n, nblocks, err, s);
}
