// lz_decode.h — what the three block decoders share (lz_decode.hip, defh.hip): one wave per block, every control value
// wave-uniform (the compiler keeps it in SGPRs), the stream held 64 dwords at a time in ONE VGPR per lane and read with
// v_readlane (a few cycles) instead of a dependent global load per token (~1 µs), the last RING bytes of output in an LDS
// ring so that several waves share a CU.
//
// Reference loops being replaced: algorithms/lz77/lz77.c:347-377 (lz77_decompress: read_bit / read_bits per token),
// algorithms/deflate/lz77.c (byte tokens written by write_literal / write_length_distance), algorithms/huffman/
// huffman.c:330-364 (tree walk per bit).
#pragma once
#include "lz_common.h"

// A window of 64 stream dwords in registers, the next 64 already in flight.  `first` is a dword-aligned pointer, `nw`
// the number of dwords that may be read (everything past it reads as zero: no access outside the block's own range).
struct WaveWords {
    const uint32_t *first;
    uint32_t nw, base, cur, nxt, lane;
    __device__ __forceinline__ uint32_t load(uint32_t i) const { return i < nw ? first[i] : 0u; }
    __device__ __forceinline__ void init(const uint32_t *p, uint32_t nwords, uint32_t lane_)
    {
        first = p; nw = nwords; base = 0; lane = lane_;
        cur = load(lane); nxt = load(64u + lane);
    }
    // dword i (wave-uniform, consumed in increasing order)
    __device__ __forceinline__ uint32_t get(uint32_t i)
    {
        if (__builtin_expect(i - base >= 64u, 0)) { base += 64u; cur = nxt; nxt = load(base + 64u + lane); }
        return (uint32_t)__builtin_amdgcn_readlane((int)cur, (int)(i - base));
    }
};

// LSB-first bit reader (the lz77 flavour's BitStream, lz77.c:110-160): `buf` holds `have` unread bits, bit 0 next
struct BitsLsb {
    WaveWords w;
    uint64_t buf; uint32_t have, widx;
    __device__ __forceinline__ void init(const uint8_t *stream, uint64_t bitpos, uint64_t nbits, uint32_t lane)
    {
        const uint64_t A = ((uint64_t)(uintptr_t)stream << 3) + bitpos;            // absolute bit address
        const uint32_t bit0 = (uint32_t)(A & 31u);
        const uint64_t words = (bit0 + nbits + 31u) >> 5;
        w.init(reinterpret_cast<const uint32_t *>((uintptr_t)((A >> 5) << 2)), words > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)words, lane);
        buf = (uint64_t)w.get(0) | ((uint64_t)w.get(1) << 32);
        have = 64u - bit0; widx = 2;
        buf >>= bit0;
    }
    // a token takes at most 32 bits: one word per refill keeps more than 32 unread bits in the buffer
    __device__ __forceinline__ void refill()
    {
        if (have <= 32u) { buf |= (uint64_t)w.get(widx) << have; have += 32u; ++widx; }
    }
    __device__ __forceinline__ uint32_t peek(uint32_t k) const { return (uint32_t)buf & ((1u << k) - 1u); }      // k <= 31
    __device__ __forceinline__ void skip(uint32_t k) { buf >>= k; have -= k; }
};

// MSB-first reader over u32 words (deflate/huffman.c:16-46 packs MSB first): bit 63 of `buf` is next
struct BitsMsb {
    WaveWords w;
    uint64_t buf; uint32_t have, widx;
    __device__ __forceinline__ void init(const uint32_t *words, uint32_t nwords, uint32_t lane)
    {
        w.init(words, nwords, lane);
        buf = ((uint64_t)w.get(0) << 32) | (uint64_t)w.get(1);
        have = 64u; widx = 2;
    }
    // called before every read of at most 32 bits: one word per refill keeps more than 32 unread bits in the buffer
    __device__ __forceinline__ void refill()
    {
        if (have <= 32u) { buf |= (uint64_t)w.get(widx) << (32u - have); have += 32u; ++widx; }
    }
    __device__ __forceinline__ uint32_t top32() const { return (uint32_t)(buf >> 32); }
    __device__ __forceinline__ void skip(uint32_t k) { buf <<= k; have -= k; }
};

// The output ring.  Position x lives in cell x & (RING-1) and is overwritten by position x + RING.  Quarters are copied out
// as soon as the write cursor has left them, so whatever has left the ring is in the output buffer already — and a match may
// reach back FARTHER than the ring: such source bytes are read from the output buffer itself (past the CU's L1, which may
// hold a line from before its bytes were written).  The ring therefore need not be the window: 4-8 KiB per wave put several
// times as many waves on a CU as a 32 / 64 KiB ring did, for a slower read on the far matches (lz_decode.hip has the A/B).
template <uint32_t RING>
struct OutRing {
    static constexpr uint32_t RM = RING - 1u, CH = RING / 4u;
    uint8_t *ring;           // LDS
    uint8_t *dst;            // global, the block's first byte
    uint32_t flushed, lane; bool aligned;
    __device__ __forceinline__ void init(uint8_t *lds, uint8_t *out, uint32_t lane_)
    {
        ring = lds; dst = out; flushed = 0; lane = lane_; aligned = (((uintptr_t)out) & 15u) == 0;
    }
    __device__ __forceinline__ void put_literal(uint32_t o, uint32_t byte) { if (lane == 0) ring[o & RM] = (uint8_t)byte; }
    // byte at position x < o: from the ring while it is there (and not about to be overwritten by the copy in progress, whose
    // last byte is position `oend - 1`), else from the output buffer
    __device__ __forceinline__ uint8_t fetch(uint32_t x, uint32_t oend) const
    {
        if (x + RING >= oend) return ring[x & RM];
        return __hip_atomic_load(dst + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // copy `take` bytes from distance d (1 <= d <= o): byte j comes from o - d + (j mod d)
    __device__ __forceinline__ void copy(uint32_t o, uint32_t d, uint32_t take)
    {
        const uint32_t oend = o + take;
        if (d >= take || d >= 64u) {
            // no source byte of a 64-byte step is written by the same step (or, for d >= 64, by an earlier lane of it)
            if (d + take <= RING) { for (uint32_t j = lane; j < take; j += 64u) ring[(o + j) & RM] = ring[(o - d + j) & RM]; }       // all of it in the ring
            else for (uint32_t j = lane; j < take; j += 64u) ring[(o + j) & RM] = fetch(o - d + j, oend);
        } else {
            for (uint32_t j = lane; j < take; j += 64u) ring[(o + j) & RM] = ring[(o - d + (j % d)) & RM];      // d < 64: near
        }
    }
    __device__ __forceinline__ void copy_out(uint32_t from, uint32_t to)
    {
        if (aligned && !((from | to) & 15u)) {
            for (uint32_t i = from + lane * 16u; i < to; i += 64u * 16u)
                *reinterpret_cast<uint4 *>(dst + i) = *reinterpret_cast<const uint4 *>(&ring[i & RM]);
        } else {
            for (uint32_t i = from + lane; i < to; i += 64u) dst[i] = ring[i & RM];
        }
    }
    __device__ __forceinline__ void advance(uint32_t o)
    {
        if (__builtin_expect(o - flushed >= CH, 0)) {                                                           // a token adds < CH bytes
            copy_out(flushed, flushed + CH); flushed += CH;
            __threadfence();                                  // far matches read these bytes back: they must have left the CU
        }
    }
    __device__ __forceinline__ void finish(uint32_t n) { copy_out(flushed, n); }
};
