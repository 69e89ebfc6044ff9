// heap_cells.h — the reference's array min-heap (algorithms/huffman/huffman.c:100-163) on ONE lane, with fewer LDS round trips.
//
// Both tree builders (k_huff_build, k_defh_lengths) replay the reference heap operation by operation — its tie-breaking (strict
// '<' on the frequency in both sifts, ties keep their places) defines the codes — and both wait for one lane's chain of dependent
// LDS reads.  A heap cell is `frequency << SH | node id`, so a comparison needs one read per node.  What this header adds
// (round 4): the sifts read AHEAD of their decisions.
//   sift down  the children AND the four grandchildren of a node are loaded together (six unconditional loads, out-of-range
//              ones replaced by an infinite cell), two levels are decided per round trip.  The step the reference takes at a
//              node — `best = me; if (l < best) best = l; if (r < best) best = r` — goes to r exactly when F(r) < F(l) (both
//              orders of its two compares give that), and goes anywhere only when that child is strictly below `me`.
//   sift up    an entry's ancestors are known before any compare: all (<= HEAP_DEPTH) are loaded at once, the stop is found in
//              registers.
// The cell count and node count live in the caller's registers, not in LDS (an `int` in the same __shared__ struct may alias
// the cell stores: it was re-read after every one of them).
#pragma once
#include <stdint.h>

#define HEAP_DEPTH 9          // cells < 512: an index has at most 8 ancestors; one spare

template <typename C, int SH>
struct HeapCells {
    static __device__ __forceinline__ C F(C c) { return (C)(c >> SH); }
    static constexpr C INF = ~(C)0;

    // cell at `idx`, or INF when the index is outside [0, n): the load itself is unconditional (address clamped)
    static __device__ __forceinline__ C at(const C *heap, int idx, int n) { const C v = heap[idx < n ? idx : 0]; return idx < n ? v : INF; }

    // place `me` at index i or below (the reference's heapify_down, huffman.c:121-140)
    static __device__ __forceinline__ void down(C *heap, const int n, int i, const C me)
    {
        for (;;) {
            const int l = 2 * i + 1;
            if (l >= n) break;
            const int ll = 2 * l + 1, rl = ll + 2;
            const C cl = heap[l];
            const C cr = at(heap, l + 1, n);
            const C cll = at(heap, ll, n), clr = at(heap, ll + 1, n), crl = at(heap, rl, n), crr = at(heap, rl + 1, n);
            const bool right1 = F(cr) < F(cl);
            const int m1 = right1 ? l + 1 : l;
            const C c1 = right1 ? cr : cl;
            if (!(F(c1) < F(me))) break;
            heap[i] = c1;
            const C a = right1 ? crl : cll, b = right1 ? crr : clr;
            const bool right2 = F(b) < F(a);
            const C c2 = right2 ? b : a;
            i = m1;
            if (!(F(c2) < F(me))) break;                 // (also when m1 has no children: INF is below nothing)
            heap[m1] = c2;
            i = 2 * m1 + 1 + (right2 ? 1 : 0);
        }
        heap[i] = me;
    }

    // the cell at the root leaves, the last one is sifted down from there (dequeue, huffman.c:150-160); n is the count BEFORE
    static __device__ __forceinline__ C pop(C *heap, int &n)
    {
        const C top = heap[0];
        const C last = heap[n - 1];
        --n;
        down(heap, n, 0, last);
        return top;
    }

    // append `me` and sift it up (enqueue + heapify_up, huffman.c:107-119, 142-148); n is the count BEFORE
    static __device__ __forceinline__ void push(C *heap, int &n, const C me)
    {
        int idx[HEAP_DEPTH + 1]; C c[HEAP_DEPTH + 1];
        idx[0] = n++;
#pragma unroll
        for (int k = 1; k <= HEAP_DEPTH; ++k) idx[k] = idx[k - 1] > 0 ? (idx[k - 1] - 1) >> 1 : 0;
#pragma unroll
        for (int k = 1; k <= HEAP_DEPTH; ++k) c[k] = heap[idx[k]];
        int pos = idx[0];
        bool going = true;
#pragma unroll
        for (int k = 1; k <= HEAP_DEPTH; ++k) {
            going = going && idx[k - 1] > 0 && F(me) < F(c[k]);
            if (going) { heap[idx[k - 1]] = c[k]; pos = idx[k]; }
        }
        heap[pos] = me;
    }
};
