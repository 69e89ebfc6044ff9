// Host harness for csrc/heap_cells.h (tests/test_heap_cells.py): the look-ahead sifts against the plain restatement of the
// reference heap (algorithms/huffman/huffman.c:100-163) on random and tie-heavy frequency sets — same array after every operation.
#define __device__
#define __forceinline__ inline
#include "../compression_algorithms_amd/csrc/heap_cells.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

template <typename C, int SH>
struct Plain {
    static C F(C c) { return (C)(c >> SH); }
    static void up(C *h, int i) { const C me = h[i]; while (i > 0) { const int p = (i - 1) >> 1; if (!(F(me) < F(h[p]))) break; h[i] = h[p]; i = p; } h[i] = me; }
    static void down(C *h, int n, int i) {
        const C me = h[i];
        for (;;) {
            const int l = 2 * i + 1, r = l + 1; int best = i; C bc = me;
            if (l < n && F(h[l]) < F(bc)) { best = l; bc = h[l]; }
            if (r < n && F(h[r]) < F(bc)) { best = r; bc = h[r]; }
            if (best == i) break;
            h[i] = bc; i = best;
        }
        h[i] = me;
    }
    static C pop(C *h, int &n) { const C t = h[0]; h[0] = h[--n]; down(h, n, 0); return t; }
    static void push(C *h, int &n, C me) { h[n++] = me; up(h, n - 1); }
};

template <typename C, int SH>
static int run(unsigned seed, int nsym, unsigned fmask)
{
    srand(seed);
    std::vector<C> a(600), b(600);
    int na = 0, nb = 0, id = 0;
    for (int s = 0; s < nsym; ++s) {
        const unsigned f = ((unsigned)rand() & fmask);
        if (!f) continue;
        const C cell = ((C)f << SH) | (C)id++;
        Plain<C, SH>::push(a.data(), na, cell);
        HeapCells<C, SH>::push(b.data(), nb, cell);
        if (na != nb || memcmp(a.data(), b.data(), sizeof(C) * na)) return 1;
    }
    while (na > 1) {
        const C l1 = Plain<C, SH>::pop(a.data(), na), r1 = Plain<C, SH>::pop(a.data(), na);
        const C l2 = HeapCells<C, SH>::pop(b.data(), nb), r2 = HeapCells<C, SH>::pop(b.data(), nb);
        if (l1 != l2 || r1 != r2 || na != nb || memcmp(a.data(), b.data(), sizeof(C) * na)) return 2;
        const C cell = ((C)(Plain<C, SH>::F(l1) + Plain<C, SH>::F(r1)) << SH) | (C)id++;
        Plain<C, SH>::push(a.data(), na, cell);
        HeapCells<C, SH>::push(b.data(), nb, cell);
        if (na != nb || memcmp(a.data(), b.data(), sizeof(C) * na)) return 3;
    }
    if (na == 1) { if (Plain<C, SH>::pop(a.data(), na) != HeapCells<C, SH>::pop(b.data(), nb)) return 4; }
    return 0;
}

int main()
{
    const unsigned masks[] = {0x1u, 0x3u, 0x7u, 0x3Fu, 0xFFu, 0xFFFu};
    for (unsigned seed = 1; seed <= 400; ++seed)
        for (unsigned m : masks)
            for (int nsym : {1, 2, 3, 7, 64, 255, 256, 286}) {
                int rc = run<uint32_t, 10>(seed, nsym, m);
                if (rc) { printf("u32 seed %u mask %x nsym %d: rc %d\n", seed, m, nsym, rc); return 1; }
                if (nsym <= 256) { rc = run<uint64_t, 16>(seed, nsym, m | (seed & 1 ? 0xFFFF0000u : 0u)); if (rc) { printf("u64 seed %u mask %x nsym %d: rc %d\n", seed, m, nsym, rc); return 1; } }
            }
    printf("ok\n");
    return 0;
}
