// Does ds_add_rtn_u32 hand out return values in LANE ORDER to the lanes of one wave instruction that hit the same LDS
// address?  (Undocumented; a stable radix scatter built on it must self-check.)  Patterns: all lanes one address, few
// addresses, random digits, same bank different addresses; 8 waves per workgroup hammering their own counters at once.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

__global__ __launch_bounds__(512)
void k_probe(const uint32_t *__restrict__ digits, uint32_t rounds, uint32_t ndig, uint32_t *__restrict__ bad)
{
    __shared__ uint32_t cnt[256 * 9];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (uint32_t i = tid; i < 256 * 9; i += 512) cnt[i] = 0;
    __syncthreads();
    uint32_t errs = 0;
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t d = digits[((size_t)blockIdx.x * rounds + r) * 512 + tid] % ndig;
        const uint32_t old = atomicAdd(&cnt[d * 9 + wave], 1u);
        // every lane learns the (digit, old) of all lanes of its wave: for equal digits, a lower lane must hold a lower value
        for (uint32_t o = 1; o < 64; ++o) {
            const uint32_t src = (lane + o) & 63u;
            const uint32_t od = __shfl(d, src), oo = __shfl(old, src);
            if (od == d && ((src < lane) != (oo < old))) ++errs;
        }
    }
    if (errs) atomicAdd(bad, errs);
}

int main()
{
    const uint32_t rounds = 64, blocks = 1024;
    size_t n = (size_t)blocks * rounds * 512;
    uint32_t *h = (uint32_t *)malloc(n * 4), *d, *bad;
    hipMalloc(&d, n * 4); hipMalloc(&bad, 4);
    int fail = 0;
    for (uint32_t ndig : {1u, 2u, 3u, 8u, 32u, 64u, 256u}) {
        for (int pat = 0; pat < 3; ++pat) {
            srand(ndig * 7 + pat);
            for (size_t i = 0; i < n; ++i) h[i] = pat == 0 ? rand() : pat == 1 ? (uint32_t)(i % 64) / 4 : ((rand() % 4) * 32);   // pat 2: same bank
            hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
            hipMemset(bad, 0, 4);
            hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(512), 0, 0, d, rounds, ndig, bad);
            uint32_t hb = 0;
            hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
            printf("ndig %3u pattern %d: %u order violations\n", ndig, pat, hb);
            fail |= hb != 0;
        }
    }
    printf(fail ? "NOT lane-ordered\n" : "lane-ordered in every trial\n");
    return fail;
}
