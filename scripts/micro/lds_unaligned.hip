// Does gfx950 serve UNALIGNED 4-byte LDS reads (ds_read_b32 at any byte address) with the right bytes?  The compiler emits
// ds_read_b32 for a 4-byte memcpy out of an align-1 LDS pointer (unaligned DS access is a target feature here); this checks
// the hardware against a byte-wise assembly for every offset class, and times both forms.
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/lds_unaligned.hip -o /tmp/lds_unaligned && /tmp/lds_unaligned
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__device__ __forceinline__ uint32_t two_dwords(const uint8_t *s, uint32_t p)
{
    const uint32_t *a = reinterpret_cast<const uint32_t *>(s + (p & ~3u));
    const uint64_t v = (uint64_t)a[0] | ((uint64_t)a[1] << 32);
    return (uint32_t)(v >> ((p & 3u) * 8u));
}
__device__ __forceinline__ uint32_t one_read(const uint8_t *s, uint32_t p) { uint32_t w; __builtin_memcpy(&w, s + p, 4); return w; }
template <int MODE>
__global__ void k(uint32_t *out, uint32_t iters, uint32_t *bad)
{
    __shared__ __attribute__((aligned(16))) uint8_t s[65536 + 16];
    for (uint32_t i = threadIdx.x; i < 65536 + 16; i += blockDim.x) s[i] = (uint8_t)((i * 2654435761u) >> 13);
    __syncthreads();
    uint32_t p = threadIdx.x * 97u + 1u, acc = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        p = (p * 1664525u + 1013904223u) & 0xFFFFu;
        const uint32_t w = MODE ? one_read(s, p) : two_dwords(s, p);
        if (MODE == 2) { const uint32_t r = (uint32_t)s[p] | ((uint32_t)s[p + 1] << 8) | ((uint32_t)s[p + 2] << 16) | ((uint32_t)s[p + 3] << 24); if (r != w) atomicAdd(bad, 1u); }
        acc += w; p ^= w & 0xFFu;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main()
{
    uint32_t *out, *bad; hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
    hipLaunchKernelGGL(k<2>, dim3(256), dim3(1024), 0, 0, out, 2000u, bad); hipDeviceSynchronize();
    uint32_t hb = 0; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    printf("mismatches of the unaligned read: %u (of %u reads)\n", hb, 256u * 1024u * 2000u);
    for (int m = 0; m < 2; ++m) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        if (m) hipLaunchKernelGGL(k<1>, dim3(256), dim3(1024), 0, 0, out, 20000u, bad); else hipLaunchKernelGGL(k<0>, dim3(256), dim3(1024), 0, 0, out, 20000u, bad);
        hipEventRecord(a);
        if (m) hipLaunchKernelGGL(k<1>, dim3(256), dim3(1024), 0, 0, out, 20000u, bad); else hipLaunchKernelGGL(k<0>, dim3(256), dim3(1024), 0, 0, out, 20000u, bad);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%s: %.3f ms\n", m ? "one unaligned ds_read_b32" : "two aligned dwords + shift", ms);
    }
    return hb ? 1 : 0;
}
