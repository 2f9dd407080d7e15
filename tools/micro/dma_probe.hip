// Micro-probe (diagnostic): semantics of __builtin_amdgcn_global_load_lds on gfx950 -- per-lane global source,
// LDS destination = wave-uniform base + lane*16; vmcnt accounting; visibility after s_waitcnt vmcnt(0) + barrier.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k(const u32x4* __restrict__ src, u32x4* dst, const int* perm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6;
    // each thread moves 2 units: unit u = tid + 256*i goes to LDS slot u (lane-linear per wave), source index = perm[u]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int u = tid + 256 * i;
        const u32x4* g = src + perm[u];
        const unsigned lbase = __builtin_amdgcn_readfirstlane((unsigned)((wave * 64 + 256 * i) * 16));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(smem + lbase), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // read back with a lane rotation so every thread reads data another wave's DMA wrote
    for (int i = 0; i < 2; ++i) {
        const int u = (tid + 256 * i + 77) & 511;
        dst[u] = *reinterpret_cast<const u32x4*>(smem + u * 16);
    }
}

int main() {
    const int n = 512;
    std::vector<unsigned> h(n * 4), out(n * 4);
    std::vector<int> perm(n);
    for (int i = 0; i < n; ++i) { perm[i] = (i * 37 + 11) % n; for (int j = 0; j < 4; ++j) h[i * 4 + j] = i * 16 + j; }
    u32x4 *s, *d; int* p;
    hipMalloc(&s, n * 16); hipMalloc(&d, n * 16); hipMalloc(&p, n * 4);
    hipMemcpy(s, h.data(), n * 16, hipMemcpyHostToDevice); hipMemcpy(p, perm.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(d, 0xff, n * 16);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), n * 16, 0, s, d, p);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(out.data(), d, n * 16, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int u = 0; u < n; ++u) for (int j = 0; j < 4; ++j) if (out[u * 4 + j] != (unsigned)(perm[u] * 16 + j)) ++bad;
    printf("status=%s mismatches=%d of %d (first: got %u want %u)\n", hipGetErrorString(e), bad, n * 4, out[0], perm[0] * 16);
    return bad != 0;
}
