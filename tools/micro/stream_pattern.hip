// Micro-benchmark: HBM read rate for the conv staging access pattern (diagnostic, not part of the product).
//   mode 0: contiguous 16 B/lane stream
//   mode 1: per pixel (stride PS bytes) read SEG contiguous bytes (4 lanes x 16 B for SEG=64), pixels consecutive
//   mode 2: like 1 but in two passes over the buffer's pixels inside the block (first bytes [0,64), later [64,128)) -- the chunk loop
// Each workgroup handles a contiguous run of pixels; loads go to registers and are reduced to defeat DCE.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_stream(const char* __restrict__ in, float* out, long npix, int PS, int SEG, int mode, int pix_per_wg) {
    const int tid = threadIdx.x;
    const long p0 = (long)blockIdx.x * pix_per_wg;
    f32x4 acc = {0, 0, 0, 0};
    const int lanes_per_pix = SEG / 16;
    const int npass = mode == 2 ? 2 : 1;
    for (int pass = 0; pass < npass; ++pass) {
        for (int u = tid; u < pix_per_wg * lanes_per_pix; u += 256) {
            const long pix = p0 + u / lanes_per_pix;
            if (pix >= npix) break;
            const long off = mode == 0 ? (pix * lanes_per_pix + u % lanes_per_pix) * 16L
                                       : pix * (long)PS + pass * SEG + (u % lanes_per_pix) * 16;
            acc += *reinterpret_cast<const f32x4*>(in + off);
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) out[blockIdx.x] = acc[0];
}

int main() {
    const long npix = 441L * 2304 * 4;          // 4 tiles of patches
    const int PS = 384;
    char* buf; float* out;
    hipMalloc(&buf, npix * PS + 4096); hipMemset(buf, 0, npix * PS + 4096);
    hipMalloc(&out, 1 << 22);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    struct Cfg { int mode, seg, ppw; const char* name; };
    Cfg cfgs[] = {{0, 64, 384, "contiguous, 64 B/pixel-equivalent"}, {1, 64, 384, "64 B of every 384 B pixel"}, {1, 128, 384, "128 B of every 384 B pixel"},
                  {2, 64, 384, "2 passes x 64 B (halves of a 128 B line, separated)"}, {1, 192, 384, "192 B of every 384 B pixel"}, {1, 384, 384, "whole 384 B pixel"},
                  {1, 64, 1536, "64 B/384 B, 1536 pixels per WG"}};
    for (auto& c : cfgs) {
        const int nwg = (int)((npix + c.ppw - 1) / c.ppw);
        for (int it = 0; it < 3; ++it) {
            hipEventRecord(a);
            hipLaunchKernelGGL(k_stream, dim3(nwg), dim3(256), 0, 0, buf, out, npix, PS, c.seg, c.mode, c.ppw);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (it == 2) {
                const double bytes = (double)npix * c.seg * (c.mode == 2 ? 2 : 1);
                printf("%-60s %8.3f ms  %7.2f TB/s useful\n", c.name, ms, bytes / ms / 1e9);
            }
        }
    }
    return 0;
}
