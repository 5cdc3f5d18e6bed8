// Variants of the LDS-DMA window copy: (a) one M0, instruction offsets 0/1024/2048/3072 -- does the offset advance the
// LDS side too?  (b) scalar base + one 32-bit per-lane offset (saddr form).
// hipcc --offload-arch=gfx950 -O3 -o lds_dma_window2 lds_dma_window2.hip && ./lds_dma_window2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ __launch_bounds__(256) void k(const float *g, int start, float2 *out) {
    __shared__ __attribute__((aligned(16))) float2 s_buf[4][576];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float2 *buf = s_buf[wave];
    for (int i = lane; i < 576; i += 64) buf[i] = make_float2(-1.f, -1.f);
    __syncthreads();
    const float *src = g + start + 1024 * wave;  // wave-uniform
    const unsigned lds_dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)buf);
    unsigned keep;
    if (MODE == 0) {
        const float *gsrc = src + 4 * lane;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n"
                     "\tglobal_load_lds_dwordx4 %1, off\n"
                     "\tglobal_load_lds_dwordx4 %1, off offset:1024\n"
                     "\tglobal_load_lds_dwordx4 %1, off offset:2048\n"
                     "\tglobal_load_lds_dwordx4 %1, off offset:3072\n"
                     "\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    } else {
        const unsigned off = 16u * lane;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n"
                     "\tglobal_load_lds_dwordx4 %1, %2\n"
                     "\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n"
                     "\tglobal_load_lds_dwordx4 %1, %2 offset:2048\n"
                     "\tglobal_load_lds_dwordx4 %1, %2 offset:3072\n"
                     "\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(off), "s"(src), "s"(lds_dst) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < 8; r++) out[512 * wave + lane + 64 * r] = buf[lane + 64 * r];
}
template <int MODE>
int run(const float *d, float2 *o) {
    int bad_total = 0;
    for (int start : {0, 1, 3, 777, 4099}) {
        hipMemset(o, 0, 4 * 512 * sizeof(float2));
        k<MODE><<<1, 256>>>(d, start, o);
        std::vector<float> r(4 * 1024);
        hipMemcpy(r.data(), o, r.size() * sizeof(float), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 4096; i++) bad += r[i] != (float)(start + i);
        printf("mode %d start %5d: %d wrong of 4096 (values at 0, 256, 512, 768: %g %g %g %g)\n", MODE, start, bad, r[0], r[256],
               r[512], r[768]);
        bad_total += bad;
    }
    return bad_total;
}
int main() {
    const int n = 1 << 16;
    std::vector<float> h(n);
    for (int i = 0; i < n; i++) h[i] = (float)i;
    float *d; float2 *o;
    hipMalloc(&d, n * sizeof(float)); hipMalloc(&o, 4 * 512 * sizeof(float2));
    hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
    const int a = run<0>(d, o), b = run<1>(d, o);
    printf("offsets advance both sides: %s; saddr form: %s\n", a ? "NO" : "yes", b ? "NO" : "yes");
    return 0;
}
