// Does a window of 1024 floats at a 4-byte-aligned (not 16-byte-aligned) global address arrive in LDS in order through four
// global_load_lds_dwordx4 of one wave (lane i of chunk k: floats 256 k + 4 i .. + 3 -> LDS base + 1024 k + 16 i)?
// hipcc --offload-arch=gfx950 -O3 -o lds_dma_window lds_dma_window.hip && ./lds_dma_window
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k(const float *g, int start, float2 *out) {
    __shared__ __attribute__((aligned(16))) float2 s_buf[4][576];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float2 *buf = s_buf[wave];
    const float *src = g + start + 1024 * wave;  // each wave its own window
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const float *gsrc = src + 256 * c + 4 * lane;
        const unsigned lds_dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)buf + 1024u * c);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < 8; r++) out[512 * wave + lane + 64 * r] = buf[lane + 64 * r];
}
int main() {
    const int n = 1 << 16;
    std::vector<float> h(n);
    for (int i = 0; i < n; i++) h[i] = (float)i;
    float *d; float2 *o;
    hipMalloc(&d, n * sizeof(float)); hipMalloc(&o, 4 * 512 * sizeof(float2));
    hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
    int bad_total = 0;
    for (int start : {0, 1, 2, 3, 5, 777, 4099}) {
        hipMemset(o, 0, 4 * 512 * sizeof(float2));
        k<<<1, 256>>>(d, start, o);
        std::vector<float> r(4 * 1024);
        hipMemcpy(r.data(), o, r.size() * sizeof(float), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 4096; i++) bad += r[i] != (float)(start + i);
        printf("start %5d: %d wrong of 4096 (first values %g %g %g %g)\n", start, bad, r[0], r[1], r[2], r[1024]);
        bad_total += bad;
    }
    printf(bad_total ? "FAILED\n" : "OK\n");
    return bad_total != 0;
}
