// VALU issue rate on gfx950: cycles per wave64 v_fma_f32 / v_pk_fma_f32 per SIMD at 1, 2, 4 waves per SIMD, with 8 or 16
// independent accumulators, and for the complex multiply-accumulate pattern of the reverb (two chains of two).
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate profiles/micro/valu_rate.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float v2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(1024) void rate_kernel(float *out, int iters, float a, float b) {
    float acc[16];
    v2 pacc[8];
    for (int i = 0; i < 16; i++) acc[i] = threadIdx.x + i;
    for (int i = 0; i < 8; i++) pacc[i] = v2{(float)threadIdx.x, (float)i};
    float x = a, y = b;
    const v2 px = v2{a, b}, py = v2{b, a};
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {  // 16 independent chains, 64 fma
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y));
        } else if (MODE == 1) {  // 4 independent chains, 64 fma
#pragma unroll
            for (int r = 0; r < 16; r++)
#pragma unroll
                for (int i = 0; i < 4; i++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y));
        } else if (MODE == 2) {  // packed, 8 independent chains, 32 pk_fma = 64 fma
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pacc[i]) : "v"(px), "v"(py));
        } else if (MODE == 3) {  // the cmac order: x-chain, y-chain, x-chain, y-chain per accumulator pair, 64 fma
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[2 * i]) : "v"(x), "v"(y));
                    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[2 * i + 1]) : "v"(x), "v"(y));
                    asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(acc[2 * i]) : "v"(y), "v"(y));
                    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[2 * i + 1]) : "v"(y), "v"(x));
                }
        } else if (MODE == 4) {  // v_add_f32, 16 chains
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int i = 0; i < 16; i++) asm volatile("v_add_f32 %0, %1, %0" : "+v"(acc[i]) : "v"(x));
        } else if (MODE == 5) {  // v_mul_f32 into fresh registers from two sources (no accumulate)
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int i = 0; i < 16; i++) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(acc[i]) : "v"(x), "v"(y));
        }
    }
    float s = 0;
    for (int i = 0; i < 16; i++) s += acc[i];
    for (int i = 0; i < 8; i++) s += pacc[i].x + pacc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void run(const char *name, int waves_per_simd) {
    const int threads = 64 * 4 * waves_per_simd;  // one workgroup per CU: 4 SIMDs x waves
    const int iters = 20000;
    float *d;
    (void)hipMalloc(&d, (2 << 20) * sizeof(float));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    rate_kernel<MODE><<<256, threads>>>(d, 100, 1.0f, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    rate_kernel<MODE><<<256, threads>>>(d, iters, 1.0f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // 64 lane-FMAs-equivalents per iteration per wave
    const double inst_per_simd = 64.0 * iters * waves_per_simd;  // fma-equivalents
    const double tflops = 2.0 * 64 * 64.0 * iters * waves_per_simd * 4 * 256 / (ms * 1e-3) / 1e12;
    printf("%-34s waves/SIMD %d: %.3f ms, %.2f ns per fma-equivalent per SIMD, %.1f TFLOP/s\n", name, waves_per_simd, ms,
           ms * 1e6 / inst_per_simd, tflops);
    (void)hipFree(d);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32 x16 chains", w);
        run<1>("v_fma_f32 x4 chains", w);
        run<2>("v_pk_fma_f32 x8 chains", w);
        run<3>("cmac pattern (fmac/fma)", w);
        run<4>("v_add_f32 x16 chains", w);
        run<5>("v_mul_f32 no accumulate", w);
    }
    return 0;
}
