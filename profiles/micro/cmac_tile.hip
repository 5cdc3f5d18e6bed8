// The reverb's tiled multiply-accumulate loop without its loads: 16 complex accumulators, a sliding window of 16 complex
// inputs, one complex coefficient per step; register operands only, all 256 CUs, 4 waves per SIMD.  Which order of the
// four FMAs of a complex multiply-accumulate does the vector unit like best?
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o /tmp/cmac_tile profiles/micro/cmac_tile.hip && /tmp/cmac_tile
#include <hip/hip_runtime.h>

#include <cstdio>

template <int ORDER>
__global__ __launch_bounds__(1024) void tile_kernel(float *out, int groups, float seed) {
    float ax[16], ay[16], xr[16], xi[16];
    for (int i = 0; i < 16; i++) {
        ax[i] = ay[i] = 0.0f;
        xr[i] = seed + threadIdx.x + i;
        xi[i] = seed - i;
    }
    float hr = seed * 0.5f, hi = seed * 0.25f;
    for (int g = 0; g < groups; g++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            // "loads": new window element and coefficient from cheap arithmetic on loop state
            xr[(16 - j) % 16] = hr + (float)j;
            xi[(16 - j) % 16] = hi - (float)j;
            hr = hr * 1.0001f;
            hi = hi * 0.9999f;
            if (ORDER == 0) {  // per accumulator: x, y, x, y (what jf_reverb.hip does)
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const float a = xr[(i + 16 - j) % 16], b = xi[(i + 16 - j) % 16];
                    ax[i] = __builtin_fmaf(a, hr, ax[i]);
                    ay[i] = __builtin_fmaf(a, hi, ay[i]);
                    ax[i] = __builtin_fmaf(-b, hi, ax[i]);
                    ay[i] = __builtin_fmaf(b, hr, ay[i]);
                }
            } else if (ORDER == 1) {  // all first halves, then all second halves
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const float a = xr[(i + 16 - j) % 16];
                    ax[i] = __builtin_fmaf(a, hr, ax[i]);
                    ay[i] = __builtin_fmaf(a, hi, ay[i]);
                }
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const float b = xi[(i + 16 - j) % 16];
                    ax[i] = __builtin_fmaf(-b, hi, ax[i]);
                    ay[i] = __builtin_fmaf(b, hr, ay[i]);
                }
            } else {  // four sweeps over the accumulators
#pragma unroll
                for (int i = 0; i < 16; i++) ax[i] = __builtin_fmaf(xr[(i + 16 - j) % 16], hr, ax[i]);
#pragma unroll
                for (int i = 0; i < 16; i++) ay[i] = __builtin_fmaf(xr[(i + 16 - j) % 16], hi, ay[i]);
#pragma unroll
                for (int i = 0; i < 16; i++) ax[i] = __builtin_fmaf(-xi[(i + 16 - j) % 16], hi, ax[i]);
#pragma unroll
                for (int i = 0; i < 16; i++) ay[i] = __builtin_fmaf(xi[(i + 16 - j) % 16], hr, ay[i]);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; i++) s += ax[i] + ay[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ORDER>
static void run(const char *name) {
    float *d;
    (void)hipMalloc(&d, 256 * 1024 * sizeof(float));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int groups = 4000;
    tile_kernel<ORDER><<<256, 1024>>>(d, 50, 1.0f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    tile_kernel<ORDER><<<256, 1024>>>(d, groups, 1.0f);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double fma = 64.0 * 16 * groups * 16.0 * 256;  // per wave: 64 FMAs x 16 steps per group; 16 waves x 256 CUs
    printf("%-44s %.3f ms, %.1f TFLOP/s, %.2f ns per FMA instruction per SIMD\n", name, ms, 2.0 * 64 * fma / (ms * 1e-3) / 1e12,
           ms * 1e6 / (64.0 * 16 * groups * 4));
    (void)hipFree(d);
}

int main() {
    run<0>("x, y, x, y per accumulator");
    run<1>("first halves of all, then second halves");
    run<2>("four sweeps over the accumulators");
    return 0;
}
