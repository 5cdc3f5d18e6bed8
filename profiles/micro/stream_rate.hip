// What a read stream achieves on this MI355X, measured the way the reverb's product kernel reads its delay line: a buffer larger
// than (or as large as) the 256 MiB Infinity Cache read once per launch, 8 or 16 bytes per lane per load, ordinary or
// non-temporal loads, one-shot or grid-stride workgroups, 4 or 8 loads in flight per wave.  Prints TB/s per form.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_rate profiles/micro/stream_rate.hip && /tmp/stream_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float v2 __attribute__((ext_vector_type(2)));
typedef float v4 __attribute__((ext_vector_type(4)));

// every wave reads DEPTH vectors per step, steps of (threads of the grid) vectors apart
template <typename V, bool NT, int DEPTH>
__global__ __launch_bounds__(256) void read_kernel(const V *__restrict__ in, size_t n_vec, float *out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    for (; i + (DEPTH - 1) * stride < n_vec; i += DEPTH * stride) {
        V v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; d++) v[d] = NT ? __builtin_nontemporal_load(in + i + d * stride) : in[i + d * stride];
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            acc += v[d].x;
            acc += v[d].y;
        }
    }
    for (; i < n_vec; i += stride) {  // what is left of a buffer that is not a whole number of steps
        const V v = in[i];
        acc += v.x + v.y;
    }
    if (acc == 12345.678f) out[0] = acc;  // (keeps the loads)
}

template <typename V, bool NT, int DEPTH>
static void run(const char *name, const void *buf, size_t bytes, int wgs, float *out) {
    const size_t n_vec = bytes / sizeof(V);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL((read_kernel<V, NT, DEPTH>), dim3(wgs), dim3(256), 0, 0, (const V *)buf, n_vec, out);
    const int reps = 20;
    (void)hipEventRecord(a, 0);
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((read_kernel<V, NT, DEPTH>), dim3(wgs), dim3(256), 0, 0, (const V *)buf, n_vec, out);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    printf("%-52s %6d workgroups  %7.1f us per launch  %5.2f TB/s\n", name, wgs, 1e3 * ms / reps, bytes * (double)reps / (ms * 1e-3) / 1e12);
}

int main() {
    float *out;
    (void)hipMalloc(&out, 64);
    for (size_t mb : {247, 512, 2048}) {
        const size_t bytes = mb << 20;
        void *buf;
        if (hipMalloc(&buf, bytes) != hipSuccess) return 1;
        (void)hipMemset(buf, 0, bytes);
        (void)hipDeviceSynchronize();
        printf("== %zu MiB read once per launch\n", mb);
        for (int wgs : {2048, 4096, 16384}) {
            run<v2, false, 4>("8 B per lane, 4 in flight", buf, bytes, wgs, out);
            run<v2, true, 4>("8 B per lane, 4 in flight, non-temporal", buf, bytes, wgs, out);
            run<v2, true, 8>("8 B per lane, 8 in flight, non-temporal", buf, bytes, wgs, out);
            run<v4, false, 4>("16 B per lane, 4 in flight", buf, bytes, wgs, out);
            run<v4, true, 4>("16 B per lane, 4 in flight, non-temporal", buf, bytes, wgs, out);
            run<v4, true, 8>("16 B per lane, 8 in flight, non-temporal", buf, bytes, wgs, out);
        }
        (void)hipFree(buf);
    }
    return 0;
}
