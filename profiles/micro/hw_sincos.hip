// Accuracy of v_sin_f32 / v_cos_f32 (input in turns) on gfx950 against double, on |x| <= 1/8 turn -- the reduced
// argument of the distance factor.  hipcc --offload-arch=gfx950 -O2 -o /tmp/hw_sincos profiles/micro/hw_sincos.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>
__global__ void k(const float *x, float *s, float *c, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    s[i] = __builtin_amdgcn_sinf(x[i]);
    c[i] = __builtin_amdgcn_cosf(x[i]);
}
int main() {
    const int n = 1 << 22;
    std::vector<float> x(n), s(n), c(n);
    for (int i = 0; i < n; i++) x[i] = (float)(((double)i / n - 0.5) * 0.25);  // [-1/8, 1/8) turns
    float *dx, *ds, *dc;
    hipMalloc(&dx, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dc, n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, ds, dc, n);
    hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
    double es = 0, ec = 0;
    for (int i = 0; i < n; i++) {
        const double a = 2.0 * M_PI * (double)x[i];
        es = fmax(es, fabs((double)s[i] - sin(a)));
        ec = fmax(ec, fabs((double)c[i] - cos(a)));
    }
    printf("max abs err: v_sin_f32 %.3e (%.2f ulp of 1), v_cos_f32 %.3e (%.2f ulp of 1)\n", es, es / 1.19e-7, ec, ec / 1.19e-7);
    return 0;
}
