// Issue-rate probe for packed f32 VALU instructions on gfx950: wave64 instructions per cycle per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o pk_rate pk_rate.hip && ./pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float c2 __attribute__((ext_vector_type(2)));
#define ITER 4096
#define REP8(x) x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(256) void probe(float *out, float s) {
    c2 a[8];
    for (int i = 0; i < 8; i++) a[i] = c2{(float)threadIdx.x + i, (float)i};
    c2 b = c2{s, 1.0f - s};
    const unsigned long long mask = __ballot(threadIdx.x & 1);
    if (MODE == 10 || MODE == 18 || MODE == 19 || MODE == 20) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(b.x), "v"(b.y) : "vcc");
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == 0) {  // 2 scalar FMAs
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(b.x), "v"(b.y));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].y) : "v"(b.x), "v"(b.y));
            } else if (MODE == 1) {  // 1 packed FMA, plain
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(b));
            } else if (MODE == 2) {  // packed FMA with op_sel / neg modifiers
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(a[i]) : "v"(b), "v"(b));
            } else if (MODE == 3) {  // packed add
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            } else if (MODE == 4) {  // packed mul
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            } else if (MODE == 5) {  // 2 scalar adds
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].y) : "v"(b.y));
            } else if (MODE == 7) {  // VOP2 fmac
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(b.y));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i].y) : "v"(b.x), "v"(b.y));
            } else if (MODE == 8) {  // VOP3-encoded add
                asm volatile("v_add_f32_e64 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x));
                asm volatile("v_add_f32_e64 %0, %0, %1" : "+v"(a[i].y) : "v"(b.y));
            } else if (MODE == 9) {  // mul
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i].y) : "v"(b.y));
            } else if (MODE == 10) {  // cndmask
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i].x) : "v"(b.x));
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i].y) : "v"(b.y));
            } else if (MODE == 11) {  // dpp add
                asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i].x));
                asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(a[i].y));
            } else if (MODE == 12) {  // fma with a repeated source (2 distinct registers)
                asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(a[i].x) : "v"(b.x));
                asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(a[i].y) : "v"(b.y));
            } else if (MODE == 13) {  // mov
                asm volatile("v_mov_b32 %0, %1" : "+v"(a[i].x) : "v"(b.x));
                asm volatile("v_mov_b32 %0, %1" : "+v"(a[i].y) : "v"(b.y));
            } else if (MODE == 14) {  // fma, all sources distinct from the destination
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "+v"(a[i].x) : "v"(b.x), "v"(b.y), "v"(a[(i + 3) & 7].y));
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "+v"(a[i].y) : "v"(b.x), "v"(b.y), "v"(a[(i + 5) & 7].x));
            } else if (MODE == 15) {  // cndmask, mask in an SGPR pair written before the loop
                asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "s"(mask));
                asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i].y) : "v"(b.y), "s"(mask));
            } else if (MODE == 16) {  // add with an SGPR source
                asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i].x) : "s"(s));
                asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i].y) : "s"(s));
            } else if (MODE == 17) {  // alternate add / fma
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].y) : "v"(b.x), "v"(b.y));
            } else if (MODE == 18) {  // cndmask on vcc set by a v_cmp before the loop
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i].x) : "v"(b.x) : );
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].y) : "v"(b.y));
            } else if (MODE == 19) {  // VOP3-encoded cndmask on vcc
                asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[i].x) : "v"(b.x));
                asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[i].y) : "v"(b.y));
            } else if (MODE == 20) {  // cndmask vcc where the false operand is not the destination
                asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "+v"(a[i].x) : "v"(b.x), "v"(b.y));
                asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "+v"(a[i].y) : "v"(b.y), "v"(b.x));
            } else if (MODE == 21) {  // cndmask e64 sgpr mask, false operand not the destination
                asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "+v"(a[i].x) : "v"(b.x), "v"(b.y), "s"(mask));
                asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "+v"(a[i].y) : "v"(b.y), "v"(b.x), "s"(mask));
            } else if (MODE == 22) {  // packed FMA, second source a scalar-register pair (the filters' weights)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "s"(b));
            } else if (MODE == 23) {  // packed multiply by a scalar-register pair
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(b));
            } else if (MODE == 24) {  // packed FMA, weight in ONE vector register broadcast by op_sel
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "v"(b), "v"(b));
            } else if (MODE == 25) {  // fma with an SGPR source
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(b.x), "s"(s));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].y) : "v"(b.y), "s"(s));
            } else if (MODE == 26) {  // mul with an SGPR source
                asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i].x) : "s"(s));
                asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i].y) : "s"(s));
            } else if (MODE == 27) {  // fma with a literal constant (v_fmamk_f32)
                asm volatile("v_fmamk_f32 %0, %1, 0x3f3504f3, %0" : "+v"(a[i].x) : "v"(b.x));
                asm volatile("v_fmamk_f32 %0, %1, 0x3f3504f3, %0" : "+v"(a[i].y) : "v"(b.y));
            } else if (MODE == 6) {  // packed add, destination different from sources (3 distinct pairs)
                asm volatile("v_pk_add_f32 %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(b));
            }
        }
    }
    float r = 0;
    for (int i = 0; i < 8; i++) r += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE>
void run(const char *name, int per_iter_instr) {
    float *d;
    const int wgs = 256 * 4;  // 4 WGs x 4 waves per CU: 4 waves per SIMD
    hipMalloc(&d, wgs * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    probe<MODE><<<wgs, 256>>>(d, 0.25f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<MODE><<<wgs, 256>>>(d, 0.25f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 4 waves x ITER x per_iter_instr wave-instructions
    const double instr = 4.0 * ITER * per_iter_instr;
    printf("%-44s %8.3f ms  %6.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", name, ms,
           ms * 1e6 / instr, ms * 1e6 / instr * 2.4);
    hipFree(d);
}
int main() {
    run<0>("2 x v_fma_f32 (16 per iteration)", 16);
    run<1>("v_pk_fma_f32 plain (8 per iteration)", 8);
    run<2>("v_pk_fma_f32 op_sel+neg (8 per iteration)", 8);
    run<3>("v_pk_add_f32 (8)", 8);
    run<4>("v_pk_mul_f32 (8)", 8);
    run<5>("2 x v_add_f32 (16)", 16);
    run<6>("v_pk_add_f32 3 distinct pairs (8)", 8);
    run<7>("2 x v_fmac_f32 VOP2 (16)", 16);
    run<8>("2 x v_add_f32_e64 VOP3 (16)", 16);
    run<9>("2 x v_mul_f32 (16)", 16);
    run<10>("2 x v_cndmask_b32 (16)", 16);
    run<11>("2 x v_add_f32_dpp (16)", 16);
    run<12>("2 x v_fma_f32 a,a,acc (16)", 16);
    run<13>("2 x v_mov_b32 (16)", 16);
    run<14>("2 x v_fma_f32 4 distinct regs (16)", 16);
    run<15>("2 x v_cndmask_b32_e64 sgpr mask (16)", 16);
    run<16>("2 x v_add_f32 sgpr src (16)", 16);
    run<17>("v_add_f32 + v_fma_f32 (16)", 16);
    run<18>("v_cndmask vcc + v_add (16)", 16);
    run<10>("2 x v_cndmask_b32 vcc set by v_cmp (16)", 16);
    run<19>("2 x v_cndmask_b32_e64 vcc (16)", 16);
    run<20>("2 x v_cndmask_b32 vcc, dst not a source (16)", 16);
    run<21>("2 x v_cndmask_b32_e64 sgpr, dst not a source (16)", 16);
    run<22>("v_pk_fma_f32, sgpr-pair source (8)", 8);
    run<23>("v_pk_mul_f32, sgpr-pair source (8)", 8);
    run<24>("v_pk_fma_f32, vgpr weight via op_sel (8)", 8);
    run<25>("2 x v_fma_f32 sgpr src (16)", 16);
    run<26>("2 x v_mul_f32 sgpr src (16)", 16);
    run<27>("2 x v_fmamk_f32 literal (16)", 16);
    run<0>("2 x v_fma_f32 again (16)", 16);
    return 0;
}
