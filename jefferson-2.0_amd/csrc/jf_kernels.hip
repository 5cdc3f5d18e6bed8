// jf_kernels.hip -- hand-written CDNA4 (gfx950) kernels of the HRTF binaural
// convolution hot path.  No rocFFT/hipFFT, no Thrust, no MFMA (the path is FFT +
// pointwise).
//
// One 64-lane wavefront owns one work item = (block b, source s) and carries it
// through the whole per-block pipeline of the reference
// (GPUSoundSource.cu:320-385 interpolateConvolve + :463-513 window handling):
//
//   window gather (signal ring / previous window)            a5
//   real FFT 1024 = complex FFT 512 (radix 8x8x8 Stockham, two LDS exchanges)
//     + split post-pass; the Hermitian partner Z[512-k] comes from lane 64 - lane through LDS (mirror8)  a6
//   distance factor D[k]: exact 64-bit fixed-point phase k*c mod 1, the nearest 1/1024 turn from the FFT's
//     twiddle table corrected by the small-angle terms of the remainder (no f64)   a7
//   sum_i w_i H_i[k] from the interleaved table, * X[k] D[k], both ears  a8
//   inverse: Z = Y_L + j Y_R, built in registers (upper half mirrored through LDS, mirror8),
//     1024-point inverse as 4 decimated 256-point transforms
//     (radix 16x16, ONE LDS exchange) + a pruned last radix-4 done as a
//     DPP quad reduce-scatter -- only the last B samples are ever formed  a9
//   crossfade old/new filter sets in registers                          a10
//   store the B stereo frames of this source                            a11
//
// fused_block_kernel: one stereo block per source (the reference's `intermediate`); fused_pair_kernel: BOTH
// filter sets of G consecutive sources summed as spectra (per ear: sum X D H_left, sum X D H_right, turned into Z once
// per unit) and inverted once; rt_block_kernel: one launch per audio block for the per-block calls.
//
// mix_kernel / mix_few_kernel sum the per-source (per-group) blocks in source order (a12), prep_kernel computes
// indices/weights (a2, a3) for every item.  A run prepares the NEXT window's descriptors itself: the pair kernel in
// trailing workgroups of its own launch (they run in the kernel's tail), the per-source kernel inside its mix launch
// (mix_prep_kernel).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "jf_device.h"
#include "jf_experiments.h"
#include "jf_packed.h"

namespace jf {

// --------------------------------------------------------------- helpers --
#define JF_DEV __device__ __forceinline__
// The value again, opaque to the optimiser: expressions of the result cannot be hoisted out of the enclosing loop.
// Used on the lane index where the compiler would otherwise keep dozens of loop-invariant per-lane addresses and
// constants alive across a whole persistent kernel -- and spill them.
JF_DEV int opaque(int x) {
    JF_EXP_OPAQUE(x);
    return x;
}
typedef float __attribute__((address_space(1))) gfloat;  // float in global memory
struct __attribute__((packed, aligned(4))) FloatPair {   // two consecutive samples, 4-byte aligned
    float x, y;
};
typedef FloatPair __attribute__((address_space(1))) gpair;
// Read-only tables of a launch (descriptors, signal table, play positions, processing order) are read through the
// constant address space: a wave-uniform load from it is a scalar load into scalar registers.  Through a generic
// pointer the compiler must assume that the kernel's own stores may alias, and issues a vector load instead -- a
// vector register per dword and a full memory latency in front of whatever needs the value.
#define JF_CONST_AS __attribute__((address_space(4)))
template <class T>
JF_DEV const T JF_CONST_AS *as_const(const T *p) {
    return (const T JF_CONST_AS *)p;
}
// an item's descriptor, copied out of global memory by scalar loads
JF_DEV ItemDesc load_desc(const ItemDesc *p) {
    const ItemDesc JF_CONST_AS *c = as_const(p);
    ItemDesc d;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        d.rows_new[t] = c->rows_new[t];
        d.w_new[t] = c->w_new[t];
        d.rows_old[t] = c->rows_old[t];
        d.w_old[t] = c->w_old[t];
    }
    d.c_fix = c->c_fix;
    d.inv_frac = c->inv_frac;
    d.n_new = c->n_new;
    d.n_old = c->n_old;
    d.flags = c->flags;
    return d;
}

// LDS traffic that is PRIVATE TO ONE WAVEFRONT (the FFT exchanges, the mirrors): LDS ops of a wave execute in
// issue order, so all that is needed is to stop the compiler from moving a
// lane's reads above other lanes' writes.
#define JF_WAVE_LDS_SYNC()                                      \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  \
        __builtin_amdgcn_wave_barrier();                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  \
    } while (0)
// LDS traffic BETWEEN THE TWO WAVEFRONTS OF A PAIR (mailboxes and their flag words): workgroup-scope release before a
// flag is written, workgroup-scope acquire after one has been read, both restricted to the local address space -- on
// gfx950 an s_waitcnt lgkmcnt(0), and no wait for the table-row loads that may be in flight (a fence over all address
// spaces would add vmcnt(0)).
#define JF_PAIR_RELEASE()                                                 \
    do {                                                                  \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");   \
        __builtin_amdgcn_wave_barrier();                                  \
    } while (0)
#define JF_PAIR_ACQUIRE()                                                 \
    do {                                                                  \
        __builtin_amdgcn_wave_barrier();                                  \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");   \
    } while (0)

JF_DEV float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
JF_DEV float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
JF_DEV float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
JF_DEV float2 cmulc(float2 a, float2 b) {  // a * conj(b)
    return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}

// twiddle from the e^{+i} table: forward transforms use the conjugate
template <int DIR>
JF_DEV float2 ctw(float2 v, float2 w) {
    return DIR > 0 ? cmul(v, w) : cmulc(v, w);
}
// The same products as a packed multiply + a packed FMA (jf_packed.h) where both operands already sit in aligned register
// pairs (LDS reads, freshly computed values): 9.9 issue cycles instead of 14.0 for two multiplies (2.7 each) and two FMAs
// (4.3 each: three vector-register sources) -- profiles/micro/pk_rate.hip, 4 waves per SIMD.
#ifndef JF_PACKED_CMUL
#define JF_PACKED_CMUL 1
#endif
#ifndef JF_PACKED_DTAIL
#define JF_PACKED_DTAIL 0
#endif
JF_DEV float2 cmul_pk(float2 a, float2 b) {
#if JF_PACKED_CMUL
    return f2_of(pcmul(c2_of(a), c2_of(b)));
#else
    return cmul(a, b);
#endif
}
JF_DEV float2 cmulc_pk(float2 a, float2 b) {  // a * conj(b)
#if JF_PACKED_CMUL
    return f2_of(pcmulc(c2_of(a), c2_of(b)));
#else
    return cmulc(a, b);
#endif
}

// v * (c + i*DIR*s)
template <int DIR>
JF_DEV float2 cmulk(float2 v, float c, float s) {
    const float d = DIR > 0 ? s : -s;
    return make_float2(v.x * c - v.y * d, v.x * d + v.y * c);
}

// v * (h + i*DIR*h) and v * (-h + i*DIR*h), h = sqrt(1/2): an add, a subtract and two multiplies
// (2.7 issue cycles each) instead of two multiplies and two FMAs (4.3)
template <int DIR>
JF_DEV float2 cmul_h(float2 v) {
    constexpr float h = 0.70710678118654752440f;
    return DIR > 0 ? make_float2(h * (v.x - v.y), h * (v.x + v.y)) : make_float2(h * (v.x + v.y), h * (v.y - v.x));
}
template <int DIR>
JF_DEV float2 cmul_nh(float2 v) {
    constexpr float h = 0.70710678118654752440f;
    return DIR > 0 ? make_float2(-h * (v.x + v.y), h * (v.x - v.y)) : make_float2(h * (v.y - v.x), -h * (v.x + v.y));
}

// v * (DIR * i)
template <int DIR>
JF_DEV float2 cmuli(float2 v) {
    return DIR > 0 ? make_float2(-v.y, v.x) : make_float2(v.y, -v.x);
}

template <int DIR>
JF_DEV void fft4(float2 &a0, float2 &a1, float2 &a2, float2 &a3) {
    const float2 t0 = cadd(a0, a2), t1 = csub(a0, a2);
    const float2 t2 = cadd(a1, a3), t3 = cmuli<DIR>(csub(a1, a3));
    a0 = cadd(t0, t2);
    a1 = cadd(t1, t3);
    a2 = csub(t0, t2);
    a3 = csub(t1, t3);
}

// in-register 8-point DFT, natural order in and out
template <int DIR>
JF_DEV void fft8(float2 (&v)[8]) {
    float2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    float2 o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    fft4<DIR>(e0, e1, e2, e3);
    fft4<DIR>(o0, o1, o2, o3);
    o1 = cmul_h<DIR>(o1);
    o2 = cmuli<DIR>(o2);
    o3 = cmul_nh<DIR>(o3);
    v[0] = cadd(e0, o0);
    v[4] = csub(e0, o0);
    v[1] = cadd(e1, o1);
    v[5] = csub(e1, o1);
    v[2] = cadd(e2, o2);
    v[6] = csub(e2, o2);
    v[3] = cadd(e3, o3);
    v[7] = csub(e3, o3);
}

JF_DEV void cswap(float2 &a, float2 &b) {
    const float2 t = a;
    a = b;
    b = t;
}

// in-register 16-point DFT, natural order in and out
template <int DIR>
JF_DEV void fft16(float2 (&v)[16]) {
    constexpr float c1 = 0.92387953251128675613f;  // cos(pi/8)
    constexpr float s1 = 0.38268343236508977173f;  // sin(pi/8)
#pragma unroll
    for (int n1 = 0; n1 < 4; n1++) fft4<DIR>(v[n1], v[n1 + 4], v[n1 + 8], v[n1 + 12]);
    // v[n1 + 4*k2] *= W16^(n1*k2)
    v[5] = cmulk<DIR>(v[5], c1, s1);     // 1
    v[9] = cmul_h<DIR>(v[9]);            // 2
    v[13] = cmulk<DIR>(v[13], s1, c1);   // 3
    v[6] = cmul_h<DIR>(v[6]);            // 2
    v[10] = cmuli<DIR>(v[10]);           // 4
    v[14] = cmul_nh<DIR>(v[14]);         // 6
    v[7] = cmulk<DIR>(v[7], s1, c1);     // 3
    v[11] = cmul_nh<DIR>(v[11]);         // 6
    v[15] = cmulk<DIR>(v[15], -c1, -s1); // 9
#pragma unroll
    for (int k2 = 0; k2 < 4; k2++) fft4<DIR>(v[4 * k2], v[4 * k2 + 1], v[4 * k2 + 2], v[4 * k2 + 3]);
    // v[k1 + 4*k2] holds X[k2 + 4*k1]: transpose the 4x4
    cswap(v[1], v[4]);
    cswap(v[2], v[8]);
    cswap(v[3], v[12]);
    cswap(v[6], v[9]);
    cswap(v[7], v[13]);
    cswap(v[11], v[14]);
}

// keep + (send of the lane quad_perm points at), as DPP adds, two complex values (or one) per asm block.
// Written as asm: the compiler turns a third of such adds into v_mov_b32_dpp + v_add_f32.  A DPP source
// written by one of the two preceding VALU instructions is a hazard the compiler cannot see inside asm:
// the two wait states are spelled out at the entry of each block.
// XOR1: partner lane = lane ^ 1 (quad_perm [1,0,3,2]); else lane ^ 2 (quad_perm [2,3,0,1]).
#define JF_QP1 "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define JF_QP2 "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1"
template <bool XOR1>
JF_DEV void quad_exchange_add(float2 &k0, float2 s0, float2 &k1, float2 s1) {
    if (XOR1)
        asm("s_nop 1\n\t"
            "v_add_f32_dpp %0, %4, %0 " JF_QP1 "\n\tv_add_f32_dpp %1, %5, %1 " JF_QP1 "\n\t"
            "v_add_f32_dpp %2, %6, %2 " JF_QP1 "\n\tv_add_f32_dpp %3, %7, %3 " JF_QP1
            : "+v"(k0.x), "+v"(k0.y), "+v"(k1.x), "+v"(k1.y)
            : "v"(s0.x), "v"(s0.y), "v"(s1.x), "v"(s1.y));
    else
        asm("s_nop 1\n\t"
            "v_add_f32_dpp %0, %4, %0 " JF_QP2 "\n\tv_add_f32_dpp %1, %5, %1 " JF_QP2 "\n\t"
            "v_add_f32_dpp %2, %6, %2 " JF_QP2 "\n\tv_add_f32_dpp %3, %7, %3 " JF_QP2
            : "+v"(k0.x), "+v"(k0.y), "+v"(k1.x), "+v"(k1.y)
            : "v"(s0.x), "v"(s0.y), "v"(s1.x), "v"(s1.y));
}
template <bool XOR1>
JF_DEV void quad_exchange_add(float2 &k0, float2 s0) {
    if (XOR1)
        asm("s_nop 1\n\tv_add_f32_dpp %0, %2, %0 " JF_QP1 "\n\tv_add_f32_dpp %1, %3, %1 " JF_QP1
            : "+v"(k0.x), "+v"(k0.y)
            : "v"(s0.x), "v"(s0.y));
    else
        asm("s_nop 1\n\tv_add_f32_dpp %0, %2, %0 " JF_QP2 "\n\tv_add_f32_dpp %1, %3, %1 " JF_QP2
            : "+v"(k0.x), "+v"(k0.y)
            : "v"(s0.x), "v"(s0.y));
}
#undef JF_QP1
#undef JF_QP2

// Hermitian mirror across the wave: out[j] = in[7 - j] of lane 64 - lane; lane 0, whose partner would be
// "lane 64", gets its own in[(8 - j) & 7] (bins 0 / 512 and the multiples of 64 live on lane 0).  Through
// LDS rather than ds_bpermute: lane l writes slot l + 64 q, lane 0 also writes in[0] to slot 512, and
// every lane reads slots (64 - l) + 64 (7 - j) -- for lane 0 that is 64 (8 - j): exactly its own values,
// so the lane-0 exception costs one masked store instead of two selects per value.
JF_DEV void mirror8(float2 *buf, const float2 (&in)[8], float2 (&out)[8], int lane) {
#pragma unroll
    for (int q = 0; q < 8; q++) buf[lane + 64 * q] = in[q];
    if (lane == 0) buf[512] = in[0];
    JF_WAVE_LDS_SYNC();
    const float2 *rd = buf + (64 - lane);
#pragma unroll
    for (int j = 0; j < 8; j++) out[j] = rd[64 * (7 - j)];
    JF_WAVE_LDS_SYNC();
}

// ------------------------------------------------------------ forward FFT --
// Real FFT of 1024 samples through a 512-point complex FFT.
// In : z[r] = (x[2(lane+64r)], x[2(lane+64r)+1]), r = 0..7
// Out: X[q] = TWICE bin (lane + 64 q), q = 0..7, unnormalised (the halving of the split pass is left to
//      the caller's scale factor: a power of two, so nothing rounds differently); on lane 0, X[0] is
//      packed as 2 (X[0].re, X[512].re) (both bins are real).
// buf: >= 576 float2 of this wave's LDS; tw: the twiddle pack (jf_device.h) in LDS.
JF_DEV void rfft1024_wave(float2 (&z)[8], float2 (&X)[8], float2 *buf, const float2 *tw, int lane) {
    // pass A (sub-length 1): no twiddles; store 8 contiguous, row padded 8 -> 9
    fft8<-1>(z);
#pragma unroll
    for (int r = 0; r < 8; r++) buf[9 * lane + r] = z[r];
    JF_WAVE_LDS_SYNC();
    // pass B (sub-length 8)
    float2 u[8];
    {
        const int base = lane + (lane >> 3);
#pragma unroll
        for (int r = 0; r < 8; r++) u[r] = buf[base + 72 * r];
        const int k = lane & 7;
#pragma unroll
        for (int r = 1; r < 8; r++) u[r] = cmulc_pk(u[r], tw[kTwWB + 8 * r + k]);
        fft8<-1>(u);
        JF_WAVE_LDS_SYNC();
        const int wbase = 72 * (lane >> 3) + k;
#pragma unroll
        for (int r = 0; r < 8; r++) buf[wbase + 8 * r] = u[r];
    }
    JF_WAVE_LDS_SYNC();
    // pass C (sub-length 64): Z[lane + 64 r']
#pragma unroll
    for (int r = 0; r < 8; r++) u[r] = buf[lane + 72 * r];
#pragma unroll
    for (int r = 1; r < 8; r++) u[r] = cmulc_pk(u[r], tw[kTwWC + 64 * r + lane]);
    fft8<-1>(u);
    JF_WAVE_LDS_SYNC();
    // split: 2 X[k] = E + (-i) W^k O, E = Z[k] + conj Z[512-k], O = Z[k] - conj Z[512-k]
    float2 um[8];  // lanes >= 1: Z[512 - k]; lane 0: Z[(512 - 64 q) mod 512]
    mirror8(buf, u, um, lane);
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const float2 zm = um[q];
        const float2 zk = u[q];
        const float2 e = make_float2(zk.x + zm.x, zk.y - zm.y);
        const float2 o = make_float2(zk.x - zm.x, zk.y + zm.y);
        // (-i) * conj(W^k) * o
        const float2 wo = cmulc_pk(o, tw[kTwU + 64 * q + lane]);
        X[q] = make_float2(e.x + wo.y, e.y - wo.x);
    }
    // lane 0: bins 0 and 512 are real: Re(Z0) +/- Im(Z0)
    X[0] = lane == 0 ? make_float2(2.0f * (u[0].x + u[0].y), 2.0f * (u[0].x - u[0].y)) : X[0];
}

// ------------------------------------------------------------ inverse FFT --
// Last quarter of the unnormalised inverse 1024-point complex transform.
// In : Zin[r] = Z[lane + 64 r], r = 0..15.
// Out: out[j] = y[1024 - B + (lane>>2) + 16 (NOUT a + j)], a = lane & 3, j < NOUT = B / 64: this lane's
//      frames of the block.  buf: this wave's LDS, >= 1088 float2, or >= 544 with SPLIT (the exchange then
//      moves real and imaginary parts one after the other through a float image of half the size).
template <int NOUT, bool SPLIT>
JF_DEV void ifft1024_lastq_wave(float2 (&v)[16], float2 (&out)[NOUT], float2 *buf, const float2 *tw, int lane) {
    const int a = lane & 3, i = lane >> 2;
    // lane (i, a) holds S_a[i + 16 r] = Z[4 (i + 16 r) + a]: 16-point inverse over r
    fft16<+1>(v);
#pragma unroll
    for (int m = 1; m < 16; m++) v[m] = cmul(v[m], tw[kTwW2 + 16 * m + i]);
    // exchange inside each group a: write rows m (padded 64 -> 68), read columns
    if (SPLIT) {
        float *fb = reinterpret_cast<float *>(buf);
#pragma unroll
        for (int m = 0; m < 16; m++) fb[68 * m + lane] = v[m].x;
        JF_WAVE_LDS_SYNC();
#pragma unroll
        for (int j = 0; j < 16; j++) v[j].x = fb[68 * i + 4 * j + a];
        JF_WAVE_LDS_SYNC();
#pragma unroll
        for (int m = 0; m < 16; m++) fb[68 * m + lane] = v[m].y;
        JF_WAVE_LDS_SYNC();
#pragma unroll
        for (int j = 0; j < 16; j++) v[j].y = fb[68 * i + 4 * j + a];
        JF_WAVE_LDS_SYNC();
    } else {
#pragma unroll
        for (int m = 0; m < 16; m++) buf[68 * m + lane] = v[m];
        JF_WAVE_LDS_SYNC();
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = buf[68 * i + 4 * j + a];
        JF_WAVE_LDS_SYNC();
    }
    fft16<+1>(v);  // s_a[i + 16 t]
    // y[768 + n] = sum_a (-i)^a e^{+2 pi i a n / 1024} s_a[n],  n = i + 16 t: the pruned last radix-4, a sum
    // over the 4 lanes of a quad.  Only t >= T0 is in the block, and lane a keeps t = T0 + a NOUT + j only, so
    // the sum is a reduce-scatter: exchange with lane ^ 2 for the half of the t's this lane's pair keeps,
    // then with lane ^ 1 for its own quarter (3 NOUT DPP adds per component instead of 8 NOUT + selects).
    constexpr int T0 = 16 - 4 * NOUT;
    const bool hi = (lane & 2) != 0, odd = (lane & 1) != 0;
    float2 s[2 * NOUT];
#pragma unroll
    for (int m = 0; m < 2 * NOUT; m++) {
        const int t0 = T0 + m, t1 = T0 + 2 * NOUT + m;
        const float2 p0 = cmul(v[t0], tw[kTwW3 + 64 * t0 + lane]);
        const float2 p1 = cmul(v[t1], tw[kTwW3 + 64 * t1 + lane]);
        s[m] = hi ? p1 : p0;      // kept
        v[m] = hi ? p0 : p1;      // sent to lane ^ 2 (v[] reused as scratch)
    }
#pragma unroll
    for (int m = 0; m < 2 * NOUT; m += 2) quad_exchange_add<false>(s[m], v[m], s[m + 1], v[m + 1]);
#pragma unroll
    for (int j = 0; j < NOUT; j++) {
        out[j] = odd ? s[NOUT + j] : s[j];
        v[j] = odd ? s[j] : s[NOUT + j];
    }
#pragma unroll
    for (int j = 0; j + 1 < NOUT; j += 2) quad_exchange_add<true>(out[j], v[j], out[j + 1], v[j + 1]);
    if (NOUT & 1) quad_exchange_add<true>(out[NOUT - 1], v[NOUT - 1]);
}

// ------------------------------------------------------- distance factor --
// D[k] = exp(-2 pi i * fsvs r' k / 513) * inv_frac (kernels.cu:116-125).  The phase is
// exact integer arithmetic: c = frac(fsvs r'/513) as a 64-bit fraction of a turn, phase(k) =
// k*c mod 1 (top 32 bits kept, 1.5e-9 rad), split into the nearest quarter turn and a
// remainder |f| <= 1/2 quarter turn that goes through float minimax kernels with an exactly
// represented argument (two floats).
// p = the phase word.  Branch-free: the quarter only swaps sin/cos and sets sign bits.
JF_DEV float2 distance_from_phase(unsigned p, float inv_frac) {
    const unsigned p2 = p + 0x20000000u;  // + 1/8 turn: round to the nearest quarter
    const int rem = (int)(p2 & 0x3FFFFFFFu) - 0x20000000;
    // x + xl = remainder in radians, |x| <= pi/4, to ~1e-16: the 30-bit remainder does not fit a float (rf rounds, rl
    // is what it drops) and neither does pi/2 / 2^30 (Kh + Kl); xl collects both residuals with exact FMAs
    constexpr float Kh = 0x1.921fb6p-30f, Kl = -0x1.777a5cp-55f;
    const float rf = (float)rem;
    const float rl = (float)(rem - (int)rf);
    const float x = rf * Kh;
    const float xl = fmaf(rl, Kh, fmaf(rf, Kl, fmaf(rf, Kh, -x)));
    const float x2 = x * x;
    // Cephes sinf/cosf kernels, ~1 ulp, then the first-order correction for xl
    const float s0 = x + x * x2 * (-1.6666654611e-1f + x2 * (8.3321608736e-3f + x2 * -1.9515295891e-4f));
    const float c0 = 1.0f - 0.5f * x2 +
                     x2 * x2 * (4.166664568298827e-2f + x2 * (-1.388731625493765e-3f + x2 * 2.443315711809948e-5f));
    const float s = fmaf(xl, c0, s0);
    const float c = fmaf(-xl, s0, c0);
    // quarter q = p2 >> 30: (cos, sin) = (c, s), (-s, c), (-c, -s), (s, -c); the result is (cos, -sin) * inv_frac
    const bool odd = (p2 & 0x40000000u) != 0;
    const float cc = odd ? s : c, ss = odd ? c : s;
    const unsigned neg_re = (p2 + 0x40000000u) & 0x80000000u;  // quarters 1, 2
    const unsigned neg_im = ~p2 & 0x80000000u;                 // quarters 0, 1
    return make_float2(__uint_as_float(__float_as_uint(cc * inv_frac) ^ neg_re),
                       __uint_as_float(__float_as_uint(ss * inv_frac) ^ neg_im));
}

// The same from the FFT's own twiddle table (LDS, kTwU: exp(+2 pi i j / 1024), j = 0..511, rounded from double):
// the phase is split into the nearest 1/1024 turn and a remainder |x| <= pi/1024, and the table value is corrected by
// cos x - 1 = -x^2/2 (next term 4e-12) and sin x = x - x^3/6 (next term 2e-15) as small addends to it -- one rounding
// on top of the table's, the argument needs no second float (x is known to 1.8e-10), 20 instructions and one LDS read
// instead of 33.
JF_DEV float2 distance_from_phase_tab(unsigned p, float inv_frac, const float2 *tw) {
    const unsigned p2 = p + 0x200000u;  // + 1/2048 turn: round to the nearest table entry
    // the remainder is the low 22 bits of p read as a signed number: ((p + 2^21) mod 2^22) - 2^21, one v_bfe_i32
    const int rem = (int)(p << 10) >> 10;
    const float2 t = tw[kTwU + ((p2 >> 22) & 511u)];  // (cos, sin) of entry j mod 512; entry j + 512 is its negative
    const float x = (float)rem * 0x1.921fb6p-30f;      // 2 pi / 2^32
    const float x2 = x * x;
    const float sd = x * fmaf(x2, -0x1.555556p-3f, 1.0f);  // sin x
    const float eh = 0.5f * x2;                            // 1 - cos x
    const float sf = __uint_as_float(__float_as_uint(inv_frac) | (p2 & 0x80000000u));  // inv_frac > 0; second half turn: -
#if JF_PACKED_DTAIL
    // cos(a + x) = cos a - (cos a (1 - cos x) + sin a sin x), sin(a + x) = sin a - (sin a (1 - cos x) - cos a sin x) on the
    // pair (cos a, sin a) as it comes from the table: four packed instructions for the eight of the scalar form below,
    // the same operations in the same order (bit-identical) -- and 0.4 % SLOWER (0.2488 against 0.2478 ms): the four
    // are one dependent chain, the eight are two.  Off.
    const c2 tt = c2_of(t), es = c2{eh, sd};
    c2 sfp;  // only its low half is read
    sfp.x = sf;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wuninitialized"
    const c2 corr = pfma_nrot_bhi(tt, es, pmul_blo(tt, es));  // (t.x eh + t.y sd, t.y eh - t.x sd)
    return f2_of(pmul_blo_conj(tt - corr, sfp));               // (re sf, -im sf)
#pragma clang diagnostic pop
#else
    const float re = t.x - fmaf(t.y, sd, t.x * eh);
    const float im = t.y - fmaf(-t.x, sd, t.y * eh);
    return make_float2(re * sf, -im * sf);
#endif
}

// D of this lane's bins lane + 64 q, q = 0..7, and Re D[512].  Phase words by 64-bit accumulation (two adds per bin
// instead of two quarter-rate 32-bit multiplies): hi32(k c mod 2^64), k = lane + 64 q; every bin evaluated on its own.
// Bin 512's factor is wave-uniform and was a ninth evaluation by all lanes; bin 0's is 1/frac exactly and needs none: LANE 0
// EVALUATES D[512] IN ITS SLOT OF BIN 0 (one select on the phase word), so dq[0] on lane 0 is D[512], not D[0] -- its
// callers take X[0] D[0] = X[0].x inv_frac and d512x there (distance_factor_bin0 restores the slot where D itself is wanted).
// JF_FAST_DISTANCE (off): D[lane + 64 q] = D[lane] E^q with the wave-uniform step E = exp(-2 pi i 64 c), whose
// powers lanes 0..8 evaluate and broadcast through scalar registers -- 135 fewer instructions per source-block, 4.5 %
// of the batch kernel's time, but one more rounding per factor: on a full-scale signal (|y| ~ 1.2) the output error
// against float64 grows from 2.06e-7 to 2.90e-7, past the reference's 2e-7 (profiles/r02_experiments.md).
JF_DEV void distance_factors(unsigned c_hi, unsigned c_lo, float inv_frac, int lane, float2 (&dq)[8], float &d512x,
                             [[maybe_unused]] const float2 *tw) {
    const unsigned long long c64 = ((unsigned long long)c_hi << 32) | c_lo;
#ifndef JF_FAST_DISTANCE
    unsigned long long ph = (unsigned long long)(unsigned)lane * c64;
    const unsigned long long step = c64 << 6;
#pragma unroll
    for (int q = 0; q < 8; q++) {
        unsigned p = (unsigned)(ph >> 32);
        if (q == 0) p = lane == 0 ? (unsigned)((c64 << 9) >> 32) : p;
#if JF_TABLE_DISTANCE
        dq[q] = distance_from_phase_tab(p, inv_frac, tw);
        if (q & 1) __builtin_amdgcn_sched_barrier(0);  // two table reads in flight, not eight (registers)
#else
        dq[q] = distance_from_phase(p, inv_frac);
#endif
        ph += step;
    }
    d512x = dq[0].x;  // on lane 0 (the only lane that uses it)
#else
    const float2 d0 = distance_from_phase((unsigned)(((unsigned long long)(unsigned)lane * c64) >> 32), inv_frac);
    const float2 e = distance_from_phase((unsigned)(((unsigned long long)(unsigned)(lane & 15) * (c64 << 6)) >> 32), 1.0f);
    dq[0] = d0;
#pragma unroll
    for (int q = 1; q < 8; q++) {
        const float ex = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e.x), q));
        const float ey = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e.y), q));
        dq[q] = cmul(d0, make_float2(ex, ey));
    }
    d512x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e.x), 8)) * inv_frac;
#endif
}

// ------------------------------------------------------ filter + inverse --
// For this lane's bins k = lane + 64 q: he = sum_t w[t] * H[rows[t]][k] (both ears in one
// float4), Y_ear = (X D)[k] * he_ear, and the two inputs of the inverse transform it yields:
// v[q] = Z[k] = Y_L + j Y_R and mir[q] = Z[N-k] = conj Y_L + j conj Y_R.
// xd[q] = X[k] D[k]; on lane 0, xd[0] = (X0*D0.re, X512*D512.re).
// use(q, zk, zm) consumes bin q's pair (stores it for a per-source inverse, or adds it to a group's sums).
template <int NT, class F>
JF_DEV void filtered_bins(const float4 *__restrict__ htab, const int *rows, const float *w,
                          const float2 (&xd)[8], int lane, F &&use) {
    const float4 *hp[NT];
    float wt[NT];
    // row addresses stay scalar (table + row, wave-uniform) with one per-lane byte offset for all rows (see filtered_half)
    unsigned boff = 16u * (unsigned)lane;
    asm("" : "+v"(boff));
#pragma unroll
    for (int t = 0; t < NT; t++) {
        hp[t] = htab + (size_t)__builtin_amdgcn_readfirstlane(rows[t]) * 512;
        wt[t] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(w[t])));
    }
    // JF_CHUNK_LOADS row loads (16 B per lane each) in flight per round; the weighted sum is
    // folded into Z right away so that only the loads of one round are live.
    constexpr int QC = (JF_CHUNK_LOADS / NT) > 8 ? 8 : (JF_CHUNK_LOADS / NT);
#pragma unroll
    for (int qc = 0; qc < 8; qc += QC) {
        float4 h[QC][NT];
#pragma unroll
        for (int q = 0; q < QC; q++)
#pragma unroll
            for (int t = 0; t < NT; t++)
                h[q][t] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(hp[t] + 64 * (qc + q)) + boff);
#pragma unroll
        for (int q = 0; q < QC; q++) {
            float4 he = make_float4(wt[0] * h[q][0].x, wt[0] * h[q][0].y, wt[0] * h[q][0].z, wt[0] * h[q][0].w);
#pragma unroll
            for (int t = 1; t < NT; t++) {
                he.x += wt[t] * h[q][t].x;
                he.y += wt[t] * h[q][t].y;
                he.z += wt[t] * h[q][t].z;
                he.w += wt[t] * h[q][t].w;
            }
            const float2 x = xd[qc + q];
            const float2 yl = cmul(x, make_float2(he.x, he.y));
            const float2 yr = cmul(x, make_float2(he.z, he.w));
            float2 zk = make_float2(yl.x - yr.y, yl.y + yr.x);
            float2 zm = make_float2(yl.x + yr.y, yr.x - yl.y);
            if (qc + q == 0) {
                // lane 0: bins 0 and 512 (real spectra; c2r drops their imaginary parts)
                const float2 z0 = make_float2(x.x * he.x, x.x * he.z);    // Z[0]
                const float2 z512 = make_float2(x.y * he.y, x.y * he.w);  // Z[512]
                zk = lane == 0 ? z0 : zk;
                zm = lane == 0 ? z512 : zm;
            }
            use(qc + q, zk, zm);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <class F>
JF_DEV void filtered_bins_nt(int nt, const float4 *__restrict__ htab, const int *rows, const float *w,
                             const float2 (&xd)[8], int lane, F &&use) {
    if (nt == 4)
        filtered_bins<4>(htab, rows, w, xd, lane, use);
    else if (nt == 2)
        filtered_bins<2>(htab, rows, w, xd, lane, use);
    else
        filtered_bins<1>(htab, rows, w, xd, lane, use);
}

// The inverse of one spectrum given as this lane's Z[k] (zk) and Z[N-k] (zm), k = lane + 64 q: the upper
// half Z[lane + 64 r], r = 8..15, lives mirrored on lane 64 - lane (lane 0: r = 8 -> Z[512], else
// Z[N - 64 (16 - r)]).
template <int NOUT, bool SPLIT>
JF_DEV void inverse_of(float2 (&v)[16], const float2 (&zm)[8], float2 (&out)[NOUT], float2 *buf, const float2 *tw,
                       int lane) {
    float2 up[8];
    mirror8(buf, zm, up, lane);
#pragma unroll
    for (int r = 8; r < 16; r++) v[r] = up[r - 8];
    ifft1024_lastq_wave<NOUT, SPLIT>(v, out, buf, tw, lane);
}

// One filter set for this lane's bins, then the inverse transform.
template <int NOUT, bool SPLIT>
JF_DEV void filter_set(int nt, const float4 *__restrict__ htab, const int *rows, const float *w,
                       const float2 (&xd)[8], float2 (&out)[NOUT], float2 *buf, const float2 *tw,
                       int lane) {
    float2 v[16], mir[8];
    filtered_bins_nt(nt, htab, rows, w, xd, lane, [&](int q, float2 zk, float2 zm) {
        v[q] = zk;
        mir[q] = zm;
    });
    inverse_of<NOUT, SPLIT>(v, mir, out, buf, tw, lane);
}

// ------------------------------------------------------------ fused kernel --
#ifndef JF_BLOCK_D_EARLY
#define JF_BLOCK_D_EARLY 1  // per-source kernels: distance factors while the window loads are in flight
#endif
constexpr bool kSplit = JF_SPLIT_EXCHANGE != 0;
constexpr int kWaveLds = kSplit ? 576 : 1088;  // float2 per wave: the inverse exchange (8704 B), or with the split
                                               // exchange the forward passes (4608 B)

// Front half of one (block b, source s) work item: window gather (Audio.cu:121-139,
// GPUSoundSource.cu:472-513), write-back of the window and counters at the last block of a call, forward
// FFT with its 1/N (GPUSoundSource.cu:344-346), times the distance factor: xd[q] = X[k] D[k], k = lane + 64 q
// (lane 0: xd[0] = (X0 D0.re, X512 D512.re)).  D_EARLY: the distance factors are computed while the window
// loads are in flight (16 more registers live across the FFT).  False if the item is silent (not
// interpolable: the reference has no defined output there).
// First part of the front half: the loads of the window, nothing that waits for them (the caller may put other work
// between this and item_finish).  count0 / L: the source's play position and (stored) signal length.
template <int NOUT>
JF_DEV void item_gather(const FusedParams &P, int b, int s, int lane, float2 (&z)[8], int &count0_out, int &L_out) {
    constexpr int B = 64 * NOUT;
    const SrcSignal JF_CONST_AS *sgc = as_const(P.sigs + s);
    SrcSignal sg;
    sg.ptr = sgc->ptr;
    sg.length = sgc->length;
    const int count0 = as_const(P.st_in + s)->count;
    const float *hist = P.hist_in + (size_t)s * kN;
    // the pointer comes out of a table in memory: tell the compiler it is global memory, or every window load is a
    // flat load that also ties up the LDS counter
    const gfloat *sigp = (const gfloat *)sg.ptr;
    // first NEW sample of this call has q = 0; window sample n has q = b*B + n - (N - B)
    // The engine stores every signal with length >= N (short ones tiled, empty ones as
    // zeros), so one conditional subtract wraps the loop.
    const int q0 = b * B - (kN - B);
    const int L = sg.length;
    const int qpos = q0 > 0 ? q0 : 0;
    const int base = (int)(((long long)count0 + qpos) % L) - qpos;  // signal index of q = 0 (mod L)
    // Where the window comes from is wave-uniform.  Usual case: all of it from one stretch of the looped signal --
    // eight loads of a sample pair per lane at scalar base + 8 lane + 512 r, no per-lane index arithmetic.  Else
    // (the first blocks of a call, whose windows reach back into the previous one; a window across the loop point of
    // the signal) every lane works out where its samples are; all loads are in flight together either way.
    int start0 = base + q0;  // signal index of the window's first sample (meaningful for q0 >= 0), < L + N
    start0 = start0 >= L ? start0 - L : start0;
    const bool one_stretch = q0 >= 0 && start0 + kN <= L;
    if (one_stretch) {
        const gpair *p = reinterpret_cast<const gpair *>(sigp + start0 + 2u * lane);
#pragma unroll
        for (int r = 0; r < 8; r++) z[r] = make_float2(p[64 * r].x, p[64 * r].y);
    } else {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int qr = q0 + 128 * r;  // first sample of this 128-sample row (wave-uniform)
            if ((NOUT % 2 == 0) && qr < 0) {
                // B is a multiple of 128: the window/signal boundary falls between rows
                z[r] = *reinterpret_cast<const float2 *>(hist + (kN + qr) + 2 * lane);
            } else if (NOUT % 2 == 0) {
                int i0 = base + qr + 2 * lane, i1 = i0 + 1;  // < L + N
                i0 = i0 >= L ? i0 - L : i0;
                i1 = i1 >= L ? i1 - L : i1;
                z[r] = make_float2(sigp[i0], sigp[i1]);
            } else {
                float xv[2];
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const int q = qr + 2 * lane + c;
                    int idx = base + q;  // < L + N for q >= 0
                    idx = idx >= L ? idx - L : idx;
                    xv[c] = q < 0 ? hist[kN + q] : sigp[idx];
                }
                z[r] = make_float2(xv[0], xv[1]);
            }
        }
    }
    count0_out = count0;
    L_out = L;
}

// Second part: write-back of the window and counters at the last block of a call, forward FFT, distance factor.
template <int NOUT, bool D_EARLY>
JF_DEV bool item_finish(const FusedParams &P, const ItemDesc *dp, const float *pos_rec, int b, int s, float2 *buf,
                        const float2 *s_tw, int lane, float2 (&z)[8], int count0, int L, float2 (&xd)[8]) {
    constexpr int B = 64 * NOUT;
    // ---- descriptor (wave-uniform -> scalar loads)
    const int n_new = dp->n_new;
    const unsigned c_hi = (unsigned)(dp->c_fix >> 32), c_lo = (unsigned)dp->c_fix;
    const float inv_frac = dp->inv_frac;
    // ---- distance factor.  The 1/N of the forward transform and the 1/2 of its split pass ride on
    // 1/frac: powers of two, exact
    float2 dq[8];
    const float sinv = inv_frac * (1.0f / 2048.0f);
    float d512x;
    if (D_EARLY) {
        distance_factors(c_hi, c_lo, sinv, lane, dq, d512x, s_tw);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (b == P.K - 1) {
        // last block of the call: leave the window and the counters for the next call
        float *ho = P.hist_out + (size_t)s * kN;
#pragma unroll
        for (int r = 0; r < 8; r++) *reinterpret_cast<float2 *>(ho + 2 * (lane + 64 * r)) = z[r];
        if (lane == 0) {
            SrcState st;
            st.count = (int)(((long long)count0 + (long long)P.K * B) % L);
            const float *pp = pos_rec;
            st.old_ele = pp[0];
            st.old_azi = pp[1];
            st.pad = 0;
            P.st_out[s] = st;
        }
    }

    if (n_new <= 0) return false;

    float2 X[8];
    rfft1024_wave(z, X, buf, s_tw, lane);
    if (!D_EARLY) distance_factors(c_hi, c_lo, sinv, lane, dq, d512x, s_tw);
#pragma unroll
    for (int q = 0; q < 8; q++) xd[q] = cmul_pk(X[q], dq[q]);
    const float2 x0 = make_float2(X[0].x * sinv, X[0].y * d512x);
    xd[0] = lane == 0 ? x0 : xd[0];
    return true;
}

template <int NOUT, bool D_EARLY>
JF_DEV bool item_front(const FusedParams &P, const ItemDesc *dp, const float *pos_rec, int b, int s,
                       float2 *buf, const float2 *s_tw, int lane, float2 (&xd)[8]) {
    float2 z[8];
    int count0, L;
    item_gather<NOUT>(P, b, s, lane, z, count0, L);
    return item_finish<NOUT, D_EARLY>(P, dp, pos_rec, b, s, buf, s_tw, lane, z, count0, L, xd);
}

// One (block b, source s) work item by one wavefront: everything from the window gather to the
// crossfaded stereo frames, which are ADDED to acc (B/64 frames per lane: frame i + 16 (NOUT a + j),
// lane = 4 i + a).  dp: this item's descriptor (global memory in the batch kernel, LDS in the
// real-time kernel); pos_rec: its latched position record.  buf: this wave's LDS; s_tw: twiddle pack.
template <int NOUT>
JF_DEV void spatialise_item(const FusedParams &P, const ItemDesc *dp, const float *pos_rec, int b, int s,
                            float2 *buf, const float2 *s_tw, int lane, float2 (&acc)[NOUT]) {
    constexpr int B = 64 * NOUT;
    const int a = lane & 3, i = lane >> 2;
    const int n_new = dp->n_new;
    const int n_old = dp->n_old;
    float2 xd[8];
    if (!item_front<NOUT, JF_BLOCK_D_EARLY != 0>(P, dp, pos_rec, b, s, buf, s_tw, opaque(lane), xd)) return;

    // ---- filter set(s) + inverse + crossfade (GPUSoundSource.cu:351-381)
    float2 res[NOUT];
#pragma unroll 1
    for (int set = (n_old > 0 ? 0 : 1); set < 2; set++) {
        const int *rows = set ? dp->rows_new : dp->rows_old;
        const float *w = set ? dp->w_new : dp->w_old;
        float2 mine[NOUT];  // frames i + 16 (NOUT a + j) of the block
        filter_set<NOUT, kSplit>(set ? n_new : n_old, P.htab, rows, w, xd, mine, buf, s_tw, opaque(lane));
#pragma unroll
        for (int j = 0; j < NOUT; j++) {
            float2 r1 = mine[j];
            if (set == 0) {
                res[j] = r1;
            } else {
                const int n_out = i + 16 * (NOUT * a + j);  // frame inside the block
                if (n_old > 0) {
                    // kernels.cu:132-137
                    const float fn = (float)n_out / ((float)B - 1.0f);
                    r1 = make_float2(res[j].x * (1.0f - fn) + r1.x * fn,
                                     res[j].y * (1.0f - fn) + r1.y * fn);
                }
                acc[j] = cadd(acc[j], r1);
            }
        }
    }
}

template <int NOUT>  // B / 64
#if JF_MIN_WAVES > 0
#define JF_FUSED_BOUNDS __launch_bounds__(64 * kWavesPerWg, JF_MIN_WAVES)
#else
#define JF_FUSED_BOUNDS __launch_bounds__(64 * kWavesPerWg)
#endif
__global__ JF_FUSED_BOUNDS void fused_block_kernel(const FusedParams P) {
    __shared__ float2 s_tw[kTwPack];
    __shared__ float2 s_buf[kWavesPerWg * kWaveLds];
    const int tid = threadIdx.x;
    for (int j = tid; j < kTwPack; j += 64 * kWavesPerWg) s_tw[j] = P.tw[j];
    __syncthreads();

    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float2 *buf = s_buf + wave * kWaveLds;
    constexpr int B = 64 * NOUT;
    // Persistent workgroups: the grid is sized to the machine (launch_fused), the twiddle pack
    // is staged once, and every wave strides over the work units.  A unit = one block of G
    // consecutive sources, processed one after the other and summed in source order in
    // registers, so only one stereo block per group is written (G = 1: per-source blocks).
    const int G = P.G, SG = P.S / G;
    const int n_units = P.K * SG;
    const int a = lane & 3, i = lane >> 2;
    [[maybe_unused]] int steps_done = 0;
#pragma unroll 1
    for (int unit = blockIdx.x * kWavesPerWg + wave; unit < n_units; unit += gridDim.x * kWavesPerWg) {
#if JF_UNIT_ORDER
        // consecutive waves take consecutive BLOCKS of the same sources: their table rows and windows overlap
        const int sg = unit / P.K;
        const int b = unit - sg * P.K;
        const int s0 = sg * G;
#else
        const int b = unit / SG;
        const int s0 = (unit - b * SG) * G;
#endif
        float2 acc[NOUT];
#pragma unroll
        for (int j = 0; j < NOUT; j++) acc[j] = make_float2(0.f, 0.f);
#pragma unroll 1
        for (int g = 0; g < G; g++) {
#if JF_PAIR_ROTATE_PRIO
            switch (3 - (steps_done++ & 3)) {  // progress-ordered priorities, see fused_pair_kernel
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
            }
#endif
            const int item = b * P.S + s0 + g;
            const ItemDesc dl = load_desc(P.desc + item);  // scalar loads
            spatialise_item<NOUT>(P, &dl, P.pos + (size_t)item * 5, b, s0 + g, buf, s_tw, lane, acc);
        }
#if JF_UNIT_ORDER
        float2 *out = reinterpret_cast<float2 *>(P.partial) + ((size_t)b * SG + sg) * B;
#else
        float2 *out = reinterpret_cast<float2 *>(P.partial) + (size_t)unit * B;
#endif
#pragma unroll
        for (int j = 0; j < NOUT; j++) out[i + 16 * (NOUT * a + j)] = acc[j];
    }
}

// ------------------------------------------------------------- pair kernel --
// The group kernel's work with BOTH filter sets of a unit's sources summed as spectra:
//     sum_s [ old_s (1 - f) + new_s f ]  =  (1 - f) IFFT( sum_s Z_old,s )  +  f IFFT( sum_s Z_new,s ),
// two inverse transforms per unit instead of G + 1, and one round of table-row loads per source for both
// sets when they share rows (a source that moved by a degree inside one grid cell interpolates between the
// same four rows with other weights).  Two spectral sums are 64 floats per lane -- more than a wavefront can
// hold at four waves per SIMD -- so a unit is worked by a PAIR of wavefronts (w and 15 - w of a workgroup) that
// split the BINS of the sums, not the transforms: every FFT stays a one-wave transform.
//   * wave `half` keeps the sums of bins lane + 64 q, q = 4 half .. 4 half + 3, of both sets: Z[k] and
//     Z[N-k], 32 registers;
//   * the waves alternate over the unit's sources: the owner of source g runs its front half (window,
//     forward FFT, distance factor), keeps its own bins of X D and leaves the partner's in a mailbox in
//     LDS; each wave filters and accumulates its bins of every source -- G / 2 forward transforms and G
//     half-filters per wave;
//   * at the end wave 0 inverts the old sum, wave 1 the new one (each fetches the other half of its sum
//     from the partner's mailbox), wave 0 hands its frames over, wave 1 cross-fades and stores the block.
// Hand-offs are sequence-numbered flags in LDS (no workgroup barrier: only the two waves of a pair wait for
// each other): pub[w] = number of hand-offs wave w has published, ack[w] = number of the partner's hand-offs
// wave w has consumed.  A wave has two mailbox slots and consumes the partner's source one step late, so it
// waits only when the partner has fallen a whole source behind.  Both waves run the same sequence of
// hand-offs, so the counts always match, and every wait is for something the partner reaches without
// waiting for anything this wave has not done yet.  Each wave adds its sources in a fixed order.
constexpr int kPairsPerWg = kWavesPerWg / 2;
constexpr int kPairWork = 576;                          // float2: a wave's FFT work space
constexpr int kPairMail = 256;                          // float2: one mailbox slot = 4 bins x 64 lanes
constexpr int kPairWave = kPairWork + 2 * kPairMail;    // per wave
constexpr int kPairLds = 2 * kPairWave + 2;             // per pair, + 4 flag words

// Flag words are read and written with explicit LDS instructions on their LDS byte address (the low half of the
// generic address): a volatile access through a generic pointer compiles to flat loads.
JF_DEV int lds_flag_read(unsigned addr) {
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return __builtin_amdgcn_readfirstlane(v);
}
JF_DEV void lds_flag_write(unsigned addr, int v) {
    asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
// Every wait is bounded (about a tenth of a second): a hand-off that never arrives -- impossible by the protocol
// above -- raises the host-visible error word and lets the grid drain instead of hanging the GPU.
JF_DEV void pair_wait(unsigned flag, int v, int *err, bool &dead) {
    if (dead) return;  // after one time-out this wave no longer waits for anything
    for (int spins = 0; lds_flag_read(flag) < v; spins++) {
        __builtin_amdgcn_s_sleep(1);
        if (spins > (1 << 20)) {
            *reinterpret_cast<volatile int *>(err) = 1;
            dead = true;
            return;
        }
    }
}

// Bins qb .. qb + 3 (lofs = 64 qb + lane) of one filter set, or of two sets that read the same rows with different
// weights (BOTH): sLa[q] += X D * H_left, sRa[q] += X D * H_right with set a's weights (and sLb, sRb with set b's).
// The sums are kept PER EAR, not as Z[k] = Y_L + i Y_R and Z[N-k] = conj Y_L + i conj Y_R: a product that is added to a
// sum is two packed FMAs, whereas forming Z[k] and Z[N-k] of every source and adding those took a multiply, an FMA and
// two packed adds per ear (8 packed instructions per bin and source fewer with two sets: a quarter of the filter's);
// the caller forms Z[k] and Z[N-k] once per unit (ear_sums_to_z).  Row addresses stay scalar (table + row, wave-uniform)
// with ONE per-lane offset register for all rows: loads in the saddr form, no 64-bit pointer pair per row.
// fetch(xh) delivers the source's X D for these bins; it is called AFTER the first two stages of row loads have been
// requested, so whatever it waits for (the partner's hand-off flag, the mailbox read) overlaps with their latency.
// he_ear = sum_t w_t H[row_t]: a packed multiply by the first weight, then one packed FMA per further row, in the rows'
// order.  The ONE place this sum is written: the half-filters below and table_interp_build_kernel (the pre-interpolated
// rows) both call it, so a pre-interpolated row holds bit for bit what a half-filter forms from the measured rows.
template <int NT>
JF_DEV void weighted_ears(const float4 (&hq)[NT], const c2 (&w)[NT], c2 &heL, c2 &heR) {
    heL = pmul_s(c2{hq[0].x, hq[0].y}, w[0]);
    heR = pmul_s(c2{hq[0].z, hq[0].w}, w[0]);
#pragma unroll
    for (int t = 1; t < NT; t++) {
        heL = pfma_s(c2{hq[t].x, hq[t].y}, w[t], heL);
        heR = pfma_s(c2{hq[t].z, hq[t].w}, w[t], heR);
    }
}

// s_ear += x * he_ear, two packed FMAs per ear (jf_packed.h).  q == 0 && special: lane 0 of the lower half, where bins 0
// and 512 travel as the two halves of one entry (real spectra; c2r drops their imaginary parts) and the product is
// element by element.  Both products are the same two packed FMAs on different first operands -- (re, re) then (im, .)
// for a complex product, (re, im) then (0, .) for the element-wise one -- so the exception costs two selects per bin 0,
// shared by the ears and sets, instead of a second product and four selects per ear and set (the selects measured 3 % of
// the kernel's time).
JF_DEV void mac_ears(int q, bool special, float2 x, c2 heL, c2 heR, c2 &sL, c2 &sR) {
    if (q == 0) {
        const c2 xa = c2{x.x, special ? x.y : x.x};
        c2 xb;
        xb.x = special ? 0.0f : x.y;
        xb.y = xb.x;
        sL = pfma_lo_rot(xb, heL, pfma_each(xa, heL, sL));
        sR = pfma_lo_rot(xb, heR, pfma_each(xa, heR, sR));
    } else {
        const c2 xc = c2_of(x);
        sL = pcmac(xc, heL, sL);
        sR = pcmac(xc, heR, sR);
    }
}

template <int NT, bool BOTH, class X>
JF_DEV void filtered_half(const float4 *__restrict__ htab, unsigned lofs, const int *rows, const float *wa, const float *wb,
                          X &&fetch, bool special, c2 (&sLa)[4], c2 (&sRa)[4], c2 (&sLb)[4], c2 (&sRb)[4]) {
    const float4 *hp[NT];
    c2 a[NT], b[NT];  // (w, w): a weight as a scalar-register pair feeds both halves of a packed operation
    unsigned boff = 16u * lofs;  // byte offset of this lane's first bin inside a row
    // opaque to the optimiser here: otherwise it folds table + lane offset into one loop-invariant 64-bit VGPR pointer
    // and adds the row to that, one VGPR pair per row
    asm("" : "+v"(boff));
#pragma unroll
    for (int t = 0; t < NT; t++) {
        // rows and weights are wave-uniform: pin them to scalar registers
        hp[t] = htab + (size_t)__builtin_amdgcn_readfirstlane(rows[t]) * 512;
        const float fa = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(wa[t])));
        const float fb = BOTH ? __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(wb[t]))) : 0.0f;
        a[t] = c2{fa, fa};
        b[t] = c2{fb, fb};
    }
    // Packed f32 throughout (jf_packed.h): this is multiply-accumulate work.
    auto mac2 = [&](int q, float2 x, c2 heL, c2 heR, c2 &sL, c2 &sR) { mac_ears(q, special, x, heL, heR, sL, sR); };
    // The four bins in stages of JF_STAGE_LOADS row loads (16 B per lane each), two stages in flight: while one
    // stage's rows are weighted and multiplied, the next stage's loads are already under way -- the load latency is
    // paid once per call, not once per stage (the loads are L2 hits; it is their latency, not their bandwidth, that
    // the kernel waits for: profiles/r02_experiments.md).
    constexpr int QC = (JF_STAGE_LOADS / NT) > 4 ? 4 : ((JF_STAGE_LOADS / NT) < 1 ? 1 : (JF_STAGE_LOADS / NT));
    constexpr int NS = 4 / QC;
    // JF_STAGE_DEPTH stages in flight (2; 3 or 4 = more landing registers: what a stage's wait costs is in profiles/r05/l1_bound.md)
    constexpr int DEPTH = JF_STAGE_DEPTH < NS ? JF_STAGE_DEPTH : NS;
    float4 h[DEPTH][QC][NT];
    auto load_stage = [&](int st) {
#pragma unroll
        for (int q = 0; q < QC; q++)
#pragma unroll
            for (int t = 0; t < NT; t++)
                h[st % DEPTH][q][t] =
                    *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(hp[t] + 64 * (QC * st + q)) + boff);
    };
#pragma unroll
    for (int st = 0; st < DEPTH; st++) load_stage(st);
    __builtin_amdgcn_sched_barrier(0);
    float2 xh[4];
    fetch(xh);
#pragma unroll
    for (int st = 0; st < NS; st++) {
        if (st >= 1 && st + DEPTH - 1 < NS) load_stage(st + DEPTH - 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < QC; q++) {
            const float4(&hq)[NT] = h[st % DEPTH][q];
            c2 haL, haR, hbL = c2{0.f, 0.f}, hbR = c2{0.f, 0.f};
            weighted_ears<NT>(hq, a, haL, haR);
            if (BOTH) weighted_ears<NT>(hq, b, hbL, hbR);
            const int qq = QC * st + q;
            mac2(qq, xh[qq], haL, haR, sLa[qq], sRa[qq]);
            if (BOTH) mac2(qq, xh[qq], hbL, hbR, sLb[qq], sRb[qq]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <bool BOTH, class X>
JF_DEV void filtered_half_nt(int nt, const float4 *__restrict__ htab, unsigned lofs, const int *rows, const float *wa,
                             const float *wb, X &&fetch, bool special, c2 (&sLa)[4], c2 (&sRa)[4], c2 (&sLb)[4],
                             c2 (&sRb)[4]) {
    if (nt == 4)
        filtered_half<4, BOTH>(htab, lofs, rows, wa, wb, fetch, special, sLa, sRa, sLb, sRb);
    else if (nt == 2)
        filtered_half<2, BOTH>(htab, lofs, rows, wa, wb, fetch, special, sLa, sRa, sLb, sRb);
    else
        filtered_half<1, BOTH>(htab, lofs, rows, wa, wb, fetch, special, sLa, sRa, sLb, sRb);
}

// The same for sets that are PRE-INTERPOLATED rows (ItemDesc flags bit 2: a whole-degree position, whose weighted sum
// table_interp_build_kernel has formed once and for all): one row per set, no weighting -- per bin one 16-byte load and
// four packed FMAs per set instead of four loads and twelve packed instructions.  row_a feeds sLa / sRa; with BOTH row_b
// feeds sLb / sRb (old and new set of a moving source: different rows).  All eight loads are in flight before fetch().
template <bool BOTH, class X>
JF_DEV void filtered_half_pre(const float4 *__restrict__ htab, unsigned lofs, int row_a, int row_b, X &&fetch, bool special,
                              c2 (&sLa)[4], c2 (&sRa)[4], c2 (&sLb)[4], c2 (&sRb)[4]) {
    unsigned boff = 16u * lofs;
    asm("" : "+v"(boff));  // see filtered_half
    const float4 *ha = htab + (size_t)__builtin_amdgcn_readfirstlane(row_a) * 512;
    const float4 *hb = htab + (size_t)__builtin_amdgcn_readfirstlane(row_b) * 512;
    float4 h[2][4];  // (h[1] is not touched without BOTH)
#pragma unroll
    for (int q = 0; q < 4; q++) {
        h[0][q] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(ha + 64 * q) + boff);
        if (BOTH) h[1][q] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(hb + 64 * q) + boff);
    }
    __builtin_amdgcn_sched_barrier(0);
    float2 xh[4];
    fetch(xh);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 4; q++) {
        mac_ears(q, special, xh[q], c2{h[0][q].x, h[0][q].y}, c2{h[0][q].z, h[0][q].w}, sLa[q], sRa[q]);
        if (BOTH) mac_ears(q, special, xh[q], c2{h[1][q].x, h[1][q].y}, c2{h[1][q].z, h[1][q].w}, sLb[q], sRb[q]);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// The per-ear sums of a unit -> Z[k] = Y_L + i Y_R and Z[N-k] = conj Y_L + i conj Y_R, the inverse transform's input
// (one c2r transform yields both ears: Y_L in the real parts, Y_R in the imaginary parts of the frames).
JF_DEV void ear_sums_to_z(const c2 (&sL)[4], const c2 (&sR)[4], bool special, c2 (&zk)[4], c2 (&zm)[4]) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
        zk[q] = padd_i(sL[q], sR[q]);
        zm[q] = pcadd_ic(sL[q], sR[q]);
    }
    // lane 0 of the lower half holds bins 0 and 512 in the halves of its first entry
    const c2 z0 = c2{sL[0].x, sR[0].x}, z512 = c2{sL[0].y, sR[0].y};
    zk[0] = special ? z0 : zk[0];
    zm[0] = special ? z512 : zm[0];
}

#ifndef JF_PAIR_MIX_AGES
#define JF_PAIR_MIX_AGES 1  // which two waves of a workgroup form a pair: 1 = w and 15 - w, 2 = same SIMD, 0 = 2i and 2i + 1
#endif
#ifndef JF_PAIR_RELOAD_PARAMS
#define JF_PAIR_RELOAD_PARAMS 1
#endif
#ifndef JF_PAIR_D_EARLY
#define JF_PAIR_D_EARLY 0
#endif
#ifndef JF_PAIR_ACK_IN_VGPR
#define JF_PAIR_ACK_IN_VGPR 1
#endif
#ifndef JF_PAIR_OVERLAP
#define JF_PAIR_OVERLAP 0  // 1: a wave's window loads fly while it filters the partner's previous source -- 16 more live
                           // registers, which spill (72 B) and cost more than the overlap gains: 0.195 vs 0.182 ms
#endif

JF_DEV void prep_body(const RingTable &rt, int mode, const float *__restrict__ pos, const SrcState *__restrict__ st,
                      ItemDesc *__restrict__ desc, int S, int K, int canon, int tid, ItemDesc *stage);

// ROWS: the instantiation for launches whose descriptors may carry pre-interpolated rows (ItemDesc flags bit 2); the other
// one does not contain that path at all (its presence alone costs the per-block weighting path registers and ~4 % more
// instructions per source-block: profiles/r04/interp_table.md).
template <int NOUT, bool ROWS>
__global__ JF_FUSED_BOUNDS void fused_pair_kernel(const FusedParams Pin) {
    __shared__ float2 s_tw[kTwPack];
    __shared__ float2 s_pair[kPairsPerWg * kPairLds];
    const int tid = threadIdx.x;
#if JF_PAIR_RELOAD_PARAMS
    // The launch parameters are read again from the kernel-argument segment (scalar loads, scalar cache) wherever the
    // source loop needs them, instead of being held in scalar registers from kernel entry: with ~30 parameter registers
    // live across the whole persistent loop the compiler ran out of scalar registers and kept 49 of them in the lanes of
    // a vector register -- every use a v_readlane, i.e. a VECTOR instruction (~40 per source-block).
    FusedParams P;
    const FusedParams JF_CONST_AS *const kernarg =
        (const FusedParams JF_CONST_AS *)__builtin_amdgcn_kernarg_segment_ptr();
    // FusedParams is this kernel's ONLY parameter, so the kernel-argument segment starts with it; the whole struct is
    // copied (every field of P is valid after a reload -- the loads of fields nobody reads afterwards are dead code).
    static_assert(std::is_trivially_copyable<FusedParams>::value && std::is_standard_layout<FusedParams>::value,
                  "FusedParams is re-read from the kernel-argument segment as plain bytes");
    auto reload_params = [&]() {
        const FusedParams JF_CONST_AS *q = kernarg;
        asm volatile("" : "+s"(q));
        // EVERY field, so that no stale or uninitialised one can be read after a reload (the size check below trips when a
        // field is added to FusedParams and not here)
        P.htab = q->htab, P.tw = q->tw, P.desc = q->desc, P.sigs = q->sigs, P.st_in = q->st_in, P.st_out = q->st_out;
        P.hist_in = q->hist_in, P.hist_out = q->hist_out, P.pos = q->pos, P.partial = q->partial;
        P.S = q->S, P.K = q->K, P.B = q->B, P.G = q->G, P.mode = q->mode, P.order = q->order, P.err = q->err;
        P.n_pair_wgs = q->n_pair_wgs, P.prep_pos = q->prep_pos, P.prep_desc = q->prep_desc, P.prep_K = q->prep_K;
        P.prep_canon = q->prep_canon;
        // (P.rt is not reloaded: nothing behind a reload reads it -- the workgroups that build descriptors take Pin.rt)
        static_assert(sizeof(FusedParams) == 10 * 8 + 5 * 4 + 4 + 2 * 8 + 4 + 4 + 2 * 8 + 2 * 4 + sizeof(RingTable),
                      "a field was added to FusedParams: reload it here too");
    };
#else
    const FusedParams &P = Pin;
    auto reload_params = []() {};
#endif
    if ((int)blockIdx.x >= Pin.n_pair_wgs) {
        // The trailing workgroups of the grid: the descriptors of the window that follows this run (prep_kernel's work).
        // They are dispatched as compute units come free, i.e. while the slowest pairs are still on their last unit: the
        // 10 us chain of the index/weight rule hides in the kernel's tail instead of standing behind it as a launch.
        static_assert(sizeof(s_pair) >= sizeof(ItemDesc) * 32 * kWavesPerWg, "staging of 32 records per wave");
        prep_body(Pin.rt, Pin.mode, Pin.prep_pos, nullptr, Pin.prep_desc, Pin.S, Pin.prep_K, Pin.prep_canon,
                  ((int)blockIdx.x - Pin.n_pair_wgs) * (64 * kWavesPerWg) + tid, reinterpret_cast<ItemDesc *>(s_pair));
        return;
    }
    for (int j = tid; j < kTwPack; j += 64 * kWavesPerWg) s_tw[j] = Pin.tw[j];
    reload_params();
    if (tid < kPairsPerWg) {
        int *f = reinterpret_cast<int *>(s_pair + tid * kPairLds + 2 * kPairWave);
        f[0] = f[1] = f[2] = f[3] = 0;
    }
    // (the barrier moved behind the first unit's descriptor scan, so that the scan's scalar loads overlap with the staging above:
    // 0.7 % SLOWER -- profiles/r04/pair_kernel_residue.md)
    __syncthreads();

    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Which waves pair up.  The SIMD's arbiter serves its oldest wave first on a tie, and a pair runs at the pace of its
    // slower wave: as (2i, 2i + 1) the last pairs of a workgroup left the kernel 12 us after the first (profiles/stamps.py)
    // and the SIMDs ran half empty meanwhile.  As (w, 15 - w) -- an old wave with a young one, on different SIMDs -- every
    // pair is held back alike: units take 5 % longer, the pairs leave within 4 us of each other, the kernel is 5 % shorter.
#if JF_PAIR_MIX_AGES == 2
    // both waves of a pair on one SIMD (wave w runs on SIMD w % 4), oldest with youngest and the middle two together: 1 % slower
    const int pair = 2 * (wave & 3) + (((wave >> 2) == 1 || (wave >> 2) == 2) ? 1 : 0), half = wave >> 3;
    static_assert(kPairsPerWg == 8, "16 waves");
#elif JF_PAIR_MIX_AGES
    const int pair = wave < kPairsPerWg ? wave : 2 * kPairsPerWg - 1 - wave, half = wave < kPairsPerWg ? 0 : 1;
#else
    const int pair = wave >> 1, half = wave & 1;
#endif
    JF_EXP_STAMP_SETUP(P, pair, half, lane);
    JF_EXP_PHASE_SETUP();
    float2 *base = s_pair + pair * kPairLds;
    float2 *buf = base + half * kPairWave;  // my FFT work space
    float2 *mail = buf + kPairWork;         // my two mailbox slots
    const float2 *pbuf = base + (half ^ 1) * kPairWave, *pmail = pbuf + kPairWork;  // the partner's
    const unsigned flags = (unsigned)(size_t)(base + 2 * kPairWave);  // LDS byte address of pub[2], ack[2]
    const unsigned my_pub = flags + 4 * half, his_pub = flags + 4 * (half ^ 1);
#if JF_PAIR_ACK_IN_VGPR
    // my_ack is needed once per source (consumed()), as the ADDRESS operand of a ds_write -- a vector register anyway: held in a
    // scalar register it is one of the scalars the compiler keeps in lanes of a vector register (v_readlane + s_nop + v_mov
    // per use: 20 of the kernel's 63 v_readlane; 808.0 -> 806.3 vector instructions per source-block, -0.1 % time)
    unsigned my_ack = flags + 8 + 4 * half;
    asm volatile("" : "+v"(my_ack));
    const unsigned his_ack = flags + 8 + 4 * (half ^ 1);
#else
    const unsigned my_ack = flags + 8 + 4 * half, his_ack = flags + 8 + 4 * (half ^ 1);
#endif
    int npub = 0, nseen = 0;  // hand-offs I published / the partner's I consumed (wave-uniform)
    [[maybe_unused]] int steps_done = 0;  // sources I have run the front half of
    bool dead = false;        // a wait timed out (pair_wait)
    auto publish = [&]() {  // my mailbox stores, then the flag: release
        JF_PAIR_RELEASE();
        npub++;
        if (lane == 0 && !JF_EXP_PUBLISH_DROPPED(npub)) lds_flag_write(my_pub, npub);
    };
    auto await_partner = [&]() {  // the partner's next hand-off is in its mailbox: flag, then the mailbox loads: acquire
        nseen++;
        pair_wait(his_pub, nseen, P.err, dead);
        JF_PAIR_ACQUIRE();
    };
    auto consumed = [&]() {  // I am done reading the partner's mailbox (my loads have returned before the flag says so)
        JF_PAIR_RELEASE();
        if (lane == 0) lds_flag_write(my_ack, nseen);
    };
    auto mail_free = [&](int upto) {  // the partner has consumed my hand-offs 1 .. upto: its flag, then my stores
        pair_wait(his_ack, upto, P.err, dead);
        JF_PAIR_ACQUIRE();
    };

    constexpr int B = 64 * NOUT;
    const int G = P.G, SG = P.S / G;
    const int n_units = P.K * SG;
    const int qb = 4 * half;
    const bool special = lane == 0 && half == 0;
    const unsigned lofs = 64u * qb + lane;
    const int n_own = (G - half + 1) / 2, n_his = (G - (half ^ 1) + 1) / 2;  // sources g = 2 j + half / + (half ^ 1)
    // Rounds of units over the persistent pairs.  Units differ in cost -- the first blocks of a call gather their windows
    // the slow way -- and so do pairs: the SIMD's arbiter serves its oldest wave first on a tie, and the last pairs of a
    // workgroup leave the kernel ~12 us after the first (profiles/stamps.py).  JF_UNIT_ZIGZAG = 2: every round the
    // pair -> unit map is rotated by one workgroup, so the expensive units (slots 0..2 of a group's blocks) always go to
    // pairs 0..2 of a workgroup, and to another workgroup every round; 1: every other round in reverse (the expensive
    // units then alternate between the first and the LAST pairs of a workgroup: 1.3 % slower); 0: plain.
    const int n_pairs = P.n_pair_wgs * kPairsPerWg, my_pair = blockIdx.x * kPairsPerWg + pair;
#pragma unroll 1
    for (int round = 0; round * n_pairs < n_units; round++) {
#if JF_UNIT_ZIGZAG == 2
        // rotation by one workgroup per round: the pair-in-workgroup index of a unit's slot stays what it is
        const int unit = round * n_pairs + (my_pair + 8 * round) % n_pairs;
#elif JF_UNIT_ZIGZAG
        const int unit = (round & 1) ? (round + 1) * n_pairs - 1 - my_pair : round * n_pairs + my_pair;
#else
        const int unit = round * n_pairs + my_pair;
#endif
        if (unit >= n_units) continue;
        reload_params();
#if JF_UNIT_ORDER
        const int sg = unit / P.K;
        const int b = unit - sg * P.K;
        const int s0 = sg * G;
#else
        const int b = unit / SG;
        const int sg = unit - b * SG;
        const int s0 = sg * G;
#endif
        // the unit's sources: slots s0 .. s0 + G - 1 of the engine's processing order (jf_engine.cpp: sources that read
        // the same table rows next to each other)
        const ItemDesc *db = P.desc + (size_t)b * P.S;
        const int JF_CONST_AS *ord = as_const(P.order + s0);
        bool any_xfade = false;
        for (int g = 0; g < G; g++) {
            const ItemDesc JF_CONST_AS *d = as_const(db + ord[g]);
            any_xfade = any_xfade || ((d->flags & 2) != 0 && d->n_new > 0);
        }
        mail_free(npub);  // the last unit's final hand-offs used both slots
        JF_EXP_PHASE(6);  // unit start: descriptor scan, waiting for the partner's last reads
        // sums over the unit's sources of X D H_left and X D H_right, bins k = lane + 64 (qb + q), old and new sets
        c2 sLo[4], sRo[4], sLn[4], sRn[4];
#pragma unroll
        for (int q = 0; q < 4; q++) sLo[q] = sRo[q] = sLn[q] = sRn[q] = c2{0.f, 0.f};
        // fetch(xh): see filtered_half.  A source with two filters (its sets do not share rows) fetches once.
        auto accumulate = [&](const ItemDesc *dp, auto &&fetch) {
            const int nn = dp->n_new;
            if (ROWS && (dp->flags & 4)) {
                // both sets are pre-interpolated rows (whole-degree positions): one row each, no weights
                if (!any_xfade)
                    filtered_half_pre<false>(P.htab, lofs, dp->rows_new[0], dp->rows_new[0], fetch, special, sLn, sRn, sLn, sRn);
                else
                    filtered_half_pre<true>(P.htab, lofs, dp->rows_old[0], dp->rows_new[0], fetch, special, sLo, sRo, sLn,
                                            sRn);
            } else if (!any_xfade) {
                filtered_half_nt<false>(nn, P.htab, lofs, dp->rows_new, dp->w_new, dp->w_new, fetch, special, sLn, sRn, sLn,
                                        sRn);
            } else if (dp->flags & 1) {
                // both sets read the same rows (prep_kernel laid them out so): one round of loads
                filtered_half_nt<true>(nn, P.htab, lofs, dp->rows_new, dp->w_old, dp->w_new, fetch, special, sLo, sRo, sLn,
                                       sRn);
            } else {
                float2 keep[4];
                filtered_half_nt<false>(dp->n_old, P.htab, lofs, dp->rows_old, dp->w_old, dp->w_old,
                                        [&](float2 (&xh)[4]) {
                                            fetch(xh);
#pragma unroll
                                            for (int q = 0; q < 4; q++) keep[q] = xh[q];
                                        },
                                        special, sLo, sRo, sLo, sRo);
                filtered_half_nt<false>(nn, P.htab, lofs, dp->rows_new, dp->w_new, dp->w_new,
                                        [&](float2 (&xh)[4]) {
#pragma unroll
                                            for (int q = 0; q < 4; q++) xh[q] = keep[q];
                                        },
                                        special, sLn, sRn, sLn, sRn);
            }
        };
        auto take_partner_source = [&](int jp) {  // his j-th source: my bins of its X D are in his mailbox
            reload_params();
            const ItemDesc dl = load_desc(db + ord[2 * jp + (half ^ 1)]);
            const ItemDesc *dp = &dl;
            if (dp->n_new <= 0) return;  // silent: he published nothing
            accumulate(dp, [&](float2 (&xh)[4]) {
                JF_EXP_PHASE(4);  // partner's source: descriptor, first row requests
                await_partner();
                JF_EXP_PHASE(7);  // ... waiting for his hand-off
                const float2 *m = pmail + (nseen & 1) * kPairMail + lane;
#pragma unroll
                for (int q = 0; q < 4; q++) xh[q] = m[64 * q];
                consumed();
            });
        };
        int jp = 0;
#pragma unroll 1
        for (int j = 0; j < n_own; j++) {
#if JF_PAIR_ROTATE_PRIO
            // The issue arbiter of a SIMD serves its oldest wave first: left alone, waves 0..7 of a workgroup run their
            // first unit in 54 us and waves 8..15 in 70-100 us (profiles/stamps.py), and the SIMDs are half empty while
            // the late ones finish.  A wave lowers its priority with every source it has done (mod 4): whoever is
            // behind is served first, and the waves of a SIMD advance together.
            switch (3 - (steps_done++ & 3)) {  // s_setprio takes an immediate
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
            }
#endif
            reload_params();
            const int src = ord[2 * j + half];
            const ItemDesc dl = load_desc(db + src);
            const ItemDesc *dp = &dl;
            const int item = b * P.S + src;
            // the loads of my source's window first; the partner's previous source is filtered while they are in
            // flight (his hand-off has been waiting for a whole source), then my source's transform and filter
            float2 z[8];
            int count0, L;
            item_gather<NOUT>(P, b, src, opaque(lane), z, count0, L);
            JF_EXP_PHASE(0);  // own source: descriptor and signal records, window requests
#if JF_PAIR_OVERLAP
            if (jp < j && jp < n_his) take_partner_source(jp++);
#endif
            float2 xd[8];
            if (item_finish<NOUT, JF_PAIR_D_EARLY != 0>(P, dp, P.pos + (size_t)item * 5, b, src, buf, s_tw, opaque(lane), z,
                                                        count0, L, xd)) {
                // (requesting my filter's first row loads before this hand-off would hold X D, 16 registers, across
                // them: it spills)
                JF_EXP_PHASE(1);  // window arrival, forward transform, distance factors
                float2 xh[4];
                mail_free(npub - 1);  // the slot of this hand-off was last used two hand-offs ago
                float2 *m = mail + ((npub + 1) & 1) * kPairMail + lane;
                // Which four bins stay and which go to the partner is wave-uniform: a branch, not sixteen selects (3 % of
                // the kernel's time went into selects on the wave's half and on lane 0's packed bins).  The asm comments
                // keep the compiler from merging the two sides into selects again or sinking the stores below the join.
                // (Two copies of the half-filters behind the branch, each reading X D where it is, save the eight moves of
                // the upper half too but spill: 16 B of scratch, slower.)
                if (half) {
                    asm volatile("; upper half keeps bins 4..7" ::: "memory");
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        m[64 * q] = xd[q];
                        xh[q] = xd[4 + q];
                    }
                    asm volatile("; upper half: stored" ::: "memory");
                } else {
                    asm volatile("; lower half keeps bins 0..3" ::: "memory");
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        m[64 * q] = xd[4 + q];
                        xh[q] = xd[q];
                    }
                    asm volatile("; lower half: stored" ::: "memory");
                }
                publish();
                JF_EXP_PHASE(2);  // hand-off (waiting for a free slot, mailbox stores)
                accumulate(dp, [&](float2 (&x)[4]) {
#pragma unroll
                    for (int q = 0; q < 4; q++) x[q] = xh[q];
                });
                JF_EXP_PHASE(3);  // own source's two half-filters
            }
#if !JF_PAIR_OVERLAP
            if (jp < j && jp < n_his) {
                take_partner_source(jp++);
                JF_EXP_PHASE(4);  // the partner's source's two half-filters (after the hand-off arrived)
            }
#endif
        }
#pragma unroll 1
        for (; jp < n_his; jp++) {
            take_partner_source(jp);
            JF_EXP_PHASE(4);
        }

        // ---- the two inverse transforms: wave 0 takes the old sum, wave 1 the new one
        reload_params();
        c2 zko[4], zkn[4], zmo[4], zmn[4];
        ear_sums_to_z(sLo, sRo, special, zko, zmo);
        ear_sums_to_z(sLn, sRn, special, zkn, zmn);
        const bool give = half == 0 || any_xfade;  // my bins of the sum the partner inverts
        const bool take = half == 1 || any_xfade;  // I invert a sum
        if (give) {
            mail_free(npub);  // both slots: Z[k] in the first, Z[N-k] in the second
#pragma unroll
            for (int q = 0; q < 4; q++) {
                mail[64 * q + lane] = f2_of(half == 0 ? zkn[q] : zko[q]);
                mail[kPairMail + 64 * q + lane] = f2_of(half == 0 ? zmn[q] : zmo[q]);
            }
        }
        if (take) {  // my own Z[N-k] go through LDS as well: the inverse needs them from lane 64 - lane
#pragma unroll
            for (int q = 0; q < 4; q++) buf[64 * q + lane] = f2_of(half == 0 ? zmo[q] : zmn[q]);
        }
        if (give) publish();
        float2 fr[NOUT];
        if (take) {
            await_partner();
            const int ln = opaque(lane);  // or the addresses below are worked out at kernel entry and kept (spilled)
            float2 v[16];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float2 theirs = pmail[64 * q + ln];
                const float2 own = f2_of(half == 0 ? zko[q] : zkn[q]);
                v[q] = half == 0 ? own : theirs;
                v[4 + q] = half == 0 ? theirs : own;
            }
            // Z[lane + 64 r], r = 8..15 = Z[N-k] of bin 7 - j on lane 64 - lane; lane 0: its own bin (8 - j) & 7
            // (see mirror8).  Bins 0..3 are wave 0's, 4..7 wave 1's: mine in my work space, his in his mailbox.
            const float2 *lo = half == 0 ? buf : pmail + kPairMail;
            const float2 *hi = half == 0 ? pmail + kPairMail : buf;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int qn = 7 - j, q0 = (8 - j) & 7;  // bin read by lanes 1..63 / by lane 0
                const float2 *pn = (qn < 4 ? lo : hi) + 64 * (qn & 3) + (64 - ln);
                const float2 *p0 = (q0 < 4 ? lo : hi) + 64 * (q0 & 3);
                v[8 + j] = *(ln == 0 ? p0 : pn);
            }
            consumed();
            ifft1024_lastq_wave<NOUT, true>(v, fr, buf, s_tw, opaque(lane));
        }
        // per-lane frame numbers, fade weights and output offsets from an opaque lane index: worked out here, once per
        // unit, instead of at kernel entry and held (or spilled) through the source loop
        const int lf = opaque(lane), af = lf & 3, fi = lf >> 2;
        if (any_xfade) {
            if (half == 0) {
                mail_free(npub);
#pragma unroll
                for (int j = 0; j < NOUT; j++) mail[64 * j + lf] = fr[j];
                publish();
            } else {
                await_partner();
#pragma unroll
                for (int j = 0; j < NOUT; j++) {
                    const float2 old = pmail[64 * j + lf];
                    // kernels.cu:132-137
                    const int n_out = fi + 16 * (NOUT * af + j);  // frame inside the block
                    const float fn = (float)n_out / ((float)B - 1.0f);
                    fr[j] = make_float2(old.x * (1.0f - fn) + fr[j].x * fn, old.y * (1.0f - fn) + fr[j].y * fn);
                }
                consumed();
            }
        }
        if (half == 1) {
            float2 *out = reinterpret_cast<float2 *>(P.partial) + ((size_t)b * SG + sg) * B;
#pragma unroll
            for (int j = 0; j < NOUT; j++) out[fi + 16 * (NOUT * af + j)] = fr[j];
        }
        JF_EXP_STAMP_ROUND(round);
        JF_EXP_PHASE(5);  // end of the unit: exchange of the sums, inverse transform, crossfade, store
    }
    JF_EXP_STAMP_END();
    JF_EXP_PHASE_END(P);
}

// ---------------------------------------------------------------- mixing --
// Audio.cu:109-110: out[i] += source->intermediate[i], sources in index order.
// Deterministic: 16 groups of consecutive sources are each summed in source order by
// their own wave, then the 16 group sums are added in group order.  One workgroup
// = 64 consecutive output floats of one block x 16 source groups.
constexpr int kMixGroups = 16;
JF_DEV void mix_body(const float *__restrict__ partial, float *__restrict__ mix, int S, int blk /* 2B */, int wg,
                     float (&red)[kMixGroups][64]) {
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int chunks = blk / 64;
    const int b = wg / chunks, n = (wg - b * chunks) * 64 + lane;
    const int per = (S + kMixGroups - 1) / kMixGroups;
    const int s0 = grp * per, s1 = min(S, s0 + per);
    const float *p = partial + (size_t)b * S * blk + n;
    float acc = 0.0f;
    for (int s = s0; s < s1; s++) acc += p[(size_t)s * blk];
    red[grp][lane] = acc;
    __syncthreads();
    if (grp == 0) {
        float t = red[0][lane];
#pragma unroll
        for (int g = 1; g < kMixGroups; g++) t += red[g][lane];
        mix[(size_t)b * blk + n] = t;
    }
}
__global__ __launch_bounds__(64 * kMixGroups) void mix_kernel(const float *__restrict__ partial,
                                                              float *__restrict__ mix, int S, int K,
                                                              int blk /* 2B */) {
    __shared__ float red[kMixGroups][64];
    mix_body(partial, mix, S, blk, blockIdx.x, red);
}

// The same sum in the same association for the pair kernel's few partial blocks per audio block (16, 32 or 64 groups of
// sources): one thread per output float, all of its 16 PER loads in flight at once, no LDS round.  Both forms are a few
// memory latencies long (4.7 us against mix_kernel's 4.2 under rocprofv3, where every launch is drained); back to back
// behind the fused kernel this one makes a step 0.5 us shorter (7.1 against 7.6 us outside the fused launch, four runs
// each).
template <int PER>
__global__ __launch_bounds__(256) void mix_few_kernel(const float *__restrict__ partial, float *__restrict__ mix, int blk,
                                                      int total /* K * blk */) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int b = t / blk, n = t - b * blk;
    const float *p = partial + (size_t)b * (kMixGroups * PER) * blk + n;
    float v[kMixGroups][PER];
#pragma unroll
    for (int g = 0; g < kMixGroups; g++)
#pragma unroll
        for (int j = 0; j < PER; j++) v[g][j] = p[(size_t)(g * PER + j) * blk];
    float tot = 0.0f;
#pragma unroll
    for (int g = 0; g < kMixGroups; g++) {
        float acc = 0.0f;  // as mix_body: every group's sum starts from 0
#pragma unroll
        for (int j = 0; j < PER; j++) acc += v[g][j];
        tot = g == 0 ? acc : tot + acc;
    }
    mix[t] = tot;
}

// ----------------------------------------------- indices and weights (a2,a3)
// SoundSource.cu:65-105 and hrtf_signals.cu:20-51, float32 exactly as written
// (no contraction), the nearest-azimuth search done locally instead of over the
// whole ring.
#pragma clang fp contract(off)
__device__ static const int d_elev_pos[kNumElev] = {-40, -30, -20, -10, 0, 10, 20, 30, 40, 50, 60, 70, 80, 90};

JF_DEV int dev_pick_azi(const RingTable &rt, int ring, float obj_azi) {
    const float inc = rt.inc[ring];
    const int n = rt.offset[ring + 1] - rt.offset[ring];
    obj_azi = roundf(obj_azi);
    int i0 = (int)floorf(obj_azi / inc) - 1;
    if (i0 > n - 4) i0 = n - 4;
    if (i0 < 0) i0 = 0;
    float dmin = 1e37f;
    int best = 0;
    for (int i = i0; i < i0 + 4 && i < n; i++) {
        float d = obj_azi - i * inc;
        d = d > 0 ? d : -d;
        if (d < dmin) {
            dmin = d;
            best = i;
        }
    }
    return rt.offset[ring] + best;
}

// the same for an integer azimuth: one look-up in the table that search filled (jf_engine.cpp), the search itself outside it
JF_DEV int dev_pick_int(const RingTable &rt, int ring, int th) {
    if (rt.pick != nullptr && (unsigned)th < (unsigned)kPickAzi) return rt.pick[ring * kPickAzi + th];
    return dev_pick_azi(rt, ring, (float)th);
}

// The nearest measurement of a grid that is not the reference's (jf_engine_create_grid; twin of host_grid_pick): the ring
// whose elevation is nearest (the lower one on a tie), on it the azimuth nearest on the circle.
JF_DEV int dev_grid_pick(const RingTable &rt, float ele, float azi) {
    float dmin = 1e37f;
    int ring = 0;
    for (int r = 0; r < rt.n_rings; r++) {
        float d = ele - rt.ele[r];
        d = d > 0 ? d : -d;
        if (d < dmin) {
            dmin = d;
            ring = r;
        }
    }
    const int n = rt.offset[ring + 1] - rt.offset[ring];
    float a = azi - 360.0f * floorf(azi / 360.0f);
    if (!(a < 360.0f)) a = 0.0f;
    int i = (int)floorf(a / rt.inc[ring] + 0.5f);
    if (i >= n) i = 0;  // nearer to 360 = the ring's first entry
    return rt.offset[ring] + i;
}

// hrtf_signals.cu:20-51 in full: nearest elevation ring, then nearest azimuth on it
JF_DEV int dev_pick_hrtf(const RingTable &rt, float obj_ele, float obj_azi) {
    if (!rt.kemar) return dev_grid_pick(rt, obj_ele, obj_azi);
    obj_ele = roundf(obj_ele / 10) * 10;
    float dmin = 1e37f;
    int ring = 0;
    for (int e = 0; e < kNumElev; e++) {
        float d = obj_ele - d_elev_pos[e];
        d = d > 0 ? d : -d;
        if (d < dmin) {
            dmin = d;
            ring = e;
        }
    }
    return dev_pick_azi(rt, ring, obj_azi);
}

// returns number of terms (1, 2, 4) or 0 when the elevation ring does not exist
// The corrected rule behind JF_FLAG_CORRECTED_INTERPOLATION (not in the reference):
// true floor of the elevation, azimuth folded into [0, 360) with a ring's last interval wrapping to its first
// entry, float azimuths (a ring's two weights sum to 1), elevations below the lowest ring clamped to it.
// Same index order and weight meaning as the reference's rule.  Float32 step by step as in the oracles.
// In its general form (any grid of rings, include/jefferson.h: jf_hrtf_grid): the ring pair is the one whose elevations
// enclose the position (elevations outside the grid clamped to its first / last ring), the elevation weight is linear between
// them.  For the reference's grid the closed form below gives the same ring, the same phi0 and the same divisor 10 -- bit for
// bit the same indices and weights (tests/test_abi.py compares the two on the host, tests/test_gpu_grid.py on the GPU).
JF_DEV bool dev_interp_corrected(const RingTable &rt, float ele, float azi, int h[4], float om[6]) {
    if (!(ele <= 90.0f) || !(ele > -1.0e6f) || !(azi > -1.0e6f && azi < 1.0e6f)) return false;
    float a = azi - 360.0f * floorf(azi / 360.0f);
    if (!(a < 360.0f)) a = 0.0f;
    int r0;
    float phi0, span;
    if (rt.kemar) {
        if (ele < -40.0f) ele = -40.0f;
        const float q = floorf(ele / 10.0f);
        phi0 = 10.0f * q;
        r0 = (int)q + 4;
        span = 10.0f;
    } else {
        const int last = rt.n_rings - 1;
        if (ele < rt.ele[0]) ele = rt.ele[0];
        if (ele > rt.ele[last]) ele = rt.ele[last];
        r0 = 0;
        for (int r = 1; r <= last; r++) r0 = rt.ele[r] <= ele ? r : r0;
        phi0 = rt.ele[r0];
        span = rt.ele[r0 < last ? r0 + 1 : r0] - phi0;
    }
    const bool on_ring = ele == phi0;
    const int ring[2] = {r0, on_ring ? r0 : r0 + 1};
    const float omE = on_ring ? 0.0f : (ele - phi0) / span;
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int r = ring[j];
        const float d = rt.inc[r];
        const int n = rt.offset[r + 1] - rt.offset[r];
        int i0 = (int)floorf(a / d);
        if (i0 > n - 1) i0 = n - 1;
        float wa = (a - (float)i0 * d) / d;
        if (wa < 0.0f) wa = 0.0f;
        if (wa > 1.0f) wa = 1.0f;
        if (n == 1) wa = 0.0f;
        int i1 = i0 + 1 == n ? 0 : i0 + 1;
        if (wa == 0.0f) i1 = i0;
        h[2 * j] = rt.offset[r] + i0;
        h[2 * j + 1] = rt.offset[r] + i1;
        om[2 * j] = wa;
        om[2 * j + 1] = 1.0f - wa;
    }
    om[4] = omE;
    om[5] = 1.0f - omE;
    return true;
}

JF_DEV int dev_flatten_terms(int h0, int h1, int h2, int h3, float omegaA, float omegaB, float omegaC, float omegaD,
                             float omegaE, float omegaF, int rows[4], float w[4]);

JF_DEV int dev_interp_terms(const RingTable &rt, float ele, float azi, int rows[4], float w[4], bool corrected = false) {
    if (corrected || !rt.kemar) {  // (the reference's rule is a rule of the reference's grid)
        int h[4];
        float om[6];
        if (!dev_interp_corrected(rt, ele, azi, h, om)) return 0;
        return dev_flatten_terms(h[0], h[1], h[2], h[3], om[0], om[1], om[2], om[3], om[4], om[5], rows, w);
    }
    if (!(ele > -50.0f && ele < 91.0f) || !(azi > -1.0e6f && azi < 1.0e6f)) return 0;  // as host_interpolation (jf_host.cpp)
    const int phi0 = (int)(ele) / 10 * 10;
    const int phi1 = (int)(ele + 9) / 10 * 10;
    const float omegaE = (ele - phi0) / 10.0f;
    const float omegaF = (phi1 - ele) / 10.0f;
    // the rings with these elevations (multiples of 10 by construction; -40 .. 90 exist)
    const int r0 = (phi0 >= -40 && phi0 <= 90) ? (phi0 + 40) / 10 : -1;
    const int r1 = (phi1 >= -40 && phi1 <= 90) ? (phi1 + 40) / 10 : -1;
    if (r0 < 0 || r1 < 0) return 0;
    const float dt1 = rt.inc[r0], dt2 = rt.inc[r1];
    const int th0 = (int)((int)(azi / dt1) * dt1);
    const int th1 = (int)((int)((azi + dt1 - 1) / dt1) * dt1);
    const int th2 = (int)((int)(azi / dt2) * dt2);
    const int th3 = (int)((int)((azi + dt2 - 1) / dt2) * dt2);
    const float omegaA = (azi - th0) / dt1;
    const float omegaB = (th1 - azi) / dt1;
    const float omegaC = (azi - th2) / dt2;
    const float omegaD = (th3 - azi) / dt2;
    const int h0 = dev_pick_int(rt, r0, th0);
    const int h1 = dev_pick_int(rt, r0, th1);
    const int h2 = dev_pick_int(rt, r1, th2);
    const int h3 = dev_pick_int(rt, r1, th3);
    return dev_flatten_terms(h0, h1, h2, h3, omegaA, omegaB, omegaC, omegaD, omegaE, omegaF, rows, w);
}

// GPUSoundSource.cu:301-316: the case by index equality, flattened to <= 4 (row, weight) terms.  Written with selects, not
// branches: the compiler merges the branches' stores `w[i] = ..` into one store at a run-time index, which puts the
// arrays into scratch memory (24 bytes that every kernel with this code inlined then carries).  The same values: the four
// products are formed whether or not case 4 uses them.
JF_DEV int dev_flatten_terms(int h0, int h1, int h2, int h3, float omegaA, float omegaB, float omegaC, float omegaD,
                             float omegaE, float omegaF, int rows[4], float w[4]) {
    const bool c1 = h0 == h1 && h1 == h2 && h2 == h3;    // one row
    const bool c2 = !c1 && h0 == h2 && h1 == h3;          // elevation on a ring: two azimuths
    const bool c3 = !c1 && !c2 && h0 == h1 && h0 != h2;   // azimuth on the grid: two rings
    const bool c4 = !c1 && !c2 && !c3;
    const float fb = omegaF * omegaB, fa = omegaF * omegaA, ed = omegaE * omegaD, ec = omegaE * omegaC;
    rows[0] = h0;
    w[0] = c1 ? 1.0f : c2 ? omegaB : c3 ? omegaF : fb;
    rows[1] = c1 ? h0 : c3 ? h2 : h1;
    w[1] = c1 ? 0.0f : c2 ? omegaA : c3 ? omegaE : fa;
    rows[2] = c4 ? h2 : h0;
    w[2] = c4 ? ed : 0.0f;
    rows[3] = c4 ? h3 : h0;
    w[3] = c4 ? ec : 0.0f;
    return c1 ? 1 : c4 ? 4 : 2;
}

// Descriptor of one work item from its latched position record and the position of the block
// before (GPUSoundSource.cu:81-90 and :325-335).
JF_DEV void make_desc(const RingTable &rt, int mode, const float *p /* ele, azi, x, y, z */, float old_ele,
                      float old_azi, ItemDesc &d) {
    const float ele = p[0], azi = p[1];
    const bool corrected = (mode & 2) != 0;  // mode: bit 0 FD_BASIC, bit 1 the corrected index/weight rule
    if (mode & 1) {
        // *_FD_BASIC (CPUSoundSource.cpp:50-52,113-142): the nearest table row, weight 1, no
        // distance factor (D = 1), no crossfade
        const bool ok = (ele > -1.0e6f && ele < 1.0e6f) && (azi > -1.0e6f && azi < 1.0e6f);
        const int row = ok ? dev_pick_hrtf(rt, ele, azi) : 0;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            d.rows_new[t] = d.rows_old[t] = row;
            d.w_new[t] = t == 0 ? 1.0f : 0.0f;
            d.w_old[t] = 0.0f;
        }
        d.n_new = ok ? 1 : 0;
        d.n_old = 0;
        d.c_fix = 0;
        d.inv_frac = 1.0f;
        d.flags = 0;
        return;
    }
    d.n_new = dev_interp_terms(rt, ele, azi, d.rows_new, d.w_new, corrected);
    d.n_old = 0;
    // GPUSoundSource.cu:331-335
    if (old_azi != azi || old_ele != ele) {
        d.n_old = dev_interp_terms(rt, old_ele, old_azi, d.rows_old, d.w_old, corrected);
        if (d.n_old == 0) d.n_new = 0;
    } else {
#pragma unroll
        for (int t = 0; t < 4; t++) {
            d.rows_old[t] = 0;
            d.w_old[t] = 0.0f;
        }
    }
    // GPUSoundSource.cu:81-90
    const float x = p[2], y = p[3], z = p[4];
    float r = sqrtf(x * x + y * y + z * z);
    r /= 5;
    const float fsvs = (float)(44100.0 / 343.0);
    const float frac = 1 + fsvs * (float)((double)r * (double)r);
    {
        // phase step per bin in turns, as a 64-bit fraction (double keeps 52+ fractional bits here)
        double c = (double)fsvs * (double)r * (1.0 / 513.0);  // 1e-16 relative: far below the 2^-32 turn the phase word keeps
        c -= floor(c);
        d.c_fix = (unsigned long long)(c * 18446744073709551616.0);
    }
    d.inv_frac = 1.0f / frac;
    if (!(frac >= 1.0f) || !(frac < 3.0e38f)) d.n_new = 0;  // NaN / inf coordinates
    d.flags = 0;
}

// Two adjacent lanes per item: the even one does the new position's rule and the distance part, the odd one
// the old position's rule (the kernel is a short dependent chain per thread at one wave per SIMD: halving
// the chain halves its time).  Same arithmetic as make_desc, which the real-time kernel uses.
// st == nullptr: the window continues the trajectory (the block before its first one is at pos - 5 S), as it does for a
// window prepared ahead of its run (mix_prep_kernel); else the old position of block 0 is the state the last run left.
// stage: 32 records per wave of the workgroup in LDS.  The lanes build their records there and the wave then writes its
// 32 consecutive records (2.8 KB) with six coalesced stores; written straight from the lanes, every store instruction
// scattered 64 pieces of 16 B over 32 records 88 B apart, and those stores, not the arithmetic, were most of the kernel's
// time (13 us for 131 072 items against 5 us for the chain of one wave).
JF_DEV void prep_body(const RingTable &rt, int mode, const float *__restrict__ pos, const SrcState *__restrict__ st,
                      ItemDesc *__restrict__ desc, int S, int K, int canon, int tid, ItemDesc *stage) {
    const int item = tid >> 1;
    const bool old_half = tid & 1;
    const bool live = item < S * K;  // both lanes of a pair agree; no early return before the shuffles
    const int it = live ? item : 0;
    const int b = it / S, s = it - b * S;
    const float *p = pos + (size_t)it * 5;
    const float ele = p[0], azi = p[1];
    float old_ele, old_azi;
    if (b == 0 && st != nullptr) {
        old_ele = st[s].old_ele;
        old_azi = st[s].old_azi;
    } else {
        old_ele = p[-5 * S];
        old_azi = p[-5 * S + 1];
    }
    const int lane = (int)threadIdx.x & 63;
    ItemDesc *wave_stage = stage + ((int)threadIdx.x >> 6) * 32;
    ItemDesc &d = wave_stage[lane >> 1];
    int rows[4] = {0, 0, 0, 0};
    float w[4] = {0.f, 0.f, 0.f, 0.f};
    int n = 0;
    const bool moved = old_azi != azi || old_ele != ele;  // GPUSoundSource.cu:331-335
    const bool corrected = (mode & 2) != 0;
    bool pre = false;  // this lane's set is a pre-interpolated row
    if (mode & 1) {
        // *_FD_BASIC (CPUSoundSource.cpp:50-52,113-142): the nearest table row, weight 1, no
        // distance factor (D = 1), no crossfade
        const bool ok = (ele > -1.0e6f && ele < 1.0e6f) && (azi > -1.0e6f && azi < 1.0e6f);
        const int row = ok ? dev_pick_hrtf(rt, ele, azi) : 0;
        rows[0] = rows[1] = rows[2] = rows[3] = row;
        if (!old_half) {
            w[0] = 1.0f;
            n = ok ? 1 : 0;
        }
    } else {
        // ONE call for both lanes of a pair, each with its own position: as two calls under `old_half` the wave ran the
        // rule twice, once with the even and once with the odd lanes masked off
        const float e_in = old_half ? old_ele : ele, a_in = old_half ? old_azi : azi;
        // A whole-degree position inside the pre-interpolated part of the table (jf_device.h: htab) is ONE row with weight 1:
        // the row holds the weighted sum the rule below would ask for, formed by the same operations in the same order.
        pre = canon && (mode & kModeInterpRows) != 0 && e_in >= (float)kInterpEleMin &&
              e_in <= (float)kInterpEleMax && a_in >= 0.0f && a_in < (float)kInterpAzi && floorf(e_in) == e_in &&
              floorf(a_in) == a_in;
        int n_in;
        if (pre) {
            rows[0] = rows[1] = rows[2] = rows[3] = rt.n_rows + ((int)e_in - kInterpEleMin) * kInterpAzi + (int)a_in;
            w[0] = 1.0f;
            n_in = 1;
        } else {
            n_in = dev_interp_terms(rt, e_in, a_in, rows, w, corrected);
        }
        const bool used = !old_half || moved;  // a source that did not move has no old set
        n = used ? n_in : 0;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            rows[t] = used ? rows[t] : 0;
            w[t] = used ? w[t] : 0.0f;
        }
    }
    // the other half's result (the even lane needs the old set for the pair-kernel layout)
    const int n_other = __shfl_xor(n, 1);
    const bool pre_other = __shfl_xor((int)pre, 1) != 0;
    int orow[4];
    float ow[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        orow[t] = __shfl_xor(rows[t], 1);
        ow[t] = __shfl_xor(w[t], 1);
    }
    if (live && old_half && !canon) {  // (in the pair-kernel layout the even lane writes the whole record)
#pragma unroll
        for (int t = 0; t < 4; t++) {
            d.rows_old[t] = rows[t];
            d.w_old[t] = w[t];
        }
        d.n_old = n;
    }
    if (live && !old_half) {
    int flags = 0;
    if (mode & 1) {
        d.c_fix = 0;
        d.inv_frac = 1.0f;
    } else {
        if (moved && n_other == 0) n = 0;  // the old position is not interpolable
        // GPUSoundSource.cu:81-90
        const float x = p[2], y = p[3], z = p[4];
        float r = sqrtf(x * x + y * y + z * z);
        r /= 5;
        const float fsvs = (float)(44100.0 / 343.0);
        const float frac = 1 + fsvs * (float)((double)r * (double)r);
        {
            // phase step per bin in turns, as a 64-bit fraction (double keeps 52+ fractional bits here)
            double c = (double)fsvs * (double)r * (1.0 / 513.0);  // 1e-16 relative: far below the 2^-32 turn the phase word keeps
            c -= floor(c);
            d.c_fix = (unsigned long long)(c * 18446744073709551616.0);
        }
        d.inv_frac = 1.0f / frac;
        if (!(frac >= 1.0f) || !(frac < 3.0e38f)) n = 0;  // NaN / inf coordinates
    }
    if (canon) {
        // Layout for fused_pair_kernel.  A source that did not move carries its new set as its old set (inside a
        // unit that crossfades it goes through both sums).  If the rows of one set are, in order, among the rows of
        // the other -- a step inside one grid cell, onto a grid line or off one -- both sets are written on the
        // larger set's rows, with weight 0 where a set does not use a row: the kernel then loads each row once.
        // A term with weight 0 adds +-0, and the other terms keep their order, so each set's weighted sum is
        // bit-identical to the sum over its own rows.
        const bool xf = moved && !(mode & 1) && n > 0;
        flags = xf ? 2 : 0;
        // both sets are whole rows (a source that did not move carries its new set as its old set): the kernel's short path.
        // One pre-interpolated set beside an ordinary one goes through the general path (a row with weight 1).
        if (pre && n > 0 && (!xf || pre_other)) flags |= 4;
        int n_o = xf ? n_other : n;
        if (!xf) {
#pragma unroll
            for (int t = 0; t < 4; t++) {
                orow[t] = rows[t];
                ow[t] = w[t];
            }
        }
        const bool new_is_big = n >= n_o;
        int big_rows[4], small_rows[4];
        float big_w[4], small_w[4], ex_w[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; t++) {
            big_rows[t] = new_is_big ? rows[t] : orow[t];
            big_w[t] = new_is_big ? w[t] : ow[t];
            small_rows[t] = new_is_big ? orow[t] : rows[t];
            small_w[t] = new_is_big ? ow[t] : w[t];
        }
        const int n_big = new_is_big ? n : n_o, n_small = new_is_big ? n_o : n;
        bool share = n > 0;
        int from = 0;
        // fully unrolled with predicates (run-time array indices would put the arrays into scratch memory)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const bool live_i = i < n_small && share;
            // the first row of the big set, from `from` on, that is the small set's row i takes its weight (written without
            // an index variable: the compiler turns `ex_w[k] = k == at ? .. : ex_w[k]` back into a store at a run-time index)
            bool found = false;
            int next_from = from;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const bool hit = !found && k >= from && k < n_big && big_rows[k] == small_rows[i];
                ex_w[k] = (live_i && hit) ? small_w[i] : ex_w[k];
                next_from = hit ? k + 1 : next_from;
                found = found || hit;
            }
            if (live_i) {
                if (!found) share = false;
                else from = next_from;
            }
        }
        if (share) {
            flags |= 1;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                rows[t] = orow[t] = big_rows[t];
                w[t] = new_is_big ? big_w[t] : ex_w[t];
                ow[t] = new_is_big ? ex_w[t] : big_w[t];
            }
            n = n_o = n_big;
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
            d.rows_old[t] = orow[t];
            d.w_old[t] = ow[t];
        }
        d.n_old = n > 0 ? n_o : 0;
    }
#pragma unroll
    for (int t = 0; t < 4; t++) {
        d.rows_new[t] = rows[t];
        d.w_new[t] = w[t];
    }
    d.flags = flags;
    d.n_new = n;
    }
    // the wave's records, as they lie in LDS, to desc[first item of the wave ...]
    JF_WAVE_LDS_SYNC();
    const int item0 = (tid - lane) >> 1;
    const int n_rec = min(32, S * K - item0);  // <= 0 for a wave past the end
    static_assert(sizeof(ItemDesc) % 8 == 0, "copied as 8-byte pieces");
    constexpr int kPieces = (int)sizeof(ItemDesc) / 8;
    const float2 *src = reinterpret_cast<const float2 *>(wave_stage);
    float2 *dst = reinterpret_cast<float2 *>(desc + item0);
    for (int c = lane; c < n_rec * kPieces; c += 64) dst[c] = src[c];
}

constexpr int kPrepThreads = 256;
__global__ __launch_bounds__(kPrepThreads) void prep_kernel(const RingTable rt, int mode, const float *__restrict__ pos,
                                                            const SrcState *__restrict__ st, ItemDesc *__restrict__ desc,
                                                            int S, int K, int canon) {
    __shared__ ItemDesc stage[kPrepThreads / 2];
    prep_body(rt, mode, pos, st, desc, S, K, canon, blockIdx.x * kPrepThreads + threadIdx.x, stage);
}

// mix_kernel of one run and prep_kernel of the next window of the trajectory in ONE launch (the first workgroups
// prepare, the other n_mix mix): both are short chains at low occupancy, and one after the other they leave the GPU idle
// twice per run.  The engine launches this form when the run came from jf_batch_run and the window that follows it
// lies inside the uploaded trajectory; the descriptors go to the engine's other descriptor buffer and are used if the
// next run asks for exactly that window (jf_engine.cpp: run_blocks).
__global__ __launch_bounds__(64 * kMixGroups) void mix_prep_kernel(const float *__restrict__ partial,
                                                                   float *__restrict__ mix, int S_groups, int blk,
                                                                   int n_mix, const RingTable rt, int mode,
                                                                   const float *__restrict__ pos,
                                                                   ItemDesc *__restrict__ desc, int S, int K, int canon) {
    __shared__ float red[kMixGroups][64];
    __shared__ ItemDesc stage[64 * kMixGroups / 2];
    // the preparing workgroups first: theirs is the long chain (12 us against 2.7 us per round of mixing workgroups),
    // and dispatched last they would start when the mix is nearly over
    const int n_prep = (int)gridDim.x - n_mix;
    if ((int)blockIdx.x < n_prep)
        prep_body(rt, mode, pos, nullptr, desc, S, K, canon, (int)blockIdx.x * (64 * kMixGroups) + (int)threadIdx.x, stage);
    else
        mix_body(partial, mix, S_groups, blk, (int)blockIdx.x - n_prep, red);
}

__global__ void interp_debug_kernel(const RingTable rt, const float *ele, const float *azi, int *rows,
                                    float *w, int *nt, int n, int corrected) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    int r4[4] = {0, 0, 0, 0};
    float w4[4] = {0, 0, 0, 0};
    nt[g] = dev_interp_terms(rt, ele[g], azi[g], r4, w4, corrected != 0);
    for (int t = 0; t < 4; t++) {
        rows[4 * g + t] = r4[t];
        w[4 * g + t] = w4[t];
    }
}
#pragma clang fp contract(fast)

// ------------------------------------------------------- real-time kernel --
// One audio block for up to a few hundred sources in ONE launch (the per-block call of the reference's
// audio callback): workgroup g of n, wave w takes sources RTW g + w, + RTW n, ...; lane 0 builds the
// descriptor in LDS (no prep launch), the wave spatialises, the waves' stereo blocks are summed
// in wave order through LDS (no mix launch) and workgroup g writes its sum to out + g B.  pos and out
// may be host-mapped pinned memory, so a block costs one launch and one synchronisation, no copies;
// the caller adds the n partial blocks (n = 1 for up to RTW sources).
// RTW = waves (= sources per turn) of a workgroup: 8 or 16.  Sixteen waves share a compute unit's four SIMDs four
// to one and a block takes 3-4 us longer than with eight (two to one; four waves, one to one, gain nothing more) -- but every
// workgroup's partial block crosses PCIe, and from ~1000 sources on the fewer, larger workgroups win
// (profiles/r04/rt_waves.md: 16 .. 128 sources 22.5-24.4 -> 18.4-21.1 us, 512 equal, 1024 30.9 against 33.6).
constexpr int kRtWavesFew = 8, kRtWavesMany = 16, kRtFewMaxSources = 512;
#include "jf_rv_small.h"

// RV: the reverb stage's head runs inside this kernel (jf_rv_small.h: rv_head_wave): the wave of source s first takes the
// block through the P partitions of R -- the head of a non-uniformly partitioned response -- and leaves it in the wet ring,
// then spatialises it: ONE launch per audio block with the reverb on (round 4: the head kernel, 8 us, a launch gap, this kernel).
template <int NOUT, int RTW, bool RV>
// done (may be null): host-mapped words, one per workgroup; workgroup g stores `seq` into done[g] once its block lies in
// `out` -- the host then polls these words instead of synchronising the stream (the runtime's completion path costs more
// than the kernel's arithmetic at one source).
__global__ __launch_bounds__(64 * RTW) void rt_block_kernel(const FusedParams P, const RingTable rt,
                                                            const float *__restrict__ pos,
                                                            float2 *__restrict__ out, int *__restrict__ done, int seq,
                                                            const ReverbParams R) {
    constexpr int kRtWaves = RTW;
    __shared__ float2 s_tw[kTwPack];
    __shared__ float2 s_buf[kRtWaves * kWaveLds];
    __shared__ ItemDesc s_desc[kRtWaves];
    __shared__ float2 s_tw1024[RV ? 1024 : 1];  // exp(2 pi i j / 1024): the head's small transforms read it in every pass
    constexpr int B = 64 * NOUT;
    static_assert(!RV || (4 * B <= kWaveLds && (B == 64 || B == 128 || B == 256)), "rv_head_wave's LDS; the reverb's block sizes");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int j = tid; j < kTwPack; j += 64 * kRtWaves) s_tw[j] = P.tw[j];
    if (RV)
        for (int j = tid; j < 1024; j += 64 * kRtWaves) s_tw1024[j] = R.tw[j];
    // the first source's descriptor before the barrier: its position comes over PCIe (host-mapped memory) and the index/
    // weight rule is a long chain in one lane -- both overlap with the twiddle loads of the other lanes
    const int s_first = blockIdx.x * kRtWaves + wave;
    if (lane == 0 && s_first < P.S)
        make_desc(rt, P.mode, pos + 5 * s_first, P.st_in[s_first].old_ele, P.st_in[s_first].old_azi, s_desc[wave]);
    __syncthreads();
    float2 *buf = s_buf + wave * kWaveLds;
    const int a = lane & 3, i = lane >> 2;
    float2 acc[NOUT];
#pragma unroll
    for (int j = 0; j < NOUT; j++) acc[j] = make_float2(0.f, 0.f);
#pragma unroll 1
    for (int s = s_first; s < P.S; s += gridDim.x * kRtWaves) {
        const float *p = pos + 5 * s;
        if (lane == 0 && s != s_first) make_desc(rt, P.mode, p, P.st_in[s].old_ele, P.st_in[s].old_azi, s_desc[wave]);
        JF_WAVE_LDS_SYNC();
        if constexpr (RV) {
            if constexpr (B == 64 || B == 128 || B == 256) rv_head_wave<B>(R, s, buf, s_tw1024, lane);
            // the block lies in the wet ring (this wave's own stores, acknowledged) before the window below is gathered from it
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            JF_WAVE_LDS_SYNC();
        }
        spatialise_item<NOUT>(P, &s_desc[wave], p, 0, s, buf, s_tw, lane, acc);
        JF_WAVE_LDS_SYNC();
    }
#pragma unroll
    for (int j = 0; j < NOUT; j++) buf[i + 16 * (NOUT * a + j)] = acc[j];
    __syncthreads();
    for (int n = tid; n < B; n += 64 * kRtWaves) {
        float2 t = s_buf[n];
#pragma unroll
        for (int w = 1; w < kRtWaves; w++) t = cadd(t, s_buf[w * kWaveLds + n]);
        out[(size_t)blockIdx.x * B + n] = t;
    }
    if (done != nullptr) {
        // Every storing wave waits for its own stores of the block (a barrier alone does not: the compiler puts no
        // s_waitcnt vmcnt(0) in front of s_barrier, and the word below could overtake another wave's stores), then the
        // barrier, then ONE lane releases at system scope and stores the word.  The second wait is spelled out because the
        // compiler may drop the one behind the write-back when it thinks nothing is outstanding.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(done + blockIdx.x, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ------------------------------------------------ table build and FFT tap --
// hrtf_signals.cu:107-153: unnormalised r2c of every zero-padded HRIR, written
// in the interleaved device layout.  One wave per table row (both ears).
__global__ __launch_bounds__(64) void table_build_kernel(const float *__restrict__ hrir, int taps,
                                                        const float2 *__restrict__ twg,
                                                        float4 *__restrict__ htab) {
    __shared__ float2 s_tw[kTwPack];
    __shared__ float2 s_buf[576];
    const int lane = threadIdx.x;
    for (int j = lane; j < kTwPack; j += 64) s_tw[j] = twg[j];
    __syncthreads();
    const int row = blockIdx.x;
    float2 Xe[2][8];
#pragma unroll
    for (int ear = 0; ear < 2; ear++) {
        const float *h = hrir + ((size_t)row * 2 + ear) * taps;
        float2 z[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int n = 2 * (lane + 64 * r);
            z[r] = make_float2(n < taps ? h[n] : 0.0f, n + 1 < taps ? h[n + 1] : 0.0f);
        }
        rfft1024_wave(z, Xe[ear], s_buf, s_tw, lane);
        JF_WAVE_LDS_SYNC();
    }
#pragma unroll
    for (int q = 0; q < 8; q++)
        htab[(size_t)row * 512 + lane + 64 * q] =  // rfft1024_wave returns twice the spectrum
            make_float4(0.5f * Xe[0][q].x, 0.5f * Xe[0][q].y, 0.5f * Xe[1][q].x, 0.5f * Xe[1][q].y);
}

// The pre-interpolated rows (jf_device.h: htab): row n_rows + (ele + 40) 360 + azi = sum_t w_t H[row_t] for the whole-degree
// position (ele, azi) -- the index/weight rule itself (dev_interp_terms, SoundSource.cu:65-105) and the half-filters' own
// weighting (weighted_ears), one wave per row.  What GPUSoundSource.cu:118-292 recomputes for every block and source
// (four scaled products summed by atomicAdd) is computed here once per position the setters can latch.
__global__ __launch_bounds__(64) void table_interp_build_kernel(const RingTable rt, int corrected, float4 *__restrict__ htab) {
    const int lane = threadIdx.x;
    const int r = blockIdx.x;  // 0 .. kInterpRows - 1
    const int ei = r / kInterpAzi, azi = r - ei * kInterpAzi;
    int rows[4] = {0, 0, 0, 0};
    float w[4] = {0.f, 0.f, 0.f, 0.f};
    const int n = __builtin_amdgcn_readfirstlane(
        dev_interp_terms(rt, (float)(ei + kInterpEleMin), (float)azi, rows, w, corrected != 0));
    const float4 *hp[4];
    c2 wv[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        hp[t] = htab + (size_t)__builtin_amdgcn_readfirstlane(rows[t]) * 512 + lane;
        const float f = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(w[t])));
        wv[t] = c2{f, f};
    }
    float4 *out = htab + (size_t)(rt.n_rows + r) * 512 + lane;
#pragma unroll
    for (int q = 0; q < 8; q++) {
        c2 heL = c2{0.f, 0.f}, heR = c2{0.f, 0.f};  // n == 0 (no such position inside the table's range: tested): zeros
        if (n == 4) {
            const float4 hq[4] = {hp[0][64 * q], hp[1][64 * q], hp[2][64 * q], hp[3][64 * q]};
            weighted_ears<4>(hq, wv, heL, heR);
        } else if (n == 2) {
            const float4 hq[2] = {hp[0][64 * q], hp[1][64 * q]};
            const c2 w2[2] = {wv[0], wv[1]};
            weighted_ears<2>(hq, w2, heL, heR);
        } else if (n == 1) {
            const float4 hq[1] = {hp[0][64 * q]};
            const c2 w1[1] = {wv[0]};
            weighted_ears<1>(hq, w1, heL, heR);
        }
        out[64 * q] = make_float4(heL.x, heL.y, heR.x, heR.y);
    }
}

// parity tap: unnormalised spectra of arbitrary windows with the same LDS FFT
__global__ __launch_bounds__(64) void rfft_debug_kernel(const float *__restrict__ win,
                                                       const float2 *__restrict__ twg,
                                                       float2 *__restrict__ spec) {
    __shared__ float2 s_tw[kTwPack];
    __shared__ float2 s_buf[576];
    const int lane = threadIdx.x;
    for (int j = lane; j < kTwPack; j += 64) s_tw[j] = twg[j];
    __syncthreads();
    const float *x = win + (size_t)blockIdx.x * kN;
    float2 z[8], X[8];
#pragma unroll
    for (int r = 0; r < 8; r++) z[r] = *reinterpret_cast<const float2 *>(x + 2 * (lane + 64 * r));
    rfft1024_wave(z, X, s_buf, s_tw, lane);
#pragma unroll
    for (int q = 0; q < 8; q++) X[q] = make_float2(0.5f * X[q].x, 0.5f * X[q].y);  // twice the spectrum
    float2 *o = spec + (size_t)blockIdx.x * kNc;
#pragma unroll
    for (int q = 0; q < 8; q++) {
        if (lane == 0 && q == 0) {
            o[0] = make_float2(X[0].x, 0.0f);
            o[512] = make_float2(X[0].y, 0.0f);
        } else {
            o[lane + 64 * q] = X[q];
        }
    }
}

// parity taps of the stages the reference's own tests compare (precision_test.cu:60-75 distance factor,
// :225-241 weighted spectra): one wave per latched position record, through the same device code as the
// fused kernels (make_desc -> distance_factors -> filtered_bins).
//   dist [n][513] float2: D[k] of kernels.cu:116-125 (bin 512: the real part only -- the only part that is
//                         ever used, c2r drops Im Y[512]; its imaginary part is written as 0)
//   spec [n][2][513] float2 (may be null): Y_ear[k] = sum_t w_t X[k] H[row_t][ear][k] D[k] for the window
//                         win[n][1024] (X includes the 1/N of GPUSoundSource.cu:344-346), NEW filter set
__global__ __launch_bounds__(64) void stage_debug_kernel(const RingTable rt, int mode, const float *__restrict__ pos,
                                                        const float *__restrict__ win, const float4 *__restrict__ htab,
                                                        const float2 *__restrict__ twg, float2 *__restrict__ dist,
                                                        float2 *__restrict__ spec) {
    __shared__ float2 s_tw[kTwPack];
    __shared__ float2 s_buf[576];
    __shared__ ItemDesc s_desc;
    const int lane = threadIdx.x;
    for (int j = lane; j < kTwPack; j += 64) s_tw[j] = twg[j];
    const float *p = pos + 5 * (size_t)blockIdx.x;
    if (lane == 0) make_desc(rt, mode, p, p[0], p[1], s_desc);  // old == new: no crossfade
    __syncthreads();
    const ItemDesc *dp = &s_desc;
    const unsigned c_hi = (unsigned)(dp->c_fix >> 32), c_lo = (unsigned)dp->c_fix;
    float2 dq[8];
    float d512x;
    distance_factors(c_hi, c_lo, dp->inv_frac, lane, dq, d512x, s_tw);
    dq[0] = lane == 0 ? make_float2(dp->inv_frac, -0.0f) : dq[0];  // D[0] = 1/frac exactly (see distance_factors)
    float2 *dout = dist + (size_t)blockIdx.x * kNc;
#pragma unroll
    for (int q = 0; q < 8; q++) dout[lane + 64 * q] = dq[q];
    if (lane == 0) dout[512] = make_float2(d512x, 0.0f);
    if (!spec) return;
    float2 *yl = spec + (size_t)blockIdx.x * 2 * kNc, *yr = yl + kNc;
    if (dp->n_new <= 0) {
        for (int k = lane; k < kNc; k += 64) yl[k] = yr[k] = make_float2(0.f, 0.f);
        return;
    }
    const float *x = win + (size_t)blockIdx.x * kN;
    float2 z[8], X[8], xd[8];
#pragma unroll
    for (int r = 0; r < 8; r++) z[r] = *reinterpret_cast<const float2 *>(x + 2 * (lane + 64 * r));
    rfft1024_wave(z, X, s_buf, s_tw, lane);
    const float sc = 1.0f / 2048.0f;  // 1/N and the split pass's 1/2 (rfft1024_wave returns twice the spectrum)
#pragma unroll
    for (int q = 0; q < 8; q++) xd[q] = cmul(make_float2(X[q].x * sc, X[q].y * sc), dq[q]);
    const float2 x0 = make_float2(X[0].x * sc * dq[0].x, X[0].y * sc * d512x);
    xd[0] = lane == 0 ? x0 : xd[0];
    filtered_bins_nt(dp->n_new, htab, dp->rows_new, dp->w_new, xd, lane, [&](int q, float2 zk, float2 zm) {
        // zk = Y_L + j Y_R, zm = conj Y_L + j conj Y_R
        if (q == 0 && lane == 0) {  // zk = (Y_L[0], Y_R[0]), zm = (Y_L[512], Y_R[512]), all real
            yl[0] = make_float2(zk.x, 0.f);
            yr[0] = make_float2(zk.y, 0.f);
            yl[512] = make_float2(zm.x, 0.f);
            yr[512] = make_float2(zm.y, 0.f);
        } else {
            yl[lane + 64 * q] = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
            yr[lane + 64 * q] = make_float2(0.5f * (zk.y + zm.y), 0.5f * (zm.x - zk.x));
        }
    });
}

// ---------------------------------------------------------------- launchers --
// 0 = product build; bit 0 = built with a switch that makes results wrong by design (jf_experiments.h): jf_engine_create
// refuses such a library unless JF_ALLOW_EXPERIMENT=1
int kernels_build_kind() { return JF_EXP_BUILD_KIND; }

hipError_t launch_stage_debug(const RingTable &rt, int mode, const float *d_pos, const float *d_win, int n,
                              const float4 *d_htab, const float2 *d_tw, float2 *d_dist, float2 *d_spec,
                              hipStream_t st) {
    hipLaunchKernelGGL(stage_debug_kernel, dim3(n), dim3(64), 0, st, rt, mode, d_pos, d_win, d_htab, d_tw, d_dist,
                       d_spec);
    return hipGetLastError();
}

hipError_t launch_table_build(const float *d_hrir, int n_rows, int taps, const float2 *d_tw, float4 *d_htab,
                              hipStream_t st) {
    hipLaunchKernelGGL(table_build_kernel, dim3(n_rows), dim3(64), 0, st, d_hrir, taps, d_tw, d_htab);
    return hipGetLastError();
}

hipError_t launch_table_interp_build(const RingTable &rt, int corrected, float4 *d_htab, hipStream_t st) {
    hipLaunchKernelGGL(table_interp_build_kernel, dim3(kInterpRows), dim3(64), 0, st, rt, corrected, d_htab);
    return hipGetLastError();
}

hipError_t launch_rfft_debug(const float *d_win, int n, const float2 *d_tw, float2 *d_spec,
                             hipStream_t st) {
    hipLaunchKernelGGL(rfft_debug_kernel, dim3(n), dim3(64), 0, st, d_win, d_tw, d_spec);
    return hipGetLastError();
}

hipError_t launch_interp_debug(const RingTable &rt, const float *d_ele, const float *d_azi, int *d_rows,
                               float *d_w, int *d_nt, int n, int corrected, hipStream_t st) {
    hipLaunchKernelGGL(interp_debug_kernel, dim3((n + 255) / 256), dim3(256), 0, st, rt, d_ele, d_azi,
                       d_rows, d_w, d_nt, n, corrected);
    return hipGetLastError();
}

hipError_t launch_prep(const RingTable &rt, int mode, const float *d_pos, const SrcState *d_st, ItemDesc *d_desc,
                       int S, int K, int canon, hipStream_t st) {
    const int n = S * K;
    hipLaunchKernelGGL(prep_kernel, dim3((2 * n + kPrepThreads - 1) / kPrepThreads), dim3(kPrepThreads), 0, st, rt, mode,
                       d_pos, d_st, d_desc, S, K, canon);
    return hipGetLastError();
}

// Resident workgroups of the fused kernel that a call with these parameters launches (per-source kernel for
// G = 1, pair kernel otherwise), on the CURRENT device: CUs x workgroups per CU for this build's LDS and register
// footprint.  The engine asks once per (kernel, block size) at creation and keeps the answer with its device.
hipError_t fused_resident_workgroups(int nb, int kind /* 0 per-source kernel, 1 pair kernel, 2 pair kernel with rows */, int *out) {
    int dev = 0, per_cu = 0;
    hipDeviceProp_t prop;
    hipError_t q = hipGetDevice(&dev);
    if (q == hipSuccess) q = hipGetDeviceProperties(&prop, dev);
    if (q != hipSuccess) return q;
    const int threads = 64 * kWavesPerWg;
    if (kind == 2) {
        switch (nb) {
        case 1: q = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fused_pair_kernel<1, true>, threads, 0); break;
        case 2: q = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fused_pair_kernel<2, true>, threads, 0); break;
        case 3: q = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fused_pair_kernel<3, true>, threads, 0); break;
        case 4: q = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fused_pair_kernel<4, true>, threads, 0); break;
        default: return hipErrorInvalidValue;
        }
    } else if (kind == 1) {
        switch (nb) {
        case 1: q = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fused_pair_kernel<1, false>, threads, 0); break;
        case 2: q = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fused_pair_kernel<2, false>, threads, 0); break;
        case 3: q = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fused_pair_kernel<3, false>, threads, 0); break;
        case 4: q = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fused_pair_kernel<4, false>, threads, 0); break;
        default: return hipErrorInvalidValue;
        }
    } else {
        switch (nb) {
        case 1: q = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fused_block_kernel<1>, threads, 0); break;
        case 2: q = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fused_block_kernel<2>, threads, 0); break;
        case 3: q = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fused_block_kernel<3>, threads, 0); break;
        case 4: q = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fused_block_kernel<4>, threads, 0); break;
        default: return hipErrorInvalidValue;
        }
    }
    if (q != hipSuccess) return q;
    if (per_cu < 1) per_cu = 1;
    *out = prop.multiProcessorCount * per_cu;
    return hipSuccess;
}

// Persistent grid: at most max_wgs workgroups (what the GPU holds at once, fused_resident_workgroups; an
// over-estimate only queues the surplus workgroups -- the unit loop is a plain stride, there is no grid
// barrier, and every wave leaves the loop once unit >= n_units).
hipError_t launch_fused(const FusedParams &P, int max_wgs, hipStream_t st) {
    if (P.G <= 0 || P.S % P.G || max_wgs < 1) return hipErrorInvalidValue;
    const int n_items = P.K * (P.S / P.G);
    const int nb = P.B / 64;
    if (nb < 1 || nb > 4) return hipErrorInvalidValue;
    const int per_wg = P.G > 1 ? kPairsPerWg : kWavesPerWg;  // units a workgroup works on at a time
    int wgs = (n_items + per_wg - 1) / per_wg;
    if (wgs > max_wgs) wgs = max_wgs;
    const dim3 block(64 * kWavesPerWg);
    // groups of sources are summed as spectra by wave pairs (two inverse transforms per group); single sources
    // keep the per-source kernel, whose blocks are the reference's per-source `intermediate`
    if (P.G > 1) {
        FusedParams Q = P;
        Q.n_pair_wgs = wgs;
        // + the workgroups that prepare the following window's descriptors (two lanes per item)
        const int n_prep = Q.prep_pos != nullptr ? (2 * Q.S * Q.prep_K + 64 * kWavesPerWg - 1) / (64 * kWavesPerWg) : 0;
        const dim3 grid(wgs + n_prep);
        const bool rows = (P.mode & kModeInterpRows) != 0;  // descriptors may name pre-interpolated rows
        switch (P.B / 64) {
        case 1: if (rows) hipLaunchKernelGGL((fused_pair_kernel<1, true>), grid, block, 0, st, Q);
                else hipLaunchKernelGGL((fused_pair_kernel<1, false>), grid, block, 0, st, Q);
                break;
        case 2: if (rows) hipLaunchKernelGGL((fused_pair_kernel<2, true>), grid, block, 0, st, Q);
                else hipLaunchKernelGGL((fused_pair_kernel<2, false>), grid, block, 0, st, Q);
                break;
        case 3: if (rows) hipLaunchKernelGGL((fused_pair_kernel<3, true>), grid, block, 0, st, Q);
                else hipLaunchKernelGGL((fused_pair_kernel<3, false>), grid, block, 0, st, Q);
                break;
        case 4: if (rows) hipLaunchKernelGGL((fused_pair_kernel<4, true>), grid, block, 0, st, Q);
                else hipLaunchKernelGGL((fused_pair_kernel<4, false>), grid, block, 0, st, Q);
                break;
        default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    const dim3 grid(wgs);
    switch (P.B / 64) {
    case 1: hipLaunchKernelGGL(fused_block_kernel<1>, grid, block, 0, st, P); break;
    case 2: hipLaunchKernelGGL(fused_block_kernel<2>, grid, block, 0, st, P); break;
    case 3: hipLaunchKernelGGL(fused_block_kernel<3>, grid, block, 0, st, P); break;
    case 4: hipLaunchKernelGGL(fused_block_kernel<4>, grid, block, 0, st, P); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

int rt_waves_per_wg(int n_sources) { return n_sources <= kRtFewMaxSources ? kRtWavesFew : kRtWavesMany; }

// n_wgs workgroups of rt_waves_per_wg(P.S) waves
// head (may be null): the reverb stage's head of this block, run inside the launch by each source's wave (rv_head_wave);
// only with rt_waves_per_wg(P.S) == kRtWavesFew and a block size the reverb has (64, 128, 256)
hipError_t launch_rt_block(const FusedParams &P, const RingTable &rt, const float *pos, float *out, int *done, int seq,
                           int n_wgs, const ReverbParams *head, hipStream_t st) {
    if (P.K != 1 || n_wgs < 1) return hipErrorInvalidValue;
    const bool few = rt_waves_per_wg(P.S) == kRtWavesFew;
    const dim3 grid(n_wgs), block(64 * (few ? kRtWavesFew : kRtWavesMany));
    float2 *o = reinterpret_cast<float2 *>(out);
    if (head != nullptr) {
        if (!few || head->B != P.B || head->K != 1) return hipErrorInvalidValue;
        switch (P.B / 64) {
        case 1: hipLaunchKernelGGL((rt_block_kernel<1, kRtWavesFew, true>), grid, block, 0, st, P, rt, pos, o, done, seq, *head); break;
        case 2: hipLaunchKernelGGL((rt_block_kernel<2, kRtWavesFew, true>), grid, block, 0, st, P, rt, pos, o, done, seq, *head); break;
        case 4: hipLaunchKernelGGL((rt_block_kernel<4, kRtWavesFew, true>), grid, block, 0, st, P, rt, pos, o, done, seq, *head); break;
        default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    const ReverbParams none{};
#define JF_RT_LAUNCH(NOUT)                                                                                                          \
    if (few) hipLaunchKernelGGL((rt_block_kernel<NOUT, kRtWavesFew, false>), grid, block, 0, st, P, rt, pos, o, done, seq, none);  \
    else hipLaunchKernelGGL((rt_block_kernel<NOUT, kRtWavesMany, false>), grid, block, 0, st, P, rt, pos, o, done, seq, none)
    switch (P.B / 64) {
    case 1: JF_RT_LAUNCH(1); break;
    case 2: JF_RT_LAUNCH(2); break;
    case 3: JF_RT_LAUNCH(3); break;
    case 4: JF_RT_LAUNCH(4); break;
    default: return hipErrorInvalidValue;
    }
#undef JF_RT_LAUNCH
    return hipGetLastError();
}

hipError_t launch_mix_prep(const float *d_partial, float *d_mix, int S_groups, int K, int B, const RingTable &rt, int mode,
                           const float *d_pos_next, ItemDesc *d_desc_next, int S, int K_next, int canon, hipStream_t st) {
    const int blk = 2 * B;
    const int n_mix = K * (blk / 64);
    const int per = 64 * kMixGroups;
    const int n_prep = (2 * S * K_next + per - 1) / per;
    hipLaunchKernelGGL(mix_prep_kernel, dim3(n_mix + n_prep), dim3(per), 0, st, d_partial, d_mix, S_groups, blk, n_mix, rt,
                       mode, d_pos_next, d_desc_next, S, K_next, canon);
    return hipGetLastError();
}

hipError_t launch_mix(const float *d_partial, float *d_mix, int S, int K, int B, hipStream_t st) {
    const int blk = 2 * B;  // multiple of 64 because B is
    if (S == kMixGroups || S == 2 * kMixGroups || S == 4 * kMixGroups) {  // the pair kernel's large calls
        const int total = K * blk;
        const dim3 grid((total + 255) / 256), block(256);
        if (S == kMixGroups) hipLaunchKernelGGL(mix_few_kernel<1>, grid, block, 0, st, d_partial, d_mix, blk, total);
        else if (S == 2 * kMixGroups) hipLaunchKernelGGL(mix_few_kernel<2>, grid, block, 0, st, d_partial, d_mix, blk, total);
        else hipLaunchKernelGGL(mix_few_kernel<4>, grid, block, 0, st, d_partial, d_mix, blk, total);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(mix_kernel, dim3(K * (blk / 64)), dim3(64 * kMixGroups), 0, st, d_partial, d_mix, S, K,
                       blk);
    return hipGetLastError();
}

}  // namespace jf
