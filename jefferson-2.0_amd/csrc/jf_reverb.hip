// jf_reverb.hip -- convolution reverb ahead of the spatialiser (SURVEY.md 8f-1,
// BASELINE.json configs[4]): uniformly partitioned overlap-save convolution with a
// frequency-domain delay line (FDL), one partition = one audio block of B samples.
//
// Replaces the reference's offline whole-signal cuFFT convolution (cudaPart.cu:65-205,
// disabled there by reverbFlag = false and broken by swapped kernel arguments, App. C#12)
// with the real-time form: per block and source
//   A  X_k = rfft([x_{k-1}, x_k])                          -> FDL slot (head + k)        1 KB written
//   B  Y_k = sum_{p<P} X_{k-p} * H_p ; y_k = irfft(Y_k)[B:] -> the source's "wet" ring    P KB read
// and the spatialiser then reads the wet ring as the source's signal.  Stage B is the
// HBM-bound part: P x B x 8 bytes of FDL per source-block (690 x 1 KB at B = 128, 2 s IR).
//
// Spectra are stored packed: B complex per partition, bin 0 holding (X[0].re, X[B].re).
//
// NON-UNIFORM PARTITIONING (round 4; ReverbBigParams in jf_device.h): for a long impulse response the stage above is only
// the HEAD -- the first 2 M = 32 partitions of B -- and the rest of the response is cut into partitions of B1 = 16 B
// taps: every 16 blocks one transform of 2 B1 samples, P1 - 1 multiply-accumulates per bin (P1 = ceil((n_ir - B1) / B1): 42
// instead of 690 x 16 for the 2 s response at B = 128) and one inverse give the tail's contribution to the 16 blocks of the
// big block AFTER the next (reverb_big_*), which the head's finishing step adds.  The reference's own form is ONE product over the whole signal
// (cudaPart.cu:87-153: ~log N work per sample); uniform partitions cost P operations per sample, two sizes P / 16 + 16.
#include <hip/hip_runtime.h>

#include "jf_device.h"
#include "jf_packed.h"

#include <algorithm>

namespace jf {

#include "jf_rv_small.h"

// One wavefront per (block k, source s).
template <int B>
__global__ __launch_bounds__(256) void reverb_fft_kernel(const ReverbParams P) {
    __shared__ float2 s_buf[4][2 * B];
    __shared__ float2 s_tw[1024];  // the transform's passes read their twiddles from LDS, not from global memory
    for (int j = threadIdx.x; j < 1024; j += 256) s_tw[j] = P.tw[j];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = blockIdx.x * 4 + wave;
    const int n_skip = P.skip_hi - P.skip_lo;  // blocks nobody needs transformed or copied (launch_fft sizes the grid)
    if (g >= (P.K - n_skip) * P.S) return;
    const int ka = g / P.S, s = g - ka * P.S;
    const int k = ka < P.skip_lo ? ka : ka + n_skip;
    rv_forward<B>(P, k, s, s_buf[wave], s_buf[wave] + B, lane, nullptr, s_tw);
}

// ---------------------------------------------------------------- stage B --
// One workgroup (16 waves) per (block k, group of T consecutive sources): the waves split the P
// partitions (so that a real-time call with K*S ~ #CUs still puts 16 waves of loads in flight on
// every CU), each lane owns B/64 consecutive bins of every source of the group; an IR spectrum H_p is
// loaded once and used for the T sources (T = 4 in batch calls: 1.25 instead of 2 loads per
// multiply-accumulate); LDS reduce; waves 0..T-1 each invert one source and write its wet block.
// FUSE (T = 1, calls of one block: the audio callback's shape): stage A runs inside this kernel.  The last wave
// transforms the new block and takes partition 0 from LDS while the other 15 stream the older partitions, which do not
// depend on it: the transform costs neither a launch nor the gap behind it (5.1 us + gap of a 35.6 us step).  A wave's
// partitions and their order are fixed, the sums deterministic.
constexpr int kMacWaves = 16;
template <int B, int T, bool FUSE = false>
__global__ __launch_bounds__(64 * kMacWaves) void reverb_mac_kernel(const ReverbParams P) {
    static_assert(!FUSE || T == 1, "the fused form is the one-source form");
    constexpr int NB = B / 64;
    __shared__ float2 s_red[kMacWaves][T][B];
    __shared__ float2 s_fft[T][2 * B];
    __shared__ float2 s_x0[FUSE ? B : 1];
    // one-block calls: the two transforms of the block are this kernel's chain, and each of their passes reads twiddles --
    // from LDS (staged by the 1024 threads, one entry each), not from global memory
    __shared__ float2 s_tw[FUSE ? 1024 : 1];
    const float2 *tw = P.tw;
    if (FUSE) {
        s_tw[threadIdx.x] = P.tw[threadIdx.x];
        static_assert(!FUSE || 64 * kMacWaves == 1024, "one twiddle per thread");
        __syncthreads();
        tw = s_tw;
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int SG = P.S / T;
    const int kl = blockIdx.x / SG, s0 = (blockIdx.x - kl * SG) * T;
    const int k = P.kb + kl;

    const float2 *fdl = P.fdl + (size_t)s0 * P.Rg * B + lane * NB;
    const float2 *hs = P.hspec + lane * NB;
    float2 acc[T][NB];
    float2 acc0[T];  // bin 0 is two packed real bins
#pragma unroll
    for (int t = 0; t < T; t++) {
        acc0[t] = make_float2(0.f, 0.f);
#pragma unroll
        for (int i = 0; i < NB; i++) acc[t][i] = make_float2(0.f, 0.f);
    }
    const bool transformer = FUSE && wave == kMacWaves - 1;
    if (transformer) {
        rv_forward<B>(P, k, s0, s_fft[0], s_fft[0] + B, lane, s_x0, tw);
        JF_RV_SYNC();
        // partition 0: the spectrum just made, from LDS
        const float2 *hp = hs;
#pragma unroll
        for (int i = 0; i < NB; i++) {
            const float2 h = hp[i], x = s_x0[lane * NB + i];
            acc[0][i].x += x.x * h.x - x.y * h.y;
            acc[0][i].y += x.x * h.y + x.y * h.x;
            if (i == 0) {
                acc0[0].x += x.x * h.x;
                acc0[0].y += x.y * h.y;
            }
        }
    }
    // this wave's partitions: p_first, p_first + stride, ...  In the fused form the transformer's chain (the block's
    // samples, seven LDS passes, the split) is as long as the other waves' whole stream, so it takes partition 0 only
    // and partitions 1 .. P-1 are dealt to the other 15 waves.
    const int stride = FUSE ? kMacWaves - 1 : kMacWaves;
    const int p_first = !FUSE ? wave : transformer ? P.P : 1 + wave;
    int slot = (P.head + k - p_first) % P.Rg;
    if (slot < 0) slot += P.Rg;
#pragma unroll 2
    for (int p = p_first; p < P.P; p += stride) {
        float2 h[NB];
        const float2 *hp = hs + (size_t)p * B;
        if (NB == 2) {
            const float4 hv = *reinterpret_cast<const float4 *>(hp);
            h[0] = make_float2(hv.x, hv.y);
            h[NB - 1] = make_float2(hv.z, hv.w);
        } else {
#pragma unroll
            for (int i = 0; i < NB; i++) h[i] = hp[i];
        }
#pragma unroll
        for (int t = 0; t < T; t++) {
            float2 x[NB];
            const float2 *xp = fdl + ((size_t)t * P.Rg + slot) * B;
            if (NB == 2) {
                const float4 xv = *reinterpret_cast<const float4 *>(xp);
                x[0] = make_float2(xv.x, xv.y);
                x[NB - 1] = make_float2(xv.z, xv.w);
            } else {
#pragma unroll
                for (int i = 0; i < NB; i++) x[i] = xp[i];
            }
#pragma unroll
            for (int i = 0; i < NB; i++) {
                acc[t][i].x += x[i].x * h[i].x - x[i].y * h[i].y;
                acc[t][i].y += x[i].x * h[i].y + x[i].y * h[i].x;
            }
            acc0[t].x += x[0].x * h[0].x;
            acc0[t].y += x[0].y * h[0].y;
        }
        slot -= stride;
        if (slot < 0) slot += P.Rg;
    }
#pragma unroll
    for (int t = 0; t < T; t++) {
        if (lane == 0) acc[t][0] = acc0[t];
#pragma unroll
        for (int i = 0; i < NB; i++) s_red[wave][t][lane * NB + i] = acc[t][i];
    }
    __syncthreads();
    if (wave >= T) return;
    mac_finish<B, kMacWaves>(&s_red[0][wave][0], T * B, s_fft[wave], P, s0 + wave, k, lane, tw);
}

// Batch form of stage B (many blocks per call): one workgroup per (source, KB consecutive blocks).
// Block k + i at partition p needs FDL slot head + k + i - p, so the KB blocks of a tile use a
// sliding window of KB spectra: per partition ONE new X load and one H load feed KB multiply-
// accumulates -- the FDL is L2/Infinity-Cache resident here and the cache bandwidth, not HBM, is what
// the loads queue on, so a tile is made as deep as the registers allow: KB = 16, with a wave covering 64
// bins (one per lane; B / 64 waves side by side) of one contiguous chunk of the partitions (a multiple of
// KB, so that the window's register indices are static).  The packed pair in bin 0 is carried as if it
// were complex; mac_finish recomputes it from the compact copies.  Waves then each finish two blocks.
constexpr int kTileWaves = 8;
#ifndef JF_RV_PROGRESS_PRIO
#define JF_RV_PROGRESS_PRIO 1
#endif


#ifndef JF_TILE_ATTR
#define JF_TILE_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))  // two workgroups per CU: 128 VGPRs
#endif
template <int B, int KB>
__global__ __launch_bounds__(64 * kTileWaves) JF_TILE_ATTR void reverb_mac_tiled_kernel(const ReverbParams P) {
    constexpr int BH = B / 64;          // waves side by side over the bins
    constexpr int NC = kTileWaves / BH;  // partition chunks
    static_assert(KB <= 2 * kTileWaves, "each wave finishes at most two blocks");
    __shared__ float2 s_red[NC][KB][B];
    __shared__ float2 s_fft[kTileWaves][2 * B];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bh = wave % BH, c = wave / BH;
    // The tiles of one source read the same stretch of its delay line, KB slots apart: they are given workgroup
    // numbers that are equal mod 8 (same XCD, same L2) and next to each other in dispatch order, so that the later
    // readers find the slots in that L2 instead of fetching them again.
    int kt, s;
    if (P.S % 8 == 0) {
        const int n_tiles = gridDim.x / P.S, per = 8 * n_tiles;
        const int grp = blockIdx.x / per, r = blockIdx.x - grp * per;
        kt = r >> 3;
        s = 8 * grp + (r & 7);
    } else {
        kt = blockIdx.x / P.S;
        s = blockIdx.x - kt * P.S;
    }
    const int k0 = P.kb + kt * KB;

    // Addresses as a wave-uniform base (scalar registers) + one per-lane byte offset: the delay-line slot of X(-p) steps
    // back by one per partition (with a wrap), the IR spectrum forward by one -- no division and no 64-bit vector
    // arithmetic in the loop.
    const char *fdl0 = reinterpret_cast<const char *>(P.fdl + (size_t)s * P.Rg * B);
    const char *hsp = reinterpret_cast<const char *>(P.hspec);
    unsigned voff = 8u * (64 * bh + lane);
    asm volatile("" : "+v"(voff));
    auto load_at = [&](const char *base) {
        const float2 *q = reinterpret_cast<const float2 *>(base + voff);
        return rv_v2{q->x, q->y};
    };
    auto slot_of = [&](int u) {  // delay-line slot of the spectrum of block k0 + u (u may be far in the past)
        int slot = (P.head + k0 + u) % P.Rg;
        return slot < 0 ? slot + P.Rg : slot;
    };
    rv_v2 acc[KB], xr[KB];
#pragma unroll
    for (int i = 0; i < KB; i++) acc[i] = rv_v2{0.f, 0.f};
    const int chunk = (P.P + NC * KB - 1) / (NC * KB) * KB;
    const int pa = c * chunk;
    const int pb = pa + chunk < P.P ? pa + chunk : P.P;
    // window before the chunk's first partition: X(i - pa), i = 1..KB-1, kept at xr[(i - pa) mod KB] = xr[i]
#pragma unroll
    for (int i = 1; i < KB; i++) {
        xr[i] = load_at(fdl0 + (size_t)slot_of(i - pa) * (B * 8));
    }
    int xslot = slot_of(-pa);                     // slot of X(-p), p = pa
    const char *hp = hsp + (size_t)pa * (B * 8);  // H_p
    auto step = [&](int j) {  // j = p mod KB, a constant after unrolling
        const rv_v2 h = load_at(hp);
        // X(-p) replaces X(KB - p), last used by block KB-1 at p-1
        xr[(KB - j) % KB] = load_at(fdl0 + (size_t)(unsigned)xslot * (B * 8));
        xslot = xslot == 0 ? P.Rg - 1 : xslot - 1;
        hp += B * 8;
        // acc[i] += X(i - p) * h.  The four FMAs of a product as two sweeps over the blocks -- the real parts of X first,
        // then the imaginary parts: the two FMAs on one accumulator component are then 32 instructions apart instead
        // of 2 (profiles/micro/cmac_tile.hip: 114.8 against 104.6 TFLOP/s for this loop on registers alone; in the
        // kernel, where the loads set the pace, the same time at 90 VGPRs instead of 124).  The same operations on
        // every accumulator in the same order: bit-identical sums.
#if JF_RV_SCALAR_MAC
#pragma unroll
        for (int i = 0; i < KB; i++) {
            const rv_v2 x = xr[(i + KB - j) % KB];
            acc[i].x = __builtin_fmaf(x.x, h.x, acc[i].x);
            acc[i].y = __builtin_fmaf(x.x, h.y, acc[i].y);
        }
#pragma unroll
        for (int i = 0; i < KB; i++) {
            const rv_v2 x = xr[(i + KB - j) % KB];
            acc[i].x = __builtin_fmaf(-x.y, h.y, acc[i].x);
            acc[i].y = __builtin_fmaf(x.y, h.x, acc[i].y);
        }
#else
#pragma unroll
        for (int i = 0; i < KB; i++) rv_cmac(acc[i], xr[(i + KB - j) % KB], h);
#endif
    };
    int p0 = pa;
    [[maybe_unused]] int groups_done = 0;
    for (; p0 + KB <= pb; p0 += KB) {  // straight-line groups: loads of later steps may move above earlier MACs
#pragma unroll
        for (int j = 0; j < KB; j++) {
#if JF_RV_PROGRESS_PRIO
            // progress-ordered priorities (see fused_pair_kernel): the SIMD's arbiter serves its oldest wave first, and a
            // workgroup finishes with its slowest wave
            if (j % (KB / JF_RV_PROGRESS_PRIO) == 0) {
                switch (3 - (groups_done++ & 3)) {
                case 0: __builtin_amdgcn_s_setprio(0); break;
                case 1: __builtin_amdgcn_s_setprio(1); break;
                case 2: __builtin_amdgcn_s_setprio(2); break;
                default: __builtin_amdgcn_s_setprio(3); break;
                }
            }
#endif
            step(j);
        }
    }
#pragma unroll
    for (int j = 0; j < KB; j++)
        if (p0 + j < pb) step(j);  // wave-uniform
#pragma unroll
    for (int i = 0; i < KB; i++) s_red[c][i][64 * bh + lane] = make_float2(acc[i].x, acc[i].y);
    __syncthreads();
#pragma unroll 1
    for (int i = wave; i < KB; i += kTileWaves)
        if (k0 + i < P.kb + P.kn) mac_finish<B, NC, true>(&s_red[0][i][0], KB * B, s_fft[wave], P, s, k0 + i, lane, P.tw);
}

// ------------------------------------------------- big partitions (level 1) --
// Complex FFT of NPT = 1024 or 2048 points by a whole workgroup of NT = 256 threads in LDS: radix 8, 8, 8 and a last pass of
// radix 2 or 4.  T2[j] = exp(+2 pi i j / (2 NPT)), j < 2 NPT.  The twiddles a thread needs do not depend on the data: they
// are loaded into registers FIRST (BigTwiddles::load, before the caller fetches its input), so that a transform waits for
// global memory once, not once per pass.
#ifndef JF_RV_BIG_TW_LDS
#define JF_RV_BIG_TW_LDS 1
#endif
template <int NPT, int NT>
struct BigTwiddles {
    static constexpr int RL = NPT / 512;               // radix of the last pass
    static constexpr int NL = NPT / RL / NT;           // its butterflies per thread
    static_assert(NPT / 8 <= NT && NL >= 1, "one radix-8 butterfly per thread at most");
    // passes with Ns = 8 and 64: exp(2 pi i r k / (8 Ns)), r = 1 .. 7 -- JF_RV_BIG_TW_LDS = 0: in registers (28 of them, held
    // through the whole kernel); 1: in LDS (504 entries, 4 KB per workgroup, staged once by stage_w8: pass Ns = 8 reads
    // eight distinct entries per wave -- broadcasts --, pass Ns = 64 one entry per lane).  The persistent transform kernels keep
    // the next item's input in registers instead.
#if !JF_RV_BIG_TW_LDS
    float2 w8[2][7];
#endif
    float2 wl[NL][RL - 1];  // last pass (Ns = 512)
    static constexpr int kW8Len = 504;
    JF_DEV static void stage_w8(const float2 *__restrict__ T2, float2 *s_w8, int tid) {
#if JF_RV_BIG_TW_LDS
        const float2 *__restrict__ pk = T2 + 2 * NPT;
        for (int k = tid; k < kW8Len; k += NT) s_w8[k] = pk[k];
#endif
    }
    JF_DEV float2 w8_at(const float2 *s_w8, int p, int r, int tid) const {  // r = 1 .. 7
#if JF_RV_BIG_TW_LDS
        const int Ns = p ? 64 : 8;
        return s_w8[(p ? 56 : 0) + (r - 1) * Ns + (tid & (Ns - 1))];
#else
        return w8[p][r - 1];
#endif
    }
    // The values are entries of the circle T2, but a wave that fetches them there gathers 64 cache lines per load (strides of
    // 8 r .. 64 r entries between neighbouring lanes) -- 17 to 21 such loads per thread were a third of a transform kernel's
    // time.  The engine lays the same values out per pass and r, neighbouring lanes side by side, BEHIND the circle
    // (big_twiddle_pack_*, jf_engine_reverb.cpp): a load touches 1 to 8 lines.  Same bits, same results: forward 41.4 -> 35.2 us,
    // inverse 43.5 -> 38.7 us per launch at config 5's batch shape (rocprofv3, 320 launches, twice).
    JF_DEV void load(const float2 *__restrict__ T2, int tid) {
        const float2 *__restrict__ pk = T2 + 2 * NPT;
#if !JF_RV_BIG_TW_LDS
        const int j = tid < NPT / 8 ? tid : 0;
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const int Ns = p ? 64 : 8;
#pragma unroll
            for (int r = 1; r < 8; r++) w8[p][r - 1] = pk[(p ? 56 : 0) + (r - 1) * Ns + (j & (Ns - 1))];
        }
#endif
#pragma unroll
        for (int u = 0; u < NL; u++)
#pragma unroll
            for (int r = 1; r < RL; r++) wl[u][r - 1] = pk[504 + (r - 1) * 512 + ((tid + u * NT) & 511)];
    }
};

// Where element i of a workgroup transform lies in its LDS buffer.  The passes read with unit stride and write with strides of
// 8 (first pass: 8 j + r) and of 8 inside runs of 64 (second pass: 64 (j >> 3) + (j & 7) + 8 r); the later passes write with
// unit stride.  JF_RV_BIG_XOR = 0: one float2 of padding per 8 (rv_at<true>) -- the strided stores are conflict-free, but a
// half-wave's unit-stride 8-byte reads then span 36 bank pairs of 32: one extra LDS cycle per read, 40 % of the transform
// kernels' LDS cycles (round 4's counters).  1: no padding, the low five index bits XORed with index bits 5-7 (low three) and
// 6-7 (bits 3-4): aligned unit-stride runs stay permutations of the 32 bank pairs, the first pass's 32 lanes (bits 3-4 =
// j & 3, low bits r) get (j >> 2) & 7 in the low bits and ((j >> 3) ^ j) & 3 above -- injective in j -- and the second
// pass's (low bits j & 7, bits 3-4 of r) get j >> 3 into bits 3-4: every access of every pass conflict-free, and a
// transform takes 16 KB instead of 18.
#ifndef JF_RV_BIG_XOR
#define JF_RV_BIG_XOR 0
#endif
JF_DEV int rv_big_at(int i) {
#if JF_RV_BIG_XOR
    return i ^ ((i >> 5) & 7) ^ (((i >> 6) & 3) << 3);
#else
    return rv_at<true>(i);
#endif
}
constexpr int rv_big_len(int n) { return JF_RV_BIG_XOR ? n : rv_buf_len<true>(n); }

// NTR transforms of NPT points at once by one workgroup, IN PLACE in NTR buffers of LDS (layout rv_big_at): the
// twiddles depend on the thread and the pass only, so the transforms share them (registers, loads) and the barriers -- and
// since every thread has read its butterflies' inputs before any thread writes (a barrier between), one buffer per transform
// is enough.  The input comes in REGISTERS: v[t][r] = x_t[tid + r NPT / 8] of threads tid < NPT / 8 (the first pass needs no
// twiddles and reads nothing from LDS: the caller loads straight from global memory).  The results lie in buf[t] in natural order.
// UPPER_HALF: only elements NPT / 2 .. NPT - 1 of the result are formed and stored (the inverse kernel's overlap-save keeps the
// second half of its 2 B1 samples = the upper half of the complex result: the last pass's other outputs are never read).
template <int NPT, int DIR, int NT, bool UPPER_HALF = false, int NTR, int LEN>
JF_DEV void cfft_wg(float2 (&v)[NTR][8], float2 (&buf)[NTR][LEN], const BigTwiddles<NPT, NT> &tw, const float2 *s_w8, int tid) {
    static_assert(!(UPPER_HALF && JF_RV_BIG_XOR), "written for the padded layout");
    constexpr int N8 = NPT / 8;  // radix-8 butterflies per pass, one per thread
    const bool on = tid < N8;
#if !JF_RV_BIG_XOR
    // Padded layout, every address spelled out as ONE per-thread base + a compile-time offset (the compiler does not see that
    // (tid + 256 r) >> 3 = (tid >> 3) + 32 r and recomputed every one of the ~70 addresses of a transform: a third of the
    // kernel's vector instructions, round 5's counters): at(i) = i + (i >> 3), so
    //   reads of the radix-8 passes   at(tid + r N8)              = rd0 + r (9 N8 / 8)
    //   first pass's stores            at(8 tid + r)               = 9 tid + r
    //   second pass's (Ns = 8)         at(64 g + k + 8 r), k < 8   = 72 g + k + 9 r
    //   third pass's (Ns = 64)         at(512 g + k + 64 r), k < 64 = 576 g + k + (k >> 3) + 72 r
    //   last pass, in place            at(tid + 256 u + r NPT / RL) = rd0 + 288 u + r (9 NPT / (8 RL))
    const int rd0 = tid + (tid >> 3);
    const int wa0 = 9 * tid;
    const int wb0 = 72 * (tid >> 3) + (tid & 7);
    const int wc0 = 576 * (tid >> 6) + (tid & 63) + ((tid & 63) >> 3);
    if (on) {
#pragma unroll
        for (int t = 0; t < NTR; t++) {
            rv_fft8<DIR>(v[t]);
            float2 *w = buf[t] + wa0;
#pragma unroll
            for (int r = 0; r < 8; r++) w[r] = v[t][r];
        }
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 2; p++) {
        if (on) {
#pragma unroll
            for (int t = 0; t < NTR; t++) {
                const float2 *rd = buf[t] + rd0;
#pragma unroll
                for (int r = 0; r < 8; r++) v[t][r] = rd[r * (9 * N8 / 8)];
#pragma unroll
                for (int r = 1; r < 8; r++) {
                    const float2 w = tw.w8_at(s_w8, p, r, tid);
                    v[t][r] = DIR > 0 ? rv_mul(v[t][r], w) : rv_mulc(v[t][r], w);
                }
                rv_fft8<DIR>(v[t]);
            }
        }
        __syncthreads();  // every input of the pass has been read
        if (on) {
#pragma unroll
            for (int t = 0; t < NTR; t++) {
                float2 *w = buf[t] + (p ? wc0 : wb0);
#pragma unroll
                for (int r = 0; r < 8; r++) w[r * (p ? 72 : 9)] = v[t][r];
            }
        }
        __syncthreads();
    }
    constexpr int RL = BigTwiddles<NPT, NT>::RL, NL = BigTwiddles<NPT, NT>::NL;
    static_assert(RL * NL <= 8 && NT == 256 && NL * NT <= 512, "the registers of v[t] hold the last pass's values; j < 512");
#pragma unroll
    for (int t = 0; t < NTR; t++) {
        float2 *io = buf[t] + rd0;
#pragma unroll
        for (int u = 0; u < NL; u++) {
            float2 x[RL];
#pragma unroll
            for (int r = 0; r < RL; r++) x[r] = io[288 * u + r * (9 * NPT / (8 * RL))];
#pragma unroll
            for (int r = 1; r < RL; r++) x[r] = DIR > 0 ? rv_mul(x[r], tw.wl[u][r - 1]) : rv_mulc(x[r], tw.wl[u][r - 1]);
            rv_fftR<RL, DIR>(x);
            // in place: butterfly j = tid + 256 u (< 512) reads and writes elements j + r NPT / RL ... no: it WRITES j + 512 r,
            // which other butterflies read when RL = 4 (NPT / RL = 512 too) -- the same elements: in place indeed
            static_assert(NPT / RL == 512, "the last pass is in place");
#pragma unroll
            for (int r = 0; r < RL; r++) v[t][u * RL + r] = x[r];
        }
    }
    // (in place and each element touched by ONE butterfly: no barrier between the reads and the writes)
#pragma unroll
    for (int t = 0; t < NTR; t++) {
        float2 *io = buf[t] + rd0;
#pragma unroll
        for (int u = 0; u < NL; u++)
#pragma unroll
            for (int r = UPPER_HALF ? RL / 2 : 0; r < RL; r++) io[288 * u + r * 576] = v[t][u * RL + r];
    }
    __syncthreads();
#else
    auto store = [&](int j, int Ns) {  // Stockham: butterfly j's outputs go to j0 + r Ns
        const int k = j & (Ns - 1), j0 = (j - k) * 8 + k;
#pragma unroll
        for (int t = 0; t < NTR; t++)
#pragma unroll
            for (int r = 0; r < 8; r++) buf[t][rv_big_at(j0 + r * Ns)] = v[t][r];
    };
    if (on) {
#pragma unroll
        for (int t = 0; t < NTR; t++) rv_fft8<DIR>(v[t]);
        store(tid, 1);
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const int Ns = p ? 64 : 8;
        if (on) {
#pragma unroll
            for (int t = 0; t < NTR; t++) {
#pragma unroll
                for (int r = 0; r < 8; r++) v[t][r] = buf[t][rv_big_at(tid + r * (NPT / 8))];
#pragma unroll
                for (int r = 1; r < 8; r++) {
                    const float2 w = tw.w8_at(s_w8, p, r, tid);
                    v[t][r] = DIR > 0 ? rv_mul(v[t][r], w) : rv_mulc(v[t][r], w);
                }
                rv_fft8<DIR>(v[t]);
            }
        }
        __syncthreads();  // every input of the pass has been read
        if (on) store(tid, Ns);
        __syncthreads();
    }
    // last pass: radix RL, NL butterflies per thread (RL NL = NPT / NT <= 8 values: the registers of v[t] again)
    constexpr int RL = BigTwiddles<NPT, NT>::RL, NL = BigTwiddles<NPT, NT>::NL;
    static_assert(RL * NL <= 8, "the registers of v[t] hold the last pass's values");
#pragma unroll
    for (int t = 0; t < NTR; t++)
#pragma unroll
        for (int u = 0; u < NL; u++) {
            const int j = tid + u * NT;
            float2 x[RL];
#pragma unroll
            for (int r = 0; r < RL; r++) x[r] = buf[t][rv_big_at(j + r * (NPT / RL))];
#pragma unroll
            for (int r = 1; r < RL; r++) x[r] = DIR > 0 ? rv_mul(x[r], tw.wl[u][r - 1]) : rv_mulc(x[r], tw.wl[u][r - 1]);
            rv_fftR<RL, DIR>(x);
#pragma unroll
            for (int r = 0; r < RL; r++) v[t][u * RL + r] = x[r];
        }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NTR; t++)
#pragma unroll
        for (int u = 0; u < NL; u++) {
            const int j = tid + u * NT, k = j & 511, j0 = (j - k) * RL + k;
#pragma unroll
            for (int r = 0; r < RL; r++) buf[t][rv_big_at(j0 + r * 512)] = v[t][u * RL + r];
        }
    __syncthreads();
#endif
}

constexpr int kBigThreads = 256;

// Scalar loads of per-source records (SrcSignal, play positions, SrcState): through a generic pointer they are VECTOR loads with
// a wait in front of everything that depends on them (the kernel's own stores might alias); none of these arrays is written by
// the kernel that reads it (the counts and states are ping-pong pairs), so they go through the constant address space.
#define JF_RV_CONST __attribute__((address_space(4)))
template <class T>
JF_DEV const T JF_RV_CONST *rv_const(const T *p) {
    return (const T JF_RV_CONST *)p;
}

// The 2 B1 samples of transform i of source s as the first pass's input -- x[t] = z[m] = x[2m] + j x[2m + 1], m = tid + r B1 / 8
// -- REQUESTED, not waited for: the persistent transform kernel asks for the next turn's samples before it works on this one's
// (round 6).  What lies before the call's first sample comes from the dry ring (written by earlier calls), the rest from the
// looped signal itself at the play position -- a batch call need not copy its own input anywhere.
template <int B1>
JF_DEV void big_fft_fetch(const ReverbBigParams &P, int i, int s, int tid, float2 (&x)[8]) {
    const float *ring = P.dryring + (size_t)s * P.Rn * B1;
    const int Rd = P.Rn * B1;
    const SrcSignal JF_RV_CONST *sgc = rv_const(P.dry + s);
    const float *sptr = sgc->ptr;
    const int L = sgc->length, dc0 = *rv_const(P.dry_count_in + s);
    const int rel0 = P.tr_rel_first + i * B1;
    // signal index of the first sample at or behind the call's start (one division per item, none per sample: the
    // signal is at least 1024 long, so 2 B1 samples wrap at most four times)
    const int first_in = rel0 < 0 ? 0 : rel0;
    const unsigned start = ((unsigned)dc0 + (unsigned)first_in) % (unsigned)L;
    // Where the 2 B1 samples lie is the same for the whole workgroup.  The two usual cases -- all of them one stretch of the
    // looped signal (transforms inside a batch call), all of them one stretch of the dry ring (the side stream's, and a
    // call's first transform) -- are eight 8-byte loads at a scalar base + 8 tid + 2 KB r: no per-sample index arithmetic.
    const float *stretch = nullptr;
    if (rel0 >= 0 && start + 2u * (unsigned)B1 <= (unsigned)L) {
        stretch = (const float *)__builtin_assume_aligned(sptr, 4) + start;
    } else if (rel0 + 2 * B1 <= 0) {
        int pos = P.dry_pos0 + rel0;
        pos = pos < 0 ? pos + Rd : pos;
        if (pos >= 0 && pos + 2 * B1 <= Rd) stretch = ring + pos;
    }
    const int m0 = tid < B1 / 8 ? tid : 0;
    if (stretch != nullptr) {
        const float JF_RV_GLOBAL *gs = (const float JF_RV_GLOBAL *)stretch;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const c2 pr = *reinterpret_cast<const c2 JF_RV_GLOBAL *>(gs + 2 * (m0 + r * (B1 / 8)));
            x[r] = make_float2(pr.x, pr.y);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int m = m0 + r * (B1 / 8);
            const int rel = rel0 + 2 * m;  // even; the ring / signal boundary (rel = 0) never splits a pair
            if (rel < 0) {
                int pos = P.dry_pos0 + rel;
                pos = pos < 0 ? pos + Rd : pos;
                x[r] = *reinterpret_cast<const float2 *>(ring + pos);
            } else {
                unsigned i0 = start + (unsigned)(rel - first_in);
#pragma unroll
                for (int w = 0; w < 4; w++) i0 = i0 >= (unsigned)L ? i0 - (unsigned)L : i0;
                const unsigned i1 = i0 + 1 == (unsigned)L ? 0u : i0 + 1;
                x[r] = make_float2(sptr[i0], sptr[i1]);
            }
        }
    }
}

#ifndef JF_RV_BIG_FFT_AHEAD
#define JF_RV_BIG_FFT_AHEAD 1  // the next turn's samples requested before this turn's passes (0: round 5's order)
#endif

// X_m of transform i of the launch (m = first + i) for source s: spectrum of the 2 B1 dry samples of big blocks m - 2, m - 1
// into fdl1 (packed: bin 0 = (X[0], X[B1])).  One transform per workgroup and turn; item g = i S + s (a source's transforms one
// after the other -- memory order, which the inverse gains 11 % from -- cost THIS kernel 15 %: 33.3 against 28.9 us, round 5).
// PERSISTENT (round 5): a workgroup takes the turns w = blockIdx.x, blockIdx.x + gridDim.x, ... -- the twiddles are loaded once
// per workgroup instead of once per transform (they were as many bytes through the vector L1 as the samples).
// SOFTWARE-PIPELINED (round 6): the kernel was a chain of waits -- the source's record (a vector load through a generic
// pointer), then its samples, then four passes, then the stores -- in which the vector unit was half idle and so was the
// memory system (profiles/r06/reverb_transforms.md).  Now the record is a scalar load and turn w + gridDim.x's samples are
// requested before turn w's passes begin: sixteen registers per thread hold them while the passes run: 29.0 -> 26.2-26.8 us per
// launch at config 5's batch shape (one box, A B A B; JF_RV_BIG_FFT_AHEAD = 0 is the order of round 5: 28.7).
template <int B1>
__global__ __launch_bounds__(kBigThreads) void reverb_big_fft_kernel(const ReverbBigParams P) {
    __shared__ float2 s_buf[1][rv_big_len(B1)];
    const int tid0 = threadIdx.x;
    BigTwiddles<B1, kBigThreads> tw;
    __shared__ float2 s_w8[BigTwiddles<B1, kBigThreads>::kW8Len];
    BigTwiddles<B1, kBigThreads>::stage_w8(P.tw1, s_w8, tid0);  // (the first pass ends with a barrier: staged before anybody reads)
    tw.load(P.tw1, tid0);
    const int n_items = P.n_tr * P.S;
    // the split's twiddles (they depend on the thread only): once per workgroup, no global load behind the last pass
#ifndef JF_RV_BIG_WIDE_STORE
#define JF_RV_BIG_WIDE_STORE 1  // the transforms' results as 16-byte stores (two adjacent values per lane); 0: 8-byte stores
#endif
    // bin q of the thread's u-th store: WIDE: q = 2 tid + (B1 / 4) u' + e with u = 2 u' + e (two adjacent bins, one 16-byte store);
    // else q = tid + 256 u
    auto split_bin = [](int tid, int u) { return JF_RV_BIG_WIDE_STORE ? 2 * tid + (u >> 1) * (2 * kBigThreads) + (u & 1) : tid + u * kBigThreads; };
    float2 wsplit[B1 / kBigThreads];
#pragma unroll
    for (int u = 0; u < B1 / kBigThreads; u++) wsplit[u] = P.tw1[split_bin(tid0, u)];
    // (i, s) of a turn, stepped without a division per turn
    const int step_i = (int)gridDim.x / P.S, step_s = (int)gridDim.x - step_i * P.S;
    int ni = (int)blockIdx.x / P.S, ns = (int)blockIdx.x - ni * P.S;
    float2 vn[8];
#if JF_RV_BIG_FFT_AHEAD
    if ((int)blockIdx.x < n_items) big_fft_fetch<B1>(P, ni, ns, tid0, vn);
#endif
#pragma unroll 1
    for (int turn = blockIdx.x; turn < n_items; turn += gridDim.x) {
        // (opaque per turn: the compiler otherwise hoists every LDS address of the four passes out of the loop: ~60 registers)
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int i = ni, s = ns;
        ns += step_s;
        ni += step_i;
        if (ns >= P.S) {
            ns -= P.S;
            ni++;
        }
        float2 v[1][8];
#if JF_RV_BIG_FFT_AHEAD
#pragma unroll
        for (int r = 0; r < 8; r++) v[0][r] = vn[r];
        if (turn + (int)gridDim.x < n_items) big_fft_fetch<B1>(P, ni, ns, tid, vn);
#else
        big_fft_fetch<B1>(P, i, s, tid, v[0]);
#endif
        // a call that puts its small transforms off: its last two big blocks of samples -- this source's last transform has them
        // in registers -- into the dry ring, its last block as `prev`, the play position behind it (ReverbBigParams::state_out)
        if (P.state_out && i == P.n_tr - 1 && tid < B1 / 8) {
            const int Rd = P.Rn * B1;
            const int rel0 = P.tr_rel_first + i * B1;
            float *ring_out = P.dryring_out + (size_t)s * Rd;
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int rel = rel0 + 2 * (tid + r * (B1 / 8));
                if (rel >= 0) {  // (what lies before the call is in the ring already)
                    int pos = P.dry_pos0 + rel;   // < 2 Rd: the ring is longer than a call
                    pos = pos >= Rd ? pos - Rd : pos;
                    *reinterpret_cast<float2 *>(ring_out + pos) = v[0][r];
                    if (rel >= P.call_samples - P.B) *reinterpret_cast<float2 *>(P.prev_out + (size_t)s * P.B + (rel - (P.call_samples - P.B))) = v[0][r];
                }
            }
            if (tid == 0) {
                const unsigned L = (unsigned)rv_const(P.dry + s)->length, dc0 = (unsigned)*rv_const(P.dry_count_in + s);
                P.dry_count_out[s] = (int)((dc0 + (unsigned)P.call_samples) % L);
            }
        }
        cfft_wg<B1, -1, kBigThreads>(v, s_buf, tw, s_w8, tid);
        {
            const float2 *Z = s_buf[0];
            const int slot = (P.tr_slot_first + i) % P.R1;
            float2 *out = P.fdl1 + ((size_t)s * P.R1 + slot) * B1;
            // Z[q] at rd0 + 288 u, Z[B1 - q] at the mirror thread's places counted down (rv_big_at(i) = i + (i >> 3); thread 0's
            // partner for u = 0 is itself)
            float2 xo[B1 / kBigThreads];
#pragma unroll
            for (int u = 0; u < B1 / kBigThreads; u++) {
                const int q = split_bin(tid, u);
                const float2 zk = Z[rv_big_at(q)];
                const float2 zm = Z[rv_big_at((B1 - q) & (B1 - 1))];
                const float2 e = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
                const float2 o = make_float2(0.5f * (zk.x - zm.x), 0.5f * (zk.y + zm.y));
                const float2 wo = rv_mulc(o, wsplit[u]);
                float2 x = make_float2(e.x + wo.y, e.y - wo.x);
                if (q == 0) {
                    x = make_float2(zk.x + zk.y, zk.x - zk.y);  // (X[0], X[B1]), both real
                    P.fdl1[(size_t)P.S * P.R1 * B1 + (size_t)s * P.R1 + slot] = x;  // compact copy of the packed pair
                }
                xo[u] = x;
            }
#if JF_RV_BIG_WIDE_STORE
#pragma unroll
            for (int u = 0; u < B1 / kBigThreads; u += 2)
                *reinterpret_cast<float4 *>(out + split_bin(tid, u)) = make_float4(xo[u].x, xo[u].y, xo[u + 1].x, xo[u + 1].y);
#else
#pragma unroll
            for (int u = 0; u < B1 / kBigThreads; u++) out[split_bin(tid, u)] = xo[u];
#endif
        }
        __syncthreads();  // the buffer is read out before the next turn's first pass writes it
    }
}

// Y_i = sum_{q < n_part} X_{anchor + i - q} H'_{h_first + q} for a tile of KB consecutive products of one source: a sliding
// window of KB spectra in registers, per partition ONE new X load and one H load feed KB multiply-accumulates (the scheme
// of reverb_mac_tiled_kernel).  A wave covers 64 bins (one per lane) of ALL partitions, in ascending order -- no reduction
// between waves, and the sums are the same whatever KB is.  The packed pair in bin 0 is carried as if it were complex;
// reverb_big_ifft_kernel recomputes it.
// Waves per workgroup: 8 for the single products of the side stream (a narrow launch beside the blocks' kernels:
// profiles/r04/rt_waves.md), 4 for the batch form's tiles of 16 -- the same 16 waves per CU in twice as many workgroups that
// start and end apart: 69.6 / 70.6 -> 67.4 / 68.2 us per launch at config 5's batch shape (profiles/r05/reverb_batch.md)
#ifndef JF_RV_BIG_MAC_WAVES
#define JF_RV_BIG_MAC_WAVES 4
#endif
constexpr int kBigMacWaves = 8, kBigMacWavesTiled = JF_RV_BIG_MAC_WAVES;
template <int B1, int KB>
JF_DEV void big_mac_item(const ReverbBigParams &P, int item) {
    constexpr int kWaves = KB == 1 ? kBigMacWaves : kBigMacWavesTiled;
    constexpr int WGS_PER_SPEC = B1 / (64 * kWaves);  // workgroups side by side over the bins
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slice = item % WGS_PER_SPEC;
    const int rest = item / WGS_PER_SPEC;
    const int n_tiles = (P.n_prod + KB - 1) / KB;
    const int s = rest / n_tiles, i0 = (rest - s * n_tiles) * KB;  // first product of the tile
    const char *fdl0 = reinterpret_cast<const char *>(P.fdl1 + (size_t)s * P.R1 * B1);
    const char *hp = reinterpret_cast<const char *>(P.hspec1 + (size_t)P.h_first * B1);
    unsigned voff = 8u * (unsigned)(64 * (kWaves * slice + wave) + lane);
    asm volatile("" : "+v"(voff));
    auto load_at = [&](const char *base) {
        const float2 *q = reinterpret_cast<const float2 *>(base + voff);
        return rv_v2{q->x, q->y};
    };
#if JF_RV_BIG_NT_X
    // the delay line is read once per launch: streamed past the caches' replacement order (non-temporal)
    auto load_x = [&](const char *base) {
        const rv_v2 *q = reinterpret_cast<const rv_v2 *>(base + voff);
        return __builtin_nontemporal_load(q);
    };
#else
    auto load_x = load_at;
#endif
    auto slot_of = [&](int u) {  // slot of X_{anchor + i0 + u} (u may be far in the past)
        int slot = (P.anchor_slot_first + i0 + u) % P.R1;
        return slot < 0 ? slot + P.R1 : slot;
    };
    rv_v2 acc[KB], xr[KB];
#pragma unroll
    for (int i = 0; i < KB; i++) acc[i] = rv_v2{0.f, 0.f};
    // the window: X(i), i = 1 .. KB - 1 (products past the end of the launch read whatever lies there and are not stored)
#pragma unroll
    for (int i = 1; i < KB; i++) xr[i] = i0 + i < P.n_prod ? load_x(fdl0 + (size_t)slot_of(i) * ((size_t)B1 * 8)) : rv_v2{0.f, 0.f};
    int xslot = slot_of(0);
#if JF_RV_BIG_SCALAR_MAC
    auto step = [&](int j) {  // j = q mod KB, a constant after unrolling
        const rv_v2 h = load_at(hp);
        xr[(KB - j) % KB] = load_at(fdl0 + (size_t)(unsigned)xslot * ((size_t)B1 * 8));  // X(-q)
        xslot = xslot == 0 ? P.R1 - 1 : xslot - 1;
        hp += (size_t)B1 * 8;
#pragma unroll
        for (int i = 0; i < KB; i++) {
            const rv_v2 x = xr[(i + KB - j) % KB];
            acc[i].x = __builtin_fmaf(x.x, h.x, acc[i].x);
            acc[i].y = __builtin_fmaf(x.x, h.y, acc[i].y);
        }
#pragma unroll
        for (int i = 0; i < KB; i++) {
            const rv_v2 x = xr[(i + KB - j) % KB];
            acc[i].x = __builtin_fmaf(-x.y, h.y, acc[i].x);
            acc[i].y = __builtin_fmaf(x.y, h.x, acc[i].y);
        }
    };
#else
    // The compiler does not move loads across the asm statements, so the loop fetches D steps ahead itself, in this order
    // (sched_barrier: nothing crosses)
    // (past the last group the H pointer stays on the last partition; the X slots just walk on round the ring)
    constexpr int D = JF_RV_BIG_PREFETCH < KB ? JF_RV_BIG_PREFETCH : KB;
    static_assert(KB % D == 0, "the queue index of a step is a constant after unrolling");
    rv_v2 hq[D], xq[D];
    const int n_steps = (P.n_part + KB - 1) / KB * KB;
    int q_pf = 0;
    auto fetch = [&](int d) {
        hq[d] = load_at(hp);
        xq[d] = load_x(fdl0 + (size_t)(unsigned)xslot * ((size_t)B1 * 8));  // X(-q)
        xslot = xslot == 0 ? P.R1 - 1 : xslot - 1;
        if (++q_pf < n_steps) hp += (size_t)B1 * 8;
    };
#pragma unroll
    for (int d = 0; d < D; d++) fetch(d);
    auto step = [&](int j) {  // j = q mod KB, a constant after unrolling
        const rv_v2 h = hq[j % D];
        xr[(KB - j) % KB] = xq[j % D];
        __builtin_amdgcn_sched_barrier(0);
        fetch(j % D);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < KB; i++) acc[i] = pfma_re(xr[(i + KB - j) % KB], h, acc[i]);
#pragma unroll
        for (int i = 0; i < KB; i++) acc[i] = pfma_im_rot(xr[(i + KB - j) % KB], h, acc[i]);
        __builtin_amdgcn_sched_barrier(0);
    };
#endif
    // whole groups of KB partitions, straight-line (loads of later steps may move above earlier multiply-accumulates): the
    // partitions behind the response's last one are zeros (hspec1), the delay-line slots they meet hold older spectra
    for (int q0 = 0; q0 < P.n_part; q0 += KB) {
#pragma unroll
        for (int j = 0; j < KB; j++) step(j);
    }
    float2 *y = P.ybig + ((size_t)s * P.n_prod + i0) * B1 + (voff >> 3);
#pragma unroll
    for (int i = 0; i < KB; i++)
        if (i0 + i < P.n_prod) y[(size_t)i * B1] = make_float2(acc[i].x, acc[i].y);
}

// The tiled form with the response's spectra through LDS (JF_RV_BIG_MAC_LDS_H): the waves of a workgroup take the SAME 64 bins
// of kBigMacWavesTiled different sources, so the H_q they multiply by are the same: each wave fetches a quarter of a group
// of KB partitions' H into LDS a group ahead, and a step reads its H with one ds_read_b64.  A wave's vector-memory
// instructions per step: 1.25 instead of 2 (the delay line's X, which nobody shares, and a quarter of an H).  Same sums in the
// same order as big_mac_item.
// 66.8 -> 59.2-60.5 us per launch at config 5's batch shape on one box, 70 -> 64.5 on another (profiles/r05/reverb_batch.md): a
// compute unit tracks a bounded number of vector-memory INSTRUCTIONS in flight, and half of them were loads of H out of the L2.
#ifndef JF_RV_BIG_MAC_LDS_H
#define JF_RV_BIG_MAC_LDS_H 1
#endif
template <int B1, int KB>
JF_DEV void big_mac_item_shared(const ReverbBigParams &P, int item) {
    constexpr int W = kBigMacWavesTiled, HPW = KB / W, kSlices = B1 / 64;
    static_assert(KB % W == 0 && KB > 1, "every wave fetches the same number of a group's partitions");
    __shared__ rv_v2 s_h[2][KB][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_tiles = (P.n_prod + KB - 1) / KB;
    const int slice = item % kSlices, rest = item / kSlices;
    const int s_raw = (rest / n_tiles) * W + wave;
    const bool live = s_raw < P.S;  // (a wave without a source works on the last one's data and stores nothing: the barriers need it)
    const int s = live ? s_raw : P.S - 1;
    const int i0 = (rest % n_tiles) * KB;
    const char *fdl0 = reinterpret_cast<const char *>(P.fdl1 + (size_t)s * P.R1 * B1);
    const char *hbase = reinterpret_cast<const char *>(P.hspec1 + (size_t)P.h_first * B1);
    unsigned voff = 8u * (unsigned)(64 * slice + lane);
    asm volatile("" : "+v"(voff));
    auto load_at = [&](const char *base) {
        const float2 *q = reinterpret_cast<const float2 *>(base + voff);
        return rv_v2{q->x, q->y};
    };
#if JF_RV_BIG_NT_X
    auto load_x = [&](const char *base) {
        const rv_v2 *q = reinterpret_cast<const rv_v2 *>(base + voff);
        return __builtin_nontemporal_load(q);
    };
#else
    auto load_x = load_at;
#endif
    auto slot_of = [&](int u) {
        int slot = (P.anchor_slot_first + i0 + u) % P.R1;
        return slot < 0 ? slot + P.R1 : slot;
    };
    const int n_groups = (P.n_part + KB - 1) / KB;
    rv_v2 hn[HPW];
#pragma unroll
    for (int k = 0; k < HPW; k++) hn[k] = load_at(hbase + (size_t)(wave * HPW + k) * ((size_t)B1 * 8));
    rv_v2 acc[KB], xr[KB];
#pragma unroll
    for (int i = 0; i < KB; i++) acc[i] = rv_v2{0.f, 0.f};
    // (one division for the tile's first slot, the others by stepping round the ring: the fifteen modulo sequences of the
    // window's slots were ~500 scalar instructions in front of every tile's first load)
    int xslot = slot_of(0);
    {
        int sl = xslot;
#pragma unroll
        for (int i = 1; i < KB; i++) {
            sl = sl + 1 == P.R1 ? 0 : sl + 1;
            xr[i] = i0 + i < P.n_prod ? load_x(fdl0 + (size_t)(unsigned)sl * ((size_t)B1 * 8)) : rv_v2{0.f, 0.f};
        }
    }
    constexpr int D = JF_RV_BIG_PREFETCH < KB ? JF_RV_BIG_PREFETCH : KB;
    static_assert(KB % D == 0, "the queue index of a step is a constant after unrolling");
    rv_v2 xq[D];
    auto fetch = [&](int d) {
        xq[d] = load_x(fdl0 + (size_t)(unsigned)xslot * ((size_t)B1 * 8));  // X(-q)
        xslot = xslot == 0 ? P.R1 - 1 : xslot - 1;
    };
#pragma unroll
    for (int d = 0; d < D; d++) fetch(d);
#pragma unroll
    for (int k = 0; k < HPW; k++) s_h[0][wave * HPW + k][lane] = hn[k];
    __syncthreads();
#pragma unroll 1
    for (int g = 0; g < n_groups; g++) {
        const bool more = g + 1 < n_groups;
        if (more) {
#pragma unroll
            for (int k = 0; k < HPW; k++) hn[k] = load_at(hbase + (size_t)((g + 1) * KB + wave * HPW + k) * ((size_t)B1 * 8));
        }
        const rv_v2 *hl = &s_h[g & 1][0][lane];
        rv_v2 hnext = hl[0];
#pragma unroll
        for (int j = 0; j < KB; j++) {
            const rv_v2 h = hnext;
            if (j + 1 < KB) hnext = hl[(j + 1) * 64];
            xr[(KB - j) % KB] = xq[j % D];
            __builtin_amdgcn_sched_barrier(0);
            fetch(j % D);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < KB; i++) acc[i] = pfma_re(xr[(i + KB - j) % KB], h, acc[i]);
#pragma unroll
            for (int i = 0; i < KB; i++) acc[i] = pfma_im_rot(xr[(i + KB - j) % KB], h, acc[i]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) {
#pragma unroll
            for (int k = 0; k < HPW; k++) s_h[(g + 1) & 1][wave * HPW + k][lane] = hn[k];
        }
        __syncthreads();
    }
    if (!live) return;
    float2 *y = P.ybig + ((size_t)s * P.n_prod + i0) * B1 + (voff >> 3);
#pragma unroll
    for (int i = 0; i < KB; i++)
        if (i0 + i < P.n_prod) y[(size_t)i * B1] = make_float2(acc[i].x, acc[i].y);
}

// Single products (one-block calls: the side stream) the same way -- JF_RV_BIG_MAC1_SHARED: the eight waves of a workgroup take
// the same 64 bins of eight sources; the response's spectra for those bins go to LDS once per workgroup (up to 64 partitions
// at a time); a wave then has nothing but the delay line's X to load, and loads it U partitions ahead (the plain form waits
// for every partition's two loads before it asks for the next: 42 round trips in a row).  Same sums in the same order.
// configs[4]'s shape: 82.9 -> 37.2 us per launch over 256 workgroups.  Used IN LINE (launch_big_products_t says why not
// beside the blocks).
// (JF_RV_BIG_MAC1_SHARED: jf_device.h)
#ifndef JF_RV_BIG_MAC1_AHEAD
#define JF_RV_BIG_MAC1_AHEAD 8
#endif
template <int B1>
JF_DEV void big_mac_single_shared(const ReverbBigParams &P, int item, rv_v2 (*s_h)[64], int &h_slice) {
    constexpr int W = kBigMacWaves, kSlices = B1 / 64, U = JF_RV_BIG_MAC1_AHEAD, kChunk = 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slice = item % kSlices, rest = item / kSlices;
    const int i0 = rest % P.n_prod;
    const int s_raw = (rest / P.n_prod) * W + wave;
    const bool live = s_raw < P.S;
    const int s = live ? s_raw : P.S - 1;
    const char *fdl0 = reinterpret_cast<const char *>(P.fdl1 + (size_t)s * P.R1 * B1);
    const char *hbase = reinterpret_cast<const char *>(P.hspec1 + (size_t)P.h_first * B1);
    unsigned voff = 8u * (unsigned)(64 * slice + lane);
    asm volatile("" : "+v"(voff));
    auto load_at = [&](const char *base) {
        const float2 *q = reinterpret_cast<const float2 *>(base + voff);
        return rv_v2{q->x, q->y};
    };
#if JF_RV_BIG_NT_X
    auto load_x = [&](const char *base) {
        const rv_v2 *q = reinterpret_cast<const rv_v2 *>(base + voff);
        return __builtin_nontemporal_load(q);
    };
#else
    auto load_x = load_at;
#endif
    int xslot = (P.anchor_slot_first + i0) % P.R1;
    xslot = xslot < 0 ? xslot + P.R1 : xslot;
    rv_v2 acc = rv_v2{0.f, 0.f};
    for (int c0 = 0; c0 < P.n_part; c0 += kChunk) {
        const int nc = P.n_part - c0 < kChunk ? P.n_part - c0 : kChunk;
        if (h_slice != slice || P.n_part > kChunk) {  // (a workgroup's items keep their bins when the grid is a multiple of kSlices)
            __syncthreads();                           // everybody is done with what is there
            for (int q = wave; q < nc; q += W) s_h[q][lane] = load_at(hbase + (size_t)(c0 + q) * ((size_t)B1 * 8));
            __syncthreads();
            h_slice = P.n_part > kChunk ? -1 : slice;
        }
        // the delay line's spectra U partitions at a time, the next U requested before these are used
        rv_v2 xa[U], xb[U];
        auto fetch = [&](rv_v2 (&x)[U], int q0) {
#pragma unroll
            for (int u = 0; u < U; u++)
                if (q0 + u < nc) {
                    x[u] = load_x(fdl0 + (size_t)(unsigned)xslot * ((size_t)B1 * 8));
                    xslot = xslot == 0 ? P.R1 - 1 : xslot - 1;
                }
        };
        auto use = [&](const rv_v2 (&x)[U], int q0) {
#pragma unroll
            for (int u = 0; u < U; u++)
                if (q0 + u < nc) {
                    const rv_v2 h = s_h[q0 + u][lane];
                    acc = pfma_re(x[u], h, acc);
                    acc = pfma_im_rot(x[u], h, acc);
                }
        };
        fetch(xa, 0);
        for (int q0 = 0; q0 < nc; q0 += 2 * U) {
            fetch(xb, q0 + U);
            use(xa, q0);
            fetch(xa, q0 + 2 * U);
            use(xb, q0 + U);
        }
    }
    if (!live) return;
    P.ybig[((size_t)s * P.n_prod + i0) * B1 + (voff >> 3)] = make_float2(acc.x, acc.y);
}

// One workgroup per item (64 bins per wave of one tile of one source) -- or, for single products on the side stream
// (mac_wgs > 0), that many workgroups taking the items in turn: a launch that does not fill the GPU's wave slots, so that the
// kernels of the blocks it runs beside find room at once (jf_engine_reverb.cpp: run_reverb_stage).
template <int B1, int KB>
__global__ __launch_bounds__(64 * (KB == 1 ? kBigMacWaves : kBigMacWavesTiled)) void reverb_big_mac_kernel(const ReverbBigParams P) {
    if constexpr (KB == 1) {
        constexpr int per_spec = B1 / (64 * kBigMacWaves);
        const int n_items = per_spec * P.n_prod * P.S;
#pragma unroll 1
        for (int item = blockIdx.x; item < n_items; item += gridDim.x) big_mac_item<B1, KB>(P, item);
    } else {
#if JF_RV_BIG_MAC_LDS_H
        big_mac_item_shared<B1, KB>(P, blockIdx.x);
#else
        big_mac_item<B1, KB>(P, blockIdx.x);
#endif
    }
}

// single products in line (big_mac_single_shared): one workgroup per 64 bins of eight sources, or fewer taking them in turn
template <int B1>
__global__ __launch_bounds__(64 * kBigMacWaves) void reverb_big_mac1_kernel(const ReverbBigParams P) {
    __shared__ rv_v2 s_h[64][64];
    int h_slice = -1;
    const int n_items = (B1 / 64) * P.n_prod * ((P.S + kBigMacWaves - 1) / kBigMacWaves);
#pragma unroll 1
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) big_mac_single_shared<B1>(P, item, s_h, h_slice);
}

// What a turn of the inverse kernel needs from memory, requested together and used in straight-line code: the product's
// spectrum Y[q], q = tid + r B1 / 8, and Y[B1 - q] (the second set of loads hits the lines the first one fetches); wave 0: the
// compact bin-0 pairs X0[anchor + i - q] of up to 128 partitions (more are fetched when they are used); the source's wet-ring
// position (a scalar load).
// The spectrum requested and untangled four values at a time (1): the second half's loads go into the registers the first half
// has freed -- 75 registers instead of 92, SIX workgroups per compute unit instead of five, one more memory round trip per
// turn: 24.5-24.7 -> 23.7-23.8 us per launch at config 5's batch shape (one box, A B A B; profiles/r06/reverb_transforms.md)
#ifndef JF_RV_BIG_IFFT_HALVES
#define JF_RV_BIG_IFFT_HALVES 1
#endif
#ifndef JF_RV_BIG_IFFT_WGS
#define JF_RV_BIG_IFFT_WGS 6
#endif
template <int B1>
struct BigIfftInput {
    float2 yk[8], ym[8];
    float2 x0[2];  // wave 0: lane's partitions q = lane, lane + 64
    int c0;        // SrcState::count of the source (to_wet)
    // the spectrum's values r0 .. r1 - 1 of the eight (JF_RV_BIG_IFFT_HALVES: requested and untangled four at a time)
    JF_DEV void fetch_y(const ReverbBigParams &P, int g, int tid, int r0, int r1) {
        const c2 JF_RV_GLOBAL *y = (const c2 JF_RV_GLOBAL *)(P.ybig + (size_t)g * B1);
        auto ld = [](const c2 JF_RV_GLOBAL *p) {
            const c2 t = *p;
            return make_float2(t.x, t.y);
        };
        const int q0 = tid < B1 / 8 ? tid : 0;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if (r < r0 || r >= r1) continue;
#if JF_RV_BIG_NT_Y
            const c2 t = __builtin_nontemporal_load(y + q0 + r * (B1 / 8));
            yk[r] = make_float2(t.x, t.y);
#else
            yk[r] = ld(y + q0 + r * (B1 / 8));
#endif
        }
#pragma unroll
        for (int r = 0; r < 8; r++)
            if (r >= r0 && r < r1) ym[r] = ld(y + ((B1 - (q0 + r * (B1 / 8))) & (B1 - 1)));
    }
    JF_DEV void fetch(const ReverbBigParams &P, int g, int tid, int r1 = 8) {
        const int s = g / P.n_prod, i = g - s * P.n_prod;  // items in memory order: a source's products one after the other
        auto ld = [](const c2 JF_RV_GLOBAL *p) {
            const c2 t = *p;
            return make_float2(t.x, t.y);
        };
        fetch_y(P, g, tid, 0, r1);
        if (tid < 64) {
            const c2 JF_RV_GLOBAL *xc = (const c2 JF_RV_GLOBAL *)(P.fdl1 + (size_t)P.S * P.R1 * B1 + (size_t)s * P.R1);
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const int q = tid + 64 * c;
                int slot = (P.anchor_slot_first + i - q) % P.R1;
                if (slot < 0) slot += P.R1;
                x0[c] = q < P.n_part ? ld(xc + slot) : make_float2(0.f, 0.f);
            }
        }
        c0 = P.to_wet ? rv_const(P.st_in + s)->count : 0;
    }
};

// Product i of source s -> B1 time samples: TAIL(m) into the fut ring, or FULL(m) straight into the wet ring.  One product per
// workgroup and turn (items g = s n_prod + i of the n_prod S in MEMORY order, turns w = blockIdx.x, blockIdx.x + gridDim.x, ...).
// Round 5 made the kernel persistent (twiddles once per workgroup).  ROUND 6: a turn was still a chain of waits -- the bin-0
// pairs (two dependent loads and a six-step reduce in front of everything), then EIGHT branches (q == 0 ?) each with its own
// three loads and its own wait, then the passes, then a vector load of the source's state in front of the stores.  Now a
// turn's whole input is requested at once (BigIfftInput: the spectrum twice, the bin-0 pairs, the wet-ring position as a
// scalar load), the untangling twiddles W^q and the response's bin-0 pairs live in registers, and the untangling is
// straight-line code (bin 0 by a select): 26.5 -> 24.3 us per launch at config 5's batch shape (one box, A B A B).  Measured and
// NOT kept (profiles/r06/reverb_transforms.md, profiles/r06_transform_experiments.patch): the next turn's input requested a turn
// ahead (126 registers, four workgroups per compute unit: 27.7 us), Y[B1 - q] from the mirror thread through LDS (28.2), W^q
// fetched per turn (27.3), the register count forced down to six workgroups per compute unit (spills: 25.8).
template <int B1>
__global__ __launch_bounds__(kBigThreads, JF_RV_BIG_IFFT_HALVES ? JF_RV_BIG_IFFT_WGS : 0) void reverb_big_ifft_kernel(const ReverbBigParams P) {
    __shared__ float2 s_buf[1][rv_big_len(B1)];
    const int tid0 = threadIdx.x;
    BigTwiddles<B1, kBigThreads> tw;
    __shared__ float2 s_w8[BigTwiddles<B1, kBigThreads>::kW8Len];
    BigTwiddles<B1, kBigThreads>::stage_w8(P.tw1, s_w8, tid0);  // (the first pass ends with a barrier: staged before anybody reads)
    tw.load(P.tw1, tid0);
    const int n_items = P.n_prod * P.S;
    static_assert(B1 / 8 == kBigThreads || B1 / 8 == kBigThreads / 2, "q = tid + r B1 / 8 of the threads tid < B1 / 8");
    // the untangling twiddles W^q of this thread's bins and the response's bin-0 pairs of wave 0's partitions: the same for
    // every item
    float2 wq[8], h0[2];
#pragma unroll
    for (int r = 0; r < 8; r++) wq[r] = P.tw1[(tid0 < B1 / 8 ? tid0 : 0) + r * (B1 / 8)];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const int q = (tid0 & 63) + 64 * c;
        h0[c] = q < P.n_part ? P.hspec1[(size_t)P.NP * B1 + P.h_first + q] : make_float2(0.f, 0.f);
    }
#pragma unroll 1
    for (int turn = blockIdx.x; turn < n_items; turn += gridDim.x) {
        // (opaque per turn: the compiler otherwise hoists every LDS and global address of the four passes out of the loop and
        // keeps them in ~60 registers)
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int g = turn;
        const int s = g / P.n_prod, i = g - s * P.n_prod;
        BigIfftInput<B1> in;
        in.fetch(P, g, tid, JF_RV_BIG_IFFT_HALVES ? 4 : 8);
        // the true packed pair of bin 0: sum_q X0[anchor + i - q] .* H0[h_first + q] from the compact copies, wave 0's lanes
        // over the partitions (thread 0, which owns bin 0 below, is one of them)
        float2 y0 = make_float2(0.f, 0.f);
        if (tid < 64) {
#pragma unroll
            for (int c = 0; c < 2; c++) {
                y0.x += in.x0[c].x * h0[c].x;
                y0.y += in.x0[c].y * h0[c].y;
            }
            if (P.n_part > 128) {  // (responses beyond 128 big partitions: 6 s at B1 = 2048)
                const float2 *xc = P.fdl1 + (size_t)P.S * P.R1 * B1 + (size_t)s * P.R1;
                const float2 *hc = P.hspec1 + (size_t)P.NP * B1 + P.h_first;
                for (int q = tid + 128; q < P.n_part; q += 64) {
                    int slot = (P.anchor_slot_first + i - q) % P.R1;
                    if (slot < 0) slot += P.R1;
                    const float2 x = xc[slot], h = hc[q];
                    y0.x += x.x * h.x;
                    y0.y += x.y * h.y;
                }
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                y0.x += __shfl_xor(y0.x, m);
                y0.y += __shfl_xor(y0.y, m);
            }
        }
        // Z[q] = E + j O with E = (Y[q] + conj Y[B1-q]) / 2, O = conj(W^q) (Y[q] - conj Y[B1-q]) / 2: the first pass's registers
        float2 v[1][8];
        auto untangle = [&](int r0, int r1) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                if (r < r0 || r >= r1) continue;
                const float2 yk = in.yk[r], ym = in.ym[r];
                const float2 e = make_float2(0.5f * (yk.x + ym.x), 0.5f * (yk.y - ym.y));
                const float2 d = make_float2(0.5f * (yk.x - ym.x), 0.5f * (yk.y + ym.y));
                const float2 o = rv_mul(d, wq[r]);
                v[0][r] = make_float2(e.x - o.y, e.y + o.x);
            }
        };
#if JF_RV_BIG_IFFT_HALVES
        untangle(0, 4);
        __builtin_amdgcn_sched_barrier(0);  // (the second half's loads go into the registers the first half has freed)
        in.fetch_y(P, g, tid, 4, 8);
        untangle(4, 8);
#else
        untangle(0, 8);
#endif
        if (tid == 0) v[0][0] = make_float2(0.5f * (y0.x + y0.y), 0.5f * (y0.x - y0.y));
        cfft_wg<B1, +1, kBigThreads, true>(v, s_buf, tw, s_w8, tid);  // (only z[m], m >= B1 / 2, is read below)
        {
            const float2 *zt = s_buf[0];
            // overlap-save: time samples B1 .. 2 B1 - 1 = z[m], m >= B1 / 2
            if (!P.to_wet) {
                float *fut = P.fut + (size_t)s * P.Fn * B1 + (size_t)((P.fut_first + i) % P.Fn) * B1;
#if JF_RV_BIG_WIDE_STORE
#pragma unroll
                for (int u = 0; u < B1 / 4 / kBigThreads; u++) {  // z[m], z[m + 1]: four consecutive samples, one 16-byte store
                    const int m = B1 / 2 + 2 * tid + u * (2 * kBigThreads);
                    const float2 a = zt[rv_big_at(m)], b = zt[rv_big_at(m + 1)];
                    *reinterpret_cast<float4 *>(fut + (2 * m - B1)) = make_float4(a.x, a.y, b.x, b.y);
                }
#else
#pragma unroll
                for (int u = 0; u < B1 / 2 / kBigThreads; u++) {
                    const int m = B1 / 2 + tid + u * kBigThreads;
                    *reinterpret_cast<float2 *>(fut + (2 * m - B1)) = zt[rv_big_at(m)];
                }
#endif
            } else {
                // the wet ring is a multiple of B long and is addressed block by block (mac_finish): a big block may wrap inside
                float *wet = P.wet + (size_t)s * P.Wr;
                const int lgB = 31 - __builtin_clz((unsigned)P.B);  // B is 64, 128 or 256
#if JF_RV_BIG_WIDE_STORE
#pragma unroll
                for (int u = 0; u < B1 / 4 / kBigThreads; u++) {  // four consecutive samples (never across a block: B >= 64)
                    const int m = B1 / 2 + 2 * tid + u * (2 * kBigThreads);
                    const int n = 2 * m - B1;                   // sample inside the big block (a multiple of 4)
                    const int kb = n >> lgB;                     // n / B
                    const int k = P.wet_k0 + P.M * i + kb;       // block of the call
                    int w0 = in.c0 + k * P.B;                    // c0 < Wr and k B < Wr: one conditional subtraction
                    w0 = w0 >= P.Wr ? w0 - P.Wr : w0;
                    const float2 a = zt[rv_big_at(m)], b = zt[rv_big_at(m + 1)];
                    *reinterpret_cast<float4 *>(wet + w0 + (n - kb * P.B)) = make_float4(a.x, a.y, b.x, b.y);
                }
#else
#pragma unroll
                for (int u = 0; u < B1 / 2 / kBigThreads; u++) {
                    const int m = B1 / 2 + tid + u * kBigThreads;
                    const int n = 2 * m - B1;                   // sample inside the big block
                    const int kb = n >> lgB;                     // n / B (once per pair)
                    const int k = P.wet_k0 + P.M * i + kb;       // block of the call
                    int w0 = in.c0 + k * P.B;                    // c0 < Wr and k B < Wr: one conditional subtraction
                    w0 = w0 >= P.Wr ? w0 - P.Wr : w0;
                    *reinterpret_cast<float2 *>(wet + w0 + (n - kb * P.B)) = zt[rv_big_at(m)];
                }
#endif
            }
        }
        __syncthreads();  // the buffer is read out before the next turn writes it
    }
}

// The H'_q: one workgroup per partition q: rfft([h[t0 + q B1 .. + B1), zeros]) * scale, packed.
template <int B1>
__global__ __launch_bounds__(kBigThreads) void reverb_big_ir_kernel(const float *__restrict__ ir, int n_ir, int t0, float scale,
                                                                   const float2 *__restrict__ tw1, float2 *__restrict__ hspec1,
                                                                   float2 *__restrict__ h0 /* [P1] compact bin-0 pairs */) {
    __shared__ float2 s_buf[1][rv_big_len(B1)];
    const int tid = threadIdx.x;
    const int q0 = blockIdx.x;
    BigTwiddles<B1, kBigThreads> tw;
    __shared__ float2 s_w8[BigTwiddles<B1, kBigThreads>::kW8Len];
    BigTwiddles<B1, kBigThreads>::stage_w8(tw1, s_w8, tid);
    tw.load(tw1, tid);
    float2 v[1][8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int m = (tid < B1 / 8 ? tid : 0) + r * (B1 / 8);
        const int n = 2 * m;
        const long long i0 = (long long)t0 + (long long)q0 * B1 + n, i1 = i0 + 1;
        const float a0 = (n < B1 && i0 < n_ir) ? ir[i0] : 0.0f;
        const float a1 = (n + 1 < B1 && i1 < n_ir) ? ir[i1] : 0.0f;
        v[0][r] = make_float2(a0, a1);
    }
    cfft_wg<B1, -1, kBigThreads>(v, s_buf, tw, s_w8, tid);
    const float2 *Z = s_buf[0];
    for (int q = tid; q < B1; q += kBigThreads) {
        const float2 zk = Z[rv_big_at(q)];
        const float2 zm = Z[rv_big_at((B1 - q) & (B1 - 1))];
        const float2 e = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
        const float2 o = make_float2(0.5f * (zk.x - zm.x), 0.5f * (zk.y + zm.y));
        const float2 wo = rv_mulc(o, tw1[q]);
        float2 x = make_float2(e.x + wo.y, e.y - wo.x);
        if (q == 0) x = make_float2(zk.x + zk.y, zk.x - zk.y);
        hspec1[(size_t)q0 * B1 + q] = make_float2(x.x * scale, x.y * scale);
        if (q == 0) h0[q0] = make_float2(x.x * scale, x.y * scale);
    }
}

// ------------------------------------------------------------- IR spectra --
// One wavefront per partition p: rfft([h_p (B taps), zeros]) * scale, packed.
template <int B>
__global__ __launch_bounds__(64) void reverb_ir_kernel(const float *__restrict__ ir, int n_ir, float scale,
                                                      const float2 *__restrict__ tw, float2 *__restrict__ hspec,
                                                      float2 *__restrict__ h0 /* [P]: the packed bin-0 pairs */) {
    __shared__ float2 s_buf[2 * B];
    const int lane = threadIdx.x;
    const int p = blockIdx.x;
    float2 *a = s_buf, *b = s_buf + B;
    for (int m = lane; m < B; m += 64) {
        const int n = 2 * m;
        const int i0 = p * B + n, i1 = i0 + 1;
        const float x0 = (n < B && i0 < n_ir) ? ir[i0] : 0.0f;
        const float x1 = (n + 1 < B && i1 < n_ir) ? ir[i1] : 0.0f;
        a[m] = make_float2(x0, x1);
    }
    JF_RV_SYNC();
    const float2 *Z = cfft_small<B, -1>(a, b, tw, lane);
    for (int q = lane; q < B; q += 64) {
        const float2 zk = Z[q];
        const float2 zm = Z[(B - q) & (B - 1)];
        const float2 e = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
        const float2 o = make_float2(0.5f * (zk.x - zm.x), 0.5f * (zk.y + zm.y));
        const float2 wo = rv_mulc(o, tw[q * (512 / B)]);
        float2 x = make_float2(e.x + wo.y, e.y - wo.x);
        if (q == 0) x = make_float2(zk.x + zk.y, zk.x - zk.y);
        hspec[(size_t)p * B + q] = make_float2(x.x * scale, x.y * scale);
        if (q == 0) h0[p] = make_float2(x.x * scale, x.y * scale);
    }
}

// ---------------------------------------------------------------- launchers --
hipError_t launch_reverb_ir(const float *d_ir, int n_ir, int P, int B, float scale, const float2 *d_tw,
                            float2 *d_hspec, hipStream_t st) {
    switch (B) {
    case 64: hipLaunchKernelGGL(reverb_ir_kernel<64>, dim3(P), dim3(64), 0, st, d_ir, n_ir, scale, d_tw, d_hspec, d_hspec + (size_t)P * 64); break;
    case 128: hipLaunchKernelGGL(reverb_ir_kernel<128>, dim3(P), dim3(64), 0, st, d_ir, n_ir, scale, d_tw, d_hspec, d_hspec + (size_t)P * 128); break;
    case 256: hipLaunchKernelGGL(reverb_ir_kernel<256>, dim3(P), dim3(64), 0, st, d_ir, n_ir, scale, d_tw, d_hspec, d_hspec + (size_t)P * 256); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

static void launch_big_transforms(const ReverbBigParams &P, hipStream_t st);
static void launch_big_products(const ReverbBigParams &P, hipStream_t st);
template <int B>
static void launch_fft(const ReverbParams &P, hipStream_t st);

// Form of stage B for blocks kb .. kb + kn - 1 by the amount of work: block-tiled when the tiles alone fill the GPU, else
// source-grouped, else (real-time calls) one workgroup per (block, source).
template <int B, int T, int KB>
static int launch_mac_range(const ReverbParams &P0, int kb, int kn, hipStream_t st) {
    if (kn <= 0) return 0;
    ReverbParams P = P0;
    P.kb = kb;
    P.kn = kn;
    const int force = P.mac_form;  // 0 = by size; 1, 2, 3 = tests pin one form
    const long long tiles = (long long)((kn + KB - 1) / KB) * P.S;
    const int form = (force == 3 || (force == 0 && kn >= KB && tiles >= 512)) ? 3
                     : (P.S % T == 0 && (force == 2 || (force == 0 && (long long)kn * P.S / T >= 512))) ? 2
                                                                                                       : 1;
    if (form == 3) hipLaunchKernelGGL((reverb_mac_tiled_kernel<B, KB>), dim3((kn + KB - 1) / KB * P.S), dim3(64 * kTileWaves), 0, st, P);
    else if (form == 2) hipLaunchKernelGGL((reverb_mac_kernel<B, T>), dim3(kn * (P.S / T)), dim3(64 * kMacWaves), 0, st, P);
    else hipLaunchKernelGGL((reverb_mac_kernel<B, 1>), dim3(kn * P.S), dim3(64 * kMacWaves), 0, st, P);
    return form;
}

// The whole stage for one call.  plan == null: uniform partitioning, every block through stage B.
template <int B, int T, int KB>
static int launch_stage(const ReverbParams &P, ReverbPlan *plan, hipStream_t st) {
    const bool big = plan && plan->big;
    if (big && plan->tail_early.n_prod > 0) launch_big_products(plan->tail_early, st);  // needs nothing of this call
    const bool few = !(P.S % T == 0 && (long long)P.S / T >= 512);  // (many sources: the source-grouped form, two kernels)
    if (plan && plan->head_fused) {
        plan->forms[0] = 5;  // the head runs inside the caller's real-time kernel
        return 5;
    }
    if (P.K == 1 && P.mac_form == 0 && few) {
        // a call of one block: stage A inside the kernel (pinning a form keeps two kernels)
        ReverbParams Q = P;
        Q.kb = 0;
        Q.kn = 1;
        hipLaunchKernelGGL((reverb_mac_kernel<B, 1, true>), dim3(P.S), dim3(64 * kMacWaves), 0, st, Q);
        if (big && plan->transforms.n_tr > 0) launch_big_transforms(plan->transforms, st);  // the block completed a big block
        if (plan) plan->forms[0] = 4;
        return 4;
    }
    launch_fft<B>(P, st);  // transforms of the blocks (where needed) and the dry ring
    int form = 0;
    if (big) {
        // The blocks in front of the call's first whole big block first: they read the fut ring at their big block's place,
        // which TAIL of the big block the call ENDS in may share (the ring has four places) -- and they need nothing of
        // what follows.
        // (A call without a whole big block inside has ONE range, whose blocks behind a boundary need TAIL_late: two
        // neighbouring big blocks, two places.)
        const bool split = plan->n_ranges > 1;
        if (split) {
            plan->forms[0] = launch_mac_range<B, T, KB>(P, plan->kb[0], plan->kn[0], st);
            if (plan->forms[0]) form = plan->forms[0];
        }
        if (plan->transforms.n_tr > 0) launch_big_transforms(plan->transforms, st);
        if (plan->middle.n_prod > 0) launch_big_products(plan->middle, st);
        if (plan->tail_late.n_prod > 0) launch_big_products(plan->tail_late, st);
        const int r = split ? 1 : 0;
        plan->forms[r] = launch_mac_range<B, T, KB>(P, plan->kb[r], plan->kn[r], st);
        if (plan->forms[r]) form = plan->forms[r];
    } else {
        form = launch_mac_range<B, T, KB>(P, 0, P.K, st);
        if (plan) plan->forms[0] = form;
    }
    return form;
}

template <int B>
static void launch_fft(const ReverbParams &P, hipStream_t st) {
    const int n = (P.K - (P.skip_hi - P.skip_lo)) * P.S;  // (a call of whole big blocks that puts its small transforms off: none)
    if (n > 0) hipLaunchKernelGGL(reverb_fft_kernel<B>, dim3((n + 3) / 4), dim3(256), 0, st, P);
}

// the small transforms a batch call put off, from the dry ring (ReverbParams::catchup)
hipError_t launch_reverb_catchup(const ReverbParams &P, hipStream_t st) {
    if (!P.catchup || P.K <= 0) return hipErrorInvalidValue;
    switch (P.B) {
    case 64: launch_fft<64>(P, st); break;
    case 128: launch_fft<128>(P, st); break;
    case 256: launch_fft<256>(P, st); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_reverb_big_ir(const float *d_ir, int n_ir, int t0, int P1, int B1, float scale, const float2 *d_tw1,
                                float2 *d_hspec1, hipStream_t st) {
    float2 *h0 = d_hspec1 + (size_t)(P1 + 16) * B1;  // behind the P1 partitions written here and 16 of zeros
    switch (B1) {
    case 1024: hipLaunchKernelGGL(reverb_big_ir_kernel<1024>, dim3(P1), dim3(kBigThreads), 0, st, d_ir, n_ir, t0, scale, d_tw1, d_hspec1, h0); break;
    case 2048: hipLaunchKernelGGL(reverb_big_ir_kernel<2048>, dim3(P1), dim3(kBigThreads), 0, st, d_ir, n_ir, t0, scale, d_tw1, d_hspec1, h0); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// workgroups of kBigThreads of `kernel` the current device holds at once (1024 if it will not say)
template <class K>
static int big_resident_wgs(K kernel) {
    int dev = 0, per_cu = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBigThreads, 0) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        return 1024;
    }
    return prop.multiProcessorCount * per_cu;
}

template <int B1>
static void launch_big_transforms_t(const ReverbBigParams &P, hipStream_t st) {
    const int n = P.n_tr * P.S;
    // persistent grid: what the device holds at once (the surplus of a larger grid would only queue)
    static const int resident = big_resident_wgs(reverb_big_fft_kernel<B1>);
    hipLaunchKernelGGL((reverb_big_fft_kernel<B1>), dim3(std::min(n, resident)), dim3(kBigThreads), 0, st, P);
}
// products of one launch (tiles of 16 when there are several, else one by one) and their inverse transforms
template <int B1>
static void launch_big_products_t(const ReverbBigParams &P, hipStream_t st) {
    [[maybe_unused]] constexpr int per_spec = B1 / (64 * kBigMacWaves);
    [[maybe_unused]] constexpr int per_spec_tiled = B1 / (64 * kBigMacWavesTiled);
    if (P.n_prod >= 4) {
        const int tiles = (P.n_prod + 15) / 16;
#if JF_RV_BIG_MAC_LDS_H
        const int grid = (B1 / 64) * tiles * ((P.S + kBigMacWavesTiled - 1) / kBigMacWavesTiled);
#else
        const int grid = per_spec_tiled * tiles * P.S;
#endif
        hipLaunchKernelGGL((reverb_big_mac_kernel<B1, 16>), dim3(grid), dim3(64 * kBigMacWavesTiled), 0, st, P);
    } else {
        // beside the blocks of one-block calls (mac_wgs > 0: the side stream) the plain form, narrow: 83 us per launch at
        // configs[4], spread thinly over four blocks -- the shared form is done in 37 us and the block it meets pays for it
        // (mean 23.9 against 24.5 us per block, p99 38 against 35.5: profiles/r05/reverb_realtime.md); in line the shared form
        if (JF_RV_BIG_MAC1_SHARED && P.mac_wgs == 0) {
            const int n_items = (B1 / 64) * P.n_prod * ((P.S + kBigMacWaves - 1) / kBigMacWaves);
            hipLaunchKernelGGL((reverb_big_mac1_kernel<B1>), dim3(n_items), dim3(64 * kBigMacWaves), 0, st, P);
        } else {
            const int n_items = per_spec * P.n_prod * P.S;
            const int wgs = P.mac_wgs > 0 && P.mac_wgs < n_items ? P.mac_wgs : n_items;
            hipLaunchKernelGGL((reverb_big_mac_kernel<B1, 1>), dim3(wgs), dim3(64 * kBigMacWaves), 0, st, P);
        }
    }
    const int n = P.n_prod * P.S;
    // persistent grid: what the device holds at once (the surplus of a larger grid would only queue)
    static const int resident = big_resident_wgs(reverb_big_ifft_kernel<B1>);
    hipLaunchKernelGGL((reverb_big_ifft_kernel<B1>), dim3(std::min(n, resident)), dim3(kBigThreads), 0, st, P);
}
static void launch_big_transforms(const ReverbBigParams &P, hipStream_t st) {
    switch (P.B1) {
    case 1024: launch_big_transforms_t<1024>(P, st); break;
    case 2048: launch_big_transforms_t<2048>(P, st); break;
    default: break;
    }
}
static void launch_big_products(const ReverbBigParams &P, hipStream_t st) {
    switch (P.B1) {
    case 1024: launch_big_products_t<1024>(P, st); break;
    case 2048: launch_big_products_t<2048>(P, st); break;
    default: break;
    }
}

// The big partitions' work of one-block calls on the engine's side stream (jf_engine_reverb.cpp: run_reverb_stage): X_m of the big
// block just completed, then the products of a TAIL with their inverse transform.
hipError_t launch_reverb_big_side(const ReverbBigParams *transforms, const ReverbBigParams *products, hipStream_t st) {
    if (transforms && transforms->n_tr > 0) launch_big_transforms(*transforms, st);
    if (products && products->n_prod > 0) launch_big_products(*products, st);
    return hipGetLastError();
}

// The twiddle pack of BigTwiddles<B1, 256> behind the circle T2[0 .. 2 B1): pack[k] = T2[big_twiddle_pack_index(B1, k)].
int big_twiddle_pack_len(int B1) { return 504 + (B1 / 512 - 1) * 512; }
int big_twiddle_pack_index(int B1, int k) {
    if (k < 56) return (k / 8 + 1) * (k % 8) * (2 * B1 / 64);             // second pass: exp(2 pi i r k / 64), r = 1 .. 7, k < 8
    if (k < 504) return ((k - 56) / 64 + 1) * ((k - 56) % 64) * (2 * B1 / 512);  // third: exp(2 pi i r k / 512), k < 64
    const int RL = B1 / 512;
    return ((k - 504) / 512 + 1) * ((k - 504) % 512) * (2 * B1 / (512 * RL));   // last: exp(2 pi i r k / (512 RL)), k < 512
}

// form_used: 1, 2, 3 = form of stage B (after reverb_fft_kernel), 4 = form 1 with stage A fused in (no reverb_fft_kernel),
// 0 = no block went through stage B (a batch call whose blocks the big partitions formed alone)
hipError_t launch_reverb(const ReverbParams &P, ReverbPlan *plan, hipStream_t st, int *form_used) {
    int form = 0;
    switch (P.B) {
    case 64: form = launch_stage<64, 4, 16>(P, plan, st); break;
    case 128: form = launch_stage<128, 4, 16>(P, plan, st); break;
    case 256: form = launch_stage<256, 2, 8>(P, plan, st); break;
    default: return hipErrorInvalidValue;
    }
    if (form_used) *form_used = form;
    return hipGetLastError();
}

}  // namespace jf
