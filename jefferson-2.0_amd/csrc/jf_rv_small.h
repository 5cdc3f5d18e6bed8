// jf_rv_small.h -- device code of the reverb stage that more than one translation unit needs: the small (one-wavefront)
// transforms, stage A of a block (rv_forward) and the last step of stage B (mac_finish).  jf_reverb.hip builds its kernels
// from them; jf_kernels.hip builds the one-launch real-time kernel's reverb head from them (rv_head_wave below: the head of a
// non-uniformly partitioned response, by the wave that then spatialises the source).  Included inside namespace jf.
#pragma once

#define JF_DEV __device__ __forceinline__

namespace {

JF_DEV float2 rv_add(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
JF_DEV float2 rv_sub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
JF_DEV float2 rv_mul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
JF_DEV float2 rv_mulc(float2 a, float2 b) { return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }

// acc += x * h (complex): four f32 FMAs.  The packed form (two v_pk_fma_f32 with op_sel broadcasts, pcmac of
// jf_packed.h; JF_RV_SCALAR_MAC=0) runs the tiled kernel in the same time (profiles/r02_experiments.md section 4) but
// needs even-aligned register pairs for the window: 128 VGPRs and 20 B of scratch where this form takes 124 and none.
typedef c2 rv_v2;
#ifndef JF_RV_SCALAR_MAC
#define JF_RV_SCALAR_MAC 1
#endif
[[maybe_unused]] JF_DEV void rv_cmac(rv_v2 &acc, rv_v2 x, rv_v2 h) {
#if JF_RV_SCALAR_MAC
    acc.x = __builtin_fmaf(x.x, h.x, acc.x);
    acc.y = __builtin_fmaf(x.x, h.y, acc.y);
    acc.x = __builtin_fmaf(-x.y, h.y, acc.x);
    acc.y = __builtin_fmaf(x.y, h.x, acc.y);
#else
    acc = pcmac(x, h, acc);
#endif
}

// The big partitions' products (reverb_big_mac_kernel): the four FMAs of a complex product as two v_pk_fma_f32 with op_sel
// broadcasts -- the same operations per product and component in the same order, so the sums are the same bit for bit as
// with four v_fma_f32 (-DJF_RV_BIG_SCALAR_MAC=1).  Round 4, config 5's batch shape, rocprofv3, 320 launches each, twice:
//   four v_fma_f32, loads scheduled by the compiler           77.4 us   (98 registers)
//   two v_pk_fma_f32, loads left where the compiler puts them 120.2 us  (it does not move loads across the asm statements:
//                                                                        every step waited for its own loads)
//   two v_pk_fma_f32, loads JF_RV_BIG_PREFETCH steps ahead    72.6 / 71.3 / 78.1 us for 2 / 4 / 8 steps (86 registers at 4)
// What is left is the stream itself: 247 MB of delay line and 67 MB of products per launch at 4.7 TB/s.
#ifndef JF_RV_BIG_SCALAR_MAC
#define JF_RV_BIG_SCALAR_MAC 0
#endif
#ifndef JF_RV_BIG_PREFETCH
#define JF_RV_BIG_PREFETCH 4
#endif
// reverb_big_mac_kernel: the delay line's spectra -- 247 MB per launch at config 5's batch shape, each read ONCE -- as non-temporal
// loads.  The product kernel itself gains 1-4 % (69.3 against 70.4-72.8 us); every kernel BEHIND it gains more, because the
// stream no longer washes their data out of the L2 and the Infinity Cache: inverse transforms 29.5 -> 26.3 us, forward 28.2 ->
// 25.9, small transforms 12.4 -> 10.8, the spatialiser 136.8 -> 132.9; the step 0.2877-0.2912 -> 0.2744 ms (one box, one
// call, A B C A B: profiles/r05/reverb_batch.md)
#ifndef JF_RV_BIG_NT_X
#define JF_RV_BIG_NT_X 1
#endif
#ifndef JF_RV_BIG_NT_Y
#define JF_RV_BIG_NT_Y 0  // the inverse transforms' reads of the products (each read once, by them)
#endif
// reverb_big_fft_kernel: the split's twiddles loaded before the transform instead of behind its last pass: 45.4 -> 42.4 us per
// launch at config 5's batch shape (rocprofv3, 320 launches, twice; -DJF_RV_BIG_SPLIT_TW_EARLY=0 is the old form)
#ifndef JF_RV_BIG_SPLIT_TW_EARLY
#define JF_RV_BIG_SPLIT_TW_EARLY 1
#endif
[[maybe_unused]] JF_DEV c2 pfma_re(c2 x, c2 h, c2 acc) {  // acc + (x.re h.re, x.re h.im)
    c2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(x), "v"(h), "v"(acc));
    return r;
}
[[maybe_unused]] JF_DEV c2 pfma_im_rot(c2 x, c2 h, c2 acc) {  // acc + (-x.im h.im, x.im h.re)
    c2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(x), "v"(h), "v"(acc));
    return r;
}

// pointers loaded from a table are generic to the compiler (flat loads): the signals and rings are global memory
#define JF_RV_GLOBAL __attribute__((address_space(1)))

// Wave-private LDS hand-off (see jf_kernels.hip)
#define JF_RV_SYNC()                                            \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  \
        __builtin_amdgcn_wave_barrier();                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  \
    } while (0)

// ---- in-register butterflies (natural order in and out); DIR = -1 forward, +1 inverse
template <int DIR>
JF_DEV float2 rv_muli(float2 v) {  // v * (DIR * i)
    return DIR > 0 ? make_float2(-v.y, v.x) : make_float2(v.y, -v.x);
}
template <int DIR>
JF_DEV void rv_fft2(float2 (&v)[2]) {
    const float2 a = v[0], b = v[1];
    v[0] = rv_add(a, b);
    v[1] = rv_sub(a, b);
}
template <int DIR>
JF_DEV void rv_fft4(float2 &a0, float2 &a1, float2 &a2, float2 &a3) {
    const float2 t0 = rv_add(a0, a2), t1 = rv_sub(a0, a2);
    const float2 t2 = rv_add(a1, a3), t3 = rv_muli<DIR>(rv_sub(a1, a3));
    a0 = rv_add(t0, t2);
    a1 = rv_add(t1, t3);
    a2 = rv_sub(t0, t2);
    a3 = rv_sub(t1, t3);
}
template <int DIR>
JF_DEV void rv_fft4(float2 (&v)[4]) {
    rv_fft4<DIR>(v[0], v[1], v[2], v[3]);
}
template <int DIR>
JF_DEV void rv_fft8(float2 (&v)[8]) {
    constexpr float h = 0.70710678118654752440f;
    float2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    float2 o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    rv_fft4<DIR>(e0, e1, e2, e3);
    rv_fft4<DIR>(o0, o1, o2, o3);
    // o_k *= exp(DIR 2 pi i k / 8)
    o1 = DIR > 0 ? make_float2(h * (o1.x - o1.y), h * (o1.x + o1.y)) : make_float2(h * (o1.x + o1.y), h * (o1.y - o1.x));
    o2 = rv_muli<DIR>(o2);
    o3 = DIR > 0 ? make_float2(-h * (o3.x + o3.y), h * (o3.x - o3.y)) : make_float2(h * (o3.y - o3.x), -h * (o3.x + o3.y));
    v[0] = rv_add(e0, o0);
    v[4] = rv_sub(e0, o0);
    v[1] = rv_add(e1, o1);
    v[5] = rv_sub(e1, o1);
    v[2] = rv_add(e2, o2);
    v[6] = rv_sub(e2, o2);
    v[3] = rv_add(e3, o3);
    v[7] = rv_sub(e3, o3);
}
template <int R, int DIR>
JF_DEV void rv_fftR(float2 (&v)[R]) {
    if constexpr (R == 8) rv_fft8<DIR>(v);
    else if constexpr (R == 4) rv_fft4<DIR>(v);
    else rv_fft2<DIR>(v);
}

// LDS index of element i of a transform buffer.  PAD: one float2 of padding per 8 -- the passes write at strides of 8 and
// 64 elements, which without it land on one or two banks (a 32-way conflict in the first pass); with it a half-wave's 8-byte
// stores cover every bank twice, which is what 256 bytes take anyway.
template <bool PAD>
JF_DEV int rv_at(int i) {
    return PAD ? i + (i >> 3) : i;
}
template <bool PAD>
constexpr int rv_buf_len(int n) {
    return PAD ? n + n / 8 : n;
}

// One pass of radix R of the Stockham autosort FFT of NPT points, a -> b, by NT threads (tid of them): sub-transforms of
// length Ns in, Ns R out.  T[j] = exp(+2 pi i j / TN), j < TN (a full circle), TN a multiple of Ns R.
template <int NPT, int R, int DIR, int NT, int TN, bool PAD>
JF_DEV void stockham_pass(const float2 *a, float2 *b, const float2 *__restrict__ T, int Ns, int tid) {
    for (int j = tid; j < NPT / R; j += NT) {
        const int k = j & (Ns - 1);
        float2 v[R];
#pragma unroll
        for (int r = 0; r < R; r++) v[r] = a[rv_at<PAD>(j + r * (NPT / R))];
        if (Ns > 1) {  // (the first pass's twiddles are all 1)
            const int t1 = k * (TN / (Ns * R));  // exp(+-2 pi i r k / (Ns R)) = T[r t1]
#pragma unroll
            for (int r = 1; r < R; r++) {
                const float2 w = T[r * t1];
                v[r] = DIR > 0 ? rv_mul(v[r], w) : rv_mulc(v[r], w);
            }
        }
        rv_fftR<R, DIR>(v);
        const int j0 = (j - k) * R + k;
#pragma unroll
        for (int r = 0; r < R; r++) b[rv_at<PAD>(j0 + r * Ns)] = v[r];
    }
}

// Complex FFT of NPT points (a power of two, >= 64) in LDS, ping-pong between a and b: radix-8 passes, then one of radix 4
// or 2 for what is left.  WG = false: one wavefront (JF_RV_SYNC between the passes); true: a workgroup of NT threads
// (a barrier between the passes; the buffers are padded, rv_at).  Returns the buffer holding the result in natural order.
template <int NPT, int DIR, int NT, int TN, bool WG>
JF_DEV float2 *cfft_lds(float2 *a, float2 *b, const float2 *__restrict__ T, int tid) {
    int Ns = 1;
    auto sync = [&]() {
        if constexpr (WG) __syncthreads();
        else JF_RV_SYNC();
    };
#pragma unroll 1
    for (; Ns * 8 <= NPT; Ns *= 8) {
        stockham_pass<NPT, 8, DIR, NT, TN, WG>(a, b, T, Ns, tid);
        sync();
        float2 *t = a;
        a = b;
        b = t;
    }
    constexpr int kLog = __builtin_ctz(NPT) % 3;  // NPT = 8^n 2^kLog
    if constexpr (kLog == 2) {
        stockham_pass<NPT, 4, DIR, NT, TN, WG>(a, b, T, NPT / 4, tid);
        sync();
        return b;
    } else if constexpr (kLog == 1) {
        stockham_pass<NPT, 2, DIR, NT, TN, WG>(a, b, T, NPT / 2, tid);
        sync();
        return b;
    }
    return a;
}

// by one wavefront; tw = exp(+2 pi i j / 1024), j < 1024
template <int NPT, int DIR>
JF_DEV float2 *cfft_small(float2 *a, float2 *b, const float2 *__restrict__ tw, int lane) {
    return cfft_lds<NPT, DIR, 64, 1024, false>(a, b, tw, lane);
}

}  // namespace

// ---------------------------------------------------------------- stage A --
// Spectrum of [x_{k-1}, x_k] of source s into the FDL, by one wavefront (a, b: 2 x B float2 of LDS).  x0: also left in
// LDS for the caller (the real-time form of stage B uses it at once), or null.
// tw: exp(+2 pi i j / 1024), j < 1024 -- P.tw, or a copy of it in LDS (every pass of the transform reads it).
template <int B>
JF_DEV void rv_forward(const ReverbParams &P, int k, int s, float2 *a, float2 *b, int lane, float2 *x0,
                       const float2 *tw) {
    const SrcSignal sg = P.dry[s];
    const int L = sg.length;  // >= 1024 (tiled / zero buffer)
    const int dc0 = P.dry_count_in[s];
    // (dc0 < L < 2^31 and K B < 2^31: the sums fit 32 unsigned bits -- a 32-bit remainder is a fifth of the 64-bit one's
    // instructions, and this is the head of the transformer's chain in one-block calls)
    const int cur0 = (int)(((unsigned)dc0 + (unsigned)(k * B)) % (unsigned)L);
    const int prv0 = (int)(((unsigned)dc0 + (unsigned)((k > 0 ? k - 1 : 0) * B)) % (unsigned)L);
    const float *prev_state = P.prev_in + (size_t)s * B;
    if (P.catchup) {
        // the block's and its predecessor's samples from the dry ring (a multiple of B long: a block never wraps inside)
        const float *ring = P.dryring + (size_t)s * P.Rd;
        int pc = (P.dry_pos0 + k * B) % P.Rd;
        int pp = pc - B;
        pp = pp < 0 ? pp + P.Rd : pp;
        for (int m = lane; m < B; m += 64) {
            const int n = 2 * m;
            a[m] = *reinterpret_cast<const float2 *>(ring + (n < B ? pp + n : pc + (n - B)));
        }
    } else {
    if (k >= P.copy_lo && k < P.copy_hi) {
        // a block whose output the big partitions form directly (ReverbBigParams: FULL) and whose spectrum nobody will read:
        // its samples go to the dry ring, nothing else (never the call's last block, which leaves the state)
        float *ring = P.dryring + (size_t)s * P.Rd + (size_t)((P.dry_pos0 + k * B) % P.Rd);
        for (int n = lane; n < B; n += 64) {
            int idx = cur0 + n;
            idx = idx >= L ? idx - L : idx;
            ring[n] = sg.ptr[idx];
        }
        return;
    }
    // z[m] = x[2m] + j x[2m+1] over x = [previous block, current block]
    for (int m = lane; m < B; m += 64) {
        float xv[2];
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int n = 2 * m + c;
            float v;
            if (n < B) {
                if (k == 0) {
                    v = prev_state[n];
                } else {
                    int idx = prv0 + n;
                    idx = idx >= L ? idx - L : idx;
                    v = sg.ptr[idx];
                }
            } else {
                int idx = cur0 + (n - B);
                idx = idx >= L ? idx - L : idx;
                v = sg.ptr[idx];
            }
            xv[c] = v;
        }
        a[m] = make_float2(xv[0], xv[1]);
        // non-uniform partitioning: the block's dry samples by absolute time (the big-partition transform reads them there;
        // the ring length is a multiple of B, so a block never wraps inside)
        if (P.dryring != nullptr && 2 * m >= B)
            *reinterpret_cast<float2 *>(P.dryring + (size_t)s * P.Rd + (size_t)((P.dry_pos0 + k * B) % P.Rd) + (2 * m - B)) =
                make_float2(xv[0], xv[1]);
    }
    if (k == P.K - 1) {
        // state for the next call: the last dry block and the advanced play position
        float *po = P.prev_out + (size_t)s * B;
        for (int n = lane; n < B; n += 64) {
            int idx = cur0 + n;
            idx = idx >= L ? idx - L : idx;
            po[n] = sg.ptr[idx];
        }
        if (lane == 0) P.dry_count_out[s] = (int)(((unsigned)dc0 + (unsigned)(P.K * B)) % (unsigned)L);
    }
    }
    JF_RV_SYNC();
    const float2 *Z = cfft_small<B, -1>(a, b, tw, lane);
    // real-FFT split: X[k] = E + (-i) W^k O, W = exp(-2 pi i / 2B)
    float2 *out = P.fdl + ((size_t)s * P.Rg + (size_t)((P.head + k) % P.Rg)) * B;
    for (int q = lane; q < B; q += 64) {
        const float2 zk = Z[q];
        const float2 zm = Z[(B - q) & (B - 1)];
        const float2 e = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
        const float2 o = make_float2(0.5f * (zk.x - zm.x), 0.5f * (zk.y + zm.y));
        const float2 wo = rv_mulc(o, tw[q * (512 / B)]);
        float2 x = make_float2(e.x + wo.y, e.y - wo.x);
        if (q == 0) {
            x = make_float2(zk.x + zk.y, zk.x - zk.y);  // (X[0], X[B]), both real
            // compact copy of the packed pair for the block-tiled form (fdl0[s][slot], behind the spectra)
            P.fdl[(size_t)P.S * P.Rg * B + (size_t)s * P.Rg + (size_t)((P.head + k) % P.Rg)] = x;
        }
        out[q] = x;
        if (x0) x0[q] = x;
    }
}

// Last step of stage B for one (block k, source s), by one wavefront: add the NW partial spectra
// (red, red + stride, ...), untangle the packed real spectrum, inverse FFT, write the wet block.
// FIX0: the partial spectra treated the packed pair in bin 0 as a complex number; the true pair
// (sum_p X0[k-p] .* H0[p]) is formed here from the compact copies, lanes over the partitions.
// tw: exp(+2 pi i j / 1024), j < 1024 -- P.tw, or a copy of it in LDS
template <int B, int NW, bool FIX0 = false>
JF_DEV void mac_finish(const float2 *red, int stride, float2 *fftbuf, const ReverbParams &P, int s, int k, int lane,
                       const float2 *tw) {
    // Y[q] (packed), then Z[q] = E + j O with E = (Y[q] + conj Y[B-q])/2, O = conj(W^q) (Y[q] - conj Y[B-q])/2
    float2 *ybuf = fftbuf, *zbuf = fftbuf + B;
    for (int q = lane; q < B; q += 64) {
        float2 a = red[q];
#pragma unroll
        for (int w = 1; w < NW; w++) a = rv_add(a, red[(size_t)w * stride + q]);
        ybuf[q] = a;
    }
    if (FIX0) {
        const float2 *x0 = P.fdl + (size_t)P.S * P.Rg * B + (size_t)s * P.Rg;
        const float2 *h0 = P.hspec + (size_t)P.P * B;
        float2 y0 = make_float2(0.f, 0.f);
        int slot = (P.head + k - lane) % P.Rg;
        if (slot < 0) slot += P.Rg;
        for (int p = lane; p < P.P; p += 64) {
            const float2 x = x0[slot], h = h0[p];
            y0.x += x.x * h.x;
            y0.y += x.y * h.y;
            slot -= 64;
            if (slot < 0) slot += P.Rg;
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            y0.x += __shfl_xor(y0.x, m);
            y0.y += __shfl_xor(y0.y, m);
        }
        JF_RV_SYNC();  // every lane's sum of partials is in ybuf before lane 0 replaces bin 0
        if (lane == 0) ybuf[0] = y0;
    }
    JF_RV_SYNC();
    for (int q = lane; q < B; q += 64) {
        const float2 yk = ybuf[q];
        const float2 ym = ybuf[(B - q) & (B - 1)];
        float2 z;
        if (q == 0) {
            z = make_float2(0.5f * (yk.x + yk.y), 0.5f * (yk.x - yk.y));  // E0 + j O0 from (Y[0], Y[B])
        } else {
            const float2 e = make_float2(0.5f * (yk.x + ym.x), 0.5f * (yk.y - ym.y));
            const float2 d = make_float2(0.5f * (yk.x - ym.x), 0.5f * (yk.y + ym.y));
            const float2 o = rv_mul(d, tw[q * (512 / B)]);  // W^-q = exp(+2 pi i q / 2B)
            z = make_float2(e.x - o.y, e.y + o.x);
        }
        zbuf[q] = z;
    }
    JF_RV_SYNC();
    const float2 *zt = cfft_small<B, +1>(zbuf, ybuf, tw, lane);
    // overlap-save: time samples B..2B-1 = z[m], m >= B/2 (even, odd interleaved)
    const int c0 = P.st_in[s].count;  // where the spatialiser will read the first new sample
    float *wet = P.wet + (size_t)s * P.Wr;
    const int w0 = (int)(((unsigned)c0 + (unsigned)(k * B)) % (unsigned)P.Wr);
    // non-uniform partitioning: + what the partitions behind the head contribute to this block (reverb_big_*)
    const float *fut = P.fut != nullptr ? P.fut + (size_t)s * P.F + (size_t)((P.fut_pos0 + k * B) % P.F) : nullptr;
    for (int m = B / 2 + lane; m < B; m += 64) {
        float2 v = zt[m];
        const int n = 2 * m - B;  // 0..B-2, even; the ring length is a multiple of B, w0 too
        if (fut != nullptr) {
            const float2 f = *reinterpret_cast<const float2 *>(fut + n);
            v = make_float2(v.x + f.x, v.y + f.y);
        }
        *reinterpret_cast<float2 *>(wet + w0 + n) = v;
    }
}


// The whole reverb stage of ONE block of ONE source by ONE wavefront (round 5; the real-time shape with a short head: the
// 2 M partitions of a non-uniformly partitioned response, or a short response as it is): stage A (rv_forward: the block's
// spectrum into the delay line, its samples into the dry ring, the play position), the multiply-accumulate over the P
// partitions IN ORDER p = 0 .. P - 1 (a lane owns B / 64 consecutive bins: one 16-byte load per spectrum at B = 128), and
// mac_finish (untangle, inverse transform, + the big partitions' share from the fut ring, the block into the wet ring).
// The caller -- rt_block_kernel's wave of source s -- then spatialises that block: one launch per audio block instead of two,
// no kernel boundary between the stages (mean 30.5 -> 26 us for config 5's 256 sources; profiles/r05/reverb_realtime.md).
// lds: >= 3 B + B float2 of this wave's LDS (transform ping-pong 2 B, the block's spectrum B, the sums B);
// tw: exp(+2 pi i j / 1024), j < 1024, in LDS.
template <int B>
JF_DEV void rv_head_wave(const ReverbParams &P, int s, float2 *lds, const float2 *tw, int lane) {
    constexpr int NB = B / 64;
    float2 *fftbuf = lds, *x0 = lds + 2 * B, *red = lds + 3 * B;
    rv_forward<B>(P, 0, s, fftbuf, fftbuf + B, lane, x0, tw);
    JF_RV_SYNC();
    float2 acc[NB];
    float2 acc0 = make_float2(0.f, 0.f);  // bin 0 is two packed real bins
    const float2 *fdl = P.fdl + (size_t)s * P.Rg * B + lane * NB;
    const float2 *hs = P.hspec + lane * NB;
    auto load_nb = [&](const float2 *p, float2 (&v)[NB]) {
        if (NB == 2) {
            const float4 q = *reinterpret_cast<const float4 *>(p);
            v[0] = make_float2(q.x, q.y);
            v[NB - 1] = make_float2(q.z, q.w);
        } else {
#pragma unroll
            for (int i = 0; i < NB; i++) v[i] = p[i];
        }
    };
    auto mac = [&](const float2 (&x)[NB], const float2 (&h)[NB]) {
#pragma unroll
        for (int i = 0; i < NB; i++) {
            acc[i].x += x[i].x * h[i].x - x[i].y * h[i].y;
            acc[i].y += x[i].x * h[i].y + x[i].y * h[i].x;
        }
        acc0.x += x[0].x * h[0].x;
        acc0.y += x[0].y * h[0].y;
    };
    {  // partition 0: the spectrum just made, from LDS
        float2 x[NB], h[NB];
#pragma unroll
        for (int i = 0; i < NB; i++) {
            x[i] = x0[lane * NB + i];
            acc[i] = make_float2(0.f, 0.f);
        }
        load_nb(hs, h);
        mac(x, h);
    }
    // partitions 1 .. P - 1 from the delay line, eight at a time in flight
    int slot = P.head - 1;
    slot = slot < 0 ? slot + P.Rg : slot;
    constexpr int CH = 8;
#pragma unroll 1
    for (int p0 = 1; p0 < P.P; p0 += CH) {
        float2 x[CH][NB], h[CH][NB];
        int sl = slot;
#pragma unroll
        for (int c = 0; c < CH; c++) {
            // (partitions past the last one: their products are not added; the loads stay inside the buffers)
            const int p = p0 + c < P.P ? p0 + c : P.P - 1;
            load_nb(hs + (size_t)p * B, h[c]);
            load_nb(fdl + (size_t)sl * B, x[c]);
            sl = sl == 0 ? P.Rg - 1 : sl - 1;
        }
#pragma unroll
        for (int c = 0; c < CH; c++)
            if (p0 + c < P.P) mac(x[c], h[c]);
        slot -= CH;
        slot = slot < 0 ? slot + P.Rg : slot;
    }
    if (lane == 0) acc[0] = acc0;
#pragma unroll
    for (int i = 0; i < NB; i++) red[lane * NB + i] = acc[i];
    JF_RV_SYNC();
    mac_finish<B, 1>(red, 0, fftbuf, P, s, 0, lane, tw);
}
