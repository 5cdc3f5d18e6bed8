// jf_experiments.h -- every build-time EXPERIMENT of the kernels in one place.
//
// The product build defines none of the JF_EXP_* / JF_RV_EXP_* switches: each hook below then expands to the product
// code (or to nothing).  A switch replaces one step of a kernel by something cheaper so that a variant build
// (`make variant TAG=... KFLAGS=-DJF_EXP_...`, loaded with JF_LIB=...) shows what that step costs; such a build gives
// WRONG RESULTS by design (except JF_EXP_STAMPS and JF_EXP_NO_OPAQUE) and never ships.  The findings are recorded in
// profiles/r02_experiments.md / r03_experiments.md.  jf_kernels.hip and jf_reverb.hip only name the hooks.
//
//   JF_EXP_NO_OPAQUE     the lane index is not hidden from the optimiser (shows the spills that opaque() avoids)
//   JF_EXP_FASTGATHER    every window takes the one-stretch path of the gather
//   JF_EXP_NOWINLOAD     no window loads on the one-stretch path
//   JF_EXP_NOFRONT       window loads only: no forward transform, no distance factor
//   JF_EXP_NOROWLOAD     the half-filter's arithmetic without its table-row loads
//   JF_EXP_NOWAIT        no hand-off waits between the waves of a pair
//   JF_EXP_NOFILTER      fronts and hand-offs only
//   JF_EXP_PHASES        per-wave cycle counters of the pair kernel's phases (correct results): profiles/phases.py
//   JF_EXP_STAMPS        per-pair time stamps (correct results): profiles/stamps.py
//   JF_EXP_DROP_PUBLISH  fault injection: one wave of the grid stops announcing its hand-offs, so its partner's bounded
//                        wait must time out and raise the host-visible error word (tests/test_gpu_engine.py)
//   JF_RV_EXP_NOXLOAD / JF_RV_EXP_NOHLOAD / JF_RV_EXP_NOFINISH   reverb tiled kernel: no delay-line loads / no IR
//                        spectrum loads / one block of a tile finished instead of all
#pragma once

// ---- opaque(): "+v" constraint on the lane index
#ifdef JF_EXP_NO_OPAQUE
#define JF_EXP_OPAQUE(x)
#else
#define JF_EXP_OPAQUE(x) asm volatile("" : "+v"(x))
#endif

// ---- item_gather
#ifdef JF_EXP_FASTGATHER
#define JF_EXP_GATHER_PATH(one_stretch, start0) \
    do {                                        \
        if (!(one_stretch)) (start0) = 0;       \
        (one_stretch) = true;                   \
    } while (0)
#else
#define JF_EXP_GATHER_PATH(one_stretch, start0)
#endif
#ifdef JF_EXP_NOWINLOAD
#define JF_EXP_WINDOW_PAIR(p, r, start0, lane) make_float2((float)((start0) + (r)), (float)(lane))
#else
#define JF_EXP_WINDOW_PAIR(p, r, start0, lane) make_float2((p)[64 * (r)].x, (p)[64 * (r)].y)
#endif

// ---- item_finish: after the window write-back
#ifdef JF_EXP_NOFRONT
#define JF_EXP_FRONT_SHORTCUT(xd, z)               \
    do {                                           \
        _Pragma("unroll") for (int q_ = 0; q_ < 8; q_++)(xd)[q_] = (z)[q_]; \
        return true;                               \
    } while (0)
#else
#define JF_EXP_FRONT_SHORTCUT(xd, z)
#endif

// ---- filtered_half: which table row.  JF_EXP_HALFTABLE: every row index folded into the first 355 rows -- what a table
// half the size (2.9 MB: inside an XCD's 4 MB L2) would do to the row loads (timing only, wrong results)
#ifdef JF_EXP_HALFTABLE
#define JF_EXP_TABLE_ROW(r) ((r) % 355)
#else
#define JF_EXP_TABLE_ROW(r) (r)
#endif

// ---- filtered_half: one table-row load
#ifdef JF_EXP_NOROWLOAD
#define JF_EXP_ROW_LOAD(ptr, st, q, boff, t) make_float4((float)((st) + (q)), 1.0f, (float)(boff), (float)(t))
#elif defined(JF_EXP_HALFROWLOAD)
// every other row of a filter is not loaded: the row traffic a kernel would have that loads a source's rows once for
// two consecutive blocks (an upper bound for what sharing them can gain; results are wrong)
#define JF_EXP_ROW_LOAD(ptr, st, q, boff, t) \
    (((t) & 1) ? make_float4((float)((st) + (q)), 1.0f, (float)(boff), (float)(t)) : *(ptr))
#else
#define JF_EXP_ROW_LOAD(ptr, st, q, boff, t) (*(ptr))
#endif

// ---- fused_pair_kernel: what the selects on the wave's half cost (which four bins it keeps, lane 0's packed bins 0/512).
// 1: both waves of a pair run the code of wave 0 -- wrong results, timing only
#ifndef JF_EXP_NO_SELECTS
#define JF_EXP_NO_SELECTS 0
#endif

// ---- fused_pair_kernel: do the waves of a SIMD (w, w + 4, w + 8, w + 12) lose by starting a launch in the same phase?
// JF_EXP_STAGGER_US = n: wave w starts (w / 4) * n / 4 microseconds late (s_sleep counts 64 cycles: ~0.03 us)
#ifdef JF_EXP_STAGGER_US
#define JF_EXP_STAGGER(wave)                                                          \
    do {                                                                              \
        for (int i_ = 0; i_ < ((wave) >> 2) * (JF_EXP_STAGGER_US) * 8; i_++) __builtin_amdgcn_s_sleep(1); \
    } while (0)
#else
#define JF_EXP_STAGGER(wave)
#endif

// ---- pair_wait
#ifdef JF_EXP_NOWAIT
#define JF_EXP_WAIT_SHORTCUT() return
#else
#define JF_EXP_WAIT_SHORTCUT()
#endif

// ---- fused_pair_kernel: accumulate()
#ifdef JF_EXP_NOFILTER
#define JF_EXP_FILTER_SHORTCUT(fetch, zkn) \
    do {                                   \
        float2 xq_[4];                     \
        fetch(xq_);                        \
        (zkn)[0] += c2_of(xq_[0]);         \
        return;                            \
    } while (0)
#else
#define JF_EXP_FILTER_SHORTCUT(fetch, zkn)
#endif

// ---- fused_pair_kernel: publish().  The dropper keeps counting its hand-offs but no longer writes the flag word.
#ifdef JF_EXP_DROP_PUBLISH
#define JF_EXP_PUBLISH_DROPPED(npub) (blockIdx.x == 0 && pair == 0 && half == 0 && (npub) > 2)
#else
#define JF_EXP_PUBLISH_DROPPED(npub) false
#endif

// ---- fused_pair_kernel: time stamps (100 MHz real-time counter): 4 per pair, written by its wave 1 behind the error word
#ifdef JF_EXP_STAMPS
#define JF_EXP_STAMP_SETUP(P, pair, half, lane)                                                                            \
    unsigned long long *stamps_ = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>((P).err) + 16);          \
    const int stamp_wid_ = blockIdx.x * kPairsPerWg + (pair);                                                              \
    const bool stamper_ = (lane) == 0 && (half) == 1 && stamp_wid_ < 2048;                                                  \
    if (stamper_) stamps_[4 * stamp_wid_] = __builtin_amdgcn_s_memrealtime()
#define JF_EXP_STAMP_ROUND(round) \
    if (stamper_ && (round) < 2) stamps_[4 * stamp_wid_ + 1 + (round)] = __builtin_amdgcn_s_memrealtime()
#define JF_EXP_STAMP_END() \
    if (stamper_) stamps_[4 * stamp_wid_ + 3] = __builtin_amdgcn_s_memrealtime()
#else
#define JF_EXP_STAMP_SETUP(P, pair, half, lane)
#define JF_EXP_STAMP_ROUND(round)
#define JF_EXP_STAMP_END()
#endif

// ---- fused_pair_kernel: where does a wave's TIME go?  JF_EXP_PHASES: every wave keeps eight cycle counters in LDS (no
// registers besides the last time stamp); JF_EXP_PHASE(k) adds the shader-clock cycles since the previous marker to
// counter k.  At the end the counters go behind the error word: [workgroup][wave][8] unsigned (profiles/phases.py).
// Correct results; the markers cost an s_memtime and its wait each.
#ifdef JF_EXP_PHASES
#define JF_EXP_PHASE_SETUP()                                                                    \
    __shared__ unsigned s_phase[kWavesPerWg][8];                                                \
    if (lane < 8) s_phase[wave][lane] = 0;                                                      \
    unsigned long long t_phase_ = __builtin_amdgcn_s_memtime()
#define JF_EXP_PHASE(k)                                                                         \
    do {                                                                                        \
        const unsigned long long t_now_ = __builtin_amdgcn_s_memtime();                         \
        if (lane == 0) atomicAdd(&s_phase[wave][k], (unsigned)(t_now_ - t_phase_));             \
        t_phase_ = t_now_;                                                                      \
    } while (0)
#define JF_EXP_PHASE_END(P)                                                                     \
    do {                                                                                        \
        unsigned *o_ = reinterpret_cast<unsigned *>(reinterpret_cast<char *>((P).err) + 16);    \
        if (lane < 8 && blockIdx.x < 64) o_[(blockIdx.x * kWavesPerWg + wave) * 8 + lane] = s_phase[wave][lane]; \
    } while (0)
#else
#define JF_EXP_PHASE_SETUP()
#define JF_EXP_PHASE(k)
#define JF_EXP_PHASE_END(P)
#endif

// ---- reverb_mac_tiled_kernel
#ifdef JF_RV_EXP_NOXLOAD
#define JF_RV_EXP_X_LOAD(expr, a, lane) rv_v2{(float)(a), (float)(lane)}
#else
#define JF_RV_EXP_X_LOAD(expr, a, lane) (expr)
#endif
#ifdef JF_RV_EXP_NOHLOAD
#define JF_RV_EXP_H_LOAD(expr, a, lane) rv_v2{(float)(a), (float)(lane)}
#else
#define JF_RV_EXP_H_LOAD(expr, a, lane) (expr)
#endif
#ifdef JF_RV_EXP_NOFINISH
#define JF_RV_EXP_FINISH_SHORTCUT(stmt) \
    do {                                \
        stmt;                           \
        return;                         \
    } while (0)
#else
#define JF_RV_EXP_FINISH_SHORTCUT(stmt)
#endif
