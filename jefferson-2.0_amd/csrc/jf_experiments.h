// jf_experiments.h -- the build-time INSTRUMENTATION of the kernels in one place.
//
// The product build defines none of the JF_EXP_* switches: each hook below then expands to the product code (or to
// nothing).  What is left here gives CORRECT results (time stamps, per-phase cycle counters, the lane index left visible
// to the optimiser) -- with one exception, the fault injection JF_EXP_DROP_PUBLISH, which exists so that a test can see
// the bounded hand-off wait time out.  A library built with it reports so (jf::kernels_build_kind() != 0) and
// jf_engine_create refuses it unless the environment says JF_ALLOW_EXPERIMENT=1 (tests/test_gpu_engine.py does).
//
// The timing-only hooks of rounds 2 and 3 -- builds that skip a step of a kernel to show what it costs, and give WRONG
// RESULTS by design (NOCHAIN, NOWINLOAD, NOFRONT, NOROWLOAD, HALFROWLOAD, HALFTABLE, FASTGATHER, NO_SELECTS, NOWAIT,
// NOFILTER, STAGGER, JF_RV_EXP_*) -- no longer live in the product's translation units: one of them read outside a
// buffer in round 3.  They are kept as profiles/r03_timing_hooks.patch (apply to a scratch copy to repeat a measurement
// of profiles/r02_experiments.md / r03_experiments.md; the patch also marks the build as an experiment).
//
//   JF_EXP_NO_OPAQUE     the lane index is not hidden from the optimiser (shows the spills that opaque() avoids)
//   JF_EXP_PHASES        per-wave cycle counters of the pair kernel's phases: profiles/phases.py
//   JF_EXP_STAMPS        per-pair time stamps: profiles/stamps.py
//   JF_EXP_DROP_PUBLISH  fault injection: one wave of the grid stops announcing its hand-offs, so its partner's bounded
//                        wait must time out and raise the host-visible error word (tests/test_gpu_engine.py)
#pragma once

// what jf::kernels_build_kind() returns: 0 = product; bit 0 = results are wrong by design (fault injection, timing hooks)
#ifdef JF_EXP_DROP_PUBLISH
#define JF_EXP_BUILD_KIND 1
#else
#define JF_EXP_BUILD_KIND 0
#endif

// ---- opaque(): "+v" constraint on the lane index
#ifdef JF_EXP_NO_OPAQUE
#define JF_EXP_OPAQUE(x)
#else
#define JF_EXP_OPAQUE(x) asm volatile("" : "+v"(x))
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
