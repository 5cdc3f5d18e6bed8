// jf_engine_internal.h -- what the translation units of the engine's host side share: the engine's state (struct jf_engine),
// the kernels' launchers (jf_kernels.hip, jf_reverb.hip), the small helpers every ABI entry uses.  Three units since round 6:
//   jf_engine.cpp         creation, the batch pipeline (run_blocks), the per-block calls: include/jefferson.h
//   jf_engine_reverb.cpp  the convolution reverb's schedule (run_reverb_stage, the side stream, the stage launched ahead) and
//                         jf_reverb_set_ir / jf_reverb_rms_gain
//   jf_engine_debug.cpp   every entry point of include/jefferson_debug.h (taps, timing hooks, tuning switches, accessors)
// Not part of any interface: nothing outside csrc/ includes this file.
#ifndef JF_ENGINE_INTERNAL_H
#define JF_ENGINE_INTERNAL_H

#include <hip/hip_runtime.h>
#include <ctype.h>
#include <math.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/jefferson.h"
#include "../../include/jefferson_debug.h"
#include "jf_device.h"
#include "jf_host.h"

namespace jf {
hipError_t launch_table_build(const float *d_hrir, int n_rows, int taps, const float2 *d_tw, float4 *d_htab, hipStream_t st);
hipError_t launch_table_interp_build(const RingTable &rt, int corrected, float4 *d_htab, hipStream_t st);
hipError_t launch_rfft_debug(const float *d_win, int n, const float2 *d_tw, float2 *d_spec, hipStream_t st);
hipError_t launch_interp_debug(const RingTable &rt, const float *d_ele, const float *d_azi, int *d_rows,
                               float *d_w, int *d_nt, int n, int corrected, hipStream_t st);
hipError_t launch_prep(const RingTable &rt, int mode, const float *d_pos, const SrcState *d_st, ItemDesc *d_desc,
                       int S, int K, int canon, hipStream_t st);
hipError_t launch_fused(const FusedParams &P, int max_wgs, hipStream_t st);
hipError_t fused_resident_workgroups(int nb, int kind, int *out);
hipError_t launch_stage_debug(const RingTable &rt, int mode, const float *d_pos, const float *d_win, int n,
                              const float4 *d_htab, const float2 *d_tw, float2 *d_dist, float2 *d_spec,
                              hipStream_t st);
hipError_t launch_mix(const float *d_partial, float *d_mix, int S, int K, int B, hipStream_t st);
hipError_t launch_mix_prep(const float *d_partial, float *d_mix, int S_groups, int K, int B, const RingTable &rt, int mode,
                           const float *d_pos_next, ItemDesc *d_desc_next, int S, int K_next, int canon, hipStream_t st);
int rt_waves_per_wg(int n_sources);
hipError_t launch_rt_block(const FusedParams &P, const RingTable &rt, const float *pos, float *out, int *done, int seq,
                           int n_wgs, const ReverbParams *head, hipStream_t st);
hipError_t launch_reverb_ir(const float *d_ir, int n_ir, int P, int B, float scale, const float2 *d_tw,
                            float2 *d_hspec, hipStream_t st);
hipError_t launch_reverb(const ReverbParams &P, ReverbPlan *plan, hipStream_t st, int *form_used);
hipError_t launch_reverb_catchup(const ReverbParams &P, hipStream_t st);
int big_twiddle_pack_len(int B1);
int big_twiddle_pack_index(int B1, int k);
hipError_t launch_reverb_big_side(const ReverbBigParams *transforms, const ReverbBigParams *products, hipStream_t st);
hipError_t launch_reverb_big_ir(const float *d_ir, int n_ir, int t0, int P1, int B1, float scale, const float2 *d_tw1,
                                float2 *d_hspec1, hipStream_t st);
int kernels_build_kind();
}  // namespace jf

using namespace jf;

#define JF_INTERNAL __attribute__((visibility("hidden")))  // shared between the engine's units, not exported

inline thread_local std::string g_create_error;

struct HostPos {  // public fields of SoundSource (SoundSource.cuh:24-36)
    float ele, azi, r, x, y, z;
};

struct EventPair {
    hipEvent_t a, b;
};

constexpr double kInterpMovedMax = 0.30;  // jf_engine::interp_use == 2: largest share of moving items a run may have to take the rows
constexpr long kRtPollNs = 2000000;  // jf_collect_block polls the real-time kernel's completion words for at most this long
constexpr int kRvFusedHeadMax = 64;  // partitions of B a wave takes a block through by itself (rv_head_wave)
constexpr int kRtMaxWgs = 128;  // workgroups (8 or 16 waves, a source per wave and turn) of the one-launch real-time kernel: 64 and 256 measure slower

struct jf_engine {
    jf_config cfg{};
    int own_mix_blocks = 0;  // blocks the last jf_batch_run left in d_mix (0: it wrote to the caller's buffer, failed or has not run)
    int B = 0, S = 0, maxK = 0;
    hipStream_t stream = nullptr;
    std::string err;

    float4 *d_htab = nullptr;
    // The kInterpRows pre-interpolated rows (jf_device.h; 386 MB behind the 710 measured rows) are built LAZILY: by the first
    // run whose policy takes them (run_blocks), or when jf_debug_set_interp_table(e, 1) / a read of those rows asks -- never for
    // an engine that only ever runs sources that move every block, and not for the eight shards of a job on one device.
    bool interp_avail = false;  // the engine may have them (no JF_FLAG_NO_INTERP_TABLE, no failed allocation)
    bool interp_built = false;  // d_htab holds them
    // ... and which batch calls use them (jf_debug_set_interp_table): 0 none, 1 all, 2 (default) decided per run.  A source
    // that stays where it is reads its one row out of the caches block after block (12-18 % faster than weighting four
    // measured rows); a source that moves streams a new 8 KB row from HBM, and when every source moves every block the
    // kernel is bound by that stream (5.8 TB/s) and 2-5 % SLOWER than the weighting.  Measured crossover: a third of the
    // items moving (profiles/r04/interp_table.md).  Runs of an uploaded trajectory take the rows unless more than
    // kInterpMovedMax of their items move; calls without a trajectory take them.
    int interp_use = 0;
    std::vector<unsigned> traj_moved;  // [traj_blocks + 1] prefix counts of the uploaded trajectory's items that move
    bool last_rows = false;     // the last batch run's descriptors could name pre-interpolated rows
    float2 *d_tw = nullptr;
    float2 *d_twpack = nullptr;
    SrcSignal *d_sigs = nullptr;
    float *d_zero = nullptr;  // PAD_LEN zeros: the "signal" of a source without one
    SrcState *d_state[2] = {nullptr, nullptr};
    float *d_hist[2] = {nullptr, nullptr};
    ItemDesc *d_desc = nullptr;
    // Descriptors of the window that follows the last jf_batch_run, written by that run itself (trailing workgroups of the
    // pair kernel's launch, or mix_prep_kernel) into the second buffer; the next run takes them instead of launching prep_kernel if it asks for exactly that window
    // of the same trajectory in the same mode and layout -- anything else that runs or touches the state in between
    // clears `ahead.valid`.
    ItemDesc *d_desc_ahead = nullptr;
    struct {
        bool valid = false;
        int first = 0, K = 0, mode = 0, canon = 0;
        unsigned long traj_gen = 0;
    } ahead;
    unsigned long traj_gen = 0;  // bumped by every jf_batch_upload_positions
    bool prep_ahead = true;      // jf_debug_set_prep_ahead
    bool last_prep_skipped = false, last_mix_prep = false, last_fused_prep = false;  // what the last run launched (jf_debug_last_kernels)
    float *d_partial = nullptr;
    float *d_mix = nullptr;
    float *d_pos_rt = nullptr;  // [S][5]
    float *d_traj = nullptr;    // [total][S][5]
    short *d_pick = nullptr;    // nearest-azimuth table of the index/weight kernels (RingTable::pick)
    RingTable rt{};             // ring_table() + this engine's device table
    int *d_order = nullptr;     // [S] processing order of the pair kernel (a permutation of the sources)
    std::vector<int> order;     // host copy
    bool sorted_order = false;  // d_order is not the identity
    int traj_blocks = 0;
    int cur = 0;  // parity of the valid state/history
    int src_group = 0;  // 0 = automatic
    int last_group = 0; // G of the last batch pipeline run
    int last_rv_form = 0;  // form of the reverb multiply-accumulate stage the last call took
    bool last_rt = false;  // the last block went through the one-launch real-time kernel
    std::string kernels;   // jf_debug_last_kernels
    int rv_form = 0;    // 0 = automatic
    // Data::type and Data::pauseStatus are written by the UI thread and read by the audio thread at every
    // block (Audio.cu:101,104)
    std::atomic<int> mode{0};  // 0 = FD_COMPLEX, 1 = FD_BASIC
    std::atomic<int> paused{0};
    int resident_wgs[3] = {0, 0, 0};  // persistent-grid size of the per-source / the pair / the pair-with-rows kernel on this device
    int grid_limit = 0;            // > 0: tests shrink the grid so that waves loop over several units
    float last_peak = 0.0f;        // max |sample| of the last block handed out (Audio.cu:111-113 clip alert)

    std::vector<float *> d_signal;  // per source
    std::vector<SrcSignal> h_sigs;

    std::mutex pos_mu;  // setters may come from another thread (graphics.cu:378)
    std::vector<HostPos> pos;

    float *h_pos_pinned = nullptr;  // [S][5]   pinned + mapped: the real-time kernel reads it in place
    float *h_out_pinned = nullptr;  // [kRtMaxWgs][2B] pinned + mapped: ... and writes its workgroups' stereo blocks in place
    int rt_wgs = 0;                 // partial blocks the block in flight left there (0: one finished block)
    float *hd_pos = nullptr, *hd_out = nullptr;  // their device addresses
    // The real-time kernel's workgroups each store a sequence number into their word of h_done (pinned + mapped) when their
    // block lies in h_out_pinned; jf_collect_block polls the words instead of synchronising the stream.
    int *h_done = nullptr, *hd_done = nullptr;
    int rt_seq = 0;
    int *h_err = nullptr, *hd_err = nullptr;     // pinned + mapped error word of the fused kernels
    int rt_max_sources = 8192;      // per-block calls with at most this many sources take the one-launch path
                                    // (profiles/latency_rt_sweep.py: 32 against 54 us at 1024 sources, 75 against 105 at 8192)
    bool in_flight = false;         // a submitted block not yet collected
    bool have_prev = false;         // jf_callback: a block is pending from the previous call

    int profiling = 0;  // 0 off, 1 = time the fused kernel only (2 events per call), 2 = every kernel
    int profile_stride = 1;    // events around every n-th batch run only (jf_profile_set_stride)
    long profile_calls = 0;
    bool timed_now = false;    // this batch run carries event records
    std::vector<EventPair> ev_prep, ev_fused, ev_mix, ev_reverb;
    size_t ev_used = 0;

    // convolution reverb stage (jf_reverb.hip); off while rv_P == 0
    int rv_P = 0, rv_Rg = 0, rv_Wr = 0, rv_head = 0;
    float2 *d_rv_hspec = nullptr;
    float2 *d_rv_fdl = nullptr;
    float *d_rv_wet = nullptr;
    float *d_rv_prev[2] = {nullptr, nullptr};
    int *d_rv_count[2] = {nullptr, nullptr};
    // non-uniform partitioning (ReverbBigParams, jf_device.h): rv_P is then the HEAD's partition count (rv_M) and the rest
    // of the impulse response lies in rv_P1 partitions of rv_B1 = rv_M * B taps.  rv_P1 == 0: uniform partitioning.
    int rv_partitioning = 0;     // jf_debug_set_reverb_partitioning: 0 by length, 1 uniform, 2 non-uniform (at the next set_ir)
    int rv_P_total = 0;          // partitions of B the impulse response has (what rv_P is under uniform partitioning)
    int rv_M = 0;                // blocks per big block (rv_big_blocks(B)): rv_B1 = rv_M * B
    int rv_P1 = 0, rv_B1 = 0, rv_R1 = 0, rv_Rn = 0, rv_Fn = 0, rv_steps_max = 0;
    long long rv_blocks = 0;     // blocks the stage has processed since it was set up: big block m = blocks 16 m .. 16 m + 15
    long long rv_fut_m = 1;      // TAIL(m) has been formed for every big block up to this one (big blocks 0 and 1 have none: zeros)
    ReverbPlan last_plan;        // what the last call did (jf_debug_last_kernels)
    float2 *d_rv_tw1 = nullptr, *d_rv_hspec1 = nullptr, *d_rv_fdl1 = nullptr, *d_rv_ybig = nullptr;
    float *d_rv_dryring = nullptr, *d_rv_fut = nullptr;
    SrcSignal *d_sigs_wet = nullptr;  // [S] the wet rings as the spatialiser's signals
    // One-block calls (the real-time shape) keep the big partitions off the block's critical path (run_reverb_stage): their
    // kernels go to a second stream, d_rv_yacc is that stream's product buffer.
    int rv_async = 1;            // jf_debug_set_reverb_async
    hipStream_t rv_side = nullptr;
    hipEvent_t rv_ev_main = nullptr, rv_ev_side = nullptr;
    bool rv_side_busy = false;   // work was put on the side stream since the engine's stream last waited for it
    bool rv_side_urgent = false; // ... some of which the very next block reads
    float2 *d_rv_yacc = nullptr; // [S][2][B1]
    std::string last_side;       // the side stream's kernels of the last call (jf_debug_last_kernels)
    // what the last stage wants run on the side stream once the block's spatialiser has been launched (submit_side)
    bool side_tr = false;
    ReverbBigParams side_p[2];   // transforms, products
    long long side_fut_m = 0;    // ... and what that work will have formed: committed to rv_fut_m / rv_side_urgent only once it
    bool side_urgent = false;    //     has been launched (submit_side)
    // One-block calls through the one-launch real-time kernel CAN run the stage's HEAD inside that launch (rt_block_kernel<..,
    // true>, jf_rv_small.h: rv_head_wave) when the head is short (<= kRvFusedHeadMax partitions: the 2 M of a non-uniformly
    // partitioned response, or a short response) and eight waves share a workgroup: one launch per audio block instead of two.
    // OFF by default: measured 5 us SLOWER per block at config 5's 256 sources (35.1 against 30.1 us mean: the head's two small
    // transforms and its 64 KB of spectra per source are then ONE wave's chain on one of 32 compute units, where the head
    // kernel spreads a source over 16 waves and the sources over every compute unit: profiles/r05/reverb_realtime.md)
    int rv_head_fused = 0;       // jf_debug_set_reverb_head_fused
    bool post_tr = false;        // transforms left in line behind the fused head (run_reverb_stage -> jf_submit_block)
    ReverbBigParams post_tr_p;
    // A batch call of whole big blocks that ENDS on a big-block boundary reads none of the small transforms of its last 2 M - 1
    // blocks: they are state for a later call's head -- and the next such call never looks at them.  They are put off
    // (rv_small_stale; the call's last transform leaves the samples in the dry ring, the previous block and the play position:
    // ReverbBigParams::state_out) and formed from the dry ring by the first call that has a block for the head
    // (launch_reverb_catchup: same samples, same transform, same bits).  12 us of config 5's 290 us batch step.
    // THE STAGE OF THE NEXT BLOCK, AHEAD (round 5).  The reverb stage of a block needs the dry signals and its own state, not the
    // positions the host sets for that block: a one-block call through the real-time kernel therefore launches the NEXT block's
    // stage right behind its own spatialiser (same stream: ordered by construction), and the next call finds the wet block
    // there and launches the spatialiser alone -- the head kernel (8 us at config 5's 256 sources) leaves the block's critical
    // path: between two audio callbacks it has 2.9 ms to itself; in calls back to back it overlaps with the host's turn-around.
    // Only for a plain head (no big block completed, no TAIL owed, nothing put off); anything that changes what the stage read
    // or wrote -- a new signal, a reset, a new response, a batch call, a switch of the stage's knobs -- DISCARDS it
    // (rv_ahead_discard: wait for the stream, take the stage's bookkeeping back; its writes are overwritten by the stage
    // done again).  Same kernels on the same data in the same order: bit-identical.
    int rv_ahead_on = 1;          // jf_debug_set_reverb_ahead
    bool rv_ahead = false;        // the next block's stage has been launched
    struct {
        int rv_head = 0, last_rv_form = 0;
        long long rv_blocks = 0, rv_fut_m = 0;
        ReverbPlan last_plan;
        std::string last_side;
        bool last_catchup = false, last_small_fft = true, rv_side_busy = false, rv_side_urgent = false;
    } rv_book;                    // the stage's bookkeeping before that launch
    std::string kernels_frozen;   // jf_debug_last_kernels of the call that launched it (the stage's fields describe the NEXT block)
    bool kernels_use_frozen = false;
    bool rv_small_stale = false;
    bool last_catchup = false;   // the last call began with the catch-up (jf_debug_last_kernels)
    bool last_small_fft = true;  // ... and launched the small transforms' kernel
    int rv_lazy_small = 1;       // jf_debug_set_reverb_lazy_state
    int rv_side_wgs = 192;       // workgroups of its product kernel (it runs beside later blocks' kernels: launched narrow;
                                 // 64 / 128 / 256 / all measure 34.5 / 34.1 / 34.1 / 35.0 us per block: profiles/r04/rt_async.md).
                                 // Set to THREE QUARTERS of the device's compute units at creation (round 6): with one
                                 // workgroup on every compute unit the block's own kernels find none to themselves; 192 of 256
                                 // measure mean 23.8-24.0 / p99 32.3-33.4 us per block against 24.4 / 35.2-35.9 with 256, 160
                                 // and fewer stretch the product over more blocks (profiles/r06/reverb_realtime.md)
};

// Host -> device copies and memsets of engine state go through the ENGINE'S stream: it is a non-blocking stream, which the null
// stream's copies and memsets are not ordered with -- a kernel launched right behind a hipMemset of the null stream could run
// before it (a reset followed at once by a block: found by the random sessions, one run in twelve).  The copy has landed when
// this returns (the host buffer may be a temporary).
inline hipError_t h2d(jf_engine *e, void *dst, const void *src, size_t bytes) {
    const hipError_t r = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, e->stream);
    return r != hipSuccess ? r : hipStreamSynchronize(e->stream);
}

inline int fail(jf_engine *e, int code, const std::string &msg) {
    if (e)
        e->err = msg;
    else
        g_create_error = msg;
    return code;
}

#define JF_HIP(e, call)                                                                        \
    do {                                                                                       \
        hipError_t _s = (call);                                                                \
        if (_s != hipSuccess)                                                                  \
            return fail((e), JF_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(_s)); \
    } while (0)

inline bool valid_src(const jf_engine *e, int s) { return e && s >= 0 && s < e->S; }

// Elevations the setters take: where the reference's rule names two measured rings, (-50, 90] (SoundSource.cu:67-68 with
// the table of hrtf_signals.cu:7); with a grid of its own the engine clamps to the grid's first and last ring: [-90, 90].
inline bool elevation_ok(const jf_engine *e, float ele) { return e->rt.kemar ? (ele > -50.0f && ele <= 90.0f) : (ele >= -90.0f && ele <= 90.0f); }
inline const char *elevation_msg(const jf_engine *e) { return e->rt.kemar ? "elevation outside (-50, 90]" : "elevation outside [-90, 90]"; }

// The error word of the fused kernels (host-mapped): set when a wait between the two wavefronts of a pair timed out
// (fused_pair_kernel; impossible by its protocol, and bounded so that a fault cannot hang the GPU).  The blocks of that
// launch are wrong and the sources' state is undefined from then on, so the condition is FATAL for the engine: every
// call that hands out or produces audio afterwards returns JF_ERR_DEVICE (jf_pa_callback: silence); the engine can
// only be destroyed.  Valid after a synchronisation of the engine's stream.
constexpr const char *kHandOffMsg = "fused_pair_kernel: a wavefront hand-off timed out (fatal: destroy the engine)";
inline bool device_fault(const jf_engine *e) { return e->h_err && *(volatile int *)e->h_err != 0; }

// Every ABI entry that reaches HIP binds the engine's device for its duration: the callback runs on
// PortAudio's thread, the setters on the UI thread, and a host with one engine per GPU switches devices
// between calls -- a thread's current device is 0 until somebody sets it.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(const jf_engine *e) {
        if (!e) return;
        if (hipGetDevice(&prev) == hipSuccess && prev != e->cfg.device)
            switched = hipSetDevice(e->cfg.device) == hipSuccess;
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

// what the kernels get as `mode`: bit 0 = FD_BASIC, bit 1 = the corrected index/weight rule
inline bool corrected_rule(const jf_engine *e) {  // (a grid that is not the reference's has no other rule)
    return (e->cfg.flags & JF_FLAG_CORRECTED_INTERPOLATION) != 0 || !e->rt.kemar;
}
inline int kernel_mode(const jf_engine *e) {
    return e->mode.load(std::memory_order_relaxed) | (corrected_rule(e) ? 2 : 0);
}

inline EventPair *next_events(jf_engine *e, std::vector<EventPair> &pool) {
    if (pool.size() <= e->ev_used) {
        EventPair p;
        if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return nullptr;
        pool.push_back(p);
    }
    return &pool[e->ev_used];
}

// =============================================================== C ABI ====
// Nothing may propagate through the C ABI: host allocations (std::vector, std::string) can throw.
template <class F>
inline int jf_guard(F &&f) noexcept {
    try {
        return f();
    } catch (const std::bad_alloc &) {
        try { g_create_error = "out of host memory"; } catch (...) {}
        return JF_ERR_NOMEM;
    } catch (const std::exception &ex) {
        try { g_create_error = ex.what(); } catch (...) {}
        return JF_ERR_DEVICE;
    } catch (...) {
        return JF_ERR_DEVICE;
    }
}

// The side stream has nothing in flight any more (host-side wait); what it had promised is forgotten.
inline void quiesce_side(jf_engine *e) {
    if (e->rv_side && e->rv_side_busy) (void)hipStreamSynchronize(e->rv_side);
    e->rv_side_busy = e->rv_side_urgent = false;
}

// ---- shared between the units (definitions: jf_engine.cpp unless noted) ------------------------------------------------------
// reverb ahead of the spatialiser: dry signal -> FDL -> wet ring, for the K blocks of this call (state parity p).  head_out
// (one-block calls through the real-time kernel; may be null): if the stage's head can run inside that kernel, it is NOT
// launched -- *head_out receives its parameters, *head_fused says so, and e->post_tr holds what must follow the kernel
JF_INTERNAL int run_reverb_stage(jf_engine *e, int p, int K, ReverbParams *head_out = nullptr, bool *head_fused = nullptr);  // jf_engine_reverb.cpp
JF_INTERNAL int submit_side(jf_engine *e);                 // jf_engine_reverb.cpp
JF_INTERNAL int rv_ahead_discard(jf_engine *e);            // jf_engine_reverb.cpp
JF_INTERNAL bool rv_ahead_possible(const jf_engine *e);    // jf_engine_reverb.cpp
JF_INTERNAL void free_reverb(jf_engine *e);                // jf_engine_reverb.cpp
JF_INTERNAL int ensure_interp_rows(jf_engine *e);
JF_INTERNAL int run_blocks(jf_engine *e, const float *d_pos, int K, float *d_mix_out, int first_block = -1);
JF_INTERNAL int reset_sources(jf_engine *e, int src);

#endif  // JF_ENGINE_INTERNAL_H
