// jf_engine.cpp -- implementation of the C ABI (include/jefferson.h) on HIP.
// One engine = one GPU (one process per GPU in multi-GPU runs).  No CPU
// fallback: every processing entry point runs the HIP kernels or fails.
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

namespace {
thread_local std::string g_create_error;

struct HostPos {  // public fields of SoundSource (SoundSource.cuh:24-36)
    float ele, azi, r, x, y, z;
};

struct EventPair {
    hipEvent_t a, b;
};
}  // namespace

constexpr double kInterpMovedMax = 0.30;  // jf_engine::interp_use == 2: largest share of moving items a run may have to take the rows
constexpr long kRtPollNs = 2000000;  // jf_collect_block polls the real-time kernel's completion words for at most this long
constexpr int kRvFusedHeadMax = 64;  // partitions of B a wave takes a block through by itself (rv_head_wave)
constexpr int kRtMaxWgs = 128;  // workgroups (8 or 16 waves, a source per wave and turn) of the one-launch real-time kernel: 64 and 256 measure slower

struct jf_engine {
    jf_config cfg{};
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

namespace {

// Host -> device copies and memsets of engine state go through the ENGINE'S stream: it is a non-blocking stream, which the null
// stream's copies and memsets are not ordered with -- a kernel launched right behind a hipMemset of the null stream could run
// before it (a reset followed at once by a block: found by the random sessions, one run in twelve).  The copy has landed when
// this returns (the host buffer may be a temporary).
hipError_t h2d(jf_engine *e, void *dst, const void *src, size_t bytes) {
    const hipError_t r = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, e->stream);
    return r != hipSuccess ? r : hipStreamSynchronize(e->stream);
}

int fail(jf_engine *e, int code, const std::string &msg) {
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

bool valid_src(const jf_engine *e, int s) { return e && s >= 0 && s < e->S; }

// Elevations the setters take: where the reference's rule names two measured rings, (-50, 90] (SoundSource.cu:67-68 with
// the table of hrtf_signals.cu:7); with a grid of its own the engine clamps to the grid's first and last ring: [-90, 90].
bool elevation_ok(const jf_engine *e, float ele) { return e->rt.kemar ? (ele > -50.0f && ele <= 90.0f) : (ele >= -90.0f && ele <= 90.0f); }
const char *elevation_msg(const jf_engine *e) { return e->rt.kemar ? "elevation outside (-50, 90]" : "elevation outside [-90, 90]"; }

// The error word of the fused kernels (host-mapped): set when a wait between the two wavefronts of a pair timed out
// (fused_pair_kernel; impossible by its protocol, and bounded so that a fault cannot hang the GPU).  The blocks of that
// launch are wrong and the sources' state is undefined from then on, so the condition is FATAL for the engine: every
// call that hands out or produces audio afterwards returns JF_ERR_DEVICE (jf_pa_callback: silence); the engine can
// only be destroyed.  Valid after a synchronisation of the engine's stream.
constexpr const char *kHandOffMsg = "fused_pair_kernel: a wavefront hand-off timed out (fatal: destroy the engine)";
bool device_fault(const jf_engine *e) { return e->h_err && *(volatile int *)e->h_err != 0; }

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
static bool corrected_rule(const jf_engine *e) {  // (a grid that is not the reference's has no other rule)
    return (e->cfg.flags & JF_FLAG_CORRECTED_INTERPOLATION) != 0 || !e->rt.kemar;
}
static int kernel_mode(const jf_engine *e) {
    return e->mode.load(std::memory_order_relaxed) | (corrected_rule(e) ? 2 : 0);
}

EventPair *next_events(jf_engine *e, std::vector<EventPair> &pool) {
    if (pool.size() <= e->ev_used) {
        EventPair p;
        if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return nullptr;
        pool.push_back(p);
    }
    return &pool[e->ev_used];
}

// reverb ahead of the spatialiser: dry signal -> FDL -> wet ring, for the K blocks of this call (state parity p)
static int submit_side(jf_engine *e);
// head_out (one-block calls through the real-time kernel; may be null): if the stage's head can run inside that kernel, it is
// NOT launched here -- *head_out receives its parameters, *head_fused says so, and e->post_tr holds what must follow the kernel
static int run_reverb_stage(jf_engine *e, int p, int K, ReverbParams *head_out = nullptr, bool *head_fused = nullptr) {
    if (head_fused) *head_fused = false;
    e->post_tr = false;
    if (e->rv_P <= 0) return JF_OK;
    if (e->side_tr) {
        // the last stage's work for the side stream was never submitted (a launch between that stage and submit_side failed
        // and the caller went on): it goes first -- the transform it holds is of samples the dry ring still has
        const int rc = submit_side(e);
        if (rc) return rc;
    }
    EventPair *er = nullptr;
    if (e->profiling >= 2 && e->timed_now) {
        er = next_events(e, e->ev_reverb);
        if (!er) return fail(e, JF_ERR_DEVICE, "hipEventCreate failed");
        JF_HIP(e, hipEventRecord(er->a, e->stream));
    }
    ReverbParams R;
    R.tw = e->d_tw;
    R.dry = e->d_sigs;
    R.dry_count_in = e->d_rv_count[p];
    R.dry_count_out = e->d_rv_count[p ^ 1];
    R.prev_in = e->d_rv_prev[p];
    R.prev_out = e->d_rv_prev[p ^ 1];
    R.fdl = e->d_rv_fdl;
    R.hspec = e->d_rv_hspec;
    R.wet = e->d_rv_wet;
    R.st_in = e->d_state[p];
    R.S = e->S;
    R.K = K;
    R.B = e->B;
    R.P = e->rv_P;
    R.Rg = e->rv_Rg;
    R.Wr = e->rv_Wr;
    R.head = e->rv_head;
    R.mac_form = e->rv_form;
    ReverbPlan plan;
    plan.big = e->rv_P1 > 0;
    bool defer_small = false, need_small = true;
    // One-block calls -- the real-time shape -- keep the big partitions' kernels off the block's critical path.  The head
    // covers TWO big blocks of taps (2 M partitions of B), so TAIL(m) = sum_{q >= 2} X_{m+1-q} H'_q needs nothing newer than
    // X_{m-1}, which exists a whole big block before big block m begins.  When a one-block call completes big block mb, the
    // transform X_{mb+1}, the products of TAIL(mb + 2) and their inverse transform go to a second stream BEHIND the block's
    // spatialiser (submit_side); the first block to read the result is seventeen blocks away, and the stage of the next call
    // that is not such a one-block call -- or the next one that puts work there -- makes the engine's stream wait for that
    // stream (an event).  (With a head of M partitions TAIL(mb + 1) needed X_{mb+1} and was needed by the very next block: in
    // line, that block and the one before it cost 40 and 9 us more than the other fourteen at configs[4], 256 sources.)
    // Calls that pin a form, batch calls and profiled calls do everything in line on the engine's stream.
    const bool async_ok = plan.big && K == 1 && e->rv_async && e->rv_form == 0 && e->profiling < 2 && e->rv_side != nullptr;
    const bool completes = plan.big && (e->rv_blocks + K) / e->rv_M > e->rv_blocks / e->rv_M;  // transforms in this call
    if (e->rv_side_busy && (!async_ok || completes || e->rv_side_urgent)) {
        JF_HIP(e, hipStreamWaitEvent(e->stream, e->rv_ev_side, 0));
        e->rv_side_busy = e->rv_side_urgent = false;
    }
    ReverbBigParams &s_tr = e->side_p[0], &s_prod = e->side_p[1];
    e->last_side.clear();
    const long long fut_m_before = e->rv_fut_m;
    bool side_wanted = false;
    if (plan.big) {
        // Absolute block indices j0 .. j1 - 1; big block m = blocks 16 m .. 16 m + 15.
        const long long j0 = e->rv_blocks;
        const int B1 = e->rv_B1, R1 = e->rv_R1, Rn = e->rv_Rn, Fn = e->rv_Fn, M = e->rv_M;
        R.dryring = e->d_rv_dryring;
        R.Rd = Rn * B1;
        R.dry_pos0 = (int)((j0 * e->B) % R.Rd);
        R.fut = e->d_rv_fut;
        R.F = Fn * B1;
        R.fut_pos0 = (int)((j0 * e->B) % R.F);
        ReverbBigParams G;
        G.tw1 = e->d_rv_tw1;
        G.dryring = e->d_rv_dryring;
        G.dry = e->d_sigs;
        G.dry_count_in = e->d_rv_count[p];
        G.dry_pos0 = R.dry_pos0;
        G.fdl1 = e->d_rv_fdl1;
        G.hspec1 = e->d_rv_hspec1;
        G.ybig = e->d_rv_ybig;
        G.fut = e->d_rv_fut;
        G.wet = e->d_rv_wet;
        G.st_in = e->d_state[p];
        G.S = e->S;
        G.B = e->B;
        G.B1 = B1;
        G.P1 = e->rv_P1;
        G.R1 = R1;
        G.Rn = Rn;
        G.Fn = Fn;
        G.Wr = e->rv_Wr;
        G.M = M;
        G.NP = e->rv_P1 + 17;
        auto mod = [](long long a, int n) { return (int)(((a % n) + n) % n); };
        const ReverbSchedule sc = host_reverb_schedule(j0, K, M, e->rv_fut_m);  // which X_m, FULL, TAIL and ranges: jf_host.cpp
        e->rv_fut_m = sc.fut_m;
        // whole big blocks up to the call's end: the small transforms of its last blocks are put off (rv_small_stale) ...
        defer_small = e->rv_lazy_small && sc.n_mid > 0 && sc.kn[1] == 0 && sc.n_tr > 0;
        // ... and a call that takes a block through the head needs the ones an earlier call put off, first
        need_small = sc.n_mid == 0 || sc.kn[0] > 0 || sc.kn[1] > 0;
        plan.transforms = G;
        if (defer_small) {
            plan.transforms.state_out = 1;
            plan.transforms.dryring_out = e->d_rv_dryring;
            plan.transforms.prev_out = e->d_rv_prev[p ^ 1];
            plan.transforms.dry_count_out = e->d_rv_count[p ^ 1];
            plan.transforms.call_samples = K * e->B;
        }
        plan.transforms.n_tr = sc.n_tr;
        plan.transforms.tr_slot_first = mod(sc.m_lo, R1);
        plan.transforms.tr_rel_first = (int)((sc.m_lo - 2) * B1 - j0 * e->B);
        const int n_mid = sc.n_mid;
        plan.middle = G;
        plan.middle.n_prod = n_mid;
        plan.middle.anchor_slot_first = mod(sc.ma + 1, R1);  // FULL(m) is anchored at X_{m+1}
        plan.middle.h_first = 0;
        plan.middle.n_part = e->rv_P1 + 1;
        plan.middle.to_wet = 1;
        plan.middle.wet_k0 = (int)(sc.ma * M - j0);
        plan.n_ranges = sc.n_ranges;
        for (int r = 0; r < 2; r++) {
            plan.kb[r] = sc.kb[r];
            plan.kn[r] = sc.kn[r];
        }
        R.copy_lo = sc.copy_lo;
        R.copy_hi = sc.copy_hi;
        R.skip_lo = sc.skip_lo;
        R.skip_hi = sc.skip_hi;
        if (defer_small) {
            R.copy_hi = R.copy_lo;   // nothing is copied, nothing behind the front blocks is transformed
            R.skip_hi = K;
        }
        auto tail_for = [&](long long m) {  // TAIL(m) = sum_{q = 2 .. P1} X_{m+1-q} H'_q: the newest spectrum is X_{m-1}
            ReverbBigParams T = G;
            T.n_prod = 1;
            T.anchor_slot_first = mod(m - 1, R1);
            T.h_first = 2;
            T.n_part = e->rv_P1 - 1;
            T.to_wet = 0;
            T.fut_first = mod(m, Fn);
            return T;
        };
        plan.tail_early = G;
        plan.tail_late = G;
        if (sc.tail_early >= 0) plan.tail_early = tail_for(sc.tail_early);
        if (sc.tail_late >= 0) plan.tail_late = tail_for(sc.tail_late);
        if (async_ok && sc.n_tr > 0) {
            // the block completes big block mb: X_{mb+1} and, with it, TAIL(mb + 2) -- which the block after the next
            // sixteen is the first to read
            const long long mb = j0 / M;
            const std::string b1 = std::to_string(B1);
            s_tr = plan.transforms;
            plan.transforms.n_tr = 0;
            // all 2 B1 samples from the dry ring -- the head kernel has just written this block's there -- and none from the
            // signal at the play position, which the next call moves on while the side stream may still be reading
            s_tr.dry_pos0 = (R.dry_pos0 + e->B) % R.Rd;
            s_tr.tr_rel_first -= e->B;
            // (if nobody has formed TAIL(mb + 1) -- the run of one-block calls began inside this big block -- both, and the
            // next call waits for them)
            const bool both = e->rv_fut_m < mb + 1;
            s_prod = tail_for(both ? mb + 1 : mb + 2);
            s_prod.n_prod = both ? 2 : 1;
            s_prod.ybig = e->d_rv_yacc;
            s_prod.mac_wgs = both ? 0 : e->rv_side_wgs;
            e->side_urgent = both;
            e->side_fut_m = mb + 2;
            side_wanted = true;
            e->last_side = "reverb_big_fft_kernel<" + b1 + ",1>@side;reverb_big_mac_kernel<" + b1 + ",1>@side;reverb_big_ifft_kernel<" +
                           b1 + ",1>@side;";
        }
        if (plan.transforms.n_tr > e->rv_steps_max || n_mid > e->rv_steps_max)
            return fail(e, JF_ERR_STATE, "reverb: more big-partition steps in a call than buffers");
    }
    e->last_small_fft = K - (R.skip_hi - R.skip_lo) > 0;
    e->last_catchup = false;
    if (e->rv_small_stale && need_small) {
        // the last 2 M - 1 blocks before this call, from the dry ring: block rv_blocks - n .. rv_blocks - 1, slots rv_head - n ..
        ReverbParams C = R;
        const int n = 2 * e->rv_M - 1;
        C.K = n;
        C.catchup = 1;
        C.head = (int)((((long long)e->rv_head - n) % e->rv_Rg + e->rv_Rg) % e->rv_Rg);
        C.dry_pos0 = (int)((((e->rv_blocks - n) * e->B) % R.Rd + R.Rd) % R.Rd);
        C.copy_lo = C.copy_hi = C.skip_lo = C.skip_hi = 0;
        JF_HIP(e, launch_reverb_catchup(C, e->stream));
        e->rv_small_stale = false;
        e->last_catchup = true;
    }
    plan.head_fused = head_out != nullptr && K == 1 && e->rv_head_fused && e->rv_form == 0 && e->profiling < 2 &&
                      e->rv_P <= kRvFusedHeadMax && rt_waves_per_wg(e->S) == 8 && (e->B == 64 || e->B == 128 || e->B == 256);
    {
        const hipError_t q = launch_reverb(R, &plan, e->stream, &e->last_rv_form);
        if (q != hipSuccess) {
            e->rv_fut_m = fut_m_before;  // nothing of this call's schedule has been formed
            JF_HIP(e, q);
        }
    }
    if (plan.head_fused) {
        R.kb = 0;
        R.kn = 1;
        *head_out = R;
        *head_fused = true;
        if (plan.transforms.n_tr > 0) {  // (in line: the block completed a big block and the side stream is not used)
            e->post_tr = true;
            e->post_tr_p = plan.transforms;
        }
    }
    e->last_plan = plan;
    e->side_tr = side_wanted;

    if (er) JF_HIP(e, hipEventRecord(er->b, e->stream));
    e->rv_head = (e->rv_head + K) % e->rv_Rg;
    e->rv_blocks += K;
    if (defer_small) e->rv_small_stale = true;  // (a stale state from before is obsolete now: older than the head reaches)
    return JF_OK;
}

// What run_reverb_stage left for the side stream, submitted once the block's own kernels (the spatialiser's too) are in the
// engine's stream: the side stream waits for them -- it then works beside what FOLLOWS the block (in real time: nothing; in a
// run of calls back to back: the next blocks, which find room because its long kernel is launched narrow) -- and the block's
// own kernels are not held up by it.
int submit_side(jf_engine *e) {
    if (!e->side_tr) return JF_OK;
    JF_HIP(e, hipEventRecord(e->rv_ev_main, e->stream));
    JF_HIP(e, hipStreamWaitEvent(e->rv_side, e->rv_ev_main, 0));
    JF_HIP(e, launch_reverb_big_side(&e->side_p[0], &e->side_p[1], e->rv_side));
    // launched: TAIL up to side_fut_m will be there (nothing before this line may claim so -- a stage whose launches failed
    // must leave the schedule asking for them again)
    e->side_tr = false;
    if (e->rv_fut_m < e->side_fut_m) e->rv_fut_m = e->side_fut_m;
    e->rv_side_urgent = e->side_urgent;
    e->rv_side_busy = true;
    JF_HIP(e, hipEventRecord(e->rv_ev_side, e->rv_side));
    return JF_OK;
}

// The pre-interpolated rows, built on first use (jf_engine::interp_avail): a table of 710 + kInterpRows rows takes the place of
// the 710-row one -- the measured rows copied, the weighted sums formed behind them on the engine's stream.  Without room
// for the 386 MB the engine goes on without rows (per-block weighting), for good.
static int ensure_interp_rows(jf_engine *e) {
    if (e->interp_built || !e->interp_avail) return JF_OK;
    float4 *big = nullptr;
    const size_t n_rows = (size_t)e->rt.n_rows;
    if (hipMalloc(&big, sizeof(float4) * (n_rows + kInterpRows) * 512) != hipSuccess) {
        (void)hipGetLastError();
        e->interp_avail = false;
        e->interp_use = 0;
        return JF_OK;
    }
    hipError_t q = hipMemcpyAsync(big, e->d_htab, sizeof(float4) * n_rows * 512, hipMemcpyDeviceToDevice, e->stream);
    if (q == hipSuccess) q = launch_table_interp_build(e->rt, corrected_rule(e) ? 1 : 0, big, e->stream);
    if (q == hipSuccess) q = hipStreamSynchronize(e->stream);  // (everything that reads the old table has finished as well)
    if (q != hipSuccess) {
        (void)hipFree(big);
        JF_HIP(e, q);
    }
    (void)hipFree(e->d_htab);
    e->d_htab = big;
    e->interp_built = true;
    return JF_OK;
}

// The next block's stage launched ahead (jf_engine::rv_ahead) is taken back: see there.
static int rv_ahead_discard(jf_engine *e) {
    if (!e->rv_ahead) return JF_OK;
    JF_HIP(e, hipStreamSynchronize(e->stream));  // nothing of it is still being written
    e->rv_head = e->rv_book.rv_head;
    e->rv_blocks = e->rv_book.rv_blocks;
    e->rv_fut_m = e->rv_book.rv_fut_m;
    e->last_rv_form = e->rv_book.last_rv_form;
    e->last_plan = e->rv_book.last_plan;
    e->last_side = e->rv_book.last_side;
    e->last_catchup = e->rv_book.last_catchup;
    e->last_small_fft = e->rv_book.last_small_fft;
    e->side_tr = false;  // (what the stage wanted on the side stream had not been submitted yet)
    e->post_tr = false;
    // the stage may have made the engine's stream wait for the side stream (and cleared these): waited it has, so leave them
    e->rv_ahead = false;
    e->kernels_use_frozen = false;
    return JF_OK;
}

// May the stage of the block after the one just launched go ahead?  A plain head only: the block completes no big block (its
// transforms would have to follow its spatialiser), owes no TAIL, the side stream has nothing urgent, nothing is put off.
static bool rv_ahead_possible(const jf_engine *e) {
    if (e->rv_P <= 0 || !e->rv_ahead_on || e->rv_form != 0 || e->profiling || e->rv_head_fused || e->rv_small_stale) return false;
    if (e->S > e->rt_max_sources || e->S >= 2048) return false;  // (the one-launch path; the one-block head kernel's range)
    if (e->paused.load(std::memory_order_relaxed)) return false;
    if (e->rv_P1 > 0) {
        if (e->rv_side_urgent || e->side_tr) return false;
        const ReverbSchedule sc = host_reverb_schedule(e->rv_blocks, 1, e->rv_M, e->rv_fut_m);
        if (sc.tail_early >= 0 || sc.tail_late >= 0) return false;
        // a block that completes a big block: only if its transforms and products go to the side stream (they are submitted
        // behind ITS spatialiser, by the call that consumes the stage: side_tr stays pending till then)
        if (sc.n_tr > 0 && !(e->rv_async && e->rv_side != nullptr)) return false;
    }
    return true;
}

// prep -> [reverb] -> fused -> mix on the engine stream, K blocks starting at d_pos.
// first_block: index of d_pos's first block in the uploaded trajectory (jf_batch_run), -1 for positions from elsewhere.
int run_blocks(jf_engine *e, const float *d_pos, int K, float *d_mix_out, int first_block = -1) {
    if (device_fault(e)) return fail(e, JF_ERR_DEVICE, kHandOffMsg);  // fatal: see device_fault
    {
        const int rc = rv_ahead_discard(e);  // (a stage launched ahead by a one-block call: this call does its own)
        if (rc) return rc;
    }
    const int p = e->cur;
    EventPair *ep = nullptr, *ef = nullptr, *em = nullptr;
    // a pair of event records costs ~7 us of stream time: they may be put around every n-th run only (the runs in
    // between launch the same kernels, untimed)
    const bool timed = e->profiling && (e->profile_stride <= 1 || e->profile_calls++ % e->profile_stride == 0);
    e->timed_now = timed;
    if (timed) {
        ef = next_events(e, e->ev_fused);
        if (!ef) return fail(e, JF_ERR_DEVICE, "hipEventCreate failed");
    }
    if (e->profiling >= 2 && timed) {
        ep = next_events(e, e->ev_prep);
        em = next_events(e, e->ev_mix);
        if (!ep || !em) return fail(e, JF_ERR_DEVICE, "hipEventCreate failed");
    }
    // sources a pair of wavefronts sums before it stores a stereo block: as many as leave about two units for every
    // resident pair (2048 on MI355X): larger groups mean fewer inverse transforms and fewer partial blocks for the
    // mix (profiles/group_sweep.py times every size against this choice)
    const long long n_items = (long long)K * e->S;
    const int G = e->src_group > 0 ? e->src_group
                  : (e->S % 32 == 0 && n_items >= 131072) ? 32
                  : (e->S % 16 == 0 && n_items >= 32768) ? 16
                  : (e->S % 8 == 0 && n_items >= 16384) ? 8
                  : (e->S % 4 == 0 && n_items >= 8192) ? 4
                  : (e->S % 2 == 0 && (n_items >= 4096 || e->S >= 1024)) ? 2
                                                         : 1;
    FusedParams P;
    P.G = (e->S % G == 0) ? G : 1;
    const int canon = P.G > 1;  // descriptors in the pair-kernel layout
    // whole-degree positions as pre-interpolated rows: the pair kernel's descriptors only
    bool rows = canon && e->interp_avail && e->interp_use != 0;
    if (rows && e->interp_use == 2 && first_block >= 0 && (size_t)(first_block + K) < e->traj_moved.size()) {
        const double moved = (double)(e->traj_moved[first_block + K] - e->traj_moved[first_block]) / (double)n_items;
        rows = moved <= kInterpMovedMax;
    }
    // a one-block call is the audio callback's (jf_submit_block with more sources than the one-launch kernel takes, or while
    // profiling): it never pays the 386 MB allocation, the build and the stream synchronisation -- it weights per block
    // (bit-identical) until a batch run, or the pre-warm call jf_debug_set_interp_table(e, 1), has built the rows
    if (rows && !e->interp_built && K == 1 && first_block < 0) rows = false;
    if (rows && !e->interp_built) {  // the first batch run that takes them builds them
        const int rc = ensure_interp_rows(e);
        if (rc) return rc;
        rows = e->interp_built;
    }
    e->last_rows = rows;
    const int mode_now = kernel_mode(e) | (rows ? kModeInterpRows : 0);
    // per-kernel timing (profiling >= 2) keeps prep and mix as launches of their own
    const bool have = e->ahead.valid && e->profiling < 2 && first_block >= 0 && e->ahead.first == first_block &&
                      e->ahead.K == K && e->ahead.mode == mode_now && e->ahead.canon == canon &&
                      e->ahead.traj_gen == e->traj_gen;
    e->ahead.valid = false;
    e->last_prep_skipped = have;
    if (have) std::swap(e->d_desc, e->d_desc_ahead);
    if (ep) JF_HIP(e, hipEventRecord(ep->a, e->stream));
    if (!have) JF_HIP(e, launch_prep(e->rt, mode_now, d_pos, e->d_state[p], e->d_desc, e->S, K, canon, e->stream));
    if (ep) JF_HIP(e, hipEventRecord(ep->b, e->stream));
    {
        const int rc = run_reverb_stage(e, p, K);
        if (rc) return rc;
    }
    P.htab = e->d_htab;
    P.tw = e->d_twpack;
    P.desc = e->d_desc;
    P.sigs = e->rv_P > 0 ? e->d_sigs_wet : e->d_sigs;
    P.st_in = e->d_state[p];
    P.st_out = e->d_state[p ^ 1];
    P.hist_in = e->d_hist[p];
    P.hist_out = e->d_hist[p ^ 1];
    P.pos = d_pos;
    P.partial = e->d_partial;
    P.S = e->S;
    P.K = K;
    P.B = e->B;
    e->last_group = P.G;
    P.mode = mode_now;
    P.err = e->hd_err;
    P.order = e->d_order;
    // the window that follows in the trajectory, if there is a whole one: its descriptors are prepared by this run --
    // inside the pair kernel's own launch (trailing workgroups, in the kernel's tail), else inside the mix launch
    const bool ahead_ok = e->prep_ahead && e->profiling < 2 && first_block >= 0 && first_block + 2 * K <= e->traj_blocks;
    const bool ahead_in_fused = ahead_ok && P.G > 1;
    P.n_pair_wgs = 0;
    P.prep_pos = ahead_in_fused ? d_pos + (size_t)K * e->S * 5 : nullptr;
    P.prep_desc = e->d_desc_ahead;
    P.prep_K = K;
    P.prep_canon = canon;
    P.rt = e->rt;
    int max_wgs = e->resident_wgs[P.G > 1 ? (rows ? 2 : 1) : 0];
    if (e->grid_limit > 0 && e->grid_limit < max_wgs) max_wgs = e->grid_limit;
    if (ef) JF_HIP(e, hipEventRecord(ef->a, e->stream));
    JF_HIP(e, launch_fused(P, max_wgs, e->stream));
    if (ef) JF_HIP(e, hipEventRecord(ef->b, e->stream));
    if (em) JF_HIP(e, hipEventRecord(em->a, e->stream));
    e->last_mix_prep = ahead_ok && !ahead_in_fused;
    e->last_fused_prep = ahead_in_fused;
    if (ahead_ok) {
        if (ahead_in_fused)
            JF_HIP(e, launch_mix(e->d_partial, d_mix_out, e->S / P.G, K, e->B, e->stream));
        else
            JF_HIP(e, launch_mix_prep(e->d_partial, d_mix_out, e->S / P.G, K, e->B, e->rt, mode_now,
                                      d_pos + (size_t)K * e->S * 5, e->d_desc_ahead, e->S, K, canon, e->stream));
        e->ahead.valid = true;
        e->ahead.first = first_block + K;
        e->ahead.K = K;
        e->ahead.mode = mode_now;
        e->ahead.canon = canon;
        e->ahead.traj_gen = e->traj_gen;
    } else {
        JF_HIP(e, launch_mix(e->d_partial, d_mix_out, e->S / P.G, K, e->B, e->stream));
    }
    if (em) JF_HIP(e, hipEventRecord(em->b, e->stream));
    if (timed) e->ev_used++;
    e->cur = p ^ 1;
    e->last_rt = false;
    return submit_side(e);
}

void snapshot_positions(jf_engine *e, float *dst /* [S][5] */) {
    std::lock_guard<std::mutex> lk(e->pos_mu);
    for (int s = 0; s < e->S; s++) {
        const HostPos &q = e->pos[s];
        float *d = dst + 5 * s;
        d[0] = q.ele;
        d[1] = q.azi;
        d[2] = q.x;
        d[3] = q.y;
        d[4] = q.z;
    }
}

// The side stream has nothing in flight any more (host-side wait); what it had promised is forgotten.
void quiesce_side(jf_engine *e) {
    if (e->rv_side && e->rv_side_busy) (void)hipStreamSynchronize(e->rv_side);
    e->rv_side_busy = e->rv_side_urgent = false;
}

void free_reverb(jf_engine *e) {
    quiesce_side(e);
    e->side_tr = false;
    e->rv_small_stale = false;
    e->last_side.clear();
    (void)hipFree(e->d_rv_yacc);
    e->d_rv_yacc = nullptr;
    (void)hipFree(e->d_rv_hspec);
    (void)hipFree(e->d_rv_fdl);
    (void)hipFree(e->d_rv_wet);
    (void)hipFree(e->d_sigs_wet);
    for (int i = 0; i < 2; i++) {
        (void)hipFree(e->d_rv_prev[i]);
        (void)hipFree(e->d_rv_count[i]);
        e->d_rv_prev[i] = nullptr;
        e->d_rv_count[i] = nullptr;
    }
    (void)hipFree(e->d_rv_tw1);
    (void)hipFree(e->d_rv_hspec1);
    (void)hipFree(e->d_rv_fdl1);
    (void)hipFree(e->d_rv_ybig);
    (void)hipFree(e->d_rv_dryring);
    (void)hipFree(e->d_rv_fut);
    e->d_rv_tw1 = e->d_rv_hspec1 = e->d_rv_fdl1 = e->d_rv_ybig = nullptr;
    e->d_rv_dryring = e->d_rv_fut = nullptr;
    e->d_rv_hspec = nullptr;
    e->d_rv_fdl = nullptr;
    e->d_rv_wet = nullptr;
    e->d_sigs_wet = nullptr;
    e->rv_P = e->rv_Rg = e->rv_Wr = e->rv_head = 0;
    e->rv_P_total = e->rv_P1 = e->rv_B1 = e->rv_M = e->rv_R1 = e->rv_Rn = e->rv_Fn = e->rv_steps_max = 0;
    e->rv_blocks = e->rv_fut_m = 0;
    e->last_plan = ReverbPlan();
}

// zero one source's (or every source's, src < 0) window, counters and reverb state
int reset_sources(jf_engine *e, int src) {
    e->ahead.valid = false;  // the old position of the next block changes
    const size_t s0 = src < 0 ? 0 : (size_t)src, ns = src < 0 ? (size_t)e->S : 1;
    const int p = e->cur;
    quiesce_side(e);  // (what it has left in the fut ring is zeroed below with the rest)
    JF_HIP(e, hipMemsetAsync(e->d_hist[p] + s0 * kN, 0, sizeof(float) * kN * ns, e->stream));
    JF_HIP(e, hipMemsetAsync(e->d_state[p] + s0, 0, sizeof(SrcState) * ns, e->stream));
    if (e->rv_P > 0) {
        const size_t B = (size_t)e->B;
        JF_HIP(e, hipMemsetAsync(e->d_rv_fdl + s0 * e->rv_Rg * B, 0, sizeof(float2) * e->rv_Rg * B * ns, e->stream));
        JF_HIP(e, hipMemsetAsync(e->d_rv_fdl + (size_t)e->S * e->rv_Rg * B + s0 * e->rv_Rg, 0, sizeof(float2) * e->rv_Rg * ns, e->stream));
        JF_HIP(e, hipMemsetAsync(e->d_rv_wet + s0 * e->rv_Wr, 0, sizeof(float) * e->rv_Wr * ns, e->stream));
        JF_HIP(e, hipMemsetAsync(e->d_rv_prev[p] + s0 * B, 0, sizeof(float) * B * ns, e->stream));
        JF_HIP(e, hipMemsetAsync(e->d_rv_count[p] + s0, 0, sizeof(int) * ns, e->stream));
        if (e->rv_P1 > 0) {  // the big partitions' delay line, the dry ring they read and what they have promised the next blocks
            const size_t B1 = (size_t)e->rv_B1;
            JF_HIP(e, hipMemsetAsync(e->d_rv_fdl1 + s0 * e->rv_R1 * B1, 0, sizeof(float2) * e->rv_R1 * B1 * ns, e->stream));
            JF_HIP(e, hipMemsetAsync(e->d_rv_fdl1 + (size_t)e->S * e->rv_R1 * B1 + s0 * e->rv_R1, 0, sizeof(float2) * e->rv_R1 * ns, e->stream));
            JF_HIP(e, hipMemsetAsync(e->d_rv_dryring + s0 * e->rv_Rn * B1, 0, sizeof(float) * e->rv_Rn * B1 * ns, e->stream));
            JF_HIP(e, hipMemsetAsync(e->d_rv_fut + s0 * e->rv_Fn * B1, 0, sizeof(float) * e->rv_Fn * B1 * ns, e->stream));
        }
    }
    return JF_OK;
}

void destroy_engine(jf_engine *e) {
    if (!e) return;
    DeviceGuard bind(e);
    if (e->rv_side) (void)hipStreamSynchronize(e->rv_side);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    free_reverb(e);
    for (float *p : e->d_signal)
        if (p) (void)hipFree(p);
    (void)hipFree(e->d_htab);
    (void)hipFree(e->d_tw);
    (void)hipFree(e->d_twpack);
    (void)hipFree(e->d_sigs);
    (void)hipFree(e->d_zero);
    for (int i = 0; i < 2; i++) {
        (void)hipFree(e->d_state[i]);
        (void)hipFree(e->d_hist[i]);
    }
    (void)hipFree(e->d_desc);
    (void)hipFree(e->d_desc_ahead);
    (void)hipFree(e->d_partial);
    (void)hipFree(e->d_mix);
    (void)hipFree(e->d_pos_rt);
    (void)hipFree(e->d_traj);
    (void)hipFree(e->d_order);
    (void)hipFree(e->d_pick);
    if (e->h_pos_pinned) (void)hipHostFree(e->h_pos_pinned);
    if (e->h_out_pinned) (void)hipHostFree(e->h_out_pinned);
    if (e->h_done) (void)hipHostFree(e->h_done);
    if (e->h_err) (void)hipHostFree(e->h_err);
    for (auto *pool : {&e->ev_prep, &e->ev_fused, &e->ev_mix, &e->ev_reverb})
        for (auto &p : *pool) {
            (void)hipEventDestroy(p.a);
            (void)hipEventDestroy(p.b);
        }
    if (e->rv_ev_main) (void)hipEventDestroy(e->rv_ev_main);
    if (e->rv_ev_side) (void)hipEventDestroy(e->rv_ev_side);
    if (e->rv_side) (void)hipStreamDestroy(e->rv_side);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

// grid: the table of the HRTF set's measurement grid (null: the reference's KEMAR grid, 710 rows)
int create_engine(const jf_config *cfg, const RingTable *grid, const float *hrir, int taps, jf_engine **out) {
    if (!cfg || !hrir || !out) return fail(nullptr, JF_ERR_ARG, "null argument");
    *out = nullptr;
    const int B = cfg->frames_per_buffer;
    if (B < 64 || B > 256 || B % 64) return fail(nullptr, JF_ERR_ARG, "frames_per_buffer must be 64, 128, 192 or 256");
    if (cfg->hrtf_len <= 0 || taps <= 0 || taps > cfg->hrtf_len)
        return fail(nullptr, JF_ERR_ARG, "need 0 < taps <= hrtf_len");
    // PAD_LEN = 2^ceil(log2(B + L - 1)) (Universal.cuh:12); the kernels are built for 1024
    const int pad = (int)pow(2, ceil(log2((double)(B + cfg->hrtf_len - 1))));
    if (pad != kN) return fail(nullptr, JF_ERR_ARG, "frames_per_buffer + hrtf_len - 1 must pad to 1024");
    if (cfg->n_sources <= 0) return fail(nullptr, JF_ERR_ARG, "n_sources must be positive");
    if (cfg->max_batch_blocks <= 0) return fail(nullptr, JF_ERR_ARG, "max_batch_blocks must be positive");
    if (cfg->flags & ~(JF_FLAG_CORRECTED_INTERPOLATION | JF_FLAG_NO_INTERP_TABLE))
        return fail(nullptr, JF_ERR_ARG, "unknown bits in flags");
    if (kernels_build_kind() != 0) {
        // a fault-injection or timing-only build of the kernels (jf_experiments.h): wrong results by design
        const char *allow = getenv("JF_ALLOW_EXPERIMENT");
        if (!allow || strcmp(allow, "1") != 0)
            return fail(nullptr, JF_ERR_STATE,
                        "this library was built with an experiment switch that gives wrong results by design "
                        "(set JF_ALLOW_EXPERIMENT=1 to use it in a test)");
    }

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, JF_ERR_DEVICE, "no HIP device available (this library has no CPU path)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, JF_ERR_ARG, "device ordinal out of range");

    jf_engine *e = new jf_engine();
    e->cfg = *cfg;
    e->B = B;
    e->S = cfg->n_sources;
    e->maxK = cfg->max_batch_blocks;
    const size_t S = (size_t)e->S, K = (size_t)e->maxK;
    int rc = JF_OK;
    int prev_dev = -1;
    (void)hipGetDevice(&prev_dev);
    auto body = [&]() -> int {
        JF_HIP(e, hipSetDevice(cfg->device));
        {
            // the engine's stream at the highest priority the device offers, the side stream (the reverb's work ahead of time,
            // run_reverb_stage) at the lowest: where the two meet, the block in hand goes first
            int lo = 0, hi = 0;
            JF_HIP(e, hipDeviceGetStreamPriorityRange(&lo, &hi));
            JF_HIP(e, hipStreamCreateWithPriority(&e->stream, hipStreamNonBlocking, hi));
            JF_HIP(e, hipStreamCreateWithPriority(&e->rv_side, hipStreamNonBlocking, lo));
        }
        JF_HIP(e, hipEventCreateWithFlags(&e->rv_ev_main, hipEventDisableTiming));
        JF_HIP(e, hipEventCreateWithFlags(&e->rv_ev_side, hipEventDisableTiming));
        for (int kind = 0; kind < 3; kind++) JF_HIP(e, fused_resident_workgroups(B / 64, kind, &e->resident_wgs[kind]));
        {
            int cus = 0;
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess && cus >= 16)
                e->rv_side_wgs = 3 * cus / 4;
            else
                (void)hipGetLastError();
        }
        // (nothing of the engine's behaviour is read from the environment: jefferson_debug.h's setters are the overrides)
        e->interp_avail = !(cfg->flags & JF_FLAG_NO_INTERP_TABLE);
        e->interp_use = e->interp_avail ? 2 : 0;
        // the 710 measured rows only; the pre-interpolated ones come with the first run that takes them (ensure_interp_rows)
        e->rt = grid ? *grid : ring_table();
        JF_HIP(e, hipMalloc(&e->d_htab, sizeof(float4) * (size_t)e->rt.n_rows * 512));
        JF_HIP(e, hipMalloc(&e->d_tw, sizeof(float2) * 1024));
        JF_HIP(e, hipMalloc(&e->d_sigs, sizeof(SrcSignal) * S));
        for (int i = 0; i < 2; i++) {
            JF_HIP(e, hipMalloc(&e->d_state[i], sizeof(SrcState) * S));
            JF_HIP(e, hipMalloc(&e->d_hist[i], sizeof(float) * S * kN));
            JF_HIP(e, hipMemsetAsync(e->d_state[i], 0, sizeof(SrcState) * S, e->stream));
            JF_HIP(e, hipMemsetAsync(e->d_hist[i], 0, sizeof(float) * S * kN, e->stream));
        }
        JF_HIP(e, hipMalloc(&e->d_desc, sizeof(ItemDesc) * S * K));
        JF_HIP(e, hipMalloc(&e->d_desc_ahead, sizeof(ItemDesc) * S * K));
        JF_HIP(e, hipMalloc(&e->d_partial, sizeof(float) * S * K * 2 * B));
        JF_HIP(e, hipMalloc(&e->d_mix, sizeof(float) * K * 2 * B));
        JF_HIP(e, hipMalloc(&e->d_pos_rt, sizeof(float) * S * 5));
        e->rt.pick = nullptr;
        if (e->rt.kemar) {
            // nearest table row per (ring, integer azimuth), by the search itself (host_pick_hrtf = hrtf_signals.cu:20-51)
            static const int elev[kNumElev] = {-40, -30, -20, -10, 0, 10, 20, 30, 40, 50, 60, 70, 80, 90};
            std::vector<short> pick((size_t)kNumElev * kPickAzi);
            for (int r = 0; r < kNumElev; r++)
                for (int a = 0; a < kPickAzi; a++) pick[(size_t)r * kPickAzi + a] = (short)host_pick_hrtf((float)elev[r], (float)a);
            JF_HIP(e, hipMalloc(&e->d_pick, sizeof(short) * pick.size()));
            JF_HIP(e, h2d(e, e->d_pick, pick.data(), sizeof(short) * pick.size()));
            e->rt.pick = e->d_pick;
        }
        JF_HIP(e, hipMalloc(&e->d_order, sizeof(int) * S));
        e->order.resize(S);
        for (size_t s = 0; s < S; s++) e->order[s] = (int)s;
        JF_HIP(e, h2d(e, e->d_order, e->order.data(), sizeof(int) * S));
        // host memory the kernels read and write in place, and whose words the host polls while a kernel runs: mapped AND
        // coherent (fine-grained) explicitly -- not left to the runtime's default or to HIP_HOST_COHERENT
        const unsigned kHostFlags = hipHostMallocMapped | hipHostMallocCoherent;
        JF_HIP(e, hipHostMalloc(&e->h_pos_pinned, sizeof(float) * S * 5, kHostFlags));
        JF_HIP(e, hipHostMalloc(&e->h_out_pinned, sizeof(float) * 2 * B * kRtMaxWgs, kHostFlags));
        JF_HIP(e, hipHostGetDevicePointer((void **)&e->hd_pos, e->h_pos_pinned, 0));
        JF_HIP(e, hipHostGetDevicePointer((void **)&e->hd_out, e->h_out_pinned, 0));
        JF_HIP(e, hipHostMalloc(&e->h_done, sizeof(int) * kRtMaxWgs, kHostFlags));
        memset(e->h_done, 0, sizeof(int) * kRtMaxWgs);
        JF_HIP(e, hipHostGetDevicePointer((void **)&e->hd_done, e->h_done, 0));
        // the error word, followed by 64 KB that timing experiments of the kernels may fill (JF_EXP_STAMPS)
        JF_HIP(e, hipHostMalloc(&e->h_err, sizeof(int) * 4 + 65536, kHostFlags));
        memset(e->h_err, 0, sizeof(int) * 4 + 65536);
        JF_HIP(e, hipHostGetDevicePointer((void **)&e->hd_err, e->h_err, 0));
        e->d_signal.assign(S, nullptr);
        JF_HIP(e, hipMalloc(&e->d_zero, sizeof(float) * kN));
        JF_HIP(e, hipMemsetAsync(e->d_zero, 0, sizeof(float) * kN, e->stream));
        e->h_sigs.assign(S, SrcSignal{e->d_zero, kN, 0});
        JF_HIP(e, h2d(e, e->d_sigs, e->h_sigs.data(), sizeof(SrcSignal) * S));
        // SoundSource::SoundSource() defaults (SoundSource.cu:3-16)
        e->pos.assign(S, HostPos{0.0f, 0.0f, 0.5f, 0.0f, 0.0f, 0.5f});

        // twiddles exp(+2 pi i j / 1024) from double
        std::vector<float2> tw(1024);
        for (int j = 0; j < 1024; j++) {
            const double a = 2.0 * 3.14159265358979323846264338327950288 * j / 1024.0;
            tw[j] = make_float2((float)cos(a), (float)sin(a));
        }
        JF_HIP(e, h2d(e, e->d_tw, tw.data(), sizeof(float2) * 1024));
        // the same values re-laid per FFT pass (jf_device.h kTw*)
        std::vector<float2> pack(kTwPack);
        for (int lane = 0; lane < 64; lane++) {
            const int a = lane & 3, i = lane >> 2;
            for (int t = 0; t < 16; t++) pack[kTwW3 + 64 * t + lane] = tw[(a * (i + 16 * t) + 768 * a) & 1023];
            for (int q = 0; q < 8; q++) pack[kTwU + 64 * q + lane] = tw[lane + 64 * q];
            for (int r = 0; r < 8; r++) pack[kTwWC + 64 * r + lane] = tw[(2 * r * lane) & 1023];
        }
        for (int m = 0; m < 16; m++)
            for (int i = 0; i < 16; i++) pack[kTwW2 + 16 * m + i] = tw[(4 * i * m) & 1023];
        for (int r = 0; r < 8; r++)
            for (int k = 0; k < 8; k++) pack[kTwWB + 8 * r + k] = tw[(16 * r * k) & 1023];
        JF_HIP(e, hipMalloc(&e->d_twpack, sizeof(float2) * kTwPack));
        JF_HIP(e, h2d(e, e->d_twpack, pack.data(), sizeof(float2) * kTwPack));

        // HRTF spectra on the GPU (read_hrtf_signals + transform_hrtfs)
        float *d_hrir = nullptr;
        const size_t hb = sizeof(float) * (size_t)e->rt.n_rows * 2 * (size_t)taps;
        JF_HIP(e, hipMalloc(&d_hrir, hb));
        hipError_t s1 = h2d(e, d_hrir, hrir, hb);
        hipError_t s2 = s1 == hipSuccess ? launch_table_build(d_hrir, e->rt.n_rows, taps, e->d_twpack, e->d_htab, e->stream) : s1;
        hipError_t s3 = s2 == hipSuccess ? hipStreamSynchronize(e->stream) : s2;
        (void)hipFree(d_hrir);
        JF_HIP(e, s3);
        return JF_OK;
    };
    rc = body();
    if (rc != JF_OK) {
        g_create_error = e->err;
        destroy_engine(e);
        e = nullptr;
    }
    if (prev_dev >= 0 && prev_dev != cfg->device) (void)hipSetDevice(prev_dev);  // leave the caller's device as it was
    *out = e;
    return rc;
}

}  // namespace

// =============================================================== C ABI ====
// Nothing may propagate through the C ABI: host allocations (std::vector, std::string) can throw.
template <class F>
static int jf_guard(F &&f) noexcept {
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

extern "C" {

int jf_engine_create(const jf_config *cfg, const float *hrir, int taps, jf_engine **out) {
    return jf_guard([&]() -> int {
    return create_engine(cfg, nullptr, hrir, taps, out);
    });
}

// ---- any grid of elevation rings ("SOFA / other HRTF sets", FuturePlans.md:21) ----
static const float kKemarEle[kNumElev] = {-40, -30, -20, -10, 0, 10, 20, 30, 40, 50, 60, 70, 80, 90};
static int g_kemar_count[kNumElev];

int jf_kemar_grid(jf_hrtf_grid *out) {
    if (!out) return JF_ERR_ARG;
    const RingTable &k = ring_table();
    for (int r = 0; r < kNumElev; r++) g_kemar_count[r] = k.offset[r + 1] - k.offset[r];  // (the same values whoever writes them)
    out->n_rings = kNumElev;
    out->ring_elevation = kKemarEle;
    out->ring_count = g_kemar_count;
    out->ring_step = kemar_ring_steps();
    return JF_OK;
}

static int grid_table(const jf_hrtf_grid *grid, RingTable *rt) {
    if (!grid) return fail(nullptr, JF_ERR_ARG, "null grid");
    std::string err;
    const int rc = host_grid_table(grid->n_rings, grid->ring_elevation, grid->ring_count, grid->ring_step, rt, &err);
    return rc ? fail(nullptr, rc, err) : JF_OK;
}

int jf_grid_from_positions(size_t n, const float *azimuth_deg, const float *elevation_deg, float tol_deg, jf_grid_layout *layout,
                           int *row_of) {
    return jf_guard([&]() -> int {
    if (!layout) return fail(nullptr, JF_ERR_ARG, "null layout");
    std::string err;
    const int rc = host_grid_from_positions(n, azimuth_deg, elevation_deg, tol_deg, &layout->n_rings, layout->ring_elevation,
                                            layout->ring_count, layout->ring_step, row_of, &err);
    return rc ? fail(nullptr, rc, err) : JF_OK;
    });
}

int jf_grid_rows(const jf_hrtf_grid *grid) {
    return jf_guard([&]() -> int {
    RingTable rt;
    const int rc = grid_table(grid, &rt);
    return rc ? rc : rt.n_rows;
    });
}

int jf_grid_interpolation(const jf_hrtf_grid *grid, float ele, float azi, int idx[4], float omegas[6]) {
    return jf_guard([&]() -> int {
    if (!idx || !omegas) return JF_ERR_ARG;
    RingTable rt;
    const int rc = grid_table(grid, &rt);
    return rc ? rc : host_grid_interpolation(rt, ele, azi, idx, omegas);
    });
}

int jf_grid_pick(const jf_hrtf_grid *grid, float ele, float azi) {
    return jf_guard([&]() -> int {
    RingTable rt;
    const int rc = grid_table(grid, &rt);
    if (rc) return rc;
    if (!(ele >= -1.0e6f && ele <= 1.0e6f) || !(azi > -1.0e6f && azi < 1.0e6f)) return JF_ERR_RANGE;
    return host_grid_pick(rt, ele, azi);
    });
}

int jf_engine_create_grid(const jf_config *cfg, const jf_hrtf_grid *grid, const float *hrir, int taps, jf_engine **out) {
    return jf_guard([&]() -> int {
    if (out) *out = nullptr;
    RingTable rt;
    const int rc = grid_table(grid, &rt);
    if (rc) return rc;
    return create_engine(cfg, &rt, hrir, taps, out);
    });
}

// ---- SOFA files (jf_sofa.cpp over jf_hdf5.c) ----
int jf_sofa_read(const char *path, jf_sofa_set *out) {
    return jf_guard([&]() -> int {
    if (!path || !out) return fail(nullptr, JF_ERR_ARG, "null argument");
    std::string err;
    const int rc = sofa_read(path, out, &err);
    return rc ? fail(nullptr, rc, err) : JF_OK;
    });
}

void jf_sofa_release(jf_sofa_set *set) { sofa_release(set); }

int jf_sofa_taps(const jf_sofa_set *set) {
    return jf_guard([&]() -> int {
    std::string err;
    const int rc = sofa_taps(set, &err);
    return rc < 0 ? fail(nullptr, rc, err.empty() ? "not a set read by jf_sofa_read" : err) : rc;
    });
}

int jf_sofa_table(const jf_sofa_set *set, float tol_deg, jf_grid_layout *layout, float *hrir, int taps) {
    return jf_guard([&]() -> int {
    std::string err;
    const int rc = sofa_table(set, tol_deg, layout, hrir, taps, &err);
    return rc ? fail(nullptr, rc, err) : JF_OK;
    });
}

int jf_engine_create_sofa(const jf_config *cfg, const char *path, float tol_deg, jf_engine **out) {
    return jf_guard([&]() -> int {
    if (out) *out = nullptr;
    if (!cfg || !path || !out) return fail(nullptr, JF_ERR_ARG, "null argument");
    jf_sofa_set set;
    std::string err;
    int rc = sofa_read(path, &set, &err);
    if (rc) return fail(nullptr, rc, err);
    struct Release {
        jf_sofa_set *s;
        ~Release() { sofa_release(s); }
    } release{&set};
    const int taps = sofa_taps(&set, &err);
    if (taps < 0) return fail(nullptr, taps, std::string(path) + ": " + err);
    if (taps > cfg->hrtf_len)
        return fail(nullptr, JF_ERR_ARG, std::string(path) + ": impulse responses of " + std::to_string(taps) + " taps, hrtf_len is " + std::to_string(cfg->hrtf_len));
    std::vector<float> hrir((size_t)set.n_measurements * 2 * (size_t)taps);
    jf_grid_layout lay;
    rc = sofa_table(&set, tol_deg, &lay, hrir.data(), taps, &err);
    if (rc) return fail(nullptr, rc, std::string(path) + ": " + err);
    RingTable rt;
    rc = host_grid_table(lay.n_rings, lay.ring_elevation, lay.ring_count, lay.ring_step, &rt, &err);
    if (rc) return fail(nullptr, rc, std::string(path) + ": " + err);
    return create_engine(cfg, &rt, hrir.data(), taps, out);
    });
}

int jf_debug_hdf5_read(const char *path, const char *dataset, double **out, int *rank, unsigned long long *dims) {
    return jf_guard([&]() -> int {
    if (!path || !dataset || !out || !rank || !dims) return fail(nullptr, JF_ERR_ARG, "null argument");
    std::string err;
    const int rc = hdf5_read(path, dataset, out, rank, dims, &err);
    return rc ? fail(nullptr, rc, err) : JF_OK;
    });
}

int jf_debug_hdf5_attr(const char *path, const char *object, const char *attr, char *out, size_t cap) {
    return jf_guard([&]() -> int {
    if (!path || !object || !attr || !out || !cap) return fail(nullptr, JF_ERR_ARG, "null argument");
    std::string err;
    const int rc = hdf5_attr(path, object, attr, out, cap, &err);
    return rc ? fail(nullptr, rc, err) : JF_OK;
    });
}

int jf_engine_create_from_dir(const jf_config *cfg, const char *hrir_dir, jf_engine **out) {
    return jf_guard([&]() -> int {
    if (!cfg || !hrir_dir || !out) return fail(nullptr, JF_ERR_ARG, "null argument");
    std::vector<float> hrir;
    int taps = 0;
    std::string err;
    int rc = load_hrir_dir(hrir_dir, &hrir, &taps, &err);
    if (rc) return fail(nullptr, rc, err);
    return create_engine(cfg, nullptr, hrir.data(), taps, out);
    });
}

void jf_engine_destroy(jf_engine *e) { destroy_engine(e); }

const char *jf_last_error(const jf_engine *e) { return e ? e->err.c_str() : g_create_error.c_str(); }

int jf_frames_per_buffer(const jf_engine *e) { return e ? e->B : JF_ERR_ARG; }
int jf_pad_len(const jf_engine *e) { return e ? kN : JF_ERR_ARG; }
int jf_num_sources(const jf_engine *e) { return e ? e->S : JF_ERR_ARG; }
int jf_table_rows(const jf_engine *e) { return e ? e->rt.n_rows : JF_ERR_ARG; }

int jf_source_set_signal(jf_engine *e, int src, const float *mono, size_t n) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!valid_src(e, src) || (n && !mono) || n > 0x7fffffffu) return fail(e, JF_ERR_ARG, "bad source or signal");
    {
        const int rc = rv_ahead_discard(e);  // (the stage launched ahead read the old signal)
        if (rc) return rc;
    }
    JF_HIP(e, hipStreamSynchronize(e->stream));
    if (e->rv_side && e->rv_side_busy) JF_HIP(e, hipStreamSynchronize(e->rv_side));  // its transforms read the signals
    // The device copy always has length >= PAD_LEN so that the kernel wraps the loop with
    // one conditional subtract: a shorter signal is stored as whole repetitions of itself
    // (the looped stream is identical), an empty one as the shared zero buffer.
    float *d_new = nullptr;
    size_t n_dev = n;
    if (n) {
        const float *src_host = mono;
        std::vector<float> tiled;
        if (n < (size_t)kN) {
            const size_t reps = ((size_t)kN + n - 1) / n;
            tiled.resize(reps * n);
            for (size_t r = 0; r < reps; r++) memcpy(tiled.data() + r * n, mono, sizeof(float) * n);
            src_host = tiled.data();
            n_dev = reps * n;
        }
        JF_HIP(e, hipMalloc(&d_new, sizeof(float) * n_dev));
        hipError_t st = h2d(e, d_new, src_host, sizeof(float) * n_dev);
        if (st != hipSuccess) {
            (void)hipFree(d_new);
            JF_HIP(e, st);
        }
    }
    if (e->d_signal[src]) (void)hipFree(e->d_signal[src]);
    e->d_signal[src] = d_new;
    e->h_sigs[src] = n ? SrcSignal{d_new, (int)n_dev, 0} : SrcSignal{e->d_zero, kN, 0};
    JF_HIP(e, h2d(e, e->d_sigs + src, &e->h_sigs[src], sizeof(SrcSignal)));
    const int zero = 0;  // count = 0 (cudaPart.cu:198-199 run with a fresh source)
    if (e->rv_P > 0)  // the play position of the dry signal lives in the reverb stage
        JF_HIP(e, h2d(e, e->d_rv_count[e->cur] + src, &zero, sizeof(int)));
    else
        JF_HIP(e, h2d(e, &e->d_state[e->cur][src].count, &zero, sizeof(int)));
    return JF_OK;
    });
}

int jf_source_set_cartesian(jf_engine *e, int src, float x, float y, float z) {
    return jf_guard([&]() -> int {
    if (!valid_src(e, src)) return fail(e, JF_ERR_ARG, "bad source index");
    float rec[5], r;
    int rc = host_from_cartesian(x, y, z, rec, &r);
    if (rc) return fail(e, rc, "zero or non-finite coordinates");
    if (!elevation_ok(e, rec[0])) return fail(e, JF_ERR_RANGE, elevation_msg(e));
    std::lock_guard<std::mutex> lk(e->pos_mu);
    e->pos[src] = HostPos{rec[0], rec[1], r, x, y, z};
    return JF_OK;
    });
}

int jf_source_set_spherical(jf_engine *e, int src, float ele, float azi, float r) {
    return jf_guard([&]() -> int {
    if (!valid_src(e, src)) return fail(e, JF_ERR_ARG, "bad source index");
    float rec[5];
    host_from_spherical(ele, azi, r, rec);
    if (!elevation_ok(e, rec[0])) return fail(e, JF_ERR_RANGE, elevation_msg(e));
    if (!(fabsf(rec[1]) < 1.0e6f) || !(fabsf(r) < 3.0e38f)) return fail(e, JF_ERR_RANGE, "non-finite azimuth or radius");
    std::lock_guard<std::mutex> lk(e->pos_mu);
    e->pos[src] = HostPos{rec[0], rec[1], r, rec[2], rec[3], rec[4]};
    return JF_OK;
    });
}

int jf_source_get_position(const jf_engine *e, int src, float out[6]) {
    return jf_guard([&]() -> int {
    if (!valid_src(e, src) || !out) return JF_ERR_ARG;
    jf_engine *m = const_cast<jf_engine *>(e);
    std::lock_guard<std::mutex> lk(m->pos_mu);
    const HostPos &q = e->pos[src];
    out[0] = q.ele;
    out[1] = q.azi;
    out[2] = q.r;
    out[3] = q.x;
    out[4] = q.y;
    out[5] = q.z;
    return JF_OK;
    });
}

int jf_source_reset(jf_engine *e, int src) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!valid_src(e, src)) return fail(e, JF_ERR_ARG, "bad source index");
    {
        const int rc = rv_ahead_discard(e);
        if (rc) return rc;
    }
    JF_HIP(e, hipStreamSynchronize(e->stream));
    return reset_sources(e, src);
    });
}

int jf_position_from_spherical(float ele, float azi, float r, float out[JF_POS_FLOATS]) {
    return jf_guard([&]() -> int {
    if (!out) return JF_ERR_ARG;
    host_from_spherical(ele, azi, r, out);
    return JF_OK;
    });
}

int jf_position_from_cartesian(float x, float y, float z, float out[JF_POS_FLOATS]) {
    return jf_guard([&]() -> int {
    if (!out) return JF_ERR_ARG;
    return host_from_cartesian(x, y, z, out, nullptr);
    });
}

int jf_positions_from_spherical(size_t n, const float *ele, const float *azi, const float *r, float *out) {
    return jf_guard([&]() -> int {
    if (n && (!ele || !azi || !r || !out)) return JF_ERR_ARG;
    for (size_t i = 0; i < n; i++) host_from_spherical(ele[i], azi[i], r[i], out + 5 * i);
    return JF_OK;
    });
}

int jf_interpolation(float ele, float azi, int idx[4], float omegas[6]) {
    return jf_guard([&]() -> int {
    if (!idx || !omegas) return JF_ERR_ARG;
    return host_interpolation(ele, azi, idx, omegas);
    });
}

int jf_interpolation_ex(float ele, float azi, unsigned flags, int idx[4], float omegas[6]) {
    return jf_guard([&]() -> int {
    if (!idx || !omegas) return JF_ERR_ARG;
    return (flags & JF_FLAG_CORRECTED_INTERPOLATION) ? host_interpolation_corrected(ele, azi, idx, omegas)
                                                     : host_interpolation(ele, azi, idx, omegas);
    });
}

int jf_pick_hrtf(float ele, float azi) { return host_pick_hrtf(ele, azi); }

// ---- per-block -----------------------------------------------------------
int jf_submit_block(jf_engine *e) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!e) return JF_ERR_ARG;
    if (e->in_flight) return fail(e, JF_ERR_STATE, "a block is already in flight");
    if (device_fault(e)) return fail(e, JF_ERR_DEVICE, kHandOffMsg);
    if (e->paused.load(std::memory_order_relaxed)) {  // Audio.cu:101: nothing is consumed, output is silence
        JF_HIP(e, hipMemsetAsync(e->d_mix, 0, sizeof(float) * 2 * e->B, e->stream));
    } else {
        snapshot_positions(e, e->h_pos_pinned);
        if (e->S <= e->rt_max_sources && !e->profiling) {
            // few sources: ONE launch does descriptors, spatialisation and mix, reading the positions
            // from and writing the stereo block to pinned host memory -- no copies, one sync
            const int p = e->cur;
            e->ahead.valid = false;  // this block moves every source's old position
            ReverbParams head;
            bool head_fused = false;
            e->kernels_use_frozen = false;
            if (e->rv_ahead) {
                e->rv_ahead = false;  // the stage of this block was launched behind the last block's spatialiser: rv_ahead
            } else {
                // the wet ring is then this block's signal (written by the stage's own kernel, or by the real-time kernel's
                // waves themselves: head_fused)
                const int rc = run_reverb_stage(e, p, 1, &head, &head_fused);
                if (rc) return rc;
            }
            FusedParams P;
            P.htab = e->d_htab;
            P.tw = e->d_twpack;
            P.desc = nullptr;
            P.sigs = e->rv_P > 0 ? e->d_sigs_wet : e->d_sigs;
            P.st_in = e->d_state[p];
            P.st_out = e->d_state[p ^ 1];
            P.hist_in = e->d_hist[p];
            P.hist_out = e->d_hist[p ^ 1];
            P.pos = e->hd_pos;
            P.partial = nullptr;
            P.S = e->S;
            P.K = 1;
            P.B = e->B;
            P.G = 1;
            P.err = e->hd_err;
            P.order = e->d_order;
            P.mode = kernel_mode(e);
            // a wave per source, 8 or 16 waves to the workgroup (jf_kernels.hip: rt_block_kernel); at most kRtMaxWgs workgroups
            // = 2048 sources: beyond that a wave takes several
            const int rtw = rt_waves_per_wg(e->S);
            int wgs = (e->S + rtw - 1) / rtw;
            if (wgs > kRtMaxWgs) wgs = kRtMaxWgs;
            e->rt_seq = e->rt_seq == 0x7fffffff ? 1 : e->rt_seq + 1;
            JF_HIP(e, launch_rt_block(P, e->rt, e->hd_pos, e->hd_out, e->hd_done, e->rt_seq, wgs, head_fused ? &head : nullptr,
                                      e->stream));
            if (e->post_tr) {  // X_m of the big block this block completed, behind the kernel that wrote the block's samples
                JF_HIP(e, launch_reverb_big_side(&e->post_tr_p, nullptr, e->stream));
                e->post_tr = false;
            }
            {
                const int rc = submit_side(e);
                if (rc) return rc;
            }
            e->cur = p ^ 1;
            e->last_rt = true;
            e->rt_wgs = wgs;
            e->in_flight = true;
            if (rv_ahead_possible(e)) {
                // the NEXT block's stage, behind this block's spatialiser (jf_engine::rv_ahead)
                e->kernels_frozen = jf_debug_last_kernels(e);
                e->rv_book.rv_head = e->rv_head;
                e->rv_book.rv_blocks = e->rv_blocks;
                e->rv_book.rv_fut_m = e->rv_fut_m;
                e->rv_book.last_rv_form = e->last_rv_form;
                e->rv_book.last_plan = e->last_plan;
                e->rv_book.last_side = e->last_side;
                e->rv_book.last_catchup = e->last_catchup;
                e->rv_book.last_small_fft = e->last_small_fft;
                const int rc = run_reverb_stage(e, e->cur, 1);
                if (rc) return rc;
                e->rv_ahead = true;
                e->kernels_use_frozen = true;
            }
            return JF_OK;
        }
        JF_HIP(e, hipMemcpyAsync(e->d_pos_rt, e->h_pos_pinned, sizeof(float) * 5 * e->S, hipMemcpyHostToDevice,
                                 e->stream));
        int rc = run_blocks(e, e->d_pos_rt, 1, e->d_mix);
        if (rc) return rc;
    }
    JF_HIP(e, hipMemcpyAsync(e->h_out_pinned, e->d_mix, sizeof(float) * 2 * e->B, hipMemcpyDeviceToHost, e->stream));
    e->rt_wgs = 0;
    e->in_flight = true;
    return JF_OK;
    });
}

int jf_collect_block(jf_engine *e, float *out) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!e || !out) return JF_ERR_ARG;
    if (!e->in_flight) return fail(e, JF_ERR_STATE, "no block in flight");
    bool landed = false;
    if (e->rt_wgs > 0) {
        // The real-time kernel says when its blocks are in host memory: poll its words (a few microseconds of spinning on the
        // audio thread, as cudaStreamSynchronize does in the reference, Audio.cu:107) -- and fall back to the stream if they
        // do not come (a faulting kernel must surface as an error, not as a spin).
        // The spin is bounded by TIME (kRtPollNs: two milliseconds, a third of a 256-sample block's real time), read every
        // 256 polls; after that the stream synchronisation below takes over.
        const volatile int *done = e->h_done;
        timespec t0;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (long spins = 0; !landed; spins++) {
            landed = true;
            for (int g = 0; g < e->rt_wgs; g++) landed = landed && done[g] == e->rt_seq;
            if (landed) break;
            __builtin_ia32_pause();
            if ((spins & 255) == 255) {
                timespec t1;
                clock_gettime(CLOCK_MONOTONIC, &t1);
                if ((t1.tv_sec - t0.tv_sec) * 1000000000L + (t1.tv_nsec - t0.tv_nsec) > kRtPollNs) break;
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (!landed) JF_HIP(e, hipStreamSynchronize(e->stream));
    if (device_fault(e)) {  // per-block calls with more than rt_max_sources sources run the pair kernel too
        e->in_flight = false;
        return fail(e, JF_ERR_DEVICE, kHandOffMsg);
    }
    memcpy(out, e->h_out_pinned, sizeof(float) * 2 * e->B);
    // the real-time kernel's workgroups each left the sum of their sources: add them in workgroup order
    for (int g = 1; g < e->rt_wgs; g++) {
        const float *pg = e->h_out_pinned + (size_t)g * 2 * e->B;
        for (int n = 0; n < 2 * e->B; n++) out[n] += pg[n];
    }
    float peak = 0.0f;
    for (int n = 0; n < 2 * e->B; n++) peak = fmaxf(peak, fabsf(out[n]));
    e->last_peak = peak;
    e->in_flight = false;
    return JF_OK;
    });
}

int jf_process_block(jf_engine *e, float *out) {
    return jf_guard([&]() -> int {
    int rc = jf_submit_block(e);
    if (rc) return rc;
    return jf_collect_block(e, out);
    });
}

int jf_callback(jf_engine *e, float *out) {
    return jf_guard([&]() -> int {
    if (!e || !out) return JF_ERR_ARG;
    int rc;
    if (e->have_prev) {
        rc = jf_collect_block(e, out);
        if (rc) return rc;
    } else {
        memset(out, 0, sizeof(float) * 2 * e->B);  // intermediate[] before the first block
    }
    rc = jf_submit_block(e);
    if (rc) return rc;
    e->have_prev = true;
    return JF_OK;
    });
}

int jf_pa_callback(const void *, void *output, unsigned long frames, const void *, unsigned long, void *user) {
    return jf_guard([&]() -> int {
    jf_engine *e = (jf_engine *)user;
    if (!output) return 0;
    // a stream opened with another buffer size, or an engine error: hand PortAudio silence, never garbage
    if (!e || frames != (unsigned long)e->B || jf_callback(e, (float *)output) != JF_OK)
        memset(output, 0, sizeof(float) * 2 * frames);
    return 0;
    });
}

int jf_set_mode(jf_engine *e, int mode) {
    return jf_guard([&]() -> int {
    if (!e || (mode != JF_MODE_FD_COMPLEX && mode != JF_MODE_FD_BASIC)) return fail(e, JF_ERR_ARG, "unknown mode");
    e->mode.store(mode, std::memory_order_relaxed);  // read at the next block, like Data::type (Audio.cu:104)
    return JF_OK;
    });
}

int jf_set_pause(jf_engine *e, int paused) {
    return jf_guard([&]() -> int {
    if (!e) return JF_ERR_ARG;
    e->paused.store(paused != 0, std::memory_order_relaxed);
    return JF_OK;
    });
}

// ---- convolution reverb ----------------------------------------------------
int jf_reverb_set_ir(jf_engine *e, const float *ir, size_t n_ir, float gain) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!e || (n_ir && !ir) || n_ir > (size_t)1 << 26) return fail(e, JF_ERR_ARG, "bad impulse response");
    if (e->in_flight) return fail(e, JF_ERR_STATE, "a per-block call is in flight");
    {
        const int rc = rv_ahead_discard(e);
        if (rc) return rc;
    }
    JF_HIP(e, hipStreamSynchronize(e->stream));
    const bool was_on = e->rv_P > 0;
    free_reverb(e);
    if (n_ir == 0) {
        if (was_on) return reset_sources(e, -1);
        return JF_OK;
    }
    const int B = e->B;
    if (B != 64 && B != 128 && B != 256)
        return fail(e, JF_ERR_ARG, "reverb needs frames_per_buffer of 64, 128 or 256 (FFT of 2 blocks)");
    if ((long long)e->maxK * B >= (1LL << 30))  // the stage's play positions are 32-bit sums of a position and K B samples
        return fail(e, JF_ERR_ARG, "max_batch_blocks too large for the reverb stage");
    const size_t S = (size_t)e->S;
    const int P_total = (int)((n_ir + B - 1) / B);
    // Non-uniform partitioning for a response of at least three big partitions (unless a uniform form is pinned, or
    // jf_debug_set_reverb_partitioning says otherwise): the stage below is then the head of rv_big_blocks(B) partitions of B
    const bool nonuniform = e->rv_partitioning == 2 ||
                            (e->rv_partitioning == 0 && e->rv_form == 0 && P_total >= 3 * rv_big_blocks(B));
    const int M = rv_big_blocks(B);
    const int P = nonuniform ? 2 * M : P_total;  // the head: two big partitions' worth of taps (run_reverb_stage says why)
    const int B1 = M * B;
    const int P1 = nonuniform ? (int)((n_ir > (size_t)B1 ? n_ir - B1 : 0) + B1 - 1) / B1 : 0;
    const int steps_max = e->maxK / M + 1;               // big blocks one call can complete
    const int R1 = P1 + 16 + steps_max + 4, Rn = steps_max + 3, Fn = 4;  // (+ 16: the product kernel reads whole groups of 16 slots)
    const int Rg = P + e->maxK;                          // slots a call may still read + the ones it writes
    const int Wr = (e->maxK + kN / B + 1) * B;           // >= PAD_LEN, multiple of B
    float *d_ir = nullptr;
    auto body = [&]() -> int {
        // each followed by the compact copies of its packed bin-0 pairs: h0[P], fdl0[S][Rg]
        JF_HIP(e, hipMalloc(&e->d_rv_hspec, sizeof(float2) * ((size_t)P * B + P)));
        JF_HIP(e, hipMalloc(&e->d_rv_fdl, sizeof(float2) * (S * Rg * B + S * Rg)));
        JF_HIP(e, hipMalloc(&e->d_rv_wet, sizeof(float) * S * Wr));
        JF_HIP(e, hipMalloc(&e->d_sigs_wet, sizeof(SrcSignal) * S));
        for (int i = 0; i < 2; i++) {
            JF_HIP(e, hipMalloc(&e->d_rv_prev[i], sizeof(float) * S * B));
            JF_HIP(e, hipMalloc(&e->d_rv_count[i], sizeof(int) * S));
            JF_HIP(e, hipMemsetAsync(e->d_rv_prev[i], 0, sizeof(float) * S * B, e->stream));
            JF_HIP(e, hipMemsetAsync(e->d_rv_count[i], 0, sizeof(int) * S, e->stream));
        }
        std::vector<SrcSignal> wet(S);
        for (size_t s = 0; s < S; s++) wet[s] = SrcSignal{e->d_rv_wet + s * Wr, Wr, 0};
        JF_HIP(e, h2d(e, e->d_sigs_wet, wet.data(), sizeof(SrcSignal) * S));
        JF_HIP(e, hipMalloc(&d_ir, sizeof(float) * n_ir));
        JF_HIP(e, h2d(e, d_ir, ir, sizeof(float) * n_ir));
        // 1/B: normalisation of the B-point inverse used for the 2B-point real transform
        JF_HIP(e, launch_reverb_ir(d_ir, (int)n_ir, P, B, gain / (float)B, e->d_tw, e->d_rv_hspec, e->stream));
        if (P1 > 0) {
            // twiddles exp(+2 pi i j / (2 B1)), j < 2 B1 (a full circle), from double
            std::vector<float2> tw1((size_t)2 * B1);
            for (int j = 0; j < 2 * B1; j++) {
                const double a = 3.14159265358979323846264338327950288 * j / (double)B1;
                tw1[j] = make_float2((float)cos(a), (float)sin(a));
            }
            // ... followed by the transforms' own selection of them, laid out the way their lanes read them (jf_reverb.hip:
            // BigTwiddles::load)
            const int n_pack = big_twiddle_pack_len(B1);
            for (int k = 0; k < n_pack; k++) tw1.push_back(tw1[(size_t)big_twiddle_pack_index(B1, k)]);
            JF_HIP(e, hipMalloc(&e->d_rv_tw1, sizeof(float2) * tw1.size()));
            JF_HIP(e, h2d(e, e->d_rv_tw1, tw1.data(), sizeof(float2) * tw1.size()));
            const size_t NP = (size_t)P1 + 17;  // H'_0 .. H'_P1 and 16 partitions of zeros
            JF_HIP(e, hipMalloc(&e->d_rv_hspec1, sizeof(float2) * (NP * B1 + NP)));
            JF_HIP(e, hipMemsetAsync(e->d_rv_hspec1, 0, sizeof(float2) * (NP * B1 + NP), e->stream));
            JF_HIP(e, hipMalloc(&e->d_rv_fdl1, sizeof(float2) * (S * R1 * B1 + S * R1)));
            JF_HIP(e, hipMalloc(&e->d_rv_ybig, sizeof(float2) * S * steps_max * B1));
            JF_HIP(e, hipMalloc(&e->d_rv_dryring, sizeof(float) * S * Rn * B1));
            JF_HIP(e, hipMalloc(&e->d_rv_fut, sizeof(float) * S * Fn * B1));
            JF_HIP(e, hipMalloc(&e->d_rv_yacc, sizeof(float2) * S * 2 * B1));
            // 1/B1: normalisation of the B1-point inverse used for the 2 B1-point real transform
            // H'_0 .. H'_P1: the response from its first tap on in partitions of B1 (ReverbBigParams)
            JF_HIP(e, launch_reverb_big_ir(d_ir, (int)n_ir, 0, P1 + 1, B1, gain / (float)B1, e->d_rv_tw1, e->d_rv_hspec1, e->stream));
        }
        JF_HIP(e, hipStreamSynchronize(e->stream));
        return JF_OK;
    };
    int rc = body();
    (void)hipFree(d_ir);
    if (rc != JF_OK) {
        const std::string msg = e->err;
        free_reverb(e);
        return fail(e, rc, msg);
    }
    e->rv_P = P;
    e->rv_Rg = Rg;
    e->rv_Wr = Wr;
    e->rv_head = 0;
    e->rv_P_total = P_total;
    e->rv_P1 = P1;
    e->rv_B1 = P1 > 0 ? B1 : 0;
    e->rv_M = P1 > 0 ? M : 0;
    e->rv_R1 = R1;
    e->rv_Rn = Rn;
    e->rv_Fn = Fn;
    e->rv_steps_max = steps_max;
    e->rv_blocks = 0;
    e->rv_fut_m = 1;  // TAIL(0) and TAIL(1) are sums over spectra of the time before the start: the zeros of the reset
    return reset_sources(e, -1);
    });
}

float jf_reverb_rms_gain(const float *signal, size_t n, const float *ir, size_t n_ir) {
    if (!signal || !ir || n == 0 || n_ir == 0) return 1.0f;
    try {
        return host_reverb_rms_gain(signal, n, ir, n_ir);
    } catch (...) {
        return 1.0f;
    }
}

int jf_profile_read_reverb(jf_engine *e, double *reverb_ms) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!e || !reverb_ms) return JF_ERR_ARG;
    JF_HIP(e, hipStreamSynchronize(e->stream));
    double r = 0;
    for (size_t i = 0; e->profiling >= 2 && e->rv_P > 0 && i < e->ev_used && i < e->ev_reverb.size(); i++) {
        float ms = 0;
        JF_HIP(e, hipEventElapsedTime(&ms, e->ev_reverb[i].a, e->ev_reverb[i].b));
        r += ms;
    }
    *reverb_ms = r;
    return JF_OK;
    });
}

// ---- batch -----------------------------------------------------------------
int jf_batch_upload_positions(jf_engine *e, int total_blocks, const float *positions) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!e || total_blocks <= 0 || !positions) return fail(e, JF_ERR_ARG, "bad trajectory");
    JF_HIP(e, hipStreamSynchronize(e->stream));
    e->traj_gen++;
    e->ahead.valid = false;
    const size_t bytes = sizeof(float) * 5 * (size_t)e->S * (size_t)total_blocks;
    if (total_blocks > e->traj_blocks) {
        (void)hipFree(e->d_traj);
        e->d_traj = nullptr;
        e->traj_blocks = 0;
        JF_HIP(e, hipMalloc(&e->d_traj, bytes));
    }
    e->traj_blocks = total_blocks;
    JF_HIP(e, h2d(e, e->d_traj, positions, bytes));
    // how many items of every block move (their (ele, azi) differ from the block before; block 0 counts as staying):
    // what decides whether a run reads pre-interpolated rows (jf_engine::interp_use)
    e->traj_moved.assign((size_t)total_blocks + 1, 0u);
    if (e->interp_avail)
        for (int b = 1; b < total_blocks; b++) {
            const float *p1 = positions + (size_t)b * e->S * 5, *p0 = p1 - (size_t)e->S * 5;
            unsigned n = 0;
            for (int s = 0; s < e->S; s++) n += p1[5 * s] != p0[5 * s] || p1[5 * s + 1] != p0[5 * s + 1];
            e->traj_moved[(size_t)b + 1] = e->traj_moved[b] + n;
        }
    // Processing order of the pair kernel: a unit sums G sources that are next to each other in this order.  With
    // automatic grouping the sources are ordered by the table row nearest to their first position, so that the units a
    // compute unit works on at a time read neighbouring rows of the 5.8 MB table (the L2 of an XCD holds 4 MB); the mix is
    // the same sum in another association.  jf_debug_set_source_group pins consecutive sources (identity order).
    const bool want_sorted = e->src_group == 0 && e->S > 1;
    if (want_sorted || e->sorted_order) {
        std::vector<std::pair<int, int>> key((size_t)e->S);
        for (int s = 0; s < e->S; s++) {
            const float *p = positions + 5 * (size_t)s;
            const bool ok = p[0] > -1.0e6f && p[0] < 1.0e6f && p[1] > -1.0e6f && p[1] < 1.0e6f;
            key[s] = {want_sorted && ok ? host_grid_pick(e->rt, p[0], p[1]) : 0, s};
        }
        std::stable_sort(key.begin(), key.end());
        for (int s = 0; s < e->S; s++) e->order[s] = key[s].second;
        JF_HIP(e, h2d(e, e->d_order, e->order.data(), sizeof(int) * e->S));
        e->sorted_order = want_sorted;
    }
    return JF_OK;
    });
}

int jf_batch_run(jf_engine *e, int first_block, int n_blocks, float *d_out_mix) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!e) return JF_ERR_ARG;
    if (n_blocks <= 0 || n_blocks > e->maxK) return fail(e, JF_ERR_ARG, "n_blocks exceeds max_batch_blocks");
    if (first_block < 0 || first_block + n_blocks > e->traj_blocks)
        return fail(e, JF_ERR_ARG, "window outside the uploaded trajectory");
    if (e->in_flight) return fail(e, JF_ERR_STATE, "a per-block call is in flight");
    return run_blocks(e, e->d_traj + (size_t)first_block * e->S * 5, n_blocks, d_out_mix ? d_out_mix : e->d_mix,
                      first_block);
    });
}

int jf_device_numa_node(int device, int *node) {
    return jf_guard([&]() -> int {
    if (!node) return JF_ERR_ARG;
    *node = -1;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) {
        (void)hipGetLastError();
        return fail(nullptr, JF_ERR_DEVICE, "no such HIP device");
    }
    for (char *c = bus; *c; c++) *c = (char)tolower((unsigned char)*c);  // sysfs spells the address in lower case
    const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/numa_node";
    if (FILE *f = fopen(path.c_str(), "r")) {
        int n = -1;
        if (fscanf(f, "%d", &n) == 1) *node = n;
        fclose(f);
    }
    return JF_OK;
    });
}

int jf_pin_thread_to_device(int device) {
    return jf_guard([&]() -> int {
    int node = -1;
    const int rc = jf_device_numa_node(device, &node);
    if (rc != JF_OK) return rc;
    if (node < 0) return fail(nullptr, JF_ERR_STATE, "the system does not say which NUMA node the device is on");
    // /sys/devices/system/node/node<N>/cpulist: "0-63,128-191"
    const std::string path = "/sys/devices/system/node/node" + std::to_string(node) + "/cpulist";
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return fail(nullptr, JF_ERR_STATE, "no CPU list for the device's NUMA node");
    char line[4096] = {0};
    const bool got = fgets(line, sizeof(line), f) != nullptr;
    fclose(f);
    if (!got) return fail(nullptr, JF_ERR_STATE, "no CPU list for the device's NUMA node");
    cpu_set_t allowed, want;
    CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return fail(nullptr, JF_ERR_STATE, "sched_getaffinity failed");
    int n_set = 0;
    for (const char *p = line; *p;) {
        char *end = nullptr;
        const long a = strtol(p, &end, 10);
        if (end == p) break;
        long b = a;
        p = end;
        if (*p == '-') {
            b = strtol(p + 1, &end, 10);
            p = end;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++)
            if (c >= 0 && CPU_ISSET((int)c, &allowed)) {
                CPU_SET((int)c, &want);
                n_set++;
            }
        while (*p == ',' || *p == ' ' || *p == '\n') p++;
    }
    if (n_set == 0) return fail(nullptr, JF_ERR_STATE, "none of the CPUs of the device's NUMA node is allowed to this process");
    if (sched_setaffinity(0, sizeof(want), &want) != 0) return fail(nullptr, JF_ERR_STATE, "sched_setaffinity failed");
    return JF_OK;
    });
}

int jf_synchronize(jf_engine *e) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!e) return JF_ERR_ARG;
    if (e->rv_side && e->rv_side_busy) JF_HIP(e, hipStreamSynchronize(e->rv_side));
    JF_HIP(e, hipStreamSynchronize(e->stream));
    if (device_fault(e)) return fail(e, JF_ERR_DEVICE, kHandOffMsg);
    return JF_OK;
    });
}

int jf_process_batch(jf_engine *e, int n_blocks, const float *positions, float *out_mix) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!e || !positions || !out_mix || n_blocks <= 0) return fail(e, JF_ERR_ARG, "bad batch arguments");
    int rc = jf_batch_upload_positions(e, n_blocks, positions);
    if (rc) return rc;
    const size_t blk = (size_t)2 * e->B;
    for (int b0 = 0; b0 < n_blocks; b0 += e->maxK) {
        const int k = n_blocks - b0 < e->maxK ? n_blocks - b0 : e->maxK;
        rc = jf_batch_run(e, b0, k, nullptr);
        if (rc) return rc;
        JF_HIP(e, hipMemcpyAsync(out_mix + (size_t)b0 * blk, e->d_mix, sizeof(float) * blk * k, hipMemcpyDeviceToHost,
                                 e->stream));
        JF_HIP(e, hipStreamSynchronize(e->stream));
        if (device_fault(e)) return fail(e, JF_ERR_DEVICE, kHandOffMsg);
    }
    // n_blocks callbacks have run: the sources stand where the last of them read them
    return jf_sources_set_latched(e, positions + (size_t)(n_blocks - 1) * e->S * JF_POS_FLOATS);
    });
}

int jf_sources_set_latched(jf_engine *e, const float *records) {
    return jf_guard([&]() -> int {
    if (!e || !records) return JF_ERR_ARG;
    std::lock_guard<std::mutex> lk(e->pos_mu);
    for (int s = 0; s < e->S; s++) {
        const float *r = records + (size_t)s * JF_POS_FLOATS;
        e->pos[s] = HostPos{r[0], r[1], sqrtf(r[2] * r[2] + r[3] * r[3] + r[4] * r[4]), r[2], r[3], r[4]};
    }
    return JF_OK;
    });
}

int jf_batch_fetch(jf_engine *e, int n_blocks, float *out_mix) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!e || !out_mix || n_blocks <= 0 || n_blocks > e->maxK) return fail(e, JF_ERR_ARG, "bad fetch arguments");
    JF_HIP(e, hipMemcpyAsync(out_mix, e->d_mix, sizeof(float) * 2 * e->B * (size_t)n_blocks, hipMemcpyDeviceToHost, e->stream));
    JF_HIP(e, hipStreamSynchronize(e->stream));
    if (device_fault(e)) return fail(e, JF_ERR_DEVICE, kHandOffMsg);
    return JF_OK;
    });
}

int jf_debug_set_reverb_side_workgroups(jf_engine *e, int workgroups) {
    if (!e || workgroups < 8 || workgroups > 65536) return JF_ERR_ARG;
    e->rv_side_wgs = workgroups;
    return JF_OK;
}

float *jf_batch_mix_device(jf_engine *e) { return e ? e->d_mix : nullptr; }
float *jf_batch_partial_device(jf_engine *e) { return e ? e->d_partial : nullptr; }
void *jf_engine_stream(jf_engine *e) { return e ? (void *)e->stream : nullptr; }

int jf_profile_enable(jf_engine *e, int enable) {
    return jf_guard([&]() -> int {
    if (e) {
        DeviceGuard bind_(e);
        const int rc_ = rv_ahead_discard(e);  // (the next block's stage may have gone ahead in the old form)
        if (rc_) return rc_;
    }
    DeviceGuard bind(e);
    if (!e) return JF_ERR_ARG;
    JF_HIP(e, hipStreamSynchronize(e->stream));
    e->profiling = enable < 0 ? 0 : (enable > 2 ? 2 : enable);
    e->ev_used = 0;
    e->profile_calls = 0;
    return JF_OK;
    });
}

int jf_profile_set_stride(jf_engine *e, int every) {
    return jf_guard([&]() -> int {
    if (!e || every < 1) return JF_ERR_ARG;
    e->profile_stride = every;
    e->profile_calls = 0;
    return JF_OK;
    });
}

int jf_profile_read(jf_engine *e, double *fused_ms, double *prep_ms, double *mix_ms, long *launches) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!e) return JF_ERR_ARG;
    JF_HIP(e, hipStreamSynchronize(e->stream));
    double f = 0, p = 0, m = 0;
    for (size_t i = 0; i < e->ev_used; i++) {
        float ms = 0;
        JF_HIP(e, hipEventElapsedTime(&ms, e->ev_fused[i].a, e->ev_fused[i].b));
        f += ms;
        if (e->profiling >= 2) {
            JF_HIP(e, hipEventElapsedTime(&ms, e->ev_prep[i].a, e->ev_prep[i].b));
            p += ms;
            JF_HIP(e, hipEventElapsedTime(&ms, e->ev_mix[i].a, e->ev_mix[i].b));
            m += ms;
        }
    }
    if (fused_ms) *fused_ms = f;
    if (prep_ms) *prep_ms = p;
    if (mix_ms) *mix_ms = m;
    if (launches) *launches = (long)e->ev_used;
    return JF_OK;
    });
}

// ---- debugging taps -----------------------------------------------------------
int jf_debug_copy_from_device(jf_engine *e, const void *device_ptr, void *host, size_t bytes) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!e || !device_ptr || !host) return JF_ERR_ARG;
    JF_HIP(e, hipStreamSynchronize(e->stream));
    JF_HIP(e, hipMemcpy(host, device_ptr, bytes, hipMemcpyDeviceToHost));
    return JF_OK;
    });
}

int jf_debug_set_rt_max_sources(jf_engine *e, int n) {
    return jf_guard([&]() -> int {
    if (e) {
        DeviceGuard bind_(e);
        const int rc_ = rv_ahead_discard(e);  // (the next block's stage may have gone ahead in the old form)
        if (rc_) return rc_;
    }
    if (!e || n < 0) return JF_ERR_ARG;
    e->rt_max_sources = n;
    return JF_OK;
    });
}

int jf_debug_set_source_group(jf_engine *e, int group) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!e || group < 0 || (group > 0 && e->S % group)) return JF_ERR_ARG;
    e->src_group = group;
    if (group > 0 && e->sorted_order) {  // a pinned group size means consecutive sources
        JF_HIP(e, hipStreamSynchronize(e->stream));
        for (int s = 0; s < e->S; s++) e->order[s] = s;
        JF_HIP(e, h2d(e, e->d_order, e->order.data(), sizeof(int) * e->S));
        e->sorted_order = false;
    }
    return JF_OK;
    });
}

int jf_debug_source_order(const jf_engine *e, int *order) {
    if (!e || !order) return JF_ERR_ARG;
    // the per-source kernel (a run that resolved to G = 1) does not go through the order: its block u is source u
    for (int s = 0; s < e->S; s++) order[s] = e->last_group == 1 ? s : e->order[s];  // (no run yet: what a grouped run takes)
    return JF_OK;
}

int jf_debug_set_reverb_form(jf_engine *e, int form) {
    return jf_guard([&]() -> int {
    if (e) {
        DeviceGuard bind_(e);
        const int rc_ = rv_ahead_discard(e);  // (the next block's stage may have gone ahead in the old form)
        if (rc_) return rc_;
    }
    if (!e || form < 0 || form > 3) return JF_ERR_ARG;
    e->rv_form = form;
    return JF_OK;
    });
}

int jf_debug_set_interp_table(jf_engine *e, int on) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!e) return JF_ERR_ARG;
    if (on < 0 || on > 2) return fail(e, JF_ERR_ARG, "0 = never, 1 = always, 2 = decided per run");
    if (on && !e->interp_avail) return fail(e, JF_ERR_STATE, "this engine was created without the pre-interpolated rows");
    e->interp_use = on;  // the mode word of the next run changes with it: descriptors prepared ahead no longer match
    if (on == 1) {       // "always" builds them now (a run under "per run" builds them when it first takes them)
        const int rc = ensure_interp_rows(e);
        if (rc) return rc;
        if (!e->interp_built) return fail(e, JF_ERR_NOMEM, "no device memory for the pre-interpolated rows");
    }
    return JF_OK;
    });
}

int jf_debug_interp_table(const jf_engine *e) { return e && e->interp_avail ? e->interp_use : 0; }
int jf_debug_interp_table_built(const jf_engine *e) { return e && e->interp_built ? 1 : 0; }
int jf_debug_last_run_used_rows(const jf_engine *e) { return e && e->last_rows ? 1 : 0; }

int jf_debug_count_desc_flags(jf_engine *e, int n_items, int mask) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!e || n_items <= 0 || (size_t)n_items > (size_t)e->S * e->maxK) return JF_ERR_ARG;
    std::vector<ItemDesc> d((size_t)n_items);
    JF_HIP(e, hipStreamSynchronize(e->stream));
    JF_HIP(e, hipMemcpy(d.data(), e->d_desc, sizeof(ItemDesc) * d.size(), hipMemcpyDeviceToHost));
    int n = 0;
    for (const ItemDesc &x : d) n += (x.flags & mask) != 0;
    return n;
    });
}

int jf_debug_read_table_rows(jf_engine *e, int first_row, int n, float *out) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (e && out && n > 0 && first_row >= 0 && first_row + (long long)n > e->rt.n_rows) {  // pre-interpolated rows: built on demand
        const int rc = ensure_interp_rows(e);
        if (rc) return rc;
    }
    const int total = e ? e->rt.n_rows + (e->interp_built ? kInterpRows : 0) : 0;
    if (!e || !out || n <= 0 || first_row < 0 || first_row > total - n) return fail(e, JF_ERR_ARG, "rows outside the table");
    JF_HIP(e, hipStreamSynchronize(e->stream));
    JF_HIP(e, hipMemcpy(out, e->d_htab + (size_t)first_row * 512, sizeof(float4) * 512 * (size_t)n, hipMemcpyDeviceToHost));
    return JF_OK;
    });
}

int jf_debug_set_reverb_partitioning(jf_engine *e, int how) {
    return jf_guard([&]() -> int {
    if (!e || how < 0 || how > 2) return JF_ERR_ARG;
    e->rv_partitioning = how;  // in effect from the next jf_reverb_set_ir
    return JF_OK;
    });
}

int jf_debug_set_reverb_ahead(jf_engine *e, int on) {
    return jf_guard([&]() -> int {
    if (!e) return JF_ERR_ARG;
    DeviceGuard bind(e);
    const int rc = rv_ahead_discard(e);
    if (rc) return rc;
    e->rv_ahead_on = on != 0;
    return JF_OK;
    });
}

int jf_debug_set_reverb_lazy_state(jf_engine *e, int on) {
    return jf_guard([&]() -> int {
    if (e) {
        DeviceGuard bind_(e);
        const int rc_ = rv_ahead_discard(e);  // (the next block's stage may have gone ahead in the old form)
        if (rc_) return rc_;
    }
    if (!e) return JF_ERR_ARG;
    e->rv_lazy_small = on != 0;  // (transforms already put off are still formed by the call that needs them)
    return JF_OK;
    });
}

int jf_debug_set_reverb_head_fused(jf_engine *e, int on) {
    return jf_guard([&]() -> int {
    if (e) {
        DeviceGuard bind_(e);
        const int rc_ = rv_ahead_discard(e);  // (the next block's stage may have gone ahead in the old form)
        if (rc_) return rc_;
    }
    if (!e) return JF_ERR_ARG;
    e->rv_head_fused = on != 0;
    return JF_OK;
    });
}

int jf_debug_set_reverb_async(jf_engine *e, int on) {
    return jf_guard([&]() -> int {
    if (e) {
        DeviceGuard bind_(e);
        const int rc_ = rv_ahead_discard(e);  // (the next block's stage may have gone ahead in the old form)
        if (rc_) return rc_;
    }
    if (!e) return JF_ERR_ARG;
    e->rv_async = on != 0;  // what the side stream has in flight is waited for by the next call's stage (run_reverb_stage)
    return JF_OK;
    });
}

int jf_debug_reverb_schedule(long long j0, int K, int M, long long fut_m, long long out[16]) {
    if (!out || K <= 0 || M <= 0 || j0 < 0) return JF_ERR_ARG;
    const ReverbSchedule s = host_reverb_schedule(j0, K, M, fut_m);
    const long long v[16] = {s.m_lo, s.n_tr, s.ma, s.n_mid, s.n_ranges, s.kb[0], s.kn[0], s.kb[1], s.kn[1], s.copy_lo, s.copy_hi,
                             s.skip_lo, s.skip_hi, s.tail_early, s.tail_late, s.fut_m};
    for (int i = 0; i < 16; i++) out[i] = v[i];
    return JF_OK;
}

int jf_debug_reverb_partitions(const jf_engine *e, int *head, int *big, int *big_taps) {
    if (!e) return JF_ERR_ARG;
    if (head) *head = e->rv_P;
    if (big) *big = e->rv_P1 > 0 ? e->rv_P1 - 1 : 0;  // H'_2 .. H'_P1 (H'_0 and H'_1 are the head's taps; FULL uses all)
    if (big_taps) *big_taps = e->rv_B1;
    return e->rv_P_total;
}

int jf_debug_read_table(jf_engine *e, float *out) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!e || !out) return JF_ERR_ARG;
    std::vector<float4> h((size_t)e->rt.n_rows * 512);
    JF_HIP(e, hipStreamSynchronize(e->stream));
    JF_HIP(e, hipMemcpy(h.data(), e->d_htab, sizeof(float4) * h.size(), hipMemcpyDeviceToHost));
    for (int j = 0; j < e->rt.n_rows; j++) {
        float *L = out + ((size_t)j * 2 + 0) * kNc * 2;
        float *R = out + ((size_t)j * 2 + 1) * kNc * 2;
        const float4 *row = h.data() + (size_t)j * 512;
        L[0] = row[0].x;
        L[1] = 0.0f;
        L[1024] = row[0].y;
        L[1025] = 0.0f;
        R[0] = row[0].z;
        R[1] = 0.0f;
        R[1024] = row[0].w;
        R[1025] = 0.0f;
        for (int k = 1; k < 512; k++) {
            L[2 * k] = row[k].x;
            L[2 * k + 1] = row[k].y;
            R[2 * k] = row[k].z;
            R[2 * k + 1] = row[k].w;
        }
    }
    return JF_OK;
    });
}

int jf_debug_interp_device(jf_engine *e, int n, const float *ele, const float *azi, int *rows, float *weights,
                           int *nterms) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!e || n <= 0 || !ele || !azi || !rows || !weights || !nterms) return JF_ERR_ARG;
    float *d_e = nullptr, *d_a = nullptr, *d_w = nullptr;
    int *d_r = nullptr, *d_n = nullptr;
    auto body = [&]() -> int {
        JF_HIP(e, hipMalloc(&d_e, sizeof(float) * n));
        JF_HIP(e, hipMalloc(&d_a, sizeof(float) * n));
        JF_HIP(e, hipMalloc(&d_w, sizeof(float) * 4 * n));
        JF_HIP(e, hipMalloc(&d_r, sizeof(int) * 4 * n));
        JF_HIP(e, hipMalloc(&d_n, sizeof(int) * n));
        JF_HIP(e, h2d(e, d_e, ele, sizeof(float) * n));
        JF_HIP(e, h2d(e, d_a, azi, sizeof(float) * n));
        JF_HIP(e, launch_interp_debug(e->rt, d_e, d_a, d_r, d_w, d_n, n, corrected_rule(e) ? 1 : 0, e->stream));
        JF_HIP(e, hipStreamSynchronize(e->stream));
        JF_HIP(e, hipMemcpy(rows, d_r, sizeof(int) * 4 * n, hipMemcpyDeviceToHost));
        JF_HIP(e, hipMemcpy(weights, d_w, sizeof(float) * 4 * n, hipMemcpyDeviceToHost));
        JF_HIP(e, hipMemcpy(nterms, d_n, sizeof(int) * n, hipMemcpyDeviceToHost));
        return JF_OK;
    };
    int rc = body();
    (void)hipFree(d_e);
    (void)hipFree(d_a);
    (void)hipFree(d_w);
    (void)hipFree(d_r);
    (void)hipFree(d_n);
    return rc;
    });
}

int jf_debug_rfft_device(jf_engine *e, int n, const float *windows, float *spectra) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!e || n <= 0 || !windows || !spectra) return JF_ERR_ARG;
    float *d_w = nullptr;
    float2 *d_s = nullptr;
    auto body = [&]() -> int {
        JF_HIP(e, hipMalloc(&d_w, sizeof(float) * (size_t)n * kN));
        JF_HIP(e, hipMalloc(&d_s, sizeof(float2) * (size_t)n * kNc));
        JF_HIP(e, h2d(e, d_w, windows, sizeof(float) * (size_t)n * kN));
        JF_HIP(e, launch_rfft_debug(d_w, n, e->d_twpack, d_s, e->stream));
        JF_HIP(e, hipStreamSynchronize(e->stream));
        JF_HIP(e, hipMemcpy(spectra, d_s, sizeof(float2) * (size_t)n * kNc, hipMemcpyDeviceToHost));
        return JF_OK;
    };
    int rc = body();
    (void)hipFree(d_w);
    (void)hipFree(d_s);
    return rc;
    });
}

int jf_debug_last_source_group(const jf_engine *e) { return e ? e->last_group : JF_ERR_ARG; }

const char *jf_debug_last_kernels(jf_engine *e) {
    if (!e) return "";
    if (e->kernels_use_frozen) return e->kernels_frozen.c_str();  // (the stage's fields describe the block launched ahead)
    try {
        const std::string nb = std::to_string(e->B / 64), bs = std::to_string(e->B);
        std::string k;
        if (!e->last_rt && !e->last_prep_skipped) k = "prep_kernel;";
        if (e->rv_P > 0) {
            if (e->last_catchup) k += "reverb_fft_kernel<" + bs + ">@ring;";
            const ReverbPlan &pl = e->last_plan;
            const std::string b1 = std::to_string(e->rv_B1);
            auto per_wg = [&](int) { return std::string(",1>;"); };  // transforms per workgroup and turn (persistent since round 5)
            auto products = [&](const ReverbBigParams &g) {
                if (g.n_prod <= 0) return std::string();
                const std::string mac = g.n_prod >= 4 ? "reverb_big_mac_kernel<" + b1 + ",16>"
                                        : JF_RV_BIG_MAC1_SHARED && g.mac_wgs == 0 ? "reverb_big_mac1_kernel<" + b1 + ">"
                                                                                  : "reverb_big_mac_kernel<" + b1 + ",1>";
                return mac + ";reverb_big_ifft_kernel<" + b1 + per_wg(g.n_prod);
            };
            auto transforms = [&](const ReverbBigParams &g) {
                return g.n_tr > 0 ? "reverb_big_fft_kernel<" + b1 + per_wg(g.n_tr) : std::string();
            };
            const int tile = e->B == 256 ? 8 : 16, grp = e->B == 256 ? 2 : 4;
            auto stage_b = [&](int form) {
                if (form == 3) return "reverb_mac_tiled_kernel<" + bs + "," + std::to_string(tile) + ">;";
                if (form == 4) return "reverb_mac_kernel<" + bs + ",1,true>;";
                if (form == 0) return std::string();
                return "reverb_mac_kernel<" + bs + "," + std::to_string(form == 2 ? grp : 1) + ">;";
            };
            if (pl.big) k += products(pl.tail_early);
            if (e->last_rv_form == 5) {
                // (the head ran inside the real-time kernel, named below; transforms left in line follow it)
            } else if (e->last_rv_form == 4) {
                k += stage_b(4);
                if (pl.big) k += transforms(pl.transforms);
            } else {
                if (e->last_small_fft) k += "reverb_fft_kernel<" + bs + ">;";
                if (pl.big) {
                    if (pl.n_ranges > 1) k += stage_b(pl.forms[0]);
                    k += transforms(pl.transforms);
                    k += products(pl.middle) + products(pl.tail_late);
                    k += stage_b(pl.forms[pl.n_ranges > 1 ? 1 : 0]);
                } else {
                    k += stage_b(e->last_rv_form);
                }
            }
        }
        if (e->rv_P > 0) k += e->last_side;
        // launch_mix: few partial blocks per audio block (16, 32 or 64 groups) take the one-thread-per-float form
        const int n_part = e->last_group > 0 ? e->S / e->last_group : e->S;
        const std::string mix_name = (n_part == 16 || n_part == 32 || n_part == 64)
                                         ? ";mix_few_kernel<" + std::to_string(n_part / 16) + ">" : std::string(";mix_kernel");
        if (e->last_rt) {
            const bool fused = e->rv_P > 0 && e->last_rv_form == 5;
            k += "rt_block_kernel<" + nb + "," + std::to_string(rt_waves_per_wg(e->S)) + (fused ? ",reverb>" : ">");
            if (fused && e->last_plan.big && e->last_plan.transforms.n_tr > 0) {
                const std::string b1 = std::to_string(e->rv_B1);
                k += ";reverb_big_fft_kernel<" + b1 + ",1>";
            }
        }
        else k += std::string(e->last_group > 1 ? "fused_pair_kernel<" : "fused_block_kernel<") + nb +
                  (e->last_fused_prep ? ">+prep" : ">") + (e->last_mix_prep ? ";mix_prep_kernel" : mix_name);
        e->kernels = k;
        return e->kernels.c_str();
    } catch (...) {
        return "";
    }
}

int jf_debug_set_prep_ahead(jf_engine *e, int on) {
    return jf_guard([&]() -> int {
    if (!e) return JF_ERR_ARG;
    e->prep_ahead = on != 0;
    e->ahead.valid = false;
    return JF_OK;
    });
}

int jf_debug_set_grid_limit(jf_engine *e, int workgroups) {
    return jf_guard([&]() -> int {
    if (!e || workgroups < 0) return JF_ERR_ARG;
    e->grid_limit = workgroups;
    return JF_OK;
    });
}

int jf_debug_stage_taps(jf_engine *e, int n, const float *positions, const float *windows, float *dist,
                        float *spectra) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!e || n <= 0 || !positions || !dist || (spectra && !windows)) return JF_ERR_ARG;
    float *d_p = nullptr, *d_w = nullptr;
    float2 *d_d = nullptr, *d_s = nullptr;
    auto body = [&]() -> int {
        JF_HIP(e, hipMalloc(&d_p, sizeof(float) * 5 * (size_t)n));
        JF_HIP(e, hipMalloc(&d_d, sizeof(float2) * (size_t)n * kNc));
        JF_HIP(e, h2d(e, d_p, positions, sizeof(float) * 5 * (size_t)n));
        if (spectra) {
            JF_HIP(e, hipMalloc(&d_w, sizeof(float) * (size_t)n * kN));
            JF_HIP(e, hipMalloc(&d_s, sizeof(float2) * (size_t)n * 2 * kNc));
            JF_HIP(e, h2d(e, d_w, windows, sizeof(float) * (size_t)n * kN));
        }
        JF_HIP(e, launch_stage_debug(e->rt, kernel_mode(e), d_p, d_w, n, e->d_htab, e->d_twpack, d_d, d_s,
                                     e->stream));
        JF_HIP(e, hipStreamSynchronize(e->stream));
        JF_HIP(e, hipMemcpy(dist, d_d, sizeof(float2) * (size_t)n * kNc, hipMemcpyDeviceToHost));
        if (spectra) JF_HIP(e, hipMemcpy(spectra, d_s, sizeof(float2) * (size_t)n * 2 * kNc, hipMemcpyDeviceToHost));
        return JF_OK;
    };
    int rc = body();
    (void)hipFree(d_p);
    (void)hipFree(d_w);
    (void)hipFree(d_d);
    (void)hipFree(d_s);
    return rc;
    });
}

float jf_last_block_peak(const jf_engine *e) { return e ? e->last_peak : 0.0f; }

int jf_debug_read_stamps(jf_engine *e, unsigned long long *out, int n) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!e || !out || n < 0 || n > 8192) return JF_ERR_ARG;
    JF_HIP(e, hipStreamSynchronize(e->stream));
    memcpy(out, (const char *)e->h_err + 16, sizeof(unsigned long long) * (size_t)n);
    return JF_OK;
    });
}

// ---- WAV -----------------------------------------------------------------------
int jf_wav_read_mono(const char *path, float **out, size_t *n_frames, int *sample_rate) {
    return jf_guard([&]() -> int {
    if (!path || !out || !n_frames) return JF_ERR_ARG;
    std::string err;
    int rc = wav_read_mono(path, out, n_frames, sample_rate, &err);
    if (rc) g_create_error = err;
    return rc;
    });
}

int jf_wav_write_stereo24(const char *path, const float *interleaved, size_t n_frames, int sample_rate) {
    return jf_guard([&]() -> int {
    if (!path || (!interleaved && n_frames)) return JF_ERR_ARG;
    std::string err;
    int rc = wav_write_stereo24(path, interleaved, n_frames, sample_rate, &err);
    if (rc) g_create_error = err;
    return rc;
    });
}

void jf_free(void *p) { free(p); }

}  // extern "C"
