// jf_device.h -- shared host/device declarations of the HIP engine (gfx950).
//
// Data layout in HBM (all engine-owned):
//   htab   float4[n_rows][512] HRTF spectra (n_rows = 710 for KEMAR), both ears interleaved per bin:
//                             k >= 1: {L.re, L.im, R.re, R.im};
//                             k == 0: {L[0].re, L[512].re, R[0].re, R[512].re}
//                             (bins 0 and 512 of a real HRIR are real), so one
//                             16-byte load per lane fetches both ears and a row
//                             is exactly 8 KiB.  Same numbers as the reference's
//                             fft_hrtf[(j*2+ear)*513+k] (hrtf_signals.cu:90-98).
//                             Rows n_rows .. n_rows + 47159 (built on first use, 386 MB): the PRE-INTERPOLATED filters of every whole-degree
//                             position the setters can latch (SoundSource.cu:33-34,42-43 round to whole degrees):
//                             row n_rows + (ele + 40) * 360 + azi = sum_t w_t H[row_t] for (ele, azi), ele -40..90,
//                             azi 0..359, by the same operations in the same order as the filters' own weighting
//                             (table_interp_build_kernel).  A filter set is then ONE row with weight 1.
//   tw     float2[1024]       exp(+2*pi*i*j/1024), from double (reverb kernels).
//   twpack float2[2368]       the same values re-laid per FFT pass (see kTw* below), staged in LDS.
//   sig    float[len_s]       one device buffer per source (looped playback).
//   hist   float[2][S][1024]  each source's last window (ping-pong per call).
//   state  SrcState[2][S]     count / old_ele / old_azi (ping-pong per call).
//   pos    float[K][S][5]     latched positions {ele, azi, x, y, z} per block.
//   desc   ItemDesc[K][S]     per (block, source) rows/weights/distance terms.
//   partial float[K][S/G][2B] stereo blocks of groups of G consecutive sources (G = 1: the
//                             reference's per-source `intermediate`).
//   mix    float[K][2B]       sum over sources in source order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace jf {

constexpr int kN = 1024;       // PAD_LEN (Universal.cuh:12)
constexpr int kNc = 513;       // PAD_LEN / 2 + 1
constexpr int kNumHrtf = 710;  // NUM_HRTF (Universal.cuh:4): rows of the reference's KEMAR table
constexpr int kNumElev = 14;   // NUM_ELEV (hrtf_signals.cuh:25): its elevation rings
constexpr int kMaxRings = 40;  // JF_MAX_RINGS (include/jefferson.h): elevation rings of any HRTF grid (5-degree rings pole to pole: 37)
// pre-interpolated rows behind the 710 measured ones (see htab above)
constexpr int kInterpEleMin = -40, kInterpEleMax = 90, kInterpAzi = 360;
constexpr int kInterpRows = (kInterpEleMax - kInterpEleMin + 1) * kInterpAzi;  // 47 160
constexpr int kModeBasic = 1, kModeCorrected = 2, kModeInterpRows = 4;  // bits of the kernels' `mode`
// Tuning knobs of the fused kernel (overridable at build time for A/B runs):
// waves (= work items) per workgroup, and the minimum waves per SIMD the register
// allocator must leave room for (__launch_bounds__ second argument; 0 = unconstrained).
#ifndef JF_WAVES_PER_WG
#define JF_WAVES_PER_WG 16
#endif
#ifndef JF_CHUNK_LOADS
#define JF_CHUNK_LOADS 8  // table-row loads (16 B per lane each) a wave keeps in flight per round
#endif
#ifndef JF_TABLE_DISTANCE
#define JF_TABLE_DISTANCE 1  // 1: distance factors from the twiddle table + small-angle correction; 0: minimax sin/cos
#endif
#ifndef JF_STAGE_LOADS
#define JF_STAGE_LOADS 4  // pair kernel: table-row loads per stage of a half-filter; two stages are in flight
#endif
#ifndef JF_SPLIT_EXCHANGE
#define JF_SPLIT_EXCHANGE 0  // 1: halve the per-wave LDS exchange buffer (re and im separately)
#endif
#ifndef JF_MIN_WAVES
#define JF_MIN_WAVES 0
#endif
#ifndef JF_XCD_MAP
#define JF_XCD_MAP 0  // group kernel: 1 = adjacent units on one XCD (stationary -2 %, moving +0.5 %: off)
#endif
#ifndef JF_UNIT_ORDER
#define JF_UNIT_ORDER 1  // group kernel: 1 = consecutive waves take consecutive blocks of the same sources (rows and windows overlap in cache: 2.8 %)
#endif
#ifndef JF_PAIR_ROTATE_PRIO
#define JF_PAIR_ROTATE_PRIO 1  // persistent kernels: 1 = progress-ordered wave priorities (0: the hardware's oldest-first: -5.5 %)
#endif
#ifndef JF_RV_BIG_MAC1_SHARED
#define JF_RV_BIG_MAC1_SHARED 1  // reverb, single products of the big partitions IN LINE: the shared form (jf_reverb.hip: big_mac_single_shared)
#endif
#ifndef JF_STAGE_DEPTH
#define JF_STAGE_DEPTH 2  // pair kernel: stages of a half-filter's row loads in flight
#endif
#ifndef JF_UNIT_ZIGZAG
#define JF_UNIT_ZIGZAG 2  // order of the units over the rounds of the pair kernel (see there): 2 rotated, 1 zigzag, 0 plain
#endif
constexpr int kWavesPerWg = JF_WAVES_PER_WG;

// Twiddle pack: every table the FFT passes need, laid out so that a wave reads consecutive
// float2 entries at compile-time offsets (no index arithmetic, no LDS bank conflicts).
// w(j) = exp(+2 pi i j / 1024); lane = 4 i + a.
constexpr int kTwW3 = 0;       // [16][64] last inverse stage  w(a (i + 16 t) + 768 a)   at [t][lane]
constexpr int kTwU = 1024;     // [8][64]  real-FFT split      w(lane + 64 q)            at [q][lane]
constexpr int kTwWC = 1536;    // [8][64]  forward pass C      w(2 r lane)               at [r][lane]
constexpr int kTwW2 = 2048;    // [16][16] inverse inter-stage w(4 i m)                  at [m][i]
constexpr int kTwWB = 2304;    // [8][8]   forward pass B      w(16 r k)                 at [r][k]
constexpr int kTwPack = 2368;  // float2 entries (18 944 B)

struct ItemDesc {
    int rows_new[4];
    float w_new[4];
    int rows_old[4];
    float w_old[4];
    unsigned long long c_fix;  // frac(fsvs * r' / 513) * 2^64: distance-delay phase step per bin, in turns
    float inv_frac;  // 1 / (1 + fsvs r'^2)
    int n_new;       // 1, 2 or 4 terms; 0 = position not interpolable -> silence
    int n_old;       // 0 = no crossfade
    int flags;       // pair-kernel layout (prep_kernel with canon = 1): bit 0 = both sets read rows_new[] (w_old[] / w_new[]
                     // are their weights on those rows, 0 where a set does not use a row; n_old == n_new), bit 1 = the
                     // source moved (crossfade); a source that did not move carries its new set as old set too;
                     // bit 2 = rows_new[0] and rows_old[0] are pre-interpolated rows: whole filters, weight 1
};
static_assert(sizeof(ItemDesc) == 88, "ItemDesc layout");

struct SrcState {
    int count;      // SoundSource::count
    float old_ele;  // SoundSource::old_ele
    float old_azi;  // SoundSource::old_azi
    int pad;
};

struct SrcSignal {
    const float *ptr;  // SoundSource::buf, device copy
    int length;        // SoundSource::length (0 = silent)
    int pad;
};

// The measurement grid of the HRTF set: elevation rings, each sampled at a uniform azimuth step from azimuth 0; table rows
// ring by ring, azimuth ascending.  For the reference's KEMAR grid (kemar = 1) these are the tables of hrtf_signals.cu:7-12,
// filled on the host by the reference's own loop (the steps are the reference's ROUNDED ones: 6.43 for 360 / 56 ...), and the
// reference's index/weight rule applies unless the corrected one is asked for; any other grid (jf_engine_create_grid)
// is worked by the corrected rule in its general form (dev_interp_corrected) and a plain nearest-measurement search.
constexpr int kPickAzi = 401;  // integer azimuths 0 .. 400 of the nearest-azimuth table
struct RingTable {
    int n_rings;  // 14 for KEMAR
    int n_rows;   // 710 for KEMAR: offset[n_rings]; the pre-interpolated rows follow row n_rows - 1
    int kemar;    // 1: the reference's grid (ring r at -40 + 10 r degrees; `pick` valid)
    int pad;
    int offset[kMaxRings + 1];
    float inc[kMaxRings];
    float ele[kMaxRings];  // elevation of ring r, ascending
    // KEMAR only: [kNumElev][kPickAzi] device table (null: search): the table row nearest to integer azimuth a on ring e --
    // what pick_hrtf's search over a ring (hrtf_signals.cu:20-51) returns for (elevation of e, a); built by that search
    const short *pick;
};

struct FusedParams {
    const float4 *htab;
    const float2 *tw;
    const ItemDesc *desc;   // [K][S]
    const SrcSignal *sigs;  // [S]
    const SrcState *st_in;  // [S]
    SrcState *st_out;       // [S]
    const float *hist_in;   // [S][1024]
    float *hist_out;        // [S][1024]
    const float *pos;       // [K][S][5] (window of the uploaded trajectory)
    float *partial;         // [K][S/G][2B]
    int S, K, B;
    int G;  // consecutive sources summed in registers by one wavefront (S % G == 0)
    int mode;  // bit 0 = FD_BASIC, bit 1 = corrected index/weight rule, bit 2 = htab holds the pre-interpolated rows: used
               // where descriptors are built in-kernel (real-time kernel, the pair kernel's trailing workgroups)
    const int *order;  // [S] pair kernel: unit u works on sources order[G u .. G u + G - 1] (identity unless the engine sorted)
    int *err;  // host-mapped word: set to 1 if a pair hand-off of fused_pair_kernel ever times out (never, by construction)
    // fused_pair_kernel only: workgroups n_pair_wgs .. gridDim.x - 1 prepare the descriptors of the window that FOLLOWS this
    // run in the uploaded trajectory (prep_kernel's work, 512 items per workgroup) while the last pairs finish
    int n_pair_wgs = 0;                // workgroups that work on units (set by launch_fused)
    const float *prep_pos = nullptr;   // [prep_K][S][5] the following window's positions (null: nothing to prepare)
    ItemDesc *prep_desc = nullptr;     // [prep_K][S] where its descriptors go
    int prep_K = 0, prep_canon = 0;
    RingTable rt = {};
};

// Convolution reverb stage (jf_reverb.hip): uniformly partitioned overlap-save with a
// frequency-domain delay line.  Spectra are packed: B complex per partition, bin 0 =
// (X[0].re, X[B].re).
//   fdl    float2[S][Rg][B]   last Rg input spectra of every source (ring, slot = block index mod Rg)
//   hspec  float2[P][B]       IR partition spectra, pre-scaled by gain / B
//   wet    float[S][Wr]       reverberated mono signal, read by the spatialiser as the source signal
//   prev   float[2][S][B]     last dry block (ping-pong per call), dry_count int[2][S]
struct ReverbParams {
    const float2 *tw;
    const SrcSignal *dry;     // [S]
    const int *dry_count_in;  // [S]
    int *dry_count_out;       // [S]
    const float *prev_in;     // [S][B]
    float *prev_out;          // [S][B]
    float2 *fdl;
    const float2 *hspec;
    float *wet;
    const SrcState *st_in;    // count = wet-ring position of the first new sample of this call
    int S, K, B, P, Rg, Wr, head;
    int mac_form;             // 0 = chosen by call size; 1 per (block, source), 2 source groups, 3 block tiles
    // Non-uniform partitioning (ReverbBigParams below): this stage is then the HEAD of the impulse response -- P = M
    // partitions of B -- and adds the tail's contribution, which the big-partition kernels left in `fut`; every block's dry
    // samples are also copied to `dryring`, from which the big-partition transform reads.  Null / 0: uniform partitioning.
    float *dryring = nullptr;  // [S][Rd] dry input by absolute sample time mod Rd
    int Rd = 0, dry_pos0 = 0;  // ring length; position of this call's first sample
    const float *fut = nullptr;  // [S][F] the tail's contribution by absolute sample time mod F
    int F = 0, fut_pos0 = 0;
    // blocks copy_lo <= k < copy_hi of the call only copy their samples to the dry ring (no transform): the blocks of a batch
    // call whose output the big partitions form directly and whose small spectra nobody reads later (0, 0: none)
    int copy_lo = 0, copy_hi = 0;
    // ... and blocks skip_lo <= k < skip_hi are not visited at all by the transform kernel: neither their spectrum nor their
    // place in the dry ring will be read (the big-partition transforms take samples of the running call from the signal)
    int skip_lo = 0, skip_hi = 0;
    // the multiply-accumulate / finishing stage works on blocks kb .. kb + kn - 1 of the call (set by launch_reverb)
    int kb = 0, kn = 0;
    // CATCH-UP (round 5): the K blocks are the last blocks of an EARLIER call whose small transforms were put off (a batch
    // call of whole big blocks leaves them to whoever needs them: ReverbBigParams::state_out): their samples are read from the
    // dry ring -- block k at ring position dry_pos0 + k B, the block before it B in front -- and only their spectra are
    // written (slots head + k): no ring, no play position, no previous block (the call that put them off left those)
    int catchup = 0;
};

// The non-uniformly partitioned reverb's second level: partitions of B1 = M * B taps (transform length 2 B1).
//   X_m   = spectrum of the dry samples of big blocks m - 2 and m - 1 (absolute block indices 16 (m - 2) .. 16 m - 1), formed
//           as soon as block 16 m - 1 has been taken in;
//   H'_0  = spectrum of the response's first B1 taps, H'_1 .. H'_P1 = of the B1-tap partitions behind them.
// A block that is worked on its own (per-block calls, the ragged ends of batch calls) gets taps [0, 2 B1) from the uniform
// stage above -- the head: 2 M partitions of B, no latency -- and the rest from  TAIL(m) = sum_{q = 2 .. P1} X_{m+1-q} H'_q, the
// B1 samples the partitions behind the head contribute to big block m (Gardner's zero-latency scheme with two sizes): per block
// (P1 - 1) / 16 + 32 multiply-accumulates per bin instead of P = 16 (P1 + 1).  TAIL(m) needs nothing newer than X_{m-1}, which
// exists a whole big block before big block m begins: that slack is what lets one-block calls form it on a second stream
// (jf_engine_reverb.cpp: run_reverb_stage).  A big block that lies
// INSIDE a batch call needs no head at all:  FULL(m) = sum_{q <= P1} X_{m+1-q} H'_q  is its whole wet signal (uniform
// partitioning at the big size; its input is all there), one transform pair per 16 blocks instead of 16 pairs.
// blocks per big block: B1 = rv_big_blocks(B) * B is 1024 or 2048 taps (a transform of 2 B1 points by one workgroup in LDS)
constexpr int rv_big_blocks(int B) { return B <= 128 ? 16 : 8; }
struct ReverbBigParams {
    const float2 *tw1;      // exp(+2 pi i j / (2 B1)), j < 2 B1 (a full circle), from double
    const float *dryring;   // [S][Rn B1] dry samples of EARLIER calls by absolute time (and of this call's blocks that the
                            // uniform stage transformed); samples of the running call are read from the signal itself
    const SrcSignal *dry;   // [S]
    const int *dry_count_in;  // [S] play position of the call's first sample
    int dry_pos0;           // ring position of the call's first sample
    float2 *fdl1;           // [S][R1][B1] the X_m (ring, slot = m mod R1), + [S][R1] compact copies of their packed bin-0 pairs
    const float2 *hspec1;   // [NP][B1] the H'_q, pre-scaled by gain / B1 (q <= P1; zeros behind them: the product kernel works
                            // in groups of 16 partitions), + [NP] compact bin-0 pairs
    float2 *ybig;           // [S][n_prod][B1] products of one launch
    float *fut;             // [S][Fn B1] TAIL(m) at ring block m mod Fn
    float *wet;             // [S][Wr] the wet ring (FULL(m) goes straight there)
    const SrcState *st_in;  // count = wet-ring position of the call's first new sample
    int S, B, B1, P1, R1, Rn, Fn, Wr;
    int M;                  // blocks per big block: B1 / B
    int NP;                 // partitions hspec1 has room for (P1 + 1 + 16)
    // transforms: X_m for m = first .. first + n_tr - 1
    int n_tr = 0;
    int tr_slot_first = 0;    // first mod R1
    int tr_rel_first = 0;     // first sample of X_first's 2 B1 samples, relative to the call's first sample (< 0: before it)
    // products: Y_i = sum_{q < n_part} X_{anchor + i - q} H'_{h_first + q}, i < n_prod
    int n_prod = 0;
    int anchor_slot_first = 0;  // anchor mod R1
    int h_first = 0, n_part = 0;
    int to_wet = 0;             // 0: Y_i -> fut ring block (fut_first + i) mod Fn;  1: -> the wet ring, blocks wet_k0 + M i .. + M - 1 of the call
    int fut_first = 0, wet_k0 = 0;
    int mac_wgs = 0;  // single products: at most this many workgroups, taking the items in turn (0: one per item)
    // transforms of a batch call that ENDS on a big-block boundary and puts its small transforms off (jf_engine.cpp:
    // rv_small_stale): the last transform of every source -- its 2 B1 samples are the call's last two big blocks -- also leaves
    // what the small transforms' kernel would have left: those samples in the dry ring (later transforms and a catch-up read
    // them there), the call's last block as `prev`, the play position behind the call
    int state_out = 0;
    float *dryring_out = nullptr;  // = dryring
    float *prev_out = nullptr;     // [S][B]
    int *dry_count_out = nullptr;  // [S]
    int call_samples = 0;          // K B
};

// What the reverb stage does in one call (host side; launch_reverb)
struct ReverbPlan {
    bool big = false;             // non-uniform partitioning
    ReverbBigParams tail_early;   // TAIL(m) for the big block the call starts in (n_prod = 0: not needed)
    ReverbBigParams transforms;   // n_tr
    ReverbBigParams middle;       // FULL(m) of the big blocks inside the call (to_wet)
    ReverbBigParams tail_late;    // TAIL(m) for the big block the call ends in
    int n_ranges = 1;             // block ranges the uniform stage's multiply-accumulate works on
    int kb[2] = {0, 0}, kn[2] = {0, 0};
    int forms[2] = {0, 0};        // out: form each range took
    // A one-block call whose head the spatialiser's one-launch kernel runs itself (rt_block_kernel<.., true>: rv_head_wave):
    // launch_reverb then launches what must come BEFORE the head only (tail_early) and reports form 5; the caller launches the
    // real-time kernel and, behind it, `transforms` if any are left in line (launch_reverb_big_side(&transforms, nullptr, stream)).
    bool head_fused = false;
};

}  // namespace jf
