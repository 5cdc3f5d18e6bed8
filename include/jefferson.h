/*
 * jefferson.h -- C ABI of the MI355X-native HRTF binaural convolution engine.
 *
 * Drop-in boundary for the audio-callback / SoundSource-update surface of
 * Cindytb/Jefferson-2.0.  The reference has no FFI layer: its boundary is a
 * PortAudio C callback plus public C++ members shared through a global
 * `Data` object (Jefferson/src/main.cu:12-13).  Every entry point below names
 * the reference interface it replaces (paths relative to Jefferson/src/).
 *
 * Plain C: opaque handle, plain pointers and sizes, int status codes.
 * Nothing here exits the process or throws (the reference prints and
 * exit(1)s: cufftDefines.cuh:69-77, Audio.cu:16-55).
 *
 * Implemented by libjefferson_hip.so (jefferson-2.0_amd/csrc).  There is no
 * CPU fallback: if no HIP device is usable, jf_engine_create fails with
 * JF_ERR_DEVICE.
 */
#ifndef JEFFERSON_H
#define JEFFERSON_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JF_NUM_HRTF 710 /* Universal.cuh:4  NUM_HRTF */
#define JF_HRTF_CHN 2   /* Universal.cuh:11 HRTF_CHN */
#define JF_POS_FLOATS 5 /* latched position record: ele, azi, x, y, z */

enum {
    JF_OK = 0,
    JF_ERR_ARG = -1,     /* bad argument / out-of-range index */
    JF_ERR_RANGE = -2,   /* position the reference cannot interpolate (ele outside (-50, 90]) or |coords| == 0 */
    JF_ERR_DEVICE = -3,  /* HIP runtime error; text in jf_last_error.  Also the kernels' own fault report: the batch kernel
                            bounds every wait between its wavefronts (~0.1 s), and a wait that runs out -- impossible by
                            its protocol -- raises a host-visible error word instead of hanging the GPU.  That condition
                            is FATAL for the engine: the next jf_synchronize / jf_collect_block / jf_process_* / jf_callback
                            returns JF_ERR_DEVICE ("hand-off timed out"), jf_pa_callback hands PortAudio silence, and so
                            does every later processing call; destroy the engine (the reference's checkCudaErrors
                            exits the process, cufftDefines.cuh:69-77) */
    JF_ERR_IO = -4,      /* HRIR / WAV file problem */
    JF_ERR_STATE = -5,   /* call out of order (e.g. collect without submit) */
    JF_ERR_NOMEM = -6
};

typedef struct jf_engine jf_engine;

/*
 * Replaces the compile-time constants of Universal.cuh:4-13 and the
 * constructor arguments of `new GPUSoundSource[num_sources]` (main.cu:60-61).
 */
/*
 * jf_config.flags.  Default 0: bug-compatible with the reference's index/weight rule
 * (SoundSource.cu:65-105: elevations truncated toward zero, so (-10, 0) interpolates as if it were
 * [0, 10) with one negative weight; azimuths truncated to whole degrees, so the two weights on the 6.43 /
 * 8 / 12 / ... degree rings do not sum to 1; no wrap from a ring's last azimuth to 360 = its first).
 * JF_FLAG_CORRECTED_INTERPOLATION: true floor of the elevation, float azimuths folded into [0, 360) with the
 * wrap, weights that sum to 1, elevations below -40 clamped to the lowest ring.  Not in the reference.
 */
#define JF_FLAG_CORRECTED_INTERPOLATION 1u
/*
 * The setters round elevation and azimuth to whole degrees (SoundSource.cu:33-34,42-43), so every position they latch is
 * one of 131 x 360.  By default an engine may therefore also hold, behind the 710 measured rows, the weighted filter
 * sum_t w_t H[row_t] of each of those positions (386 MB of HBM, built by the same operations in the same order as the
 * per-block weighting: results are bit-identical) and batch calls read ONE row per filter set instead of up to four rows
 * and their weights (what GPUSoundSource.cu:118-292 recomputes for every block).  The rows are built LAZILY -- by the first
 * batch call of more than one block whose policy takes them, ~0.1 ms of kernel time and one allocation on that call; never
 * by a one-block call, which is the audio callback's -- so an engine whose sources move every block, and each of several
 * engines of a job on one device, never holds them.  Positions that are not whole degrees inside -40..90 x 0..359 keep the
 * per-block weighting, and so do runs in which most sources move every block.  JF_FLAG_NO_INTERP_TABLE: never build them.
 * (jefferson_debug.h: jf_debug_set_interp_table overrides the policy and, with 1, builds the rows at once -- the pre-warm
 * call for a host that wants them before its first batch.)
 */
#define JF_FLAG_NO_INTERP_TABLE 2u

typedef struct jf_config {
    int frames_per_buffer; /* FRAMES_PER_BUFFER (Universal.cuh:10): 128 or 256 (any multiple of 64 up to 256) */
    int hrtf_len;          /* HRTF_LEN (Universal.cuh:9): 512 -> PAD_LEN 1024 (Universal.cuh:12) */
    int n_sources;         /* num_sources (main.cu:60) */
    int device;            /* HIP device ordinal */
    int max_batch_blocks;  /* capacity of jf_process_batch / jf_batch_run (>= 1) */
    unsigned flags;        /* 0 = the reference's behaviour; JF_FLAG_* above */
} jf_config;

/* ---- init / teardown ------------------------------------------------- */

/*
 * Replaces read_hrtf_signals() + transform_hrtfs() (hrtf_signals.cu:107-153,
 * :248) and the GPUSoundSource constructors (GPUSoundSource.cu:17-71).
 * hrir: [JF_NUM_HRTF][2][taps] float32, row order of the reference loader
 * (elevation-major, azimuth ascending; ear 0 = left), taps <= hrtf_len.
 * The engine builds the unnormalised 513-bin spectra on the GPU and keeps its
 * own copies; the caller's buffer is not retained.
 */
int jf_engine_create(const jf_config *cfg, const float *hrir, int taps, jf_engine **out);

/*
 * Same, loading the KEMAR set from a directory with libsndfile-compatible
 * scaling (int16 / 32768): either the reference's "full" layout
 * (full/elev%d/L%de%03da.wav + R..., hrtf_signals.cu:124,131) or the "compact"
 * layout shipped in the reference repo (compact/elev%d/H%de%03da.wav, stereo,
 * mirrored for azimuth > 180: hrtf_signals.cpp:80-126).
 */
int jf_engine_create_from_dir(const jf_config *cfg, const char *hrir_dir, jf_engine **out);

/*
 * Any HRTF set measured on a grid of elevation rings -- the author's TODO "Add compatibility for any HRTF database"
 * (FuturePlans.md:21); the reference hard-codes KEMAR's 14 rings in hrtf_signals.cu:7-12 and its loader loop :107-153.
 * Ring r lies at ring_elevation[r] degrees (ascending, within [-90, 90]) and holds ring_count[r] measurements, measurement
 * i at azimuth i * ring_step[r] degrees (ring_step == NULL: 360 / count; a ring of one measurement is the pole).  Table
 * rows run ring by ring, azimuth ascending: jf_grid_rows() of them, hrir [rows][2][taps].
 *   - The reference's own grid -- jf_kemar_grid(): 14 rings at -40 .. 90, 56 + 60 + 72 + ... + 1 = 710 rows, the ROUNDED
 *     steps of hrtf_signals.cu:8 (6.43 for 360 / 56 ...) -- is recognised: jf_engine_create_grid with it IS jf_engine_create
 *     (the reference's index/weight rule by default, bit-identical output; tested).
 *   - Any other grid is worked by the corrected rule of JF_FLAG_CORRECTED_INTERPOLATION in its general form: the two rings
 *     whose elevations enclose the position (positions outside the grid clamped to its first / last ring), linear
 *     weights in elevation, on each ring the two measurements that enclose the azimuth with the wrap at 360 and weights
 *     that sum to 1; JF_MODE_FD_BASIC takes the nearest ring's nearest measurement.  The setters accept elevations in
 *     [-90, 90].  jf_grid_interpolation / jf_grid_pick are the host twins of the kernels' rule (same float32 steps).
 * A set in a SOFA file: jf_engine_create_sofa below.
 */
#define JF_MAX_RINGS 40
typedef struct jf_hrtf_grid {
    int n_rings;                 /* 1 .. JF_MAX_RINGS */
    const float *ring_elevation; /* [n_rings] */
    const int *ring_count;       /* [n_rings] */
    const float *ring_step;      /* [n_rings] or NULL */
} jf_hrtf_grid;
int jf_kemar_grid(jf_hrtf_grid *out);            /* pointers into static storage of the library */
/*
 * The rings of a set (what hrtf_signals.cu:7-12 hard-codes for KEMAR) from the directions of its measurements -- e.g. the (azimuth, elevation) columns of a SOFA file's
 * SourcePosition, in degrees, read with any HDF5 tool: measurements whose elevations lie within tol_deg of each other form a
 * ring; a ring's measurements must sit at i * 360 / count from azimuth 0 within tol_deg (any order; 359.99 counts as 0; the
 * azimuths are taken as the engine's own -- convert the set's convention first).  layout receives the rings (point a
 * jf_hrtf_grid at its arrays; ring_step holds 360 / count), row_of[i] the table row of measurement i: hrir[row_of[i]] = IR[i].
 * JF_ERR_ARG (text in jf_last_error(NULL)) for a set that is not such a grid.
 */
typedef struct jf_grid_layout {
    int n_rings;
    float ring_elevation[JF_MAX_RINGS];
    int ring_count[JF_MAX_RINGS];
    float ring_step[JF_MAX_RINGS];
} jf_grid_layout;
int jf_grid_from_positions(size_t n, const float *azimuth_deg, const float *elevation_deg, float tol_deg, jf_grid_layout *layout,
                           int *row_of);
/* table rows of the grid (NUM_HRTF, Universal.cuh:4, is KEMAR's 710), or a JF_ERR_* code */
int jf_grid_rows(const jf_hrtf_grid *grid);
/* jf_engine_create (read_hrtf_signals + transform_hrtfs, hrtf_signals.cu:107-153, :248) for a set on `grid` */
int jf_engine_create_grid(const jf_config *cfg, const jf_hrtf_grid *grid, const float *hrir, int taps, jf_engine **out);
/* interpolationCalculations (SoundSource.cu:65-105) on `grid`: idx = {ring0 low, ring0 high, ring1 low, ring1 high} rows,
 * omegas = {A, B, C, D, E, F} as in jf_interpolation */
int jf_grid_interpolation(const jf_hrtf_grid *grid, float ele, float azi, int idx[4], float omegas[6]);
/* pick_hrtf (hrtf_signals.cu:20-51) on `grid`: the nearest measurement's row */
int jf_grid_pick(const jf_hrtf_grid *grid, float ele, float azi);
/* rows of this engine's HRTF table (NUM_HRTF, Universal.cuh:4: 710 for KEMAR) */
int jf_table_rows(const jf_engine *e);

/*
 * HRTF sets in SOFA files (AES69 "Spatially Oriented Format for Acoustics": a netCDF-4, i.e. HDF5, container) -- the rest of
 * "any HRTF database" (FuturePlans.md:21).  The library reads the container itself (csrc/jf_hdf5.c: own reader of the HDF5
 * structures such files are made of, checked against files written by libhdf5; zlib for deflated chunks) -- no HDF5 or
 * netCDF library is needed.
 *   jf_sofa_read: Data.IR [M][R][N], SourcePosition (spherical "degree, degree, metre", or Cartesian: converted), the
 *     sampling rate, Data.Delay (one row repeated when the file holds one), the SOFAConventions attribute.  Refused with
 *     JF_ERR_IO and a text in jf_last_error(NULL): files that are not HDF5 / use HDF5 structures outside the reader's set,
 *     DataType other than "FIR", missing variables, shapes that do not agree.  Buffers are the library's: jf_sofa_release.
 *   jf_sofa_table: the set as a table for jf_engine_create_grid -- the rings from the measurements' directions
 *     (jf_grid_from_positions: JF_ERR_ARG for a set that is not measured on rings of uniform azimuth steps), rows in ring
 *     order, hrir [M][2][taps] (taps >= jf_sofa_taps: the file's N + the largest Data.Delay).  SOFA azimuths run
 *     counter-clockwise (90 = left); the table's run the way KEMAR's file names do (90 = right): row azimuth =
 *     360 - SOFA azimuth.  Receiver 0 is the left ear.  A set on KEMAR's own rings gets the reference's description of them
 *     (jf_kemar_grid: the rounded steps), so an engine created from it IS jf_engine_create -- the reference's rule and blocks.  Whole-sample delays shift their impulse response; fractional ones
 *     are refused (JF_ERR_IO), and so are sets with other than two receivers or a sampling rate other than 44100 Hz (the
 *     reference's own check of its HRIR files, hrtf_signals.cu:68-75: the distance factor is written for that rate).
 *   jf_engine_create_sofa: the two, then jf_engine_create_grid.  cfg->hrtf_len must hold jf_sofa_taps.
 */
typedef struct jf_sofa_set {
    int n_measurements;   /* M */
    int n_receivers;      /* R */
    int n_samples;        /* N */
    double sample_rate;   /* Hz */
    float *ir;            /* [M][R][N] */
    float *azimuth;       /* [M] degrees, SOFA's sense (counter-clockwise from the front) */
    float *elevation;     /* [M] degrees up */
    float *distance;      /* [M] metres */
    float *delay;         /* [M][R] samples */
    char conventions[48]; /* SOFAConventions, "" if the file has none */
} jf_sofa_set;
int jf_sofa_read(const char *path, jf_sofa_set *out);
void jf_sofa_release(jf_sofa_set *set);
int jf_sofa_taps(const jf_sofa_set *set);
int jf_sofa_table(const jf_sofa_set *set, float tol_deg, jf_grid_layout *layout, float *hrir, int taps);
int jf_engine_create_sofa(const jf_config *cfg, const char *path, float tol_deg, jf_engine **out);

/* closeEverything() / cleanup_hrtf_buffers() / ~GPUSoundSource (hrtf_signals.cu:100-105, GPUSoundSource.cu:532-548). */
void jf_engine_destroy(jf_engine *e);

/* Text of the last error on this engine (or of the last failed create when e == NULL); the reference prints such texts and
 * exits (cufftDefines.cuh:69-77, Audio.cu:16-55). */
const char *jf_last_error(const jf_engine *e);

/*
 * Where the audio thread should run.  The per-block calls are host call -> one or two launches -> blocks written to pinned
 * host memory -> the host's poll: on a two-socket host every step of that crosses the sockets when the calling thread runs
 * on the other one than the GPU hangs off.  Measured (MI355X on a 2 x EPYC 9575F host, profiles/r04/rt_numa.md): 15.8 against
 * 16.9-17.8 us per block for one source, 22.6 against 25.8-27.8 us for 256.  The reference leaves its callback where PortAudio
 * starts it (Audio.cu:94-175); a host that cares pins the thread that calls jf_process_block / jf_callback.
 *   jf_device_numa_node: *node = NUMA node of HIP device `device` (from its PCI address, sysfs), -1 if the system does not
 *     say; JF_ERR_DEVICE if there is no such device.
 *   jf_pin_thread_to_device: restricts the CALLING thread to the CPUs of that node (sched_setaffinity; nothing else in the
 *     library touches affinities).  JF_ERR_STATE if the node or its CPUs are not known, or none of them is allowed to this
 *     process.  Call it before jf_engine_create, so that the engine's pinned buffers are first touched from there too.
 */
int jf_device_numa_node(int device, int *node);
int jf_pin_thread_to_device(int device);

int jf_frames_per_buffer(const jf_engine *e); /* FRAMES_PER_BUFFER */
int jf_pad_len(const jf_engine *e);           /* PAD_LEN */
int jf_num_sources(const jf_engine *e);       /* Data::num_sources (DataTag.cuh:14) */

/* ---- source signal and position (SoundSource.cuh:9-46) ---------------- */

/*
 * Replaces `source.buf / source.length / source.count = 0` set by cudaFFT()
 * (cudaPart.cu:198-199).  mono float32, looped playback as in
 * copyIncomingBlock (GPUSoundSource.cu:481-513).  The engine copies (host ->
 * device); n == 0 silences the source.  Like jf_source_reset and jf_reverb_set_ir it waits for the
 * engine's stream: call it from the thread that processes blocks, between blocks.
 */
int jf_source_set_signal(jf_engine *e, int src, const float *mono, size_t n);

/* SoundSource::updateFromCartesian(float3) (SoundSource.cu:20-36); callable from a
 * different thread than the audio thread; latched at the next block boundary. */
int jf_source_set_cartesian(jf_engine *e, int src, float x, float y, float z);

/* SoundSource::updateFromSpherical(ele, azi, r) (SoundSource.cu:41-54). */
int jf_source_set_spherical(jf_engine *e, int src, float ele, float azi, float r);

/* Reads back {ele, azi, r, x, y, z} (the public fields of SoundSource.cuh:24-36). */
int jf_source_get_position(const jf_engine *e, int src, float out[6]);

/* test() preamble (precision_test.cu:2097-2107): zero the window, count = 0, old_azi = old_ele = 0. */
int jf_source_reset(jf_engine *e, int src);

/* The same conversions (SoundSource.cu:20-54) without an engine, producing the latched record
 * {ele, azi, x, y, z} used by the batch calls. */
int jf_position_from_spherical(float ele, float azi, float r, float out[JF_POS_FLOATS]);
int jf_position_from_cartesian(float x, float y, float z, float out[JF_POS_FLOATS]);
/* updateFromSpherical (SoundSource.cu:41-54) for n records at once: out[n][JF_POS_FLOATS] (trajectory construction for the
 * batch calls). */
int jf_positions_from_spherical(size_t n, const float *ele, const float *azi, const float *r, float *out);

/* SoundSource::interpolationCalculations (SoundSource.cu:65-105) + pick_hrtf
 * (hrtf_signals.cu:20-51), host side, for inspection/tests. */
int jf_interpolation(float ele, float azi, int hrtf_indices[4], float omegas[6]);
/* The same (SoundSource.cu:65-105) for an engine created with `flags` (JF_FLAG_CORRECTED_INTERPOLATION selects the corrected
 * rule). */
int jf_interpolation_ex(float ele, float azi, unsigned flags, int hrtf_indices[4], float omegas[6]);
/* pick_hrtf (hrtf_signals.cu:20-51): the nearest measurement's table row, what the *_FD_BASIC / *_TD modes filter with. */
int jf_pick_hrtf(float ele, float azi);

/* ---- per-block processing (Audio.cu:94-175) --------------------------- */

/*
 * callback_func(output, p, false) with the CPU path's timing (Audio.cu:118-158):
 * block k's input produces block k's output.  out: interleaved stereo
 * float32, 2 * frames_per_buffer values, fully overwritten.  Synchronous.
 */
int jf_process_block(jf_engine *e, float *out);

/*
 * The CUDA path's pipelining (Audio.cu:104-117): jf_submit_block enqueues
 * block k (chunkProcess, GPUSoundSource.cu:463-471) and returns at once;
 * jf_collect_block waits for it (where the reference calls cudaStreamSynchronize,
 * Audio.cu:107) and returns it.  With the one-launch kernel the wait is a poll
 * of completion words the kernel stores into host memory behind the block; the
 * calling thread spins for the ~10 us the block takes, as it would inside the
 * runtime's synchronisation.
 */
int jf_submit_block(jf_engine *e);
int jf_collect_block(jf_engine *e, float *out);

/*
 * callback_func for GPU_FD_COMPLEX exactly as the reference orders it (Audio.cu:104-117): output
 * the block submitted by the PREVIOUS call (zeros on the first call), then
 * submit the next -> one block of latency.
 */
int jf_callback(jf_engine *e, float *out);

/*
 * paCallback (Audio.cu:164-175) with PortAudio's PaStreamCallback signature
 * (opaque pointers so portaudio.h is not needed); userData is the jf_engine*
 * (the reference passes &data).  Returns 0 (paContinue).
 */
int jf_pa_callback(const void *input, void *output, unsigned long frames_per_buffer,
                   const void *time_info, unsigned long status_flags, void *user_data);

/*
 * Data::type (DataTag.cuh:16, enum processes Universal.cuh:25-32), read at every block (Audio.cu:104).
 * JF_MODE_FD_COMPLEX = GPU_FD_COMPLEX / CPU_FD_COMPLEX, the interpolated path (default).
 * JF_MODE_FD_BASIC   = *_FD_BASIC (CPUSoundSource.cpp:113-142): nearest HRTF (pick_hrtf), no
 *   interpolation, no distance factor, no crossfade.  The time-domain modes (*_TD, 512-tap direct
 *   convolution with the same nearest HRIR, CPUSoundSource.cpp:66-112) compute the same samples:
 *   B + taps - 1 <= PAD_LEN makes the circular product a linear convolution (tested).
 */
enum { JF_MODE_FD_COMPLEX = 0, JF_MODE_FD_BASIC = 1, JF_MODE_TD = JF_MODE_FD_BASIC /* CPU_TD / GPU_TD: same samples */ };
int jf_set_mode(jf_engine *e, int mode);

/* Data::pauseStatus (DataTag.cuh:15, Audio.cu:101): while paused, blocks are silence and no input is consumed.
 * Like jf_set_mode, callable from another thread than the audio thread (an atomic flag read at every block). */
int jf_set_pause(jf_engine *e, int paused);

/* The clip alert of callback_func (Audio.cu:111-113 prints "ALERT" when a mixed sample exceeds 1.0): max |sample|
 * of the last block handed out by jf_collect_block / jf_process_block / jf_callback / jf_pa_callback. */
float jf_last_block_peak(const jf_engine *e);

/* ---- convolution reverb -------------------------------- */

/*
 * Replaces the offline whole-signal reverb of cudaFFT() (cudaPart.cu:65-205: mono input (*) mono
 * impulse response, then a gain that matches the output RMS to the input RMS, :118,161-165)
 * by real-time uniformly partitioned convolution ahead of the spatialiser: every source's
 * signal is convolved with `ir` (n_ir taps, mono; partitions of frames_per_buffer taps,
 * frames_per_buffer must be 64, 128 or 256) and scaled by `gain`.  n_ir == 0 turns the
 * stage off.  Call before processing starts or between blocks; it resets every source's
 * window and play position (like jf_source_reset).
 */
int jf_reverb_set_ir(jf_engine *e, const float *ir, size_t n_ir, float gain);

/* The reference's gain rule (cudaPart.cu:118,161-165): rms(x) / rms(x (*) ir) over the whole
 * signal (x zero-padded by ceil(n_ir/2), circular convolution of that length: cudaPart.cu:170-186),
 * evaluated on the host in double.  Returns 1 for degenerate inputs. */
float jf_reverb_rms_gain(const float *signal, size_t n, const float *ir, size_t n_ir);

/* ---- batch (offline / throughput) processing -------------------------- */

/*
 * n_blocks consecutive callbacks (callback_func, Audio.cu:94-163) in one call.  positions:
 * [n_blocks][n_sources][JF_POS_FLOATS] latched records, i.e. what the
 * reference's audio thread would have read from each source at each block
 * (crossfade state carries across blocks and across calls).
 * out_mix: [n_blocks][2 * frames_per_buffer] host buffer.
 * Afterwards the sources stand where the last callback read them, as they would had the setters been called before each
 * block (jf_sources_set_latched with the last block's records): a per-block call that follows continues from there, not from
 * what the setters held before the batch.  (Setter calls from another thread DURING the batch are overwritten by this.)
 */
int jf_process_batch(jf_engine *e, int n_blocks, const float *positions, float *out_mix);
/* Every source's position := its latched record {ele, azi, x, y, z} in records[n_sources][JF_POS_FLOATS] -- what n_sources
 * setter calls (SoundSource.cu:20-54) with these (already rounded) values leave behind.  jf_batch_run, whose positions live on the device, does
 * not move the sources; a host that follows it with per-block calls says where they stand with this or with the setters. */
int jf_sources_set_latched(jf_engine *e, const float *records);

/*
 * Device-resident form of the same loop of callbacks (Audio.cu:104-117; jf_synchronize is its
 * cudaStreamSynchronize, Audio.cu:107): positions are uploaded once, then any window of them
 * is processed with no host<->device traffic.  jf_batch_run launches on the
 * engine's stream and returns without waiting; d_out_mix is a DEVICE pointer
 * to [n_blocks][2*B] floats, or NULL -> the engine's own buffer, which
 * jf_batch_fetch below copies to the host.  Blocks first_block .. first_block + n_blocks - 1 of
 * the uploaded trajectory are consumed.
 */
int jf_batch_upload_positions(jf_engine *e, int total_blocks, const float *positions);
int jf_batch_run(jf_engine *e, int first_block, int n_blocks, float *d_out_mix);
int jf_synchronize(jf_engine *e);
/*
 * The mix of the last jf_batch_run into host memory: waits for the engine's stream, then copies blocks 0 .. n_blocks - 1 of
 * the engine's own mix buffer (the one jf_batch_run fills when d_out_mix == NULL) to out_mix[n_blocks][2 * frames_per_buffer].
 * JF_ERR_STATE if the last jf_batch_run was given a device pointer of the caller's, failed, or left fewer than n_blocks there.
 * With jf_batch_upload_positions / jf_batch_run / jf_synchronize this completes the device-resident form of callback_func's
 * loop (Audio.cu:104-117 over many callbacks) without a device pointer in the host's hands; hosts that keep the mix on the
 * device (a reduce over several GPUs: jf_group.c) use the accessors of jefferson_debug.h.
 */
int jf_batch_fetch(jf_engine *e, int n_blocks, float *out_mix);

/* ---- WAV I/O (cudaPart.cu:21-63 readFile; main.cu:77-82 output file) --- */

/* 16/24/32-bit PCM or float32 WAV -> mono float32 with libsndfile scaling;
 * stereo is mixed L/2 + R/2 (cudaPart.cu:50-52).  *out is malloc'd; free with jf_free. */
int jf_wav_read_mono(const char *path, float **out, size_t *n_frames, int *sample_rate);
/* Interleaved stereo float32 -> 24-bit PCM WAV (SF_FORMAT_PCM_24, main.cu:79). */
int jf_wav_write_stereo24(const char *path, const float *interleaved, size_t n_frames, int sample_rate);
void jf_free(void *p);

#ifdef __cplusplus
}
#endif
#endif /* JEFFERSON_H */
