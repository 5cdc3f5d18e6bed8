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
 * batch run whose policy takes them (jf_debug_set_interp_table), ~0.1 ms of kernel time and one allocation on that run --
 * so an engine whose sources move every block, and each of several engines of a job on one device, never holds them.
 * Positions that are not whole degrees inside -40..90 x 0..359 keep the per-block weighting, and so do runs in which most
 * sources move every block.  JF_FLAG_NO_INTERP_TABLE: never build them (the environment variable JF_INTERP_TABLE=0 does
 * the same for every engine of a process).
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
 * The rings of a set from the directions of its measurements -- e.g. the (azimuth, elevation) columns of a SOFA file's
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
int jf_grid_rows(const jf_hrtf_grid *grid);      /* table rows of the grid, or a JF_ERR_* code */
int jf_engine_create_grid(const jf_config *cfg, const jf_hrtf_grid *grid, const float *hrir, int taps, jf_engine **out);
/* idx = {ring0 low, ring0 high, ring1 low, ring1 high} rows, omegas = {A, B, C, D, E, F} as in jf_interpolation */
int jf_grid_interpolation(const jf_hrtf_grid *grid, float ele, float azi, int idx[4], float omegas[6]);
int jf_grid_pick(const jf_hrtf_grid *grid, float ele, float azi);   /* nearest measurement's row */
/* rows of this engine's HRTF table (710 for KEMAR) */
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
/* tests: a numeric dataset of any HDF5 file the reader understands, as doubles (malloc'd: jf_free); dims[8] */
int jf_debug_hdf5_read(const char *path, const char *dataset, double **out, int *rank, unsigned long long *dims);
/* tests: a string attribute of an object ("" or "/": the root group); JF_ERR_ARG if there is none */
int jf_debug_hdf5_attr(const char *path, const char *object, const char *attr, char *out, size_t cap);

/* closeEverything() / cleanup_hrtf_buffers() / ~GPUSoundSource (hrtf_signals.cu:100-105, GPUSoundSource.cu:532-548). */
void jf_engine_destroy(jf_engine *e);

/* Text of the last error on this engine (or of the last failed create when e == NULL). */
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

/* The same conversions without an engine, producing the latched record
 * {ele, azi, x, y, z} used by the batch calls. */
int jf_position_from_spherical(float ele, float azi, float r, float out[JF_POS_FLOATS]);
int jf_position_from_cartesian(float x, float y, float z, float out[JF_POS_FLOATS]);
/* n records at once: out[n][JF_POS_FLOATS] (trajectory construction for the batch calls). */
int jf_positions_from_spherical(size_t n, const float *ele, const float *azi, const float *r, float *out);

/* SoundSource::interpolationCalculations (SoundSource.cu:65-105) + pick_hrtf
 * (hrtf_signals.cu:20-51), host side, for inspection/tests. */
int jf_interpolation(float ele, float azi, int hrtf_indices[4], float omegas[6]);
/* The same for an engine created with `flags` (JF_FLAG_CORRECTED_INTERPOLATION selects the corrected rule). */
int jf_interpolation_ex(float ele, float azi, unsigned flags, int hrtf_indices[4], float omegas[6]);
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
 * callback_func for GPU_FD_COMPLEX exactly as the reference orders it: output
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
 * n_blocks consecutive callbacks in one call.  positions:
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
 * setter calls with these (already rounded) values leave behind.  jf_batch_run, whose positions live on the device, does
 * not move the sources; a host that follows it with per-block calls says where they stand with this or with the setters. */
int jf_sources_set_latched(jf_engine *e, const float *records);

/*
 * Device-resident form: positions are uploaded once, then any window of them
 * is processed with no host<->device traffic.  jf_batch_run launches on the
 * engine's stream and returns without waiting; d_out_mix is a DEVICE pointer
 * to [n_blocks][2*B] floats (NULL -> the engine's own buffer, see
 * jf_batch_mix_device).  Blocks first_block .. first_block + n_blocks - 1 of
 * the uploaded trajectory are consumed.
 */
int jf_batch_upload_positions(jf_engine *e, int total_blocks, const float *positions);
int jf_batch_run(jf_engine *e, int first_block, int n_blocks, float *d_out_mix);
int jf_synchronize(jf_engine *e);
/* Device pointers owned by the engine (valid until destroy). */
float *jf_batch_mix_device(jf_engine *e);     /* [max_batch_blocks][2*B] */
float *jf_batch_partial_device(jf_engine *e); /* [blocks][n_sources / G][2*B]: stereo blocks of the last run, summed over
                                                 groups of G consecutive sources (jf_debug_set_source_group) */
/* hipStream_t the engine launches on, as void*. */
void *jf_engine_stream(jf_engine *e);

/* Timing with HIP events recorded on the engine's stream.  enable = 1 brackets the fused kernel
 * only (two events per call: what bench.py needs for the roofline), 2 brackets every kernel (prep,
 * reverb, fused, mix), 0 switches it off; jf_profile_read waits and returns the accumulated
 * milliseconds and launch count since arming (prep/mix are 0 at level 1). */
int jf_profile_enable(jf_engine *e, int enable);
int jf_profile_read(jf_engine *e, double *fused_ms, double *prep_ms, double *mix_ms, long *launches);
/* Put the event records around every `every`-th batch run only (default 1: around all; the runs in between launch the
 * same kernels).  A pair of records costs ~7 us of stream time, 2.7 % of a 0.25 ms run; jf_profile_read's `launches`
 * counts the runs that were timed. */
int jf_profile_set_stride(jf_engine *e, int every);
int jf_profile_read_reverb(jf_engine *e, double *reverb_ms); /* the two reverb kernels, same launches */

/* ---- debugging / parity taps ------------------------------------------ */

/* Sources one unit of work sums (as spectra) before it writes a stereo block (the reference's per-source
 * `intermediate` corresponds to 1).  0 = automatic (2 to 16 for large batches, and the sources are taken in the order
 * jf_debug_source_order reports); a value > 0 must divide n_sources and groups CONSECUTIVE sources.  The mix is the
 * same sum in a different association. */
int jf_debug_set_source_group(jf_engine *e, int group);
/* order[n_sources]: unit u of the LAST batch run summed sources order[G u] .. order[G u + G - 1], G =
 * jf_debug_last_source_group.  With automatic grouping jf_batch_upload_positions orders the sources by the table row
 * nearest to their first position (units that run side by side then read neighbouring rows of the table); with a
 * pinned group size, and whenever the last run resolved to G = 1 (per-source blocks: block u of
 * jf_batch_partial_device is source u), the identity.  Before the first run: the order a grouped run will take. */
int jf_debug_source_order(const jf_engine *e, int *order);
/* Form of the reverb's multiply-accumulate stage: 0 = by call size (default); 1 = one workgroup per
 * (block, source) -- what real-time calls use; 2 = groups of sources share each IR partition spectrum;
 * 3 = tiles of consecutive blocks share a sliding window of input spectra (large batch calls).  The
 * forms add the same products in different associations. */
int jf_debug_set_reverb_form(jf_engine *e, int form);
/* Per-block calls (jf_process_block / jf_submit_block / jf_callback) with at most n sources use the
 * one-launch real-time kernel (descriptors + spatialisation + mix per workgroup of 8 sources -- 16 beyond 512 --, pinned host
 * I/O, the workgroups' blocks added on the host in order); above that, the batch pipeline with one block.
 * Default 8192; 0 disables the real-time kernel. */
int jf_debug_set_rt_max_sources(jf_engine *e, int n);
/* ';'-separated names of the kernels the last processing call launched, in launch order (bench.py labels its
 * roofline with them).  The string is owned by the engine and valid until the next call of this function. */
const char *jf_debug_last_kernels(jf_engine *e);
/* G the last batch pipeline run used (1 = fused_block_kernel, > 1 = fused_pair_kernel). */
int jf_debug_last_source_group(const jf_engine *e);
/* Caps the persistent grid of the fused kernel at `workgroups` (0 = what the device holds): with a small cap every
 * wavefront loops over several work units, which full-size calls do only beyond 4096 units. */
int jf_debug_set_grid_limit(jf_engine *e, int workgroups);
/* jf_batch_run prepares the descriptors of the window that FOLLOWS its own in the uploaded trajectory -- in trailing
 * workgroups of the pair kernel's own launch ("fused_pair_kernel<n>+prep" in jf_debug_last_kernels), or for single
 * sources inside its mix launch (mix_prep_kernel) -- and the next jf_batch_run uses them if it asks for exactly that
 * window; anything else that runs or changes the engine's state in between discards them.  on = 0 switches this off
 * (every run launches prep_kernel and mix_kernel); default on.  Results are bit-identical either way. */
int jf_debug_set_prep_ahead(jf_engine *e, int on);
/*
 * Stage taps the reference's own tests compare (precision_test.cu:60-75 distance factor, :225-241 and :374-404
 * weighted spectra), through the device code of the fused kernels.  positions[n][JF_POS_FLOATS];
 * dist[n][513][2]: D[k] of generateDistanceFactor (kernels.cu:116-125; bin 512: real part only, imaginary 0 --
 * c2r never reads it).  If spectra != NULL: windows[n][1024] -> spectra[n][2][513][2] = Y_ear[k] =
 * sum_t w_t X[k] H[row_t][ear][k] D[k] with X = rfft(window)/N, the filter set of positions[i].
 */
int jf_debug_stage_taps(jf_engine *e, int n, const float *positions, const float *windows, float *dist,
                        float *spectra);
/* Timing experiments (kernels built with -DJF_EXP_STAMPS): n 64-bit time stamps the last launch left (n <= 8192). */
int jf_debug_read_stamps(jf_engine *e, unsigned long long *out, int n);
/* Synchronous device-to-host copy of an engine-owned buffer (jf_batch_mix_device, ...). */
int jf_debug_copy_from_device(jf_engine *e, const void *device_ptr, void *host, size_t bytes);

/* Partitioning of the convolution reverb, in effect from the next jf_reverb_set_ir.  With M = blocks per big partition (16 for
 * frames_per_buffer 64 and 128, 8 for 256: big partitions of M * frames_per_buffer = 1024 or 2048 taps) and P = partitions
 * of frames_per_buffer the response has: 0 = by the response's length (default: non-uniform from P >= 3 M on, unless
 * jf_debug_set_reverb_form pins a uniform form), 1 = uniform (one partition per block: P multiply-accumulates per bin
 * and block), 2 = non-uniform (a head of 2 M partitions of one block + partitions of M blocks for the rest: about
 * P / M + 2 M - 2 per block; Gardner's zero-latency scheme with two sizes, the large size starting two of its partitions
 * into the response, so that its work for a big block can be done a whole big block early; forced on a response shorter than
 * 3 M blocks it degenerates gracefully -- no, one or two big partitions behind the zero-padded head: tested).  The reference's
 * own form is one product over the whole signal (cudaPart.cu:87-153).  Same results to float32 rounding. */
int jf_debug_set_reverb_partitioning(jf_engine *e, int how);
/* One-block calls with the non-uniformly partitioned reverb (the real-time shape) run the big partitions' kernels on a second
 * stream, off the block's critical path: when a block completes a big block, the spectrum of that big block, the products of
 * the big block after the next and their inverse transform are launched there behind the block's spatialiser, sixteen blocks
 * before their result is first read (jf_engine.cpp: run_reverb_stage).
 * on = 0: everything in line on the engine's stream, as batch calls, calls with a pinned form and profiled calls do anyway
 * (the last block of a big block then costs ~9 us more than the others at configs[4], the first ~40 us).  Default 1.  Same
 * kernels, same order of every sum: bit-identical results. */
int jf_debug_set_reverb_async(jf_engine *e, int on);
/* One-block calls with a short head (the 2 M partitions of a non-uniformly partitioned response, or a response of at most 64
 * blocks) and at most 512 sources CAN run the reverb's head INSIDE the one-launch real-time kernel (on = 1): the wave that
 * spatialises a source first takes its block through the head's partitions (in order) and leaves it in the wet ring -- one
 * launch per audio block instead of two.  Off by default (the head as a kernel of its own in front): the one launch measured
 * 5 us slower per block at 256 sources, a head being one wave's chain there (profiles/r05/reverb_realtime.md).  Same sums of
 * the same products in another order: equal to float32 rounding, not bit for bit. */
int jf_debug_set_reverb_head_fused(jf_engine *e, int on);
/* A batch call of whole big blocks that ends on a big-block boundary reads none of the small transforms of its last blocks --
 * they are state for a later call's head, and the next such call never looks at them -- so by default (on) it puts them off:
 * its last transform leaves the samples in the dry ring, and the first call that takes a block through the head forms them
 * from there (same samples, same transform: the same bits; 12 us of config 5's batch step).  on = 0: formed by every call. */
int jf_debug_set_reverb_lazy_state(jf_engine *e, int on);
/* One-block calls through the one-launch real-time kernel launch the reverb stage of the NEXT block right behind their own
 * spatialiser (on, the default): the stage needs the dry signals and its own state, not the positions the host sets for that
 * block, so the next call finds the wet block there and launches the spatialiser alone -- the head kernel leaves the block's
 * critical path (between two audio callbacks it has the whole block period).  Only a plain head goes ahead (no big block
 * completed, no TAIL owed); a new signal, a reset, a new response, a batch call or a switch of the stage's knobs takes it back
 * (the stream is waited for, the stage is done again by the call that needs it).  Bit-identical.  on = 0: every call runs its
 * own stage first. */
int jf_debug_set_reverb_ahead(jf_engine *e, int on);
/* The schedule of the non-uniformly partitioned reverb for a call of K blocks that starts at absolute block j0, with big blocks
 * of M blocks and TAIL formed up to big block fut_m (host logic only: no engine, no GPU; tests/test_reverb_plan.py replays
 * runs of calls against a model of the rings).  out = {m_lo, n_tr, ma, n_mid, n_ranges, kb0, kn0, kb1, kn1, copy_lo, copy_hi,
 * skip_lo, skip_hi, tail_early (-1: none), tail_late (-1: none), new fut_m} -- jf_host.h: ReverbSchedule. */
int jf_debug_reverb_schedule(long long j0, int K, int M, long long fut_m, long long out[16]);
/* Returns the number of partitions of frames_per_buffer the impulse response has (0: stage off); *head = partitions of that
 * size in use (the head: two big partitions' worth), *big = partitions of *big_taps taps behind them (0, 0: uniform
 * partitioning) -- the decomposition blocks take that go through head + TAIL; whole big blocks inside a batch call are formed
 * from *big + 2 partitions of *big_taps alone. */
int jf_debug_reverb_partitions(const jf_engine *e, int *head, int *big, int *big_taps);
/* Which batch calls use the pre-interpolated rows (JF_FLAG_NO_INTERP_TABLE above): 0 = none, 1 = all, 2 = decided per run
 * (the default when the rows were built): a run of an uploaded trajectory takes them unless more than 30 % of its items
 * move -- a source that stays reads its row out of the caches (12-18 % faster), one that moves streams 8 KB per block from
 * HBM, and a run in which every source moves every block is 2-5 % slower with the rows than with the weighting of the
 * cached measured rows; measured crossover: a third of the items moving --; calls without a trajectory take them.
 * on != 0 for an engine that may not build them (JF_FLAG_NO_INTERP_TABLE): JF_ERR_STATE.  on == 1 builds them now
 * (JF_ERR_NOMEM without room for them), on == 2 leaves that to the first run that takes them.  Results are bit-identical
 * whatever the choice (JF_INTERP_TABLE=0/1/2 in the environment sets it for every engine of a process). */
int jf_debug_set_interp_table(jf_engine *e, int on);
/* 1 once the engine holds the pre-interpolated rows (built on first use). */
int jf_debug_interp_table_built(const jf_engine *e);
/* 1 if the last batch run's descriptors could name pre-interpolated rows (the kernel instantiation that reads them ran). */
int jf_debug_last_run_used_rows(const jf_engine *e);
/* The setting above (0, 1 or 2); 0 for an engine without the rows. */
int jf_debug_interp_table(const jf_engine *e);
/* How many of the first n_items descriptors of the last batch run (items b * n_sources + s) carry any bit of `mask` in
 * their flags (pair-kernel layout: 1 = both sets on one row list, 2 = crossfade, 4 = both sets are pre-interpolated
 * rows); < 0: error. */
int jf_debug_count_desc_flags(jf_engine *e, int n_items, int mask);
/* n rows of the device table in the DEVICE layout (512 x {L.re, L.im, R.re, R.im} per row, bin 0 = {L[0], L[512], R[0],
 * R[512]}), starting at `first_row`: rows 0..709 are the measured ones, row 710 + (ele + 40) * 360 + azi the
 * pre-interpolated filter of the whole-degree position (ele, azi). */
int jf_debug_read_table_rows(jf_engine *e, int first_row, int n, float *out /* n*512*4 */);
/* Copy of the device HRTF spectrum table in the REFERENCE layout
 * fft_hrtf[(j*2 + ear)*Nc + k] (hrtf_signals.cu:90-98), complex64 -> 2 floats. */
int jf_debug_read_table(jf_engine *e, float *out /* 710*2*Nc*2 */);
/* Runs only the index/weight kernel on n latched (ele, azi) pairs:
 * rows[n][4], weights[n][4], nterms[n] (<= 0: not interpolable). */
int jf_debug_interp_device(jf_engine *e, int n, const float *ele, const float *azi,
                           int *rows, float *weights, int *nterms);
/* Forward real FFT of n windows of PAD_LEN samples with the kernel's LDS FFT
 * (unnormalised, Nc complex bins each). */
int jf_debug_rfft_device(jf_engine *e, int n, const float *windows, float *spectra);

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
