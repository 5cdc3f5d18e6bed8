/*
 * jf_oracle.h -- CPU oracle for the HRTF binaural convolution hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (include/, the
 * jefferson-2.0_amd package, libjefferson_hip.so) may include, link or call
 * this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, and only as the checker / the timed CPU baseline.
 *
 * What it is: a plain-C float32 restatement of the reference's
 * frequency-domain interpolated path (GPU_FD_COMPLEX / CPU_FD_COMPLEX of
 * Cindytb/Jefferson-2.0).  Each function cites the reference file:line it
 * follows (paths relative to the reference's Jefferson/src/).
 *
 * PARITY PIN STATUS.  The reference cannot be built in this image (it needs
 * FFTW3, libsndfile, PortAudio and the CUDA SDK; no stand-ins are written),
 * and it ships no golden outputs (media/ofile.wav is empty).  What IS pinned
 * by data the reference itself holds:
 *   - the ring sizes 56+60+72+72+72+72+72+60+56+45+36+24+12+1 = 710
 *     (hrtf_signals.cu:10, Universal.cuh:4) -> azimuth_offset[];
 *   - the 368 compact KEMAR HRIR files (Jefferson/compact), from which the
 *     710x2 table is rebuilt (tests/golden/kemar_*.npy);
 *   - the test scenarios and tolerances of precision_test.cu.
 * The FFT arithmetic lives in un-vendored FFTW3f / cuFFT (CUDA 10.1), so the
 * float pipeline's numeric outputs are "parity unpinned": they are anchored
 * to a float64 model of the same formulas (oracle/model64.py) within the
 * reference's own CPU-vs-GPU tolerance (2e-7 abs, precision_test.cu:2158).
 *
 * Variant choices where the reference's CPU and CUDA paths differ
 * (SURVEY.md App. C): case predicate and (xH, xw, xD) operand order follow
 * the CUDA path (GPUSoundSource.cu:267-305); the term sum is in the fixed
 * order ((t0+t1)+t2)+t3 of the CPU path (CPUSoundSource.cpp:244-253).
 */
#ifndef JF_ORACLE_H
#define JF_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define JFO_NUM_ELEV 14
#define JFO_NUM_HRTF 710

typedef struct jfo_engine jfo_engine;

/* hrtf_signals.cu:119-140 -- ring offsets from the float-accumulated loop. */
void jfo_azimuth_offsets(int off[JFO_NUM_ELEV + 1]);
/* hrtf_signals.cu:119-124 -- (elevation, (int)round(azi)) of table row j. */
void jfo_table_positions(int ele[JFO_NUM_HRTF], int azi[JFO_NUM_HRTF]);
/* hrtf_signals.cu:20-51 */
int jfo_pick_hrtf(float obj_ele, float obj_azi);
/* SoundSource.cu:65-105; returns -1 if the elevation ring does not exist
 * (the reference reads an uninitialised deltaTheta there). */
int jfo_interp(float ele, float azi, int idx[4], float omegas[6]);
/* the corrected rule behind the drop-in's JF_FLAG_CORRECTED_INTERPOLATION (not in the reference) */
int jfo_interp_corrected(float ele, float azi, int idx[4], float omegas[6]);
/* The corrected rule for ANY grid of elevation rings (the drop-in's jf_engine_create_grid; the reference hard-codes KEMAR's
 * rings, hrtf_signals.cu:7-12, and lists other HRTF sets as a TODO, FuturePlans.md:21): ring r at ring_ele[r] degrees
 * (ascending) with ring_count[r] measurements at azimuths i * ring_step[r] (ring_step NULL: 360 / count); rows ring by ring.
 * Elevations outside the grid are clamped to its first / last ring; with KEMAR's grid and steps this IS jfo_interp_corrected
 * (tested bit for bit).  jfo_grid_pick: the nearest ring's nearest measurement (what FD_BASIC plays on such a grid). */
#define JFO_MAX_RINGS 40
int jfo_grid_rows(int n_rings, const int *ring_count);
int jfo_grid_interp(int n_rings, const float *ring_ele, const int *ring_count, const float *ring_step,
                    float ele, float azi, int idx[4], float omegas[6]);
int jfo_grid_pick(int n_rings, const float *ring_ele, const int *ring_count, const float *ring_step, float ele, float azi);
/* KEMAR as such a grid: elevations, counts and the reference's rounded steps (hrtf_signals.cu:7-10) */
void jfo_kemar_grid(float ring_ele[JFO_NUM_ELEV], int ring_count[JFO_NUM_ELEV], float ring_step[JFO_NUM_ELEV]);
/* GPUSoundSource.cu:301-316 predicate -> 1..4 */
int jfo_case(const int idx[4]);
/* Flatten (idx, omegas) into <=4 (row, weight) terms in accumulation order
 * (GPUSoundSource.cu:118-292); returns the number of terms. */
int jfo_terms(const int idx[4], const float omegas[6], int rows[4], float w[4]);
/* SoundSource.cu:41-54: out = {ele, azi, x, y, z} */
void jfo_from_spherical(float ele, float azi, float r, float out[5]);
/* SoundSource.cu:20-36: out = {ele, azi, r}; returns -1 when r == 0 */
int jfo_from_cartesian(float x, float y, float z, float out[3]);
/* GPUSoundSource.cu:81-95 + kernels.cu:116-125: D[k] (re,im) k<nc. */
void jfo_distance_factor(float x, float y, float z, int nc, float *D);
/* hrtf_signals.cu:107-153: unnormalised r2c of each zero-padded HRIR.
 * hrir [n][2][taps] -> table [n][2][N/2+1][2]. */
void jfo_build_table(const float *hrir, int n_hrtf, int taps, int N, float *table);

/* Unnormalised forward r2c / inverse c2r of length N (power of two) with the
 * oracle's own float32 FFT -- exposed so tests can check it against numpy. */
void jfo_rfft(const float *x, int N, float *X /* (N/2+1)*2 */);
void jfo_irfft(const float *X, int N, float *y /* N */);

/* Engine = Data + sources (DataTag.cuh:9-17, SoundSource.cuh, CPUSoundSource.h). */
jfo_engine *jfo_create(int frames_per_buffer, int hrtf_len, int n_sources,
                       const float *hrir /* [710][2][taps] */, int taps);
/* the same engine on a grid of its own: hrir [jfo_grid_rows][2][taps]; the grid's rule and pick replace KEMAR's in every
 * mode (mode bit 1 is then implied) */
jfo_engine *jfo_create_grid(int frames_per_buffer, int hrtf_len, int n_sources, int n_rings, const float *ring_ele,
                            const int *ring_count, const float *ring_step, const float *hrir, int taps);
void jfo_destroy(jfo_engine *e);
int jfo_pad_len(const jfo_engine *e);
/* Data::type (DataTag.cuh:16): 0 = *_FD_COMPLEX, 1 = *_FD_BASIC (CPUSoundSource.cpp:113-142) */
void jfo_set_mode(jfo_engine *e, int mode); /* bit 0: FD_BASIC; bit 1: corrected index/weight rule */
/* cudaPart.cu:198-199 (buf/length); the engine copies. */
int jfo_source_set_signal(jfo_engine *e, int s, const float *mono, int n);
int jfo_source_set_spherical(jfo_engine *e, int s, float ele, float azi, float r);
int jfo_source_set_cartesian(jfo_engine *e, int s, float x, float y, float z);
/* precision_test.cu:2097-2107: zero window, count = 0, old = (0,0). */
void jfo_source_reset(jfo_engine *e, int s);
/* Audio.cu:94-163 with CPU-path timing (zero latency): out[2*B] overwritten. */
void jfo_process_block(jfo_engine *e, float *out);
/* Last per-source stereo block (2*B floats) of source s, for stage tests. */
const float *jfo_source_last_block(const jfo_engine *e, int s);
/*
 * Batch form used for the timed CPU baseline and large parity cases:
 * n_blocks blocks for every source, positions given per (source, block) as
 * latched values {ele, azi, x, y, z} (pos[(b*n_sources + s)*5 ..]); sources are
 * processed in parallel with OpenMP (n_threads <= 0 -> all), each into its own
 * partial, then mixed in source order.  out_mix [n_blocks][2*B];
 * out_partial (may be NULL) [n_sources][n_blocks][2*B].
 */
void jfo_process_batch(jfo_engine *e, int n_blocks, const float *pos,
                       float *out_mix, float *out_partial, int n_threads);
int jfo_num_threads(void);

/*
 * Convolution reverb ahead of the spatialiser -- the intent of the reference's offline cudaFFT
 * (cudaPart.cu:65-205; disabled there and called with swapped arguments).
 * jfo_reverb_set_ir: stream form.  From the next block on every source's dry signal is convolved with
 *   gain * ir before it enters the spatialiser's window (uniformly partitioned overlap-save, partitions of
 *   frames_per_buffer taps, float32); n_ir = 0 switches the stage off.  Resets every source.  Returns -1
 *   for a block size that is not a power of two.
 * jfo_reverb_offline: the reference's whole-signal form -- out[jfo_reverb_padded_size(n, n_ir)] = the
 *   circular convolution of the zero-padded signal and impulse response (cudaPart.cu:87-153) scaled by
 *   rms(x) / rms(x (*) ir) (:118,161-165); returns that gain.  The product path's counterpart of the gain
 *   is jf_reverb_rms_gain.
 */
int jfo_reverb_set_ir(jfo_engine *e, const float *ir, int n_ir, float gain);
int jfo_reverb_padded_size(int n, int n_ir);
float jfo_reverb_offline(const float *x, int n, const float *ir, int n_ir, float *out);

#ifdef __cplusplus
}
#endif
#endif
