/*
 * jefferson_debug.h -- everything of libjefferson_hip.so that is NOT the drop-in boundary: parity taps for the tests,
 * timing hooks for bench.py, tuning switches of the A/B scripts under profiles/, and the accessors through which HIP-aware
 * code of this repository (jf_group.c's reduce over the GPUs of a node, bench.py's RCCL leg) reaches the engine's stream
 * and device buffers.  Same library, same symbols as before round 6; a host that binds the reference's interface
 * (callback_func Audio.cu:94-175, the SoundSource setters SoundSource.cuh:19-22, readFile / the output
 * file cudaPart.cu:21-63) includes jefferson.h only and never sees any of this.  None of these calls is needed for correct
 * results: every switch defaults to the shipped behaviour, and nothing here is read from the environment.
 */
#ifndef JEFFERSON_DEBUG_H
#define JEFFERSON_DEBUG_H

#include "jefferson.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- interop with HIP code of this repository ---------------------------------------------------------------------- */

/* Device pointers owned by the engine (valid until destroy). */
float *jf_batch_mix_device(jf_engine *e);     /* [max_batch_blocks][2*B] */
float *jf_batch_partial_device(jf_engine *e); /* [blocks][n_sources / G][2*B]: stereo blocks of the last run, summed over
                                                 groups of G consecutive sources (jf_debug_set_source_group) */
/* hipStream_t the engine launches on, as void*. */
void *jf_engine_stream(jf_engine *e);

/* ---- timing hooks (bench.py) ---------------------------------------------------------------------------------------- */

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

/* ---- HDF5 reader taps (tests/test_sofa.py) -------------------------------------------------------------------------- */

/* tests: a numeric dataset of any HDF5 file the reader understands, as doubles (malloc'd: jf_free); dims[8] */
int jf_debug_hdf5_read(const char *path, const char *dataset, double **out, int *rank, unsigned long long *dims);
/* tests: a string attribute of an object ("" or "/": the root group); JF_ERR_ARG if there is none */
int jf_debug_hdf5_attr(const char *path, const char *object, const char *attr, char *out, size_t cap);

/* ---- parity taps and tuning switches -------------------------------------------------------------------------------- */

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
 * before their result is first read (jf_engine_reverb.cpp: run_reverb_stage).
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
/* Workgroups of the product kernel on the reverb's side stream (jf_engine_reverb.cpp: submit_side; default: three quarters of the
 * device's compute units, one workgroup of 8 waves each, so that the kernels of the blocks it runs beside keep a quarter of them to
 * themselves: profiles/r06/reverb_realtime.md).  8 .. 65536; tuning runs
 * only (profiles/rt_ab.sh; was the environment variable JF_RV_SIDE_WGS until round 6). */
int jf_debug_set_reverb_side_workgroups(jf_engine *e, int workgroups);

/* ---- libjefferson_group.so (jefferson_group.h): test support ------------------------------------------------------------ */

typedef struct jf_group jf_group;
/*
 * Test support: what of the several-GPU host code a one-GPU box can exercise with MORE THAN ONE shard.
 * jf_group_create_shards_on_device: n_shards engines, all on `device`, no RCCL communicator (RCCL refuses duplicate
 * devices); batch runs leave every shard's mix in its buffer and jf_group_batch_fetch adds them on the host in shard order
 * -- the sharding, the per-shard repack of the trajectory, the routing of the per-source calls, the job-wide controls and the
 * FAILED transitions are the production code, only the wire is replaced.
 * jf_group_debug_fail_next: the next processing step or control call that reaches shard `shard` fails with JF_ERR_DEVICE
 * without touching the engine (-1 disarms).
 */
int jf_group_create_shards_on_device(const jf_config *cfg, int n_shards, int device, const float *hrir, int taps,
                                     jf_group **out);
int jf_group_debug_fail_next(jf_group *g, int shard);

#ifdef __cplusplus
}
#endif
#endif /* JEFFERSON_DEBUG_H */
