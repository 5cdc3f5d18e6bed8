/*
 * jefferson_group.h -- one HRTF convolution job over the GPUs of a node, from a C host.
 *
 * The reference runs ONE process that mixes its sources in a C loop (Jefferson/src/Audio.cu:109-110:
 * `output[i] += source->intermediate[i]` over all sources).  Sources are independent until that sum
 * (SURVEY.md 8e), so the multi-GPU form is: one jf_engine (jefferson.h) per GPU, each holding a contiguous
 * range of the sources with its own copy of the HRTF table, no data-path collective, and ONE exchange -- the sum
 * of the per-GPU stereo mixes.  This library does that from a single host process, in plain C on top of the C ABI
 * of jefferson.h, the HIP runtime and RCCL (ncclCommInitAll + ncclReduce on the engines' own streams); no
 * Python, no torch.  bench.py's N-process form (one rank per GPU, torch.distributed) shards the same way.
 *
 * Implemented by libjefferson_group.so (jefferson-2.0_amd/csrc/jf_group.c), which links libjefferson_hip.so
 * and librccl.so.  Source indices below are GLOBAL (0 .. n_sources - 1 of the whole job).
 */
#ifndef JEFFERSON_GROUP_H
#define JEFFERSON_GROUP_H

#include <stddef.h>

#include "jefferson.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct jf_group jf_group;

/* Contiguous source range [*lo, *hi) of part `part` of `n_parts`: sizes differ by at most one (the partition
 * bench.py and the tests use).  Returns JF_ERR_ARG for n_parts < 1 or part outside 0 .. n_parts - 1. */
int jf_shard_range(int n_total, int n_parts, int part, int *lo, int *hi);

/*
 * `new GPUSoundSource[num_sources]` + read_hrtf_signals() + transform_hrtfs() (main.cu:60-75) for n_gpus GPUs:
 * cfg->n_sources is the TOTAL number of sources, cfg->device is ignored; devices[n_gpus] are HIP device ordinals
 * (NULL: 0 .. n_gpus - 1).  n_gpus may not exceed n_sources.  Creates the engines and one RCCL communicator per GPU.
 */
int jf_group_create(const jf_config *cfg, int n_gpus, const int *devices, const float *hrir, int taps, jf_group **out);
/* The same for an HRTF set on a grid of elevation rings of its own (jefferson.h: jf_engine_create_grid; the table is replicated
 * on every GPU like KEMAR's): hrir [jf_grid_rows(grid)][2][taps]. */
int jf_group_create_grid(const jf_config *cfg, int n_gpus, const int *devices, const jf_hrtf_grid *grid, const float *hrir,
                         int taps, jf_group **out);
/* ... and for a set in a SOFA file (jefferson.h: jf_engine_create_sofa: read once on the host, the table replicated) */
int jf_group_create_sofa(const jf_config *cfg, int n_gpus, const int *devices, const char *path, float tol_deg, jf_group **out);
void jf_group_destroy(jf_group *g);
/* Text of the last error on this group (or of the last failed create when g == NULL). */
const char *jf_group_last_error(const jf_group *g);

int jf_group_num_gpus(const jf_group *g);
int jf_group_num_sources(const jf_group *g);
/* The engine of shard i and the first global source index it holds (for anything jefferson.h offers per engine). */
jf_engine *jf_group_engine(jf_group *g, int i);
int jf_group_first_source(const jf_group *g, int i);

/* jf_source_set_signal / _spherical / _cartesian of jefferson.h, routed to the engine that holds `src`. */
int jf_group_source_set_signal(jf_group *g, int src, const float *mono, size_t n);
int jf_group_source_set_spherical(jf_group *g, int src, float ele, float azi, float r);
int jf_group_source_set_cartesian(jf_group *g, int src, float x, float y, float z);

/* precision_test.cu:2097-2107 (jf_source_reset): window, play position and old position of global source `src`. */
int jf_group_source_reset(jf_group *g, int src);

/* Data::type (DataTag.cuh:16, read at every block, Audio.cu:104) and Data::pauseStatus (DataTag.cuh:15, Audio.cu:101)
 * of the one job: forwarded to every engine (jf_set_mode / jf_set_pause), in effect from the next block on. */
int jf_group_set_mode(jf_group *g, int mode);
int jf_group_set_pause(jf_group *g, int paused);

/* The convolution reverb stage ahead of the spatialiser (jf_reverb_set_ir; the reference's offline cudaFFT,
 * cudaPart.cu:65-205) on every engine: the impulse response is one per job, every GPU convolves its own sources
 * (the delay lines shard with the sources, nothing is exchanged).  n_ir = 0 switches the stage off.  Resets every
 * source's state, like jf_reverb_set_ir. */
int jf_group_reverb_set_ir(jf_group *g, const float *ir, size_t n_ir, float gain);

/* max |sample| of the last block jf_group_process_block handed out -- the clip alert of callback_func
 * (Audio.cu:111-113) is taken on the SUM over the sources, i.e. here on the sum over the GPUs. */
float jf_group_last_block_peak(const jf_group *g);

/*
 * callback_func with the CPU path's timing (Audio.cu:118-158) over all GPUs: every engine is handed its block at
 * once (jf_submit_block), then the blocks are collected and added in shard order ON THE HOST -- 2 * frames_per_buffer
 * floats per GPU, the reference's own loop.  out: 2 * frames_per_buffer floats.
 */
int jf_group_process_block(jf_group *g, float *out);

/*
 * n_blocks consecutive callbacks: positions [n_blocks][n_sources][JF_POS_FLOATS] (global source order),
 * out_mix [n_blocks][2 * frames_per_buffer] on the host.  Every engine processes its shard with no
 * host <-> device traffic besides its positions; the per-GPU mixes [n][2B] are summed by RCCL
 * (ncclReduce(sum, float32) to the first GPU, enqueued on each engine's stream behind its kernels), and
 * the first GPU's result is copied out.  n_blocks may exceed max_batch_blocks (processed in runs).  Afterwards the sources
 * stand where the last callback read them (jefferson.h: jf_process_batch).
 */
int jf_group_process_batch(jf_group *g, int n_blocks, const float *positions, float *out_mix);

/* The device-resident form (jf_batch_upload_positions / jf_batch_run of jefferson.h): upload once, then run windows
 * of the trajectory; jf_group_batch_run returns without waiting, jf_group_batch_fetch waits for the reduce of the
 * last run and copies its n_blocks x 2B floats to the host.  One run may be in flight. */
int jf_group_batch_upload_positions(jf_group *g, int total_blocks, const float *positions);
int jf_group_batch_run(jf_group *g, int first_block, int n_blocks);
int jf_group_batch_fetch(jf_group *g, float *out_mix);
/* Waits for everything enqueued on every engine's stream. */
int jf_group_synchronize(jf_group *g);

/*
 * Failure semantics.  The shards advance together: if a call that moves audio state (process_block, batch_run,
 * batch_fetch, process_batch) fails on one GPU after others have already advanced, the shards are out of step and the
 * group is marked FAILED: every later processing call returns JF_ERR_STATE ("group failed") until the job is rebuilt
 * (jf_group_destroy + jf_group_create).  Setters and getters keep working so that the host can read
 * jf_group_last_error / jf_last_error(jf_group_engine(g, i)).  jf_group_set_mode / _set_pause that fail on engine i > 0 set
 * engines 0 .. i - 1 back, so the shards never render different algorithms.  Several GPUs (n_gpus > 1) have only run on a
 * one-GPU box as a communicator of size 1 and as several shards on the one device (below) so far.
 */
int jf_group_failed(const jf_group *g);

/* (Test support -- several shards of a job on ONE device, forced failures -- is declared in jefferson_debug.h.) */

#ifdef __cplusplus
}
#endif
#endif /* JEFFERSON_GROUP_H */
