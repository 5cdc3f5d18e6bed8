/* jf_group.c -- include/jefferson_group.h: one job over the GPUs of a node from a single C host process.
 * Plain C over the C ABI of jefferson.h, the HIP runtime API and RCCL.  One jf_engine per GPU, sources in contiguous
 * shards, no data-path collective; the only exchange is the sum of the per-GPU stereo mixes (Audio.cu:109-110). */
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/jefferson_group.h"
#include "../../include/jefferson_debug.h" /* the engines' streams: the reduce is enqueued behind their kernels */

struct jf_group {
    int n;         /* GPUs */
    int S, B, maxK;
    int *dev;      /* [n] HIP device ordinals */
    int *lo;       /* [n + 1] first global source of each shard */
    jf_engine **eng;
    ncclComm_t *comm;
    float **d_red; /* [n] device buffers [maxK][2B]: each engine's mix, reduced in place into d_red[0] */
    float *blk;    /* [2B] host scratch of jf_group_process_block */
    int traj_blocks;
    int run_blocks; /* > 0: a batch run is in flight (not yet fetched) */
    int failed;     /* a processing call failed part-way: the shards are out of step (jefferson_group.h) */
    int mode, paused; /* what jf_group_set_mode / _set_pause last set on every engine (for the roll-back of a partial failure) */
    int host_sum;   /* jf_group_create_shards_on_device: all shards on ONE device, no communicator; the mixes are added on the host */
    float *hmix;    /* host_sum: [maxK][2B] scratch of jf_group_batch_fetch */
    int inject;     /* jf_group_debug_fail_next: shard + 1 whose next processing step is made to fail (0: none) */
    float last_peak; /* max |sample| of the last block handed out by jf_group_process_block */
    char err[256];
};

static _Thread_local char g_create_err[256];

static int fail(jf_group *g, int code, const char *what, const char *detail) {
    char *dst = g ? g->err : g_create_err;
    snprintf(dst, 256, "%s%s%s", what, detail ? ": " : "", detail ? detail : "");
    return code;
}

#define JG_HIP(g, call)                                                              \
    do {                                                                             \
        hipError_t s_ = (call);                                                      \
        if (s_ != hipSuccess) return fail((g), JF_ERR_DEVICE, #call, hipGetErrorString(s_)); \
    } while (0)
#define JG_NCCL(g, call)                                                              \
    do {                                                                              \
        ncclResult_t s_ = (call);                                                     \
        if (s_ != ncclSuccess) return fail((g), JF_ERR_DEVICE, #call, ncclGetErrorString(s_)); \
    } while (0)
#define JG_ENG(g, i, call)                                                       \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != JF_OK) return fail((g), rc_, #call, jf_last_error((g)->eng[i])); \
    } while (0)

int jf_shard_range(int n_total, int n_parts, int part, int *lo, int *hi) {
    if (n_total < 0 || n_parts < 1 || part < 0 || part >= n_parts || !lo || !hi) return JF_ERR_ARG;
    const int base = n_total / n_parts, rem = n_total % n_parts;
    *lo = part * base + (part < rem ? part : rem);
    *hi = *lo + base + (part < rem ? 1 : 0);
    return JF_OK;
}

/* shard that holds global source `src` */
static int shard_of(const jf_group *g, int src) {
    for (int i = 0; i < g->n; i++)
        if (src >= g->lo[i] && src < g->lo[i + 1]) return i;
    return -1;
}

void jf_group_destroy(jf_group *g) {
    if (!g) return;
    int prev_dev = -1;
    (void)hipGetDevice(&prev_dev);
    for (int i = 0; i < g->n; i++) {
        if (g->eng && g->eng[i]) (void)jf_synchronize(g->eng[i]);
        if (g->comm && g->comm[i]) (void)ncclCommDestroy(g->comm[i]);
        if (g->d_red && g->d_red[i]) {
            (void)hipSetDevice(g->dev[i]);
            (void)hipFree(g->d_red[i]);
        }
        if (g->eng && g->eng[i]) jf_engine_destroy(g->eng[i]);
    }
    free(g->dev);
    free(g->lo);
    free(g->eng);
    free(g->comm);
    free(g->d_red);
    free(g->blk);
    free(g->hmix);
    free(g);
    if (prev_dev >= 0) (void)hipSetDevice(prev_dev);
}

/* one_device >= 0: every shard on that device, no communicator (jf_group_create_shards_on_device) */
/* grid: the HRTF set's own grid of elevation rings (jf_engine_create_grid), or NULL for the reference's KEMAR grid */
static int create_group(const jf_config *cfg, int n_gpus, const int *devices, int one_device, const jf_hrtf_grid *grid,
                        const float *hrir, int taps, jf_group **out) {
    if (!cfg || !hrir || !out) return fail(NULL, JF_ERR_ARG, "null argument", NULL);
    *out = NULL;
    if (n_gpus < 1 || n_gpus > cfg->n_sources) return fail(NULL, JF_ERR_ARG, "need 1 <= n_gpus <= n_sources", NULL);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(NULL, JF_ERR_DEVICE, "no HIP device available (this library has no CPU path)", NULL);
    jf_group *g = (jf_group *)calloc(1, sizeof(*g));
    if (!g) return fail(NULL, JF_ERR_NOMEM, "out of host memory", NULL);
    int prev_dev = -1;
    (void)hipGetDevice(&prev_dev); /* the caller's current device is left as it was */
    g->n = n_gpus;
    g->S = cfg->n_sources;
    g->B = cfg->frames_per_buffer;
    g->maxK = cfg->max_batch_blocks;
    g->mode = JF_MODE_FD_COMPLEX;
    g->host_sum = one_device >= 0;
    g->dev = (int *)calloc(n_gpus, sizeof(int));
    g->lo = (int *)calloc(n_gpus + 1, sizeof(int));
    g->eng = (jf_engine **)calloc(n_gpus, sizeof(*g->eng));
    g->comm = (ncclComm_t *)calloc(n_gpus, sizeof(*g->comm));
    g->d_red = (float **)calloc(n_gpus, sizeof(*g->d_red));
    g->blk = (float *)calloc(2 * (size_t)(g->B > 0 ? g->B : 1), sizeof(float));
    if (g->host_sum) g->hmix = (float *)calloc(2 * (size_t)(g->B > 0 ? g->B : 1) * (size_t)(g->maxK > 0 ? g->maxK : 1), sizeof(float));
    int rc = JF_OK;
    if (!g->dev || !g->lo || !g->eng || !g->comm || !g->d_red || !g->blk || (g->host_sum && !g->hmix))
        rc = fail(NULL, JF_ERR_NOMEM, "out of host memory", NULL);
    for (int i = 0; rc == JF_OK && i < n_gpus; i++) {
        g->dev[i] = g->host_sum ? one_device : (devices ? devices[i] : i);
        if (g->dev[i] < 0 || g->dev[i] >= ndev) rc = fail(NULL, JF_ERR_ARG, "device ordinal out of range", NULL);
        for (int k = 0; rc == JF_OK && !g->host_sum && k < i; k++)
            if (g->dev[k] == g->dev[i]) rc = fail(NULL, JF_ERR_ARG, "a device is listed twice (RCCL needs distinct devices)", NULL);
    }
    for (int i = 0; rc == JF_OK && i < n_gpus; i++) {
        int lo, hi;
        jf_shard_range(g->S, n_gpus, i, &lo, &hi);
        g->lo[i] = lo;
        g->lo[i + 1] = hi;
        jf_config c = *cfg;
        c.n_sources = hi - lo;
        c.device = g->dev[i];
        rc = grid ? jf_engine_create_grid(&c, grid, hrir, taps, &g->eng[i]) : jf_engine_create(&c, hrir, taps, &g->eng[i]);
        if (rc != JF_OK) {
            fail(NULL, rc, grid ? "jf_engine_create_grid" : "jf_engine_create", jf_last_error(NULL));
            break;
        }
        if (hipSetDevice(g->dev[i]) != hipSuccess ||
            hipMalloc((void **)&g->d_red[i], sizeof(float) * 2 * (size_t)g->B * (size_t)g->maxK) != hipSuccess)
            rc = fail(NULL, JF_ERR_DEVICE, "hipMalloc of the reduce buffer failed", NULL);
    }
    if (rc == JF_OK && !g->host_sum) {
        ncclResult_t s = ncclCommInitAll(g->comm, n_gpus, g->dev);
        if (s != ncclSuccess) rc = fail(NULL, JF_ERR_DEVICE, "ncclCommInitAll", ncclGetErrorString(s));
    }
    if (rc != JF_OK) {
        jf_group_destroy(g);
        if (prev_dev >= 0) (void)hipSetDevice(prev_dev);
        return rc;
    }
    if (prev_dev >= 0) (void)hipSetDevice(prev_dev);
    *out = g;
    return JF_OK;
}

int jf_group_create(const jf_config *cfg, int n_gpus, const int *devices, const float *hrir, int taps, jf_group **out) {
    return create_group(cfg, n_gpus, devices, -1, NULL, hrir, taps, out);
}

int jf_group_create_grid(const jf_config *cfg, int n_gpus, const int *devices, const jf_hrtf_grid *grid, const float *hrir,
                         int taps, jf_group **out) {
    if (!grid) return fail(NULL, JF_ERR_ARG, "null grid", NULL);
    return create_group(cfg, n_gpus, devices, -1, grid, hrir, taps, out);
}

int jf_group_create_sofa(const jf_config *cfg, int n_gpus, const int *devices, const char *path, float tol_deg, jf_group **out) {
    if (out) *out = NULL;
    if (!cfg || !path || !out) return fail(NULL, JF_ERR_ARG, "null argument", NULL);
    jf_sofa_set set;
    int rc = jf_sofa_read(path, &set);
    if (rc != JF_OK) return fail(NULL, rc, path, jf_last_error(NULL));
    const int taps = jf_sofa_taps(&set);
    float *hrir = taps > 0 ? (float *)malloc(sizeof(float) * (size_t)set.n_measurements * 2 * (size_t)taps) : NULL;
    jf_grid_layout lay;
    if (taps < 0) rc = fail(NULL, taps, path, jf_last_error(NULL));
    else if (!hrir) rc = fail(NULL, JF_ERR_NOMEM, "out of host memory", NULL);
    else if ((rc = jf_sofa_table(&set, tol_deg, &lay, hrir, taps)) != JF_OK) rc = fail(NULL, rc, path, jf_last_error(NULL));
    if (rc == JF_OK) {
        const jf_hrtf_grid grid = {lay.n_rings, lay.ring_elevation, lay.ring_count, lay.ring_step};
        rc = create_group(cfg, n_gpus, devices, -1, &grid, hrir, taps, out);
    }
    free(hrir);
    jf_sofa_release(&set);
    return rc;
}

int jf_group_create_shards_on_device(const jf_config *cfg, int n_shards, int device, const float *hrir, int taps,
                                     jf_group **out) {
    if (device < 0) return fail(NULL, JF_ERR_ARG, "device ordinal out of range", NULL);
    return create_group(cfg, n_shards, NULL, device, NULL, hrir, taps, out);
}

int jf_group_debug_fail_next(jf_group *g, int shard) {
    if (!g || shard < -1 || shard >= g->n) return JF_ERR_ARG;
    g->inject = shard + 1;
    return JF_OK;
}

/* jf_group_debug_fail_next: 1 once for the armed shard */
static int injected(jf_group *g, int i) {
    if (g->inject != i + 1) return 0;
    g->inject = 0;
    snprintf(g->err, sizeof(g->err), "injected failure on shard %d (jf_group_debug_fail_next)", i);
    return 1;
}

const char *jf_group_last_error(const jf_group *g) { return g ? g->err : g_create_err; }
int jf_group_num_gpus(const jf_group *g) { return g ? g->n : JF_ERR_ARG; }
int jf_group_num_sources(const jf_group *g) { return g ? g->S : JF_ERR_ARG; }
jf_engine *jf_group_engine(jf_group *g, int i) { return (g && i >= 0 && i < g->n) ? g->eng[i] : NULL; }
int jf_group_first_source(const jf_group *g, int i) { return (g && i >= 0 && i < g->n) ? g->lo[i] : JF_ERR_ARG; }

int jf_group_source_set_signal(jf_group *g, int src, const float *mono, size_t n) {
    const int i = g ? shard_of(g, src) : -1;
    if (i < 0) return fail(g, JF_ERR_ARG, "bad source index", NULL);
    JG_ENG(g, i, jf_source_set_signal(g->eng[i], src - g->lo[i], mono, n));
    return JF_OK;
}

int jf_group_source_set_spherical(jf_group *g, int src, float ele, float azi, float r) {
    const int i = g ? shard_of(g, src) : -1;
    if (i < 0) return fail(g, JF_ERR_ARG, "bad source index", NULL);
    JG_ENG(g, i, jf_source_set_spherical(g->eng[i], src - g->lo[i], ele, azi, r));
    return JF_OK;
}

int jf_group_source_set_cartesian(jf_group *g, int src, float x, float y, float z) {
    const int i = g ? shard_of(g, src) : -1;
    if (i < 0) return fail(g, JF_ERR_ARG, "bad source index", NULL);
    JG_ENG(g, i, jf_source_set_cartesian(g->eng[i], src - g->lo[i], x, y, z));
    return JF_OK;
}

/* a processing step on shard i: a failure once any shard may have advanced marks the group failed */
#define JG_STEP(g, i, call)                                                          \
    do {                                                                             \
        if (injected((g), (i))) {                                                    \
            (g)->failed = 1;                                                         \
            return JF_ERR_DEVICE;                                                    \
        }                                                                            \
        int rc_ = (call);                                                            \
        if (rc_ != JF_OK) {                                                          \
            (g)->failed = 1;                                                         \
            return fail((g), rc_, #call, jf_last_error((g)->eng[i]));                \
        }                                                                            \
    } while (0)
#define JG_ALIVE(g) \
    if ((g)->failed) return fail_keep((g), JF_ERR_STATE, "group failed: the shards are out of step; destroy and re-create it")

static int fail_keep(jf_group *g, int code, const char *what) { /* keeps the text of the failure that caused it */
    if (!strstr(g->err, "group failed")) {
        char first[128];
        memcpy(first, g->err, sizeof(first) - 1);
        first[sizeof(first) - 1] = 0;
        snprintf(g->err, 256, "%s (first failure: %s)", what, first);
    }
    return code;
}

int jf_group_failed(const jf_group *g) { return g ? g->failed : JF_ERR_ARG; }
float jf_group_last_block_peak(const jf_group *g) { return g ? g->last_peak : 0.0f; }

int jf_group_source_reset(jf_group *g, int src) {
    const int i = g ? shard_of(g, src) : -1;
    if (i < 0) return fail(g, JF_ERR_ARG, "bad source index", NULL);
    if (g->run_blocks) return fail(g, JF_ERR_STATE, "a batch run is in flight (jf_group_batch_fetch first)", NULL);
    JG_ENG(g, i, jf_source_reset(g->eng[i], src - g->lo[i]));
    return JF_OK;
}

/* Forwarded to every engine; if engine i > 0 refuses, engines 0 .. i - 1 are set back to what they had, so the shards never
 * render different algorithms -- and if even that fails the group is marked failed. */
int jf_group_set_mode(jf_group *g, int mode) {
    if (!g) return JF_ERR_ARG;
    if (mode != JF_MODE_FD_COMPLEX && mode != JF_MODE_FD_BASIC) return fail(g, JF_ERR_ARG, "unknown mode", NULL);
    for (int i = 0; i < g->n; i++) {
        const int rc = injected(g, i) ? JF_ERR_DEVICE : jf_set_mode(g->eng[i], mode);
        if (rc != JF_OK) {
            for (int k = 0; k < i; k++)
                if (jf_set_mode(g->eng[k], g->mode) != JF_OK) g->failed = 1;
            return rc;
        }
    }
    g->mode = mode;
    return JF_OK;
}

int jf_group_set_pause(jf_group *g, int paused) {
    if (!g) return JF_ERR_ARG;
    for (int i = 0; i < g->n; i++) {
        const int rc = injected(g, i) ? JF_ERR_DEVICE : jf_set_pause(g->eng[i], paused);
        if (rc != JF_OK) {
            for (int k = 0; k < i; k++)
                if (jf_set_pause(g->eng[k], g->paused) != JF_OK) g->failed = 1;
            return rc;
        }
    }
    g->paused = paused != 0;
    return JF_OK;
}

int jf_group_reverb_set_ir(jf_group *g, const float *ir, size_t n_ir, float gain) {
    if (!g || (n_ir && !ir)) return fail(g, JF_ERR_ARG, "bad impulse response", NULL);
    if (g->run_blocks) return fail(g, JF_ERR_STATE, "a batch run is in flight (jf_group_batch_fetch first)", NULL);
    JG_ALIVE(g);
    /* all or nothing would need a second set of delay lines: a failure part-way (out of device memory on one GPU)
     * leaves engines with and without the stage, i.e. a failed group */
    for (int i = 0; i < g->n; i++) {
        const int rc = jf_reverb_set_ir(g->eng[i], ir, n_ir, gain);
        if (rc != JF_OK) {
            if (i > 0 || rc != JF_ERR_ARG) g->failed = 1;
            return fail(g, rc, "jf_reverb_set_ir", jf_last_error(g->eng[i]));
        }
    }
    return JF_OK;
}

int jf_group_process_block(jf_group *g, float *out) {
    if (!g || !out) return fail(g, JF_ERR_ARG, "null argument", NULL);
    if (g->run_blocks) return fail(g, JF_ERR_STATE, "a batch run is in flight (jf_group_batch_fetch first)", NULL);
    JG_ALIVE(g);
    /* every GPU gets its block before any is waited for */
    for (int i = 0; i < g->n; i++) {
        const int inj = injected(g, i);
        const int rc = inj ? JF_ERR_DEVICE : jf_submit_block(g->eng[i]);
        if (rc != JF_OK) {
            if (i > 0) g->failed = 1; /* shards 0 .. i-1 have a block in flight */
            return inj ? rc : fail(g, rc, "jf_submit_block", jf_last_error(g->eng[i]));
        }
    }
    memset(out, 0, sizeof(float) * 2 * (size_t)g->B);
    for (int i = 0; i < g->n; i++) {
        JG_STEP(g, i, jf_collect_block(g->eng[i], i == 0 ? out : g->blk));
        if (i > 0)
            for (int k = 0; k < 2 * g->B; k++) out[k] += g->blk[k]; /* Audio.cu:109-110, shard by shard */
    }
    float peak = 0.0f;
    for (int k = 0; k < 2 * g->B; k++) {
        const float a = out[k] < 0.0f ? -out[k] : out[k];
        if (a > peak) peak = a;
    }
    g->last_peak = peak; /* Audio.cu:111-113 looks at the summed output */
    return JF_OK;
}

int jf_group_batch_upload_positions(jf_group *g, int total_blocks, const float *positions) {
    if (!g || total_blocks <= 0 || !positions) return fail(g, JF_ERR_ARG, "bad trajectory", NULL);
    if (g->run_blocks) return fail(g, JF_ERR_STATE, "a batch run is in flight (jf_group_batch_fetch first)", NULL);
    if (g->n == 1) { /* the whole trajectory as it stands */
        JG_ENG(g, 0, jf_batch_upload_positions(g->eng[0], total_blocks, positions));
        g->traj_blocks = total_blocks;
        return JF_OK;
    }
    for (int i = 0; i < g->n; i++) {
        const size_t ns = (size_t)(g->lo[i + 1] - g->lo[i]);
        float *p = (float *)malloc(sizeof(float) * JF_POS_FLOATS * ns * (size_t)total_blocks);
        if (!p) return fail(g, JF_ERR_NOMEM, "out of host memory", NULL);
        for (int b = 0; b < total_blocks; b++)
            memcpy(p + (size_t)b * ns * JF_POS_FLOATS, positions + ((size_t)b * g->S + g->lo[i]) * JF_POS_FLOATS,
                   sizeof(float) * JF_POS_FLOATS * ns);
        const int rc = jf_batch_upload_positions(g->eng[i], total_blocks, p);
        free(p);
        if (rc != JF_OK) return fail(g, rc, "jf_batch_upload_positions", jf_last_error(g->eng[i]));
    }
    g->traj_blocks = total_blocks;
    return JF_OK;
}

int jf_group_batch_run(jf_group *g, int first_block, int n_blocks) {
    if (!g) return JF_ERR_ARG;
    if (g->run_blocks) return fail(g, JF_ERR_STATE, "a batch run is in flight (jf_group_batch_fetch first)", NULL);
    if (n_blocks <= 0 || n_blocks > g->maxK) return fail(g, JF_ERR_ARG, "n_blocks exceeds max_batch_blocks", NULL);
    if (first_block < 0 || first_block + n_blocks > g->traj_blocks)
        return fail(g, JF_ERR_ARG, "window outside the uploaded trajectory", NULL);
    JG_ALIVE(g);
    /* every engine's kernels, each on its own stream, its mix into its reduce buffer */
    for (int i = 0; i < g->n; i++) {
        const int inj = injected(g, i);
        const int rc = inj ? JF_ERR_DEVICE : jf_batch_run(g->eng[i], first_block, n_blocks, g->d_red[i]);
        if (rc != JF_OK) {
            if (i > 0) g->failed = 1; /* shards 0 .. i-1 have advanced */
            return inj ? rc : fail(g, rc, "jf_batch_run", jf_last_error(g->eng[i]));
        }
    }
    if (g->host_sum) { /* shards of one device (test form): no communicator, jf_group_batch_fetch adds the mixes */
        g->run_blocks = n_blocks;
        return JF_OK;
    }
    /* the one exchange of the path: sum of the mixes to the first GPU, behind the kernels on the same streams.
     * From here on every shard has advanced: any failure leaves the group failed, and the collective group is always
     * closed (ncclGroupEnd) so that no stream is left with half a collective. */
    const size_t count = (size_t)n_blocks * 2 * (size_t)g->B;
    ncclResult_t s = ncclGroupStart(), s_end;
    for (int i = 0; s == ncclSuccess && i < g->n; i++)
        s = ncclReduce(g->d_red[i], g->d_red[i], count, ncclFloat, ncclSum, 0, g->comm[i],
                       (hipStream_t)jf_engine_stream(g->eng[i]));
    s_end = ncclGroupEnd();
    if (s == ncclSuccess) s = s_end;
    if (s != ncclSuccess) {
        g->failed = 1;
        return fail(g, JF_ERR_DEVICE, "ncclReduce of the mixes", ncclGetErrorString(s));
    }
    g->run_blocks = n_blocks;
    return JF_OK;
}

int jf_group_batch_fetch(jf_group *g, float *out_mix) {
    if (!g || !out_mix) return fail(g, JF_ERR_ARG, "null argument", NULL);
    if (!g->run_blocks) return fail(g, JF_ERR_STATE, "no batch run in flight", NULL);
    JG_ALIVE(g);
    int prev = -1;
    (void)hipGetDevice(&prev);
    JG_HIP(g, hipSetDevice(g->dev[0]));
    hipStream_t st = (hipStream_t)jf_engine_stream(g->eng[0]);
    const size_t n_mix = 2 * (size_t)g->B * (size_t)g->run_blocks;
    hipError_t s = hipMemcpyAsync(out_mix, g->d_red[0], sizeof(float) * n_mix, hipMemcpyDeviceToHost, st);
    if (s == hipSuccess) s = hipStreamSynchronize(st);
    /* shards of one device: the other shards' mixes are added here in shard order (Audio.cu:109-110), where the several-GPU
     * form has had ncclReduce do it on the devices */
    for (int i = 1; g->host_sum && s == hipSuccess && i < g->n; i++) {
        hipStream_t si = (hipStream_t)jf_engine_stream(g->eng[i]);
        s = hipMemcpyAsync(g->hmix, g->d_red[i], sizeof(float) * n_mix, hipMemcpyDeviceToHost, si);
        if (s == hipSuccess) s = hipStreamSynchronize(si);
        for (size_t k = 0; s == hipSuccess && k < n_mix; k++) out_mix[k] += g->hmix[k];
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    g->run_blocks = 0;
    if (s != hipSuccess) {
        g->failed = 1;
        return fail(g, JF_ERR_DEVICE, "copy of the reduced mix", hipGetErrorString(s));
    }
    /* the other GPUs' part of the reduce ends with the root's; their kernels' own errors (the batch kernel's error
     * word: a hand-off that timed out, fatal for that engine) surface here */
    for (int i = 0; i < g->n; i++) JG_STEP(g, i, jf_synchronize(g->eng[i]));
    return JF_OK;
}

int jf_group_process_batch(jf_group *g, int n_blocks, const float *positions, float *out_mix) {
    if (!g || !positions || !out_mix || n_blocks <= 0) return fail(g, JF_ERR_ARG, "bad batch arguments", NULL);
    int rc = jf_group_batch_upload_positions(g, n_blocks, positions);
    if (rc != JF_OK) return rc;
    for (int b0 = 0; b0 < n_blocks; b0 += g->maxK) {
        const int k = n_blocks - b0 < g->maxK ? n_blocks - b0 : g->maxK;
        rc = jf_group_batch_run(g, b0, k);
        if (rc != JF_OK) return rc;
        rc = jf_group_batch_fetch(g, out_mix + (size_t)b0 * 2 * g->B);
        if (rc != JF_OK) return rc;
    }
    /* as jf_process_batch: the sources stand where the last callback read them */
    const float *last = positions + (size_t)(n_blocks - 1) * g->S * JF_POS_FLOATS;
    for (int i = 0; i < g->n; i++) JG_ENG(g, i, jf_sources_set_latched(g->eng[i], last + (size_t)g->lo[i] * JF_POS_FLOATS));
    return JF_OK;
}

int jf_group_synchronize(jf_group *g) {
    if (!g) return JF_ERR_ARG;
    for (int i = 0; i < g->n; i++) JG_ENG(g, i, jf_synchronize(g->eng[i]));
    return JF_OK;
}
