/* jf_ctest.c -- plain-C checks of the boundary, linked only against the C ABI (jefferson.h, jefferson_group.h).
 *
 *   jf_ctest pa      drives jf_pa_callback with PortAudio's argument list (paCallback, Audio.cu:164-175) for 200
 *                    blocks -- positions changed and the stream paused/resumed from the "UI side" in between -- and
 *                    compares every block with jf_callback on a twin engine fed the same calls
 *   jf_ctest group N one job over N GPUs (jefferson_group.h: one engine per GPU, RCCL reduce of the mixes) against
 *                    one engine holding all sources: per-block calls (host sum) and batch calls (ncclReduce); then the
 *                    same with the job-wide controls -- mode switch, reverb stage, pause, source reset, clip peak
 *   jf_ctest shards N N shards of one job on the ONE device (jf_group_create_shards_on_device: everything of the several-GPU
 *                    host code but the wire) against one engine, and the FAILED transitions under forced failures
 *   jf_ctest bench N [steps] the bench workload from a C host: 1024 moving sources per GPU, 256-sample blocks, 128
 *                    blocks per jf_group_batch_run / _fetch; prints source-frames/s (bench.py's metric)
 *
 * Synthetic HRIRs and signals (no files).  Exit code 0 = all comparisons hold; prints one line per check.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../../include/jefferson.h"
#include "../../include/jefferson_group.h"
#include "../../include/jefferson_debug.h" /* `shards N`: several shards on the one device, forced failures */

static unsigned long long rng_state = 88172645463325252ULL;
static float frand(void) { /* xorshift64*, uniform in [-0.5, 0.5) */
    rng_state ^= rng_state >> 12;
    rng_state ^= rng_state << 25;
    rng_state ^= rng_state >> 27;
    return (float)((rng_state * 2685821657736338717ULL) >> 40) / 16777216.0f - 0.5f;
}

static float *make_hrir(int taps) {
    float *h = (float *)malloc(sizeof(float) * JF_NUM_HRTF * 2 * (size_t)taps);
    for (int j = 0; j < JF_NUM_HRTF * 2; j++)
        for (int n = 0; n < taps; n++) h[(size_t)j * taps + n] = 0.3f * frand() * expf(-(float)n / 24.0f);
    return h;
}

static float *make_signal(size_t n) {
    float *s = (float *)malloc(sizeof(float) * n);
    for (size_t i = 0; i < n; i++) s[i] = frand();
    return s;
}

static double max_abs_diff(const float *a, const float *b, size_t n, double *peak) {
    double d = 0, p = 0;
    for (size_t i = 0; i < n; i++) {
        const double e = fabs((double)a[i] - (double)b[i]);
        if (e > d) d = e;
        if (fabs((double)b[i]) > p) p = fabs((double)b[i]);
    }
    if (peak) *peak = p;
    return d;
}

#define CHECK(call)                                                                       \
    do {                                                                                  \
        int rc_ = (call);                                                                 \
        if (rc_ != JF_OK) {                                                               \
            fprintf(stderr, "%s -> %d (%s / %s)\n", #call, rc_, jf_last_error(NULL), jf_group_last_error(NULL)); \
            return 2;                                                                     \
        }                                                                                 \
    } while (0)

static int test_pa(void) {
    enum { B = 256, S = 3, BLOCKS = 200, TAPS = 128 };
    float *hrir = make_hrir(TAPS);
    jf_config cfg = {B, 512, S, 0, 1, 0};
    jf_engine *pa = NULL, *ref = NULL;
    CHECK(jf_engine_create(&cfg, hrir, TAPS, &pa));
    CHECK(jf_engine_create(&cfg, hrir, TAPS, &ref));
    for (int s = 0; s < S; s++) {
        float *sig = make_signal(30000 + 777 * (size_t)s);
        CHECK(jf_source_set_signal(pa, s, sig, 30000 + 777 * (size_t)s));
        CHECK(jf_source_set_signal(ref, s, sig, 30000 + 777 * (size_t)s));
        free(sig);
    }
    float out_pa[2 * B], out_ref[2 * B];
    double worst = 0, peak = 0, pk = 0;
    int silent_blocks = 0;
    for (int k = 0; k < BLOCKS; k++) {
        /* the UI side: positions every 7th block, pause for blocks 50..59 (Data::pauseStatus, Audio.cu:101) */
        if (k % 7 == 0)
            for (int s = 0; s < S; s++) {
                const float azi = (float)((37 * s + 5 * k) % 360), ele = (float)(-20 + 10 * s);
                CHECK(jf_source_set_spherical(pa, s, ele, azi, 0.5f + 0.2f * s));
                CHECK(jf_source_set_spherical(ref, s, ele, azi, 0.5f + 0.2f * s));
            }
        if (k == 50 || k == 60) {
            CHECK(jf_set_pause(pa, k == 50));
            CHECK(jf_set_pause(ref, k == 50));
        }
        memset(out_pa, 0x7f, sizeof(out_pa)); /* must be fully overwritten */
        /* PortAudio's call: (input, output, framesPerBuffer, timeInfo, statusFlags, userData) */
        if (jf_pa_callback(NULL, out_pa, B, NULL, 0, pa) != 0) {
            fprintf(stderr, "jf_pa_callback did not return paContinue\n");
            return 1;
        }
        CHECK(jf_callback(ref, out_ref));
        const double d = max_abs_diff(out_pa, out_ref, 2 * B, &pk);
        if (d > worst) worst = d;
        if (pk > peak) peak = pk;
        if (pk == 0) silent_blocks++;
        if (fabs((double)jf_last_block_peak(pa) - pk) > 0) {
            fprintf(stderr, "jf_last_block_peak %g != %g at block %d\n", jf_last_block_peak(pa), pk, k);
            return 1;
        }
    }
    /* a stream opened with another buffer size gets silence, not garbage */
    memset(out_pa, 0x7f, sizeof(out_pa));
    (void)jf_pa_callback(NULL, out_pa, B / 2, NULL, 0, pa);
    int bad = 0;
    for (int i = 0; i < B; i++) bad += out_pa[i] != 0.0f;
    printf("pa: %d blocks, max |jf_pa_callback - jf_callback| = %g, peak %g, %d silent blocks (1 primed + 10 paused), "
           "wrong-size call -> %s\n", BLOCKS, worst, peak, silent_blocks, bad ? "GARBAGE" : "silence");
    jf_engine_destroy(pa);
    jf_engine_destroy(ref);
    free(hrir);
    return (worst == 0 && peak > 0.01 && silent_blocks == 11 && !bad) ? 0 : 1;
}

static int test_group(int n_gpus) {
    enum { B = 128, S = 40, K = 6, RUNS = 3, TAPS = 128 };
    float *hrir = make_hrir(TAPS);
    jf_config cfg = {B, 512, S, 0, K, 0};
    jf_engine *one = NULL;
    jf_group *grp = NULL;
    CHECK(jf_engine_create(&cfg, hrir, TAPS, &one));
    CHECK(jf_group_create(&cfg, n_gpus, NULL, hrir, TAPS, &grp));
    for (int s = 0; s < S; s++) {
        const size_t n = 9000 + 131 * (size_t)s;
        float *sig = make_signal(n);
        CHECK(jf_source_set_signal(one, s, sig, n));
        CHECK(jf_group_source_set_signal(grp, s, sig, n));
        free(sig);
    }
    float *pos = (float *)malloc(sizeof(float) * JF_POS_FLOATS * S * K * RUNS);
    for (int k = 0; k < K * RUNS; k++)
        for (int s = 0; s < S; s++)
            CHECK(jf_position_from_spherical((float)(-40 + (7 * s) % 121), (float)((37 * s + k) % 360), 0.5f + 0.05f * s,
                                             pos + ((size_t)k * S + s) * JF_POS_FLOATS));
    float *a = (float *)malloc(sizeof(float) * 2 * B * K * RUNS), *b = (float *)malloc(sizeof(float) * 2 * B * K * RUNS);
    CHECK(jf_process_batch(one, K * RUNS, pos, a));
    CHECK(jf_group_process_batch(grp, K * RUNS, pos, b));
    double peak = 0;
    const double d_batch = max_abs_diff(b, a, (size_t)2 * B * K * RUNS, &peak);
    /* per-block calls on top of the carried state */
    double d_block = 0;
    for (int k = 0; k < 4; k++) {
        for (int s = 0; s < S; s++) {
            CHECK(jf_source_set_spherical(one, s, 10.0f, (float)((20 * s + 9 * k) % 360), 1.0f));
            CHECK(jf_group_source_set_spherical(grp, s, 10.0f, (float)((20 * s + 9 * k) % 360), 1.0f));
        }
        CHECK(jf_process_block(one, a));
        CHECK(jf_group_process_block(grp, b));
        const double d = max_abs_diff(b, a, 2 * B, NULL);
        if (d > d_block) d_block = d;
    }
    /* one GPU: the same sums in the same order -> identical; several: one more association of float32 adds */
    const double tol = n_gpus == 1 ? 0.0 : 4e-7 * S;
    printf("group: %d GPU(s), %d sources: batch max diff %g, per-block max diff %g (tolerance %g), peak %g\n",
           jf_group_num_gpus(grp), S, d_batch, d_block, tol, peak);

    /* ---- the job-wide controls (Data::type, Data::pauseStatus, the reverb stage, reset), forwarded to every GPU */
    enum { NIR = 3 * B + 17 };
    float ir[NIR];
    for (int n = 0; n < NIR; n++) ir[n] = 0.2f * frand() * expf(-(float)n / 100.0f);
    CHECK(jf_reverb_set_ir(one, ir, NIR, 0.8f));
    CHECK(jf_group_reverb_set_ir(grp, ir, NIR, 0.8f));
    double d_ctl = 0, pk = 0, peak_ctl = 0, d_peak = 0;
    int silent = 0;
    for (int k = 0; k < 24; k++) {
        if (k == 6 || k == 12) { /* nearest-HRTF mode for blocks 6..11 */
            CHECK(jf_set_mode(one, k == 6 ? JF_MODE_FD_BASIC : JF_MODE_FD_COMPLEX));
            CHECK(jf_group_set_mode(grp, k == 6 ? JF_MODE_FD_BASIC : JF_MODE_FD_COMPLEX));
        }
        if (k == 15 || k == 18) { /* paused for blocks 15..17: silence, nothing consumed */
            CHECK(jf_set_pause(one, k == 15));
            CHECK(jf_group_set_pause(grp, k == 15));
        }
        if (k == 20) { /* sources on both sides of a shard boundary start over */
            CHECK(jf_source_reset(one, 0));
            CHECK(jf_group_source_reset(grp, 0));
            CHECK(jf_source_reset(one, S - 1));
            CHECK(jf_group_source_reset(grp, S - 1));
        }
        for (int s = 0; s < S; s++) {
            CHECK(jf_source_set_spherical(one, s, (float)(-30 + (11 * s) % 100), (float)((20 * s + 7 * k) % 360), 1.0f));
            CHECK(jf_group_source_set_spherical(grp, s, (float)(-30 + (11 * s) % 100), (float)((20 * s + 7 * k) % 360), 1.0f));
        }
        CHECK(jf_process_block(one, a));
        CHECK(jf_group_process_block(grp, b));
        const double d = max_abs_diff(b, a, 2 * B, &pk);
        if (d > d_ctl) d_ctl = d;
        if (pk > peak_ctl) peak_ctl = pk;
        if (pk == 0) silent++;
        /* the clip alert's quantity: on one GPU the engine's own figure, always the peak of what was handed out */
        double pb = 0;
        max_abs_diff(a, b, 2 * B, &pb);
        const double dp = fabs((double)jf_group_last_block_peak(grp) - pb);
        if (dp > d_peak) d_peak = dp;
    }
    const int bad_mode = jf_group_set_mode(grp, 7) != JF_ERR_ARG;
    const int bad_src = jf_group_source_reset(grp, S) != JF_ERR_ARG;
    /* three partitions of float32 sums on top of the spatialiser's tolerance */
    const double tol_ctl = n_gpus == 1 ? 0.0 : 8e-7 * S;
    printf("group controls: reverb + mode switch + pause + reset over 24 blocks: max diff %g (tolerance %g), peak %g, "
           "%d silent blocks (3 paused), clip-peak mismatch %g, failed flag %d\n",
           d_ctl, tol_ctl, peak_ctl, silent, d_peak, jf_group_failed(grp));
    const int ok_ctl = d_ctl <= tol_ctl && peak_ctl > 0.02 && silent == 3 && d_peak == 0 && !bad_mode && !bad_src &&
                       jf_group_failed(grp) == 0;
    jf_group_destroy(grp);
    jf_engine_destroy(one);
    free(pos);
    free(a);
    free(b);
    free(hrir);
    return (d_batch <= tol && d_block <= tol && peak > 0.05 && ok_ctl) ? 0 : 1;
}

/* N > 1 shards on the ONE device (jf_group_create_shards_on_device: the production sharding, repack, routing, controls and
 * failure handling with a host sum where the several-GPU form has ncclReduce) against one engine holding all sources; then
 * forced failures on shard 1: a control call must leave every shard as it was, a processing call must leave the group
 * FAILED -- JF_ERR_STATE from every later processing call, setters and error texts still there, destroy clean. */
static int test_shards(int n_shards) {
    enum { B = 128, S = 37, K = 5, RUNS = 3, TAPS = 128 }; /* 37 sources: shards of unequal size */
    float *hrir = make_hrir(TAPS);
    jf_config cfg = {B, 512, S, 0, K, 0};
    jf_engine *one = NULL;
    jf_group *grp = NULL;
    CHECK(jf_engine_create(&cfg, hrir, TAPS, &one));
    CHECK(jf_group_create_shards_on_device(&cfg, n_shards, 0, hrir, TAPS, &grp));
    int covered = 0;
    for (int i = 0; i < n_shards; i++) {
        int lo, hi;
        CHECK(jf_shard_range(S, n_shards, i, &lo, &hi));
        if (jf_group_first_source(grp, i) != lo || jf_num_sources(jf_group_engine(grp, i)) != hi - lo) return 1;
        covered += hi - lo;
    }
    if (covered != S || jf_group_num_gpus(grp) != n_shards) return 1;
    for (int s = 0; s < S; s++) {
        const size_t n = 7000 + 97 * (size_t)s;
        float *sig = make_signal(n);
        CHECK(jf_source_set_signal(one, s, sig, n));
        CHECK(jf_group_source_set_signal(grp, s, sig, n));
        free(sig);
    }
    /* every source on its own trajectory: a repack that put a record into the wrong shard or slot would show */
    float *pos = (float *)malloc(sizeof(float) * JF_POS_FLOATS * S * K * RUNS);
    for (int k = 0; k < K * RUNS; k++)
        for (int s = 0; s < S; s++)
            CHECK(jf_position_from_spherical((float)(-40 + (11 * s) % 121), (float)((53 * s + 3 * k) % 360), 0.4f + 0.07f * s,
                                             pos + ((size_t)k * S + s) * JF_POS_FLOATS));
    float *a = (float *)malloc(sizeof(float) * 2 * B * K * RUNS), *b = (float *)malloc(sizeof(float) * 2 * B * K * RUNS);
    CHECK(jf_process_batch(one, K * RUNS, pos, a));
    CHECK(jf_group_process_batch(grp, K * RUNS, pos, b));
    double peak = 0;
    const double d_batch = max_abs_diff(b, a, (size_t)2 * B * K * RUNS, &peak);
    /* the device-resident form, a window in the middle of the trajectory */
    CHECK(jf_batch_upload_positions(one, K * RUNS, pos));
    CHECK(jf_group_batch_upload_positions(grp, K * RUNS, pos));
    CHECK(jf_batch_run(one, K, K, NULL));
    CHECK(jf_batch_fetch(one, K, a));
    CHECK(jf_group_batch_run(grp, K, K));
    if (jf_group_batch_run(grp, 0, K) != JF_ERR_STATE) return 1; /* one run in flight */
    CHECK(jf_group_batch_fetch(grp, b));
    const double d_run = max_abs_diff(b, a, (size_t)2 * B * K, NULL);
    /* per-block calls, the controls with more than one engine behind them */
    double d_block = 0, pk = 0;
    int silent = 0;
    for (int k = 0; k < 12; k++) {
        if (k == 3 || k == 6) {
            CHECK(jf_set_mode(one, k == 3 ? JF_MODE_FD_BASIC : JF_MODE_FD_COMPLEX));
            CHECK(jf_group_set_mode(grp, k == 3 ? JF_MODE_FD_BASIC : JF_MODE_FD_COMPLEX));
        }
        if (k == 8 || k == 10) {
            CHECK(jf_set_pause(one, k == 8));
            CHECK(jf_group_set_pause(grp, k == 8));
        }
        for (int s = 0; s < S; s++) {
            CHECK(jf_source_set_spherical(one, s, (float)(-30 + (13 * s) % 100), (float)((29 * s + 11 * k) % 360), 0.8f));
            CHECK(jf_group_source_set_spherical(grp, s, (float)(-30 + (13 * s) % 100), (float)((29 * s + 11 * k) % 360), 0.8f));
        }
        CHECK(jf_process_block(one, a));
        CHECK(jf_group_process_block(grp, b));
        const double d = max_abs_diff(b, a, 2 * B, &pk);
        if (d > d_block) d_block = d;
        if (pk == 0) silent++;
    }
    const double tol = n_shards == 1 ? 0.0 : 4e-7 * S;
    printf("shards: %d shards of %d sources on one device: batch max diff %g, run max diff %g, per-block max diff %g "
           "(tolerance %g), peak %g, %d silent blocks (2 paused)\n", n_shards, S, d_batch, d_run, d_block, tol, peak, silent);
    int ok = d_batch <= tol && d_run <= tol && d_block <= tol && peak > 0.05 && silent == 2;

    /* ---- forced failures on the last shard */
    const int last = n_shards - 1;
    int ok_fail = 1;
    if (n_shards > 1) {
        /* a control call that fails on shard `last` leaves the earlier shards as they were: the next block is still the
         * single engine's in the OLD mode, and the group is not failed */
        CHECK(jf_group_debug_fail_next(grp, last));
        ok_fail &= jf_group_set_mode(grp, JF_MODE_FD_BASIC) == JF_ERR_DEVICE && jf_group_failed(grp) == 0;
        CHECK(jf_process_block(one, a));
        CHECK(jf_group_process_block(grp, b));
        ok_fail &= max_abs_diff(b, a, 2 * B, NULL) <= tol;
        CHECK(jf_group_debug_fail_next(grp, last));
        ok_fail &= jf_group_set_pause(grp, 1) == JF_ERR_DEVICE && jf_group_failed(grp) == 0;
        CHECK(jf_process_block(one, a));
        CHECK(jf_group_process_block(grp, b));
        ok_fail &= max_abs_diff(b, a, 2 * B, &pk) <= tol && pk > 0;
        /* a batch run that fails on shard `last` after the earlier shards have advanced */
        CHECK(jf_group_debug_fail_next(grp, last));
        ok_fail &= jf_group_batch_run(grp, 0, K) == JF_ERR_DEVICE && jf_group_failed(grp) == 1;
        ok_fail &= strstr(jf_group_last_error(grp), "injected failure") != NULL;
        ok_fail &= jf_group_batch_run(grp, 0, K) == JF_ERR_STATE && jf_group_process_block(grp, b) == JF_ERR_STATE;
        ok_fail &= strstr(jf_group_last_error(grp), "group failed") != NULL && strstr(jf_group_last_error(grp), "injected") != NULL;
        ok_fail &= jf_group_source_set_spherical(grp, S - 1, 0.0f, 0.0f, 1.0f) == JF_OK; /* setters stay usable */
        ok_fail &= jf_group_failed(grp) == 1;
        /* a second group: the failure in the collect loop of a per-block call */
        jf_group *g2 = NULL;
        CHECK(jf_group_create_shards_on_device(&cfg, n_shards, 0, hrir, TAPS, &g2));
        CHECK(jf_group_process_block(g2, b));
        CHECK(jf_group_debug_fail_next(g2, 0));   /* submit of shard 0 fails: nothing has advanced, not fatal */
        ok_fail &= jf_group_process_block(g2, b) == JF_ERR_DEVICE && jf_group_failed(g2) == 0;
        CHECK(jf_group_process_block(g2, b));
        CHECK(jf_group_debug_fail_next(g2, last)); /* submit of the last shard fails: the others have a block in flight */
        ok_fail &= jf_group_process_block(g2, b) == JF_ERR_DEVICE && jf_group_failed(g2) == 1;
        jf_group_destroy(g2);
        printf("shards: forced failures on shard %d: %s\n", last, ok_fail ? "as specified" : "WRONG");
    }
    jf_group_destroy(grp);
    jf_engine_destroy(one);
    free(pos);
    free(a);
    free(b);
    free(hrir);
    return (ok && ok_fail) ? 0 : 1;
}

static double now_s(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

/* The bench workload driven from a plain-C host through jefferson_group.h: 1024 looped noise
 * sources per GPU, azimuth + 1 degree per block (a crossfade every block), 128 blocks of 256 samples per run; run i + 1
 * is enqueued as soon as run i has been fetched.  Prints one line with bench.py's metric. */
static int test_bench(int n_gpus, int steps) {
    enum { B = 256, PER_GPU = 1024, KB = 128, TAPS = 128 };
    const int S = PER_GPU * n_gpus;
    /* the trajectory is periodic in 360 blocks; lcm(360, 128) = 5760 blocks are uploaded once and walked round */
    const int n_pos = 5760;
    float *hrir = make_hrir(TAPS);
    jf_config cfg = {B, 512, S, 0, KB, 0};
    jf_group *grp = NULL;
    CHECK(jf_group_create(&cfg, n_gpus, NULL, hrir, TAPS, &grp));
    float *sig = (float *)malloc(sizeof(float) * 44100);
    for (int s = 0; s < S; s++) {
        for (int i = 0; i < 44100; i++) sig[i] = frand();
        CHECK(jf_group_source_set_signal(grp, s, sig, 44100));
    }
    free(sig);
    float *pos = (float *)malloc(sizeof(float) * JF_POS_FLOATS * (size_t)S * n_pos);
    for (int s = 0; s < S; s++) {
        const float r = 0.5f + 3.0f * (frand() + 0.5f), ele = (float)(-40 + (7 * s) % 121);
        for (int k = 0; k < n_pos; k++)
            CHECK(jf_position_from_spherical(ele, (float)((37 * s + k) % 360), r, pos + ((size_t)k * S + s) * JF_POS_FLOATS));
    }
    CHECK(jf_group_batch_upload_positions(grp, n_pos, pos));
    free(pos);
    float *mix = (float *)malloc(sizeof(float) * 2 * B * KB);
    const int warm = 64;
    double t0 = 0, peak = 0;
    for (int i = 0; i < warm + steps; i++) {
        if (i == warm) t0 = now_s();
        CHECK(jf_group_batch_run(grp, (i * KB) % n_pos, KB));
        CHECK(jf_group_batch_fetch(grp, mix));
    }
    const double dt = now_s() - t0;
    for (int i = 0; i < 2 * B * KB; i++)
        if (fabs((double)mix[i]) > peak) peak = fabs((double)mix[i]);
    printf("bench: %d GPU(s) x %d sources, %d steps of %d blocks of %d: %.4e source-frames/s, %.4f ms per step, "
           "real-time factor %.0f, |mix| peak %.3f (plain C host: jf_group_batch_run + _fetch per step)\n",
           n_gpus, PER_GPU, steps, KB, B, (double)S * KB * B * steps / dt, dt / steps * 1e3,
           (double)KB * steps * B / 44100.0 / dt, peak);
    jf_group_destroy(grp);
    free(mix);
    free(hrir);
    return peak > 0.1 ? 0 : 1;
}

int main(int argc, char **argv) {
    if (argc >= 2 && !strcmp(argv[1], "pa")) return test_pa();
    if (argc >= 2 && !strcmp(argv[1], "group")) return test_group(argc >= 3 ? atoi(argv[2]) : 1);
    if (argc >= 2 && !strcmp(argv[1], "shards")) return test_shards(argc >= 3 ? atoi(argv[2]) : 2);
    if (argc >= 2 && !strcmp(argv[1], "bench"))
        return test_bench(argc >= 3 ? atoi(argv[2]) : 1, argc >= 4 ? atoi(argv[3]) : 256);
    fprintf(stderr, "usage: jf_ctest pa | group [n_gpus] | shards [n_shards] | bench [n_gpus [steps]]\n");
    return 64;
}
