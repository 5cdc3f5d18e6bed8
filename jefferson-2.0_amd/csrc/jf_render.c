/*
 * jf_render -- offline WAV-in / WAV-out driver over the C ABI (plain C: what a host
 * program written against include/jefferson.h looks like).
 *
 * Reproduces the reference's audio-only runs without PortAudio/OpenGL:
 *   - main.cu:60-82 init order (sources, input file, HRTF set, 24-bit stereo output file)
 *   - the waveFileTesting / benchmarkTesting trajectory (precision_test.cu:2093-2152,
 *     :2203-2250): start at (azi, ele), dwell `--dwell` blocks, then azimuth += `--step`
 *     degrees `--rounds` times; radius 0.5 (SoundSource.cu:12)
 *
 * usage: jf_render <hrir_dir> <in.wav> <out.wav> [--block 256] [--azi 3] [--ele 5]
 *                  [--dwell 172] [--rounds 72] [--step 5] [--radius 0.5] [--latency]
 *   --latency  use jf_callback (the CUDA path's one-block latency, Audio.cu:104-117)
 *              instead of jf_process_block (the CPU path's ordering)
 *   --batch N  hand the engine N callbacks at a time (jf_process_batch: the same blocks, the positions the
 *              audio thread would have latched given up front) -- what an offline render should use: a
 *              single source cannot fill a GPU one block at a time
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../../include/jefferson.h"

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int main(int argc, char **argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: %s <hrir_dir> <in.wav> <out.wav> [--block B] [--azi A] [--ele E] "
                        "[--dwell N] [--rounds R] [--step D] [--radius r] [--latency] [--batch N]\n", argv[0]);
        return 2;
    }
    int block = 256, dwell = 172, rounds = 72, latency = 0, batch = 0;
    float azi = 3, ele = 5, step = 5, radius = 0.5f;
    for (int i = 4; i < argc; i++) {
        if (!strcmp(argv[i], "--latency")) latency = 1;
        else if (i + 1 < argc && !strcmp(argv[i], "--block")) block = atoi(argv[++i]);
        else if (i + 1 < argc && !strcmp(argv[i], "--batch")) batch = atoi(argv[++i]);
        else if (i + 1 < argc && !strcmp(argv[i], "--azi")) azi = (float)atof(argv[++i]);
        else if (i + 1 < argc && !strcmp(argv[i], "--ele")) ele = (float)atof(argv[++i]);
        else if (i + 1 < argc && !strcmp(argv[i], "--dwell")) dwell = atoi(argv[++i]);
        else if (i + 1 < argc && !strcmp(argv[i], "--rounds")) rounds = atoi(argv[++i]);
        else if (i + 1 < argc && !strcmp(argv[i], "--step")) step = (float)atof(argv[++i]);
        else if (i + 1 < argc && !strcmp(argv[i], "--radius")) radius = (float)atof(argv[++i]);
        else {
            fprintf(stderr, "unknown option %s\n", argv[i]);
            return 2;
        }
    }

    float *sig = NULL;
    size_t n = 0;
    int fs = 0;
    if (jf_wav_read_mono(argv[2], &sig, &n, &fs) != JF_OK) {
        fprintf(stderr, "input: %s\n", jf_last_error(NULL));
        return 1;
    }
    jf_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.frames_per_buffer = block;
    cfg.hrtf_len = 512;
    cfg.n_sources = 1; /* main.cu:60 */
    cfg.device = 0;
    cfg.max_batch_blocks = batch > 0 ? batch : 1;
    jf_engine *e = NULL;
    if (jf_engine_create_from_dir(&cfg, argv[1], &e) != JF_OK) {
        fprintf(stderr, "engine: %s\n", jf_last_error(NULL));
        return 1;
    }
    jf_source_set_signal(e, 0, sig, n);
    jf_free(sig);

    const size_t total = (size_t)dwell * (size_t)(rounds + 1);
    float *out = (float *)malloc(sizeof(float) * 2 * (size_t)block * total);
    if (!out) return 1;
    jf_source_reset(e, 0);
    jf_source_set_spherical(e, 0, ele, azi, radius);
    size_t k = 0;
    int rc = JF_OK;
    const double t0 = now_s();
    if (batch > 0) {
        /* the whole trajectory as latched records, then N blocks per call */
        float *pos = (float *)malloc(sizeof(float) * JF_POS_FLOATS * total);
        if (!pos) return 1;
        float a = azi;
        for (int r = 0; r <= rounds; r++) {
            if (r > 0) {
                a += step;
                if (a >= 360) a -= 360;
            }
            float rec[JF_POS_FLOATS];
            if (jf_position_from_spherical(ele, a, radius, rec) != JF_OK) {
                fprintf(stderr, "position: %s\n", jf_last_error(NULL));
                return 1;
            }
            for (int j = 0; j < dwell; j++, k++) memcpy(pos + JF_POS_FLOATS * k, rec, sizeof(rec));
        }
        for (k = 0; k < total && rc == JF_OK; k += (size_t)batch) {
            const int nb = total - k < (size_t)batch ? (int)(total - k) : batch;
            rc = jf_process_batch(e, nb, pos + JF_POS_FLOATS * k, out + 2 * (size_t)block * k);
        }
        free(pos);
        rounds = -1; /* done: skip the per-block loop */
    }
    if (latency && batch <= 0) rc = jf_callback(e, out); /* priming call, precision_test.cu:2110 */
    for (int r = 0; r <= rounds && rc == JF_OK; r++) {
        if (r > 0) {
            azi += step;
            if (azi >= 360) azi -= 360;
            jf_source_set_spherical(e, 0, ele, azi, radius);
        }
        for (int j = 0; j < dwell && rc == JF_OK; j++, k++)
            rc = latency ? jf_callback(e, out + 2 * (size_t)block * k)
                         : jf_process_block(e, out + 2 * (size_t)block * k);
    }
    const double dt = now_s() - t0;
    if (rc != JF_OK) {
        fprintf(stderr, "processing: %s\n", jf_last_error(e));
        return 1;
    }
    if (jf_wav_write_stereo24(argv[3], out, (size_t)block * total, fs ? fs : 44100) != JF_OK) {
        fprintf(stderr, "output: %s\n", jf_last_error(NULL));
        return 1;
    }
    fprintf(stderr, "%zu blocks of %d frames in %.3f s: %.1f us per block, real-time factor %.1f\n", total, block, dt,
            1e6 * dt / (double)total, ((double)total * block / 44100.0) / dt);
    free(out);
    jf_engine_destroy(e);
    return 0;
}
