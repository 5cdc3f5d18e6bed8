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
 *   - --script debugmode2: the scripted-motion run of the audio-only build (DEBUGMODE 2, main.cu:101-149): the source
 *     starts at the constructor's position (ele 0, azi 0, r 0.5, SoundSource.cu:3-16) and the main thread sets the
 *     way-points (ele, azi) = (4,2), (3,1), (2,4), (9,7), (0,0) one after the other, the k-th as soon as the source's
 *     play position `count` has reached (k * 44100) % length (main.cu:105,113,121,128,135), then lets the stream run for two
 *     more seconds (main.cu:142).  The reference polls `count` every 100 ms from another thread; here it is looked at
 *     before every block (the audio thread's own granularity), so a run is reproducible: a way-point is latched by the
 *     first block that starts with `count` at or past its mark.  `count` follows Audio.cu:121-139 (wraps at `length`).
 *
 * usage: jf_render <hrir_dir | set.sofa> <in.wav> <out.wav> [--block 256] [--azi 3] [--ele 5]
 *                  [--dwell 172] [--rounds 72] [--step 5] [--radius 0.5] [--latency] [--script debugmode2]
 *   <set.sofa>  a name that ends in ".sofa": the HRTF set of a SOFA file instead of the KEMAR directory
 *               (jf_engine_create_sofa; --sofa-tol T: degrees a measurement may lie off its ring's uniform steps, 0.51 --
 *               what sets whose azimuths were rounded to whole degrees, like KEMAR's, need)
 *   --latency  use jf_callback (the CUDA path's one-block latency, Audio.cu:104-117)
 *              instead of jf_process_block (the CPU path's ordering)
 *   --no-pin   leave the thread where the system put it (default: jf_pin_thread_to_device -- on a two-socket host a block
 *              costs 1-5 us more from the socket the GPU does not hang off)
 *   --batch N  hand the engine N callbacks at a time (jf_process_batch: the same blocks, the positions the
 *              audio thread would have latched given up front) -- what an offline render should use: a
 *              single source cannot fill a GPU one block at a time
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../../include/jefferson.h"

/* DEBUGMODE 2 (main.cu:101-149): way-points in the order the main thread sets them */
static const float kScriptEle[5] = {4, 3, 2, 9, 0}, kScriptAzi[5] = {2, 1, 4, 7, 0};

/* The scripted run as latched records: pos[total][JF_POS_FLOATS], block by block.  Returns the number of blocks, or 0. */
static size_t script_debugmode2(size_t length, int block, float radius, float **pos_out) {
    if (length == 0) return 0;
    const size_t tail = (2 * 44100 + (size_t)block - 1) / (size_t)block; /* std::this_thread::sleep_for(2 s), main.cu:142 */
    size_t cap = 6 * 44100 / (size_t)block + 5 * (length / (size_t)block + 2) + tail + 16, n = 0, left = 0;
    float *pos = (float *)malloc(sizeof(float) * JF_POS_FLOATS * cap);
    if (!pos) return 0;
    float ele = 0, azi = 0; /* SoundSource::SoundSource() */
    size_t count = 0;
    int counter = 1;
    while (n < cap) {
        /* what the main thread has done by the time this block's callback runs */
        while (counter <= 5 && count >= ((size_t)counter * 44100) % length) {
            ele = kScriptEle[counter - 1];
            azi = kScriptAzi[counter - 1];
            if (++counter == 6) left = tail;
        }
        if (counter == 6 && left-- == 0) break;
        if (jf_position_from_spherical(ele, azi, radius, pos + JF_POS_FLOATS * n) != JF_OK) {
            free(pos);
            return 0;
        }
        n++;
        /* Audio.cu:121-139 */
        if (count + (size_t)block < length) count += (size_t)block;
        else count = (size_t)block - (length - count);
    }
    *pos_out = pos;
    return n;
}

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int main(int argc, char **argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: %s <hrir_dir | set.sofa> <in.wav> <out.wav> [--block B] [--azi A] [--ele E] "
                        "[--dwell N] [--rounds R] [--step D] [--radius r] [--latency] [--batch N] [--script debugmode2] [--no-pin] "
                        "[--sofa-tol T]\n", argv[0]);
        return 2;
    }
    int block = 256, dwell = 172, rounds = 72, latency = 0, batch = 0, script = 0, pin = 1;
    float azi = 3, ele = 5, step = 5, radius = 0.5f, sofa_tol = 0.51f;
    for (int i = 4; i < argc; i++) {
        if (!strcmp(argv[i], "--latency")) latency = 1;
        else if (!strcmp(argv[i], "--no-pin")) pin = 0;
        else if (i + 1 < argc && !strcmp(argv[i], "--block")) block = atoi(argv[++i]);
        else if (i + 1 < argc && !strcmp(argv[i], "--batch")) batch = atoi(argv[++i]);
        else if (i + 1 < argc && !strcmp(argv[i], "--azi")) azi = (float)atof(argv[++i]);
        else if (i + 1 < argc && !strcmp(argv[i], "--ele")) ele = (float)atof(argv[++i]);
        else if (i + 1 < argc && !strcmp(argv[i], "--dwell")) dwell = atoi(argv[++i]);
        else if (i + 1 < argc && !strcmp(argv[i], "--rounds")) rounds = atoi(argv[++i]);
        else if (i + 1 < argc && !strcmp(argv[i], "--step")) step = (float)atof(argv[++i]);
        else if (i + 1 < argc && !strcmp(argv[i], "--radius")) radius = (float)atof(argv[++i]);
        else if (i + 1 < argc && !strcmp(argv[i], "--sofa-tol")) sofa_tol = (float)atof(argv[++i]);
        else if (i + 1 < argc && !strcmp(argv[i], "--script") && !strcmp(argv[i + 1], "debugmode2")) script = 1, i++;
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
    if (pin && jf_pin_thread_to_device(0) != JF_OK) fprintf(stderr, "not pinned: %s\n", jf_last_error(NULL));
    jf_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.frames_per_buffer = block;
    cfg.hrtf_len = 512;
    cfg.n_sources = 1; /* main.cu:60 */
    cfg.device = 0;
    cfg.max_batch_blocks = batch > 0 ? batch : 1;
    jf_engine *e = NULL;
    const size_t len1 = strlen(argv[1]);
    const int sofa = len1 > 5 && !strcmp(argv[1] + len1 - 5, ".sofa");
    if ((sofa ? jf_engine_create_sofa(&cfg, argv[1], sofa_tol, &e) : jf_engine_create_from_dir(&cfg, argv[1], &e)) != JF_OK) {
        fprintf(stderr, "engine: %s\n", jf_last_error(NULL));
        return 1;
    }
    jf_source_set_signal(e, 0, sig, n);
    jf_free(sig);

    if (script) {
        /* DEBUGMODE 2: the way-points as the blocks latch them; per block through the setter (what the main thread calls,
         * SoundSource::updateFromSpherical) or, with --batch, as records handed over up front */
        float *spos = NULL;
        const size_t nb = script_debugmode2(n, block, radius, &spos);
        float *sout = nb ? (float *)malloc(sizeof(float) * 2 * (size_t)block * nb) : NULL;
        if (!nb || !sout) {
            fprintf(stderr, "script: empty input or out of memory\n");
            return 1;
        }
        int src = JF_OK;
        const double ts = now_s();
        jf_source_reset(e, 0);
        if (batch > 0) {
            for (size_t k0 = 0; k0 < nb && src == JF_OK; k0 += (size_t)batch) {
                const int kb = nb - k0 < (size_t)batch ? (int)(nb - k0) : batch;
                src = jf_process_batch(e, kb, spos + JF_POS_FLOATS * k0, sout + 2 * (size_t)block * k0);
            }
        } else {
            if (latency) src = jf_callback(e, sout); /* priming call: the CUDA path hands out block k - 1 */
            for (size_t k0 = 0; k0 < nb && src == JF_OK; k0++) {
                const float *r = spos + JF_POS_FLOATS * k0; /* {ele, azi, x, y, z}: the setter takes (ele, azi, radius) */
                jf_source_set_spherical(e, 0, r[0], r[1], radius);
                src = latency ? jf_callback(e, sout + 2 * (size_t)block * k0) : jf_process_block(e, sout + 2 * (size_t)block * k0);
            }
        }
        const double ds = now_s() - ts;
        if (src != JF_OK) {
            fprintf(stderr, "processing: %s\n", jf_last_error(e));
            return 1;
        }
        if (jf_wav_write_stereo24(argv[3], sout, (size_t)block * nb, fs ? fs : 44100) != JF_OK) {
            fprintf(stderr, "output: %s\n", jf_last_error(NULL));
            return 1;
        }
        fprintf(stderr, "debugmode2: %zu blocks of %d frames in %.3f s: %.1f us per block, real-time factor %.1f\n", nb, block,
                ds, 1e6 * ds / (double)nb, ((double)nb * block / 44100.0) / ds);
        free(spos);
        free(sout);
        jf_engine_destroy(e);
        return 0;
    }

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
