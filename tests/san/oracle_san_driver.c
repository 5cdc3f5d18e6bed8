/* Sanitizer driver for the CPU oracle (oracle/jf_oracle.c) -- built by tests/test_sanitizers.py with
 * -fsanitize=address,undefined and run on the CPU: the checker itself must not read or write out of bounds. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../oracle/jf_oracle.h"

static unsigned lcg(unsigned *s) { return *s = *s * 1664525u + 1013904223u; }
static float unit(unsigned *s) { return (float)((int)(lcg(s) >> 8) % 2001 - 1000) / 1000.f; }

int main(void) {
    unsigned seed = 7;
    int bad = 0;
    float *hrir = (float *)malloc(sizeof(float) * JFO_NUM_HRTF * 2 * 128);
    for (int i = 0; i < JFO_NUM_HRTF * 2 * 128; i++) hrir[i] = 0.1f * unit(&seed) * expf(-0.03f * (float)(i % 128));
    for (int B = 64; B <= 256; B *= 2) {
        const int S = 3, K = 9;
        jfo_engine *e = jfo_create(B, 512, S, hrir, 128);
        if (!e) return 2;
        float sig[3000];
        for (int i = 0; i < 3000; i++) sig[i] = 0.5f * unit(&seed);
        jfo_source_set_signal(e, 0, sig, 3000);
        jfo_source_set_signal(e, 1, sig, 37);   /* shorter than a block: wraps several times per block */
        jfo_source_set_signal(e, 2, sig, 0);    /* empty */
        float *out = (float *)malloc(sizeof(float) * 2 * B);
        for (int k = 0; k < K; k++) {
            jfo_source_set_spherical(e, 0, -40 + 13 * k, (float)(50 * k), 0.5f);
            jfo_source_set_cartesian(e, 1, 1.f - 0.3f * k, 0.2f * k, -1.f);
            jfo_source_set_spherical(e, 2, 95.f, 400.f, 0.f);  /* outside the range */
            if (k == 4) jfo_source_reset(e, 0);
            if (k == 6) jfo_set_mode(e, 1);
            if (k == 7) jfo_set_mode(e, 2);
            jfo_process_block(e, out);
            for (int i = 0; i < 2 * B; i++) bad += !isfinite(out[i]);
            (void)jfo_source_last_block(e, k % S);
        }
        jfo_set_mode(e, 0);
        float *pos = (float *)malloc(sizeof(float) * 5 * S * K), *mix = (float *)malloc(sizeof(float) * 2 * B * K);
        float *part = (float *)malloc(sizeof(float) * 2 * B * K * S);
        for (int k = 0; k < K; k++)
            for (int s = 0; s < S; s++) jfo_from_spherical((float)(10 * s - 20), (float)((40 * s + 7 * k) % 360), 0.4f + s, pos + 5 * (k * S + s));
        jfo_process_batch(e, K, pos, mix, part, 2);
        jfo_process_batch(e, K, pos, mix, NULL, 0);
        /* the reverb stage: a ragged response, then off again */
        float ir[700];
        for (int i = 0; i < 700; i++) ir[i] = unit(&seed) * expf(-0.01f * (float)i);
        if (jfo_reverb_set_ir(e, ir, 700, 0.5f) != 0) bad++;
        jfo_source_set_signal(e, 0, sig, 3000);
        for (int k = 0; k < 8; k++) jfo_process_block(e, out);
        jfo_process_batch(e, K, pos, mix, part, 0);
        if (jfo_reverb_set_ir(e, ir, 1, 1.f) != 0) bad++;
        jfo_process_block(e, out);
        jfo_reverb_set_ir(e, NULL, 0, 1.f);
        jfo_process_block(e, out);
        free(pos), free(mix), free(part), free(out);
        jfo_destroy(e);
    }
    {   /* the offline form and the stand-alone pieces */
        float x[500], ir[90];
        for (int i = 0; i < 500; i++) x[i] = unit(&seed);
        for (int i = 0; i < 90; i++) ir[i] = unit(&seed);
        const int n = jfo_reverb_padded_size(500, 90);
        float *y = (float *)malloc(sizeof(float) * (size_t)n);
        const float g = jfo_reverb_offline(x, 500, ir, 90, y);
        bad += !(g > 0.f && isfinite(g));
        free(y);
        float X[2 * 513], t[1024], D[2 * 513];
        for (int i = 0; i < 1024; i++) t[i] = unit(&seed);
        jfo_rfft(t, 1024, X);
        jfo_irfft(X, 1024, t);
        jfo_distance_factor(0.1f, -0.2f, 3.f, 513, D);
        int idx[4], rows[4];
        float om[6], w[4];
        for (int e = -60; e <= 100; e++)
            for (int a = -5; a <= 365; a++) {
                if (jfo_interp((float)e, (float)a, idx, om) == 0) (void)jfo_terms(idx, om, rows, w), (void)jfo_case(idx);
                (void)jfo_interp_corrected((float)e + 0.5f, (float)a + 0.25f, idx, om);
                (void)jfo_pick_hrtf((float)e, (float)a);
            }
        float *table = (float *)malloc(sizeof(float) * JFO_NUM_HRTF * 2 * 513 * 2);
        jfo_build_table(hrir, JFO_NUM_HRTF, 128, 1024, table);
        free(table);
    }
    free(hrir);
    printf("oracle sanitizer driver: %d bad values\n", bad);
    return bad ? 1 : 0;
}
