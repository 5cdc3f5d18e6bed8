/*
 * jf_oracle.c -- float32 CPU restatement of the reference's HRTF convolution
 * path.  TEST INFRASTRUCTURE ONLY (see jf_oracle.h for the rules and for the
 * parity pin status: numeric outputs are "parity unpinned", anchored to
 * oracle/model64.py within the reference's own 2e-7 tolerance).
 *
 * Citations are relative to /root/reference/Jefferson/src/.
 * Compile with -ffp-contract=off so a*b+c is never fused (the reference's CPU
 * build does not fuse; its CUDA build may -- SURVEY.md App. A).
 */
#include "jf_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Universal.cuh:14-16 */
#define JFO_PI 3.14159265358979323846264338327950288

/* hrtf_signals.cu:7-10 */
static const int elevation_pos[JFO_NUM_ELEV] = {-40, -30, -20, -10, 0, 10, 20,
                                                30,  40,  50,  60,  70, 80, 90};
static const float azimuth_inc[JFO_NUM_ELEV] = {6.43f, 6.00f, 5.00f,  5.00f,  5.00f,
                                                5.00f, 5.00f, 6.00f,  6.43f,  8.00f,
                                                10.00f, 15.00f, 30.00f, 361.0f};
static int azimuth_offset[JFO_NUM_ELEV + 1];
static int offsets_ready = 0;

/* hrtf_signals.cu:119-140: the loader's double loop, counting only. */
static void build_offsets(void) {
    if (offsets_ready) return;
    int j = 0;
    azimuth_offset[0] = 0;
    for (int i = 0; i < JFO_NUM_ELEV; i++) {
        float azi;
        for (azi = 0; azi < 360; azi += azimuth_inc[i]) j++;
        azimuth_offset[i + 1] = j;
    }
    offsets_ready = 1;
}

void jfo_azimuth_offsets(int off[JFO_NUM_ELEV + 1]) {
    build_offsets();
    memcpy(off, azimuth_offset, sizeof(azimuth_offset));
}

void jfo_table_positions(int ele[JFO_NUM_HRTF], int azi_out[JFO_NUM_HRTF]) {
    int j = 0;
    for (int i = 0; i < JFO_NUM_ELEV; i++) {
        float azi;
        for (azi = 0; azi < 360; azi += azimuth_inc[i]) {
            if (j < JFO_NUM_HRTF) {
                ele[j] = elevation_pos[i];
                azi_out[j] = (int)round(azi); /* hrtf_signals.cu:124 */
            }
            j++;
        }
    }
}

/* hrtf_signals.cu:20-51 */
int jfo_pick_hrtf(float obj_ele, float obj_azi) {
    build_offsets();
    int i, n, ele_idx = 0, hrtf_idx = 0;
    float d, dmin;
    obj_ele = roundf(obj_ele / 10) * 10;
    dmin = 1e37f;
    for (i = 0; i < JFO_NUM_ELEV; i++) {
        d = obj_ele - elevation_pos[i];
        d = d > 0 ? d : -d;
        if (d < dmin) {
            dmin = d;
            ele_idx = i;
        }
    }
    obj_azi = roundf(obj_azi);
    dmin = 1e37f;
    n = azimuth_offset[ele_idx + 1] - azimuth_offset[ele_idx];
    for (i = 0; i < n; i++) {
        d = obj_azi - i * azimuth_inc[ele_idx];
        d = d > 0 ? d : -d;
        if (d < dmin) {
            dmin = d;
            hrtf_idx = azimuth_offset[ele_idx] + i;
        }
    }
    return hrtf_idx;
}

/* SoundSource.cu:65-105 */
int jfo_interp(float ele, float azi, int hrtf_indices[4], float omegas[6]) {
    float omegaA, omegaB, omegaC, omegaD, omegaE, omegaF;
    int phi[2];
    int theta[4];
    float deltaTheta1 = 0.0f, deltaTheta2 = 0.0f;
    int found1 = 0, found2 = 0;
    phi[0] = (int)(ele) / 10 * 10;
    phi[1] = (int)(ele + 9) / 10 * 10;
    omegaE = (ele - phi[0]) / 10.0f;
    omegaF = (phi[1] - ele) / 10.0f;
    for (int i = 0; i < JFO_NUM_ELEV; i++) {
        if (phi[0] == elevation_pos[i]) {
            deltaTheta1 = azimuth_inc[i];
            found1 = 1;
        }
        if (phi[1] == elevation_pos[i]) {
            deltaTheta2 = azimuth_inc[i];
            found2 = 1;
            break;
        }
    }
    if (!found1 || !found2) return -1; /* reference: uninitialised read */
    theta[0] = (int)((int)(azi / deltaTheta1) * deltaTheta1);
    theta[1] = (int)((int)((azi + deltaTheta1 - 1) / deltaTheta1) * deltaTheta1);
    theta[2] = (int)((int)(azi / deltaTheta2) * deltaTheta2);
    theta[3] = (int)((int)((azi + deltaTheta2 - 1) / deltaTheta2) * deltaTheta2);
    omegaA = (azi - theta[0]) / deltaTheta1;
    omegaB = (theta[1] - azi) / deltaTheta1;
    omegaC = (azi - theta[2]) / deltaTheta2;
    omegaD = (theta[3] - azi) / deltaTheta2;
    hrtf_indices[0] = jfo_pick_hrtf((float)phi[0], (float)theta[0]);
    hrtf_indices[1] = jfo_pick_hrtf((float)phi[0], (float)theta[1]);
    hrtf_indices[2] = jfo_pick_hrtf((float)phi[1], (float)theta[2]);
    hrtf_indices[3] = jfo_pick_hrtf((float)phi[1], (float)theta[3]);
    omegas[0] = omegaA;
    omegas[1] = omegaB;
    omegas[2] = omegaC;
    omegas[3] = omegaD;
    omegas[4] = omegaE;
    omegas[5] = omegaF;
    return 0;
}

/* The corrected rule the drop-in offers behind JF_FLAG_CORRECTED_INTERPOLATION (not in
 * the reference): true floor/ceil of the elevation (no truncation toward zero below 0), azimuth folded into
 * [0, 360) with the last interval of a ring wrapping to its first entry, azimuths kept in float so that a
 * ring's two weights sum to 1, elevations below the lowest ring clamped to it.  Same index order and weight
 * meaning as jfo_interp: idx = {ring0 low, ring0 high, ring1 low, ring1 high}, omegas = {A, B, C, D, E, F} with
 * A/C the weight of the high azimuth, B/D of the low one, E of ring1, F of ring0. */
int jfo_interp_corrected(float ele, float azi, int hrtf_indices[4], float omegas[6]) {
    build_offsets();
    if (!(ele <= 90.0f) || !(ele > -1.0e6f) || !(azi > -1.0e6f && azi < 1.0e6f)) return -1;
    if (ele < -40.0f) ele = -40.0f;
    float a = azi - 360.0f * floorf(azi / 360.0f);
    if (!(a < 360.0f)) a = 0.0f;
    const float q = floorf(ele / 10.0f);
    const float phi0 = 10.0f * q;
    const int on_ring = (ele == phi0);
    const int r0 = (int)q + 4;
    const int r1 = on_ring ? r0 : r0 + 1;
    const float omE = on_ring ? 0.0f : (ele - phi0) / 10.0f;
    const int ring[2] = {r0, r1};
    for (int j = 0; j < 2; j++) {
        const int r = ring[j];
        const float d = azimuth_inc[r];
        const int n = azimuth_offset[r + 1] - azimuth_offset[r];
        int i0 = (int)floorf(a / d);
        if (i0 > n - 1) i0 = n - 1;
        float wa = (a - (float)i0 * d) / d;
        if (wa < 0.0f) wa = 0.0f;
        if (wa > 1.0f) wa = 1.0f;
        if (n == 1) wa = 0.0f;
        int i1 = i0 + 1 == n ? 0 : i0 + 1;
        if (wa == 0.0f) i1 = i0;
        hrtf_indices[2 * j] = azimuth_offset[r] + i0;
        hrtf_indices[2 * j + 1] = azimuth_offset[r] + i1;
        omegas[2 * j] = wa;
        omegas[2 * j + 1] = 1.0f - wa;
    }
    omegas[4] = omE;
    omegas[5] = 1.0f - omE;
    return 0;
}

/* ---- any grid of elevation rings (jf_oracle.h) ---- */
typedef struct {
    int n_rings, n_rows;
    int offset[JFO_MAX_RINGS + 1];
    float step[JFO_MAX_RINGS], ele[JFO_MAX_RINGS];
} jfo_grid;

static int grid_fill(jfo_grid *g, int n_rings, const float *ring_ele, const int *ring_count, const float *ring_step) {
    if (n_rings < 1 || n_rings > JFO_MAX_RINGS || !ring_ele || !ring_count) return -1;
    g->n_rings = n_rings;
    g->offset[0] = 0;
    for (int r = 0; r < n_rings; r++) {
        if (ring_count[r] < 1 || (r > 0 && !(ring_ele[r] > ring_ele[r - 1]))) return -1;
        g->ele[r] = ring_ele[r];
        g->step[r] = ring_step ? ring_step[r] : 360.0f / (float)ring_count[r];
        if (ring_count[r] == 1 && !(g->step[r] >= 360.0f)) g->step[r] = 361.0f; /* the pole: one measurement is the ring */
        g->offset[r + 1] = g->offset[r] + ring_count[r];
    }
    g->n_rows = g->offset[n_rings];
    return 0;
}

int jfo_grid_rows(int n_rings, const int *ring_count) {
    int n = 0;
    for (int r = 0; r < n_rings; r++) n += ring_count[r];
    return n;
}

void jfo_kemar_grid(float ring_ele[JFO_NUM_ELEV], int ring_count[JFO_NUM_ELEV], float ring_step[JFO_NUM_ELEV]) {
    build_offsets();
    for (int r = 0; r < JFO_NUM_ELEV; r++) {
        ring_ele[r] = (float)elevation_pos[r];
        ring_count[r] = azimuth_offset[r + 1] - azimuth_offset[r];
        ring_step[r] = azimuth_inc[r];
    }
}

/* Same index order and weight meaning as jfo_interp_corrected; float32 step by step. */
static int grid_interp(const jfo_grid *g, float ele, float azi, int hrtf_indices[4], float omegas[6]) {
    if (!(ele <= 90.0f) || !(ele > -1.0e6f) || !(azi > -1.0e6f && azi < 1.0e6f)) return -1;
    const int last = g->n_rings - 1;
    if (ele < g->ele[0]) ele = g->ele[0];
    if (ele > g->ele[last]) ele = g->ele[last];
    float a = azi - 360.0f * floorf(azi / 360.0f);
    if (!(a < 360.0f)) a = 0.0f;
    int r0 = 0;
    while (r0 < last && g->ele[r0 + 1] <= ele) r0++; /* the highest ring at or below the position */
    const float phi0 = g->ele[r0];
    const int on_ring = (ele == phi0);
    const int r1 = on_ring ? r0 : r0 + 1;
    const float omE = on_ring ? 0.0f : (ele - phi0) / (g->ele[r0 + 1] - phi0);
    const int ring[2] = {r0, r1};
    for (int j = 0; j < 2; j++) {
        const int r = ring[j];
        const float d = g->step[r];
        const int n = g->offset[r + 1] - g->offset[r];
        int i0 = (int)floorf(a / d);
        if (i0 > n - 1) i0 = n - 1;
        float wa = (a - (float)i0 * d) / d;
        if (wa < 0.0f) wa = 0.0f;
        if (wa > 1.0f) wa = 1.0f;
        if (n == 1) wa = 0.0f;
        int i1 = i0 + 1 == n ? 0 : i0 + 1;
        if (wa == 0.0f) i1 = i0;
        hrtf_indices[2 * j] = g->offset[r] + i0;
        hrtf_indices[2 * j + 1] = g->offset[r] + i1;
        omegas[2 * j] = wa;
        omegas[2 * j + 1] = 1.0f - wa;
    }
    omegas[4] = omE;
    omegas[5] = 1.0f - omE;
    return 0;
}

static int grid_pick(const jfo_grid *g, float ele, float azi) {
    int ring = 0;
    float dmin = 1e37f;
    for (int r = 0; r < g->n_rings; r++) {
        float d = ele - g->ele[r];
        d = d > 0 ? d : -d;
        if (d < dmin) {
            dmin = d;
            ring = r;
        }
    }
    const int n = g->offset[ring + 1] - g->offset[ring];
    float a = azi - 360.0f * floorf(azi / 360.0f);
    if (!(a < 360.0f)) a = 0.0f;
    int i = (int)floorf(a / g->step[ring] + 0.5f);
    if (i >= n) i = 0;
    return g->offset[ring] + i;
}

int jfo_grid_interp(int n_rings, const float *ring_ele, const int *ring_count, const float *ring_step,
                    float ele, float azi, int idx[4], float omegas[6]) {
    jfo_grid g;
    if (grid_fill(&g, n_rings, ring_ele, ring_count, ring_step)) return -2;
    return grid_interp(&g, ele, azi, idx, omegas);
}

int jfo_grid_pick(int n_rings, const float *ring_ele, const int *ring_count, const float *ring_step, float ele, float azi) {
    jfo_grid g;
    if (grid_fill(&g, n_rings, ring_ele, ring_count, ring_step)) return -2;
    return grid_pick(&g, ele, azi);
}

/* GPUSoundSource.cu:301-316 (the CUDA path's predicate; CPUSoundSource.cpp:262
 * tests only idx0==idx2 for case 2 -- SURVEY.md App. C#6). */
int jfo_case(const int h[4]) {
    if (h[0] == h[1] && h[1] == h[2] && h[2] == h[3]) return 1;
    if (h[0] == h[2] && h[1] == h[3]) return 2;
    if (h[0] == h[1] && h[0] != h[2]) return 3;
    return 4;
}

/* GPUSoundSource.cu:118-292: which rows, which scale, in buf_no order. */
int jfo_terms(const int h[4], const float om[6], int rows[4], float w[4]) {
    switch (jfo_case(h)) {
    case 1:
        rows[0] = h[0];
        w[0] = 1.0f; /* ComplexPointwiseMul: no scale (exact x1) */
        return 1;
    case 2:
        rows[0] = h[0]; w[0] = om[1];
        rows[1] = h[1]; w[1] = om[0];
        return 2;
    case 3:
        rows[0] = h[0]; w[0] = om[5];
        rows[1] = h[2]; w[1] = om[4];
        return 2;
    default:
        rows[0] = h[0]; w[0] = om[5] * om[1];
        rows[1] = h[1]; w[1] = om[5] * om[0];
        rows[2] = h[2]; w[2] = om[4] * om[3];
        rows[3] = h[3]; w[3] = om[4] * om[2];
        return 4;
    }
}

/* SoundSource.cu:41-54 */
void jfo_from_spherical(float ele, float azi, float r, float out[5]) {
    ele = roundf(ele);
    azi = roundf(azi);
    out[0] = ele;
    out[1] = azi;
    out[2] = (float)(r * sin(azi * JFO_PI / 180.0f));  /* x */
    out[4] = (float)(r * -cos(azi * JFO_PI / 180.0f)); /* z */
    out[3] = (float)(r * sin(ele * JFO_PI / 180.0f));  /* y */
}

/* SoundSource.cu:20-36 */
int jfo_from_cartesian(float x, float y, float z, float out[3]) {
    float r = sqrtf(x * x + z * z + y * y);
    float horizR = sqrtf(x * x + z * z);
    if (r == 0.0f) return -1;
    float ele = (float)(atan2f(y, horizR) * 180.0f / JFO_PI);
    float azi = (float)(atan2f(-x / r, -z / r) * 180.0f / JFO_PI);
    if (azi < 0.0f) azi += 360;
    out[0] = roundf(ele);
    out[1] = roundf(azi);
    out[2] = r;
    return 0;
}

/* GPUSoundSource.cu:81-95 (host part) + kernels.cu:116-125 (per bin). */
void jfo_distance_factor(float x, float y, float z, int nc, float *D) {
    float r = sqrtf(x * x + y * y + z * z);
    r /= 5;
    float fsvs = (float)(44100.0 / 343.0);
    float frac = 1 + fsvs * (float)pow(r, 2);
    for (int i = 0; i < nc; i++) {
        double ph = 2 * JFO_PI * fsvs * r * i / nc;
        D[2 * i] = (float)(cos(ph) / frac);
        D[2 * i + 1] = (float)(-sin(ph) / frac);
    }
}

/* ------------------------------------------------------------------ FFT --
 * The reference calls FFTW3f / cuFFT (not vendored).  This is the oracle's
 * own radix-2 float32 FFT with double-derived twiddles; definitions follow
 * FFTW's: r2c forward exp(-), c2r inverse exp(+), both unnormalised, c2r
 * ignoring the imaginary parts of bins 0 and N/2.
 */
typedef struct {
    int n;
    float *tw; /* exp(-2 pi i k / n), k < n/2, interleaved */
    int *rev;
} jfo_plan;

static void plan_init(jfo_plan *p, int n) {
    p->n = n;
    p->tw = (float *)malloc(sizeof(float) * (size_t)n);
    p->rev = (int *)malloc(sizeof(int) * (size_t)n);
    for (int k = 0; k < n / 2; k++) {
        double a = -2.0 * JFO_PI * k / n;
        p->tw[2 * k] = (float)cos(a);
        p->tw[2 * k + 1] = (float)sin(a);
    }
    int bits = 0;
    while ((1 << bits) < n) bits++;
    for (int i = 0; i < n; i++) {
        int r = 0;
        for (int b = 0; b < bits; b++)
            if (i & (1 << b)) r |= 1 << (bits - 1 - b);
        p->rev[i] = r;
    }
}
static void plan_free(jfo_plan *p) {
    free(p->tw);
    free(p->rev);
}

/* in-place complex FFT, sign = -1 forward, +1 inverse (unnormalised) */
static void cfft(const jfo_plan *p, float *z, int sign) {
    const int n = p->n;
    for (int i = 0; i < n; i++) {
        int r = p->rev[i];
        if (r > i) {
            float tr = z[2 * i], ti = z[2 * i + 1];
            z[2 * i] = z[2 * r];
            z[2 * i + 1] = z[2 * r + 1];
            z[2 * r] = tr;
            z[2 * r + 1] = ti;
        }
    }
    for (int len = 2; len <= n; len <<= 1) {
        const int half = len >> 1, step = n / len;
        for (int base = 0; base < n; base += len) {
            for (int k = 0; k < half; k++) {
                float wr = p->tw[2 * k * step];
                float wi = p->tw[2 * k * step + 1];
                if (sign > 0) wi = -wi;
                float *a = z + 2 * (base + k), *b = z + 2 * (base + k + half);
                float tr = b[0] * wr - b[1] * wi;
                float ti = b[0] * wi + b[1] * wr;
                b[0] = a[0] - tr;
                b[1] = a[1] - ti;
                a[0] = a[0] + tr;
                a[1] = a[1] + ti;
            }
        }
    }
}

/* r2c of N reals through one N/2-point complex FFT (even/odd packing). */
static void rfft_plan(const jfo_plan *ph /* n = N/2 */, const jfo_plan *pf /* n = N */,
                      const float *x, float *X, float *work /* N floats */) {
    const int N = pf->n, H = N / 2;
    memcpy(work, x, sizeof(float) * (size_t)N); /* z[n] = x[2n] + j x[2n+1] */
    cfft(ph, work, -1);
    for (int k = 0; k <= H; k++) {
        int a = k % H, b = (H - k) % H;
        float zr = work[2 * a], zi = work[2 * a + 1];
        float cr = work[2 * b], ci = -work[2 * b + 1]; /* conj Z[H-k] */
        float er = 0.5f * (zr + cr), ei = 0.5f * (zi + ci);
        /* O = -j (Z - conj Z')/2 */
        float orr = 0.5f * (zi - ci), oi = -0.5f * (zr - cr);
        float wr, wi;
        if (k < H) {
            wr = pf->tw[2 * k];
            wi = pf->tw[2 * k + 1];
        } else {
            wr = -1.0f;
            wi = 0.0f;
        }
        X[2 * k] = er + (orr * wr - oi * wi);
        X[2 * k + 1] = ei + (orr * wi + oi * wr);
    }
}

/* Two c2r transforms at once (cufftPlanMany C2R batch 2, ostride 2:
 * GPUSoundSource.cu:53-66; fftwf_plan_many_dft_c2r: CPUSoundSource.cpp:15-21):
 * z = yL + j yR from Z[k] = YL[k] + j YR[k] (Hermitian-extended), so the
 * complex output IS the interleaved stereo buffer. */
static void irfft2_plan(const jfo_plan *pf, const float *YL, const float *YR, float *z) {
    const int N = pf->n, H = N / 2;
    for (int k = 0; k <= H; k++) {
        float lr = YL[2 * k], li = YL[2 * k + 1];
        float rr = YR ? YR[2 * k] : 0.0f, ri = YR ? YR[2 * k + 1] : 0.0f;
        if (k == 0 || k == H) li = ri = 0.0f; /* c2r ignores these */
        z[2 * k] = lr - ri;
        z[2 * k + 1] = li + rr;
        if (k > 0 && k < H) {
            z[2 * (N - k)] = lr + ri;
            z[2 * (N - k) + 1] = -li + rr;
        }
    }
    cfft(pf, z, +1);
}

void jfo_rfft(const float *x, int N, float *X) {
    jfo_plan ph, pf;
    plan_init(&ph, N / 2);
    plan_init(&pf, N);
    float *work = (float *)malloc(sizeof(float) * (size_t)N);
    rfft_plan(&ph, &pf, x, X, work);
    free(work);
    plan_free(&ph);
    plan_free(&pf);
}

void jfo_irfft(const float *X, int N, float *y) {
    jfo_plan pf;
    plan_init(&pf, N);
    float *z = (float *)malloc(sizeof(float) * 2 * (size_t)N);
    irfft2_plan(&pf, X, NULL, z);
    for (int n = 0; n < N; n++) y[n] = z[2 * n];
    free(z);
    plan_free(&pf);
}

/* hrtf_signals.cu:107-153 */
void jfo_build_table(const float *hrir, int n_hrtf, int taps, int N, float *table) {
    jfo_plan ph, pf;
    plan_init(&ph, N / 2);
    plan_init(&pf, N);
    const int nc = N / 2 + 1;
    float *x = (float *)malloc(sizeof(float) * (size_t)N);
    float *work = (float *)malloc(sizeof(float) * (size_t)N);
    for (int r = 0; r < n_hrtf * 2; r++) {
        memset(x, 0, sizeof(float) * (size_t)N);
        memcpy(x, hrir + (size_t)r * taps, sizeof(float) * (size_t)(taps < N ? taps : N));
        rfft_plan(&ph, &pf, x, table + (size_t)r * nc * 2, work);
    }
    free(x);
    free(work);
    plan_free(&ph);
    plan_free(&pf);
}

/* --------------------------------------------------------------- engine -- */
typedef struct {
    float *buf; /* SoundSource.cuh:11 */
    int length, count;
    float ele, azi, r;
    float coords[3];
    float old_ele, old_azi;
    float *x;    /* CPUSoundSource.h:22, window of N samples */
    float *last; /* last 2*B output */
    /* convolution reverb ahead of the spatialiser (jfo_reverb_set_ir): frequency-domain delay line of the last P
     * input spectra (planar re / im, P x (B + 1) each, slot = block index mod P) and the previous dry block */
    float *rv_xr, *rv_xi, *rv_prev;
    int rv_head;
} jfo_source;

struct jfo_engine {
    int B, L, N, Nc, n_sources;
    int mode; /* bit 0: 0 = FD_COMPLEX (interpolated), 1 = FD_BASIC (nearest HRTF); bit 1: corrected index/weight rule */
    float *table; /* [710][2][Nc][2] ([grid.n_rows] with a grid of its own) */
    int has_grid; /* jfo_create_grid: the grid's rule and pick in every mode */
    jfo_grid grid;
    jfo_source *src;
    jfo_plan ph, pf;
    /* reverb stage: P partitions of B taps, spectra of [h_p, 0] (2B-point r2c, B + 1 bins, planar), pre-scaled by
     * gain / (2B) (the c2r is unnormalised) */
    int rv_P;
    float *rv_hr, *rv_hi;
    jfo_plan rv_ph, rv_pf; /* n = B and n = 2B */
};

static jfo_engine *create_rows(int B, int hrtf_len, int n_sources, int n_rows, const float *hrir, int taps);

jfo_engine *jfo_create(int B, int hrtf_len, int n_sources, const float *hrir, int taps) {
    return create_rows(B, hrtf_len, n_sources, JFO_NUM_HRTF, hrir, taps);
}

jfo_engine *jfo_create_grid(int B, int hrtf_len, int n_sources, int n_rings, const float *ring_ele,
                            const int *ring_count, const float *ring_step, const float *hrir, int taps) {
    jfo_grid g;
    if (grid_fill(&g, n_rings, ring_ele, ring_count, ring_step)) return NULL;
    jfo_engine *e = create_rows(B, hrtf_len, n_sources, g.n_rows, hrir, taps);
    if (e) {
        e->has_grid = 1;
        e->grid = g;
    }
    return e;
}

static jfo_engine *create_rows(int B, int hrtf_len, int n_sources, int n_rows, const float *hrir, int taps) {
    if (B <= 0 || hrtf_len <= 0 || n_sources <= 0 || taps > hrtf_len) return NULL;
    jfo_engine *e = (jfo_engine *)calloc(1, sizeof(*e));
    e->B = B;
    e->L = hrtf_len;
    /* Universal.cuh:12 */
    e->N = (int)pow(2, ceil(log2((double)(B + hrtf_len - 1))));
    e->Nc = e->N / 2 + 1;
    e->n_sources = n_sources;
    plan_init(&e->ph, e->N / 2);
    plan_init(&e->pf, e->N);
    e->table = (float *)malloc(sizeof(float) * (size_t)n_rows * 2 * e->Nc * 2);
    jfo_build_table(hrir, n_rows, taps, e->N, e->table);
    e->src = (jfo_source *)calloc((size_t)n_sources, sizeof(jfo_source));
    for (int s = 0; s < n_sources; s++) {
        jfo_source *q = &e->src[s];
        q->x = (float *)calloc((size_t)e->N + 2, sizeof(float));
        q->last = (float *)calloc((size_t)2 * B, sizeof(float));
        /* SoundSource.cu:3-16 */
        q->coords[0] = 0;
        q->coords[1] = 0;
        q->coords[2] = 0.5f;
        q->azi = 0;
        q->ele = 0;
        q->r = 0.5f;
        q->old_azi = 0;
        q->old_ele = 0;
    }
    return e;
}

static void reverb_free(jfo_engine *e);

void jfo_destroy(jfo_engine *e) {
    if (!e) return;
    for (int s = 0; s < e->n_sources; s++) {
        free(e->src[s].buf);
        free(e->src[s].x);
        free(e->src[s].last);
    }
    reverb_free(e);
    free(e->src);
    free(e->table);
    plan_free(&e->ph);
    plan_free(&e->pf);
    free(e);
}

int jfo_pad_len(const jfo_engine *e) { return e->N; }

void jfo_set_mode(jfo_engine *e, int mode) { e->mode = mode; }

int jfo_source_set_signal(jfo_engine *e, int s, const float *mono, int n) {
    if (s < 0 || s >= e->n_sources || n < 0) return -1;
    jfo_source *q = &e->src[s];
    free(q->buf);
    q->buf = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    memcpy(q->buf, mono, sizeof(float) * (size_t)n);
    q->length = n;
    q->count = 0;
    return 0;
}

int jfo_source_set_spherical(jfo_engine *e, int s, float ele, float azi, float r) {
    if (s < 0 || s >= e->n_sources) return -1;
    float o[5];
    jfo_from_spherical(ele, azi, r, o);
    jfo_source *q = &e->src[s];
    q->ele = o[0];
    q->azi = o[1];
    q->r = r;
    q->coords[0] = o[2];
    q->coords[1] = o[3];
    q->coords[2] = o[4];
    return 0;
}

int jfo_source_set_cartesian(jfo_engine *e, int s, float x, float y, float z) {
    if (s < 0 || s >= e->n_sources) return -1;
    float o[3];
    if (jfo_from_cartesian(x, y, z, o)) return -1;
    jfo_source *q = &e->src[s];
    q->coords[0] = x;
    q->coords[1] = y;
    q->coords[2] = z;
    q->ele = o[0];
    q->azi = o[1];
    q->r = o[2];
    return 0;
}

void jfo_source_reset(jfo_engine *e, int s) {
    jfo_source *q = &e->src[s];
    memset(q->x, 0, sizeof(float) * ((size_t)e->N + 2));
    q->count = 0;
    q->old_azi = 0.0f;
    q->old_ele = 0.0f;
    if (e->rv_P > 0) {
        const size_t nb = (size_t)e->rv_P * (size_t)(e->B + 1);
        memset(q->rv_xr, 0, sizeof(float) * nb);
        memset(q->rv_xi, 0, sizeof(float) * nb);
        memset(q->rv_prev, 0, sizeof(float) * (size_t)e->B);
        q->rv_head = 0;
    }
}

static size_t scratch_floats_spat(const jfo_engine *e) {
    return (size_t)e->Nc * 8 + (size_t)e->N * 5 + 16;
}

/* ------------------------------------------------------ convolution reverb --
 * The reference convolves the whole input file with a reverb impulse response before playback and
 * matches the result's RMS to the input's (cudaPart.cu:65-205: PadData :87-88, r2c of both :138-139,
 * pointwise product scaled by 1/new_size :146, c2r :153, rms / rms2 gain :118,161-165); the result
 * becomes the source's looped `buf` (:171-172).  That code is disabled (reverbFlag = false, :20) and
 * its kernel calls have swapped arguments (:146,165): what is restated here is
 * its intent.  Two forms:
 *   jfo_reverb_offline -- the reference's own whole-signal form: circular convolution of length
 *     new_size with the RMS gain, as cudaFFT would leave it in `buf`;
 *   jfo_reverb_set_ir  -- the same convolution as a stream ahead of the spatialiser (what a real-time
 *     engine has to do): every block of B dry samples goes through a uniformly partitioned
 *     overlap-save convolution (partitions of B taps, 2B-point transforms, a delay line of the last P
 *     input spectra) and the window of the spatialiser is fed the reverberated block.  Float32
 *     throughout, the products added in partition order p = 0 .. P-1. */
static void reverb_free(jfo_engine *e) {
    if (e->rv_P <= 0) return;
    for (int s = 0; s < e->n_sources; s++) {
        free(e->src[s].rv_xr);
        free(e->src[s].rv_xi);
        free(e->src[s].rv_prev);
        e->src[s].rv_xr = e->src[s].rv_xi = e->src[s].rv_prev = NULL;
    }
    free(e->rv_hr);
    free(e->rv_hi);
    e->rv_hr = e->rv_hi = NULL;
    plan_free(&e->rv_ph);
    plan_free(&e->rv_pf);
    e->rv_P = 0;
}

int jfo_reverb_set_ir(jfo_engine *e, const float *ir, int n_ir, float gain) {
    const int B = e->B, nb = B + 1;
    if (n_ir < 0 || (n_ir > 0 && !ir) || (B & (B - 1))) return -1;
    reverb_free(e);
    for (int s = 0; s < e->n_sources; s++) jfo_source_reset(e, s); /* as the engine does when the stage changes */
    if (n_ir == 0) return 0;
    const int P = (n_ir + B - 1) / B;
    plan_init(&e->rv_ph, B);
    plan_init(&e->rv_pf, 2 * B);
    e->rv_hr = (float *)calloc((size_t)P * nb, sizeof(float));
    e->rv_hi = (float *)calloc((size_t)P * nb, sizeof(float));
    float *x = (float *)malloc(sizeof(float) * 2 * (size_t)B);
    float *X = (float *)malloc(sizeof(float) * 2 * (size_t)nb);
    float *work = (float *)malloc(sizeof(float) * 2 * (size_t)B);
    const float scale = gain / (float)(2 * B);
    for (int p = 0; p < P; p++) {
        memset(x, 0, sizeof(float) * 2 * (size_t)B);
        for (int n = 0; n < B && p * B + n < n_ir; n++) x[n] = ir[p * B + n];
        rfft_plan(&e->rv_ph, &e->rv_pf, x, X, work);
        for (int k = 0; k < nb; k++) {
            e->rv_hr[(size_t)p * nb + k] = X[2 * k] * scale;
            e->rv_hi[(size_t)p * nb + k] = X[2 * k + 1] * scale;
        }
    }
    free(x);
    free(X);
    free(work);
    for (int s = 0; s < e->n_sources; s++) {
        jfo_source *q = &e->src[s];
        q->rv_xr = (float *)calloc((size_t)P * nb, sizeof(float));
        q->rv_xi = (float *)calloc((size_t)P * nb, sizeof(float));
        q->rv_prev = (float *)calloc((size_t)B, sizeof(float));
        q->rv_head = 0;
    }
    e->rv_P = P;
    return 0;
}

/* One block of one source through the stream form: blk[B] holds the dry block on entry, the reverberated block on
 * return.  scratch: 12 B + 16 floats. */
static void reverb_block(const jfo_engine *e, jfo_source *q, float *blk, float *scratch) {
    const int B = e->B, nb = B + 1, P = e->rv_P;
    float *in = scratch;             /* 2B: [previous dry block, this dry block] */
    float *X = in + 2 * B;           /* 2 nb */
    float *Y = X + 2 * nb;           /* 2 nb */
    float *z = Y + 2 * nb;           /* 4B: complex output of the c2r */
    float *work = z + 4 * B;         /* 2B */
    memcpy(in, q->rv_prev, sizeof(float) * (size_t)B);
    memcpy(in + B, blk, sizeof(float) * (size_t)B);
    memcpy(q->rv_prev, blk, sizeof(float) * (size_t)B);
    rfft_plan(&e->rv_ph, &e->rv_pf, in, X, work);
    float *xr = q->rv_xr + (size_t)q->rv_head * nb, *xi = q->rv_xi + (size_t)q->rv_head * nb;
    for (int k = 0; k < nb; k++) {
        xr[k] = X[2 * k];
        xi[k] = X[2 * k + 1];
    }
    float *restrict yr = Y, *restrict yi = Y + nb;
    for (int k = 0; k < nb; k++) yr[k] = yi[k] = 0.0f;
    int slot = q->rv_head;
    for (int p = 0; p < P; p++) { /* planar and restrict-qualified so that the compiler vectorises over the bins */
        const float *restrict ar = q->rv_xr + (size_t)slot * nb, *restrict ai = q->rv_xi + (size_t)slot * nb;
        const float *restrict hr = e->rv_hr + (size_t)p * nb, *restrict hi = e->rv_hi + (size_t)p * nb;
        for (int k = 0; k < nb; k++) {
            yr[k] += ar[k] * hr[k] - ai[k] * hi[k];
            yi[k] += ar[k] * hi[k] + ai[k] * hr[k];
        }
        slot = slot == 0 ? P - 1 : slot - 1;
    }
    q->rv_head = q->rv_head + 1 == P ? 0 : q->rv_head + 1;
    for (int k = 0; k < nb; k++) {
        X[2 * k] = yr[k];
        X[2 * k + 1] = yi[k];
    }
    irfft2_plan(&e->rv_pf, X, NULL, z);
    for (int n = 0; n < B; n++) blk[n] = z[2 * (B + n)]; /* overlap-save: the last B samples are valid */
}

/* cudaPart.cu:87-88 / kernels.cu:169-188: both signals zero-padded to new_size */
int jfo_reverb_padded_size(int n, int n_ir) { return n + (n_ir - n_ir / 2); }

/* cudaPart.cu:118 and :161: sqrt(sum(x^2) / new_size).  (thrust::transform_reduce adds floats in an unspecified
 * tree order; the sum is formed in double here and rounded once.) */
static float rms_of(const float *x, int n) {
    double acc = 0.0;
    for (int i = 0; i < n; i++) acc += (double)x[i] * (double)x[i];
    return (float)sqrt(acc / (double)n);
}

/* The whole-signal form, cudaPart.cu:87-172: out[new_size] = (rms / rms2) * c2r(r2c(x) * r2c(ir) / new_size), a
 * CIRCULAR convolution of length new_size (the tail of the linear convolution wraps onto the start).  new_size is
 * not a power of two in general and this oracle's FFT is radix-2, so the linear convolution is formed with a
 * power-of-two transform and folded -- the same numbers up to float32 rounding.  Returns rms / rms2 (1 if either is 0). */
float jfo_reverb_offline(const float *x, int n, const float *ir, int n_ir, float *out) {
    if (n <= 0 || n_ir <= 0 || n > (1 << 28) || n_ir > (1 << 28)) return 1.0f;
    const int new_size = jfo_reverb_padded_size(n, n_ir);
    int m = 1;
    while (m < n + n_ir) m <<= 1;
    jfo_plan ph, pf;
    plan_init(&ph, m / 2);
    plan_init(&pf, m);
    float *a = (float *)calloc((size_t)m, sizeof(float));
    float *b = (float *)calloc((size_t)m, sizeof(float));
    float *A = (float *)malloc(sizeof(float) * ((size_t)m + 2));
    float *Bs = (float *)malloc(sizeof(float) * ((size_t)m + 2));
    float *work = (float *)malloc(sizeof(float) * (size_t)m);
    float *z = (float *)malloc(sizeof(float) * 2 * (size_t)m);
    memcpy(a, x, sizeof(float) * (size_t)n);
    memcpy(b, ir, sizeof(float) * (size_t)n_ir);
    rfft_plan(&ph, &pf, a, A, work);
    rfft_plan(&ph, &pf, b, Bs, work);
    const float scale = 1.0f / (float)m;
    for (int k = 0; k <= m / 2; k++) { /* kernels.cu:44-53: (a * b) * scale */
        const float cr = A[2 * k] * Bs[2 * k] - A[2 * k + 1] * Bs[2 * k + 1];
        const float ci = A[2 * k] * Bs[2 * k + 1] + A[2 * k + 1] * Bs[2 * k];
        A[2 * k] = scale * cr;
        A[2 * k + 1] = scale * ci;
    }
    irfft2_plan(&pf, A, NULL, z);
    for (int i = 0; i < new_size; i++) out[i] = 0.0f;
    for (int i = 0; i < n + n_ir - 1; i++) out[i % new_size] += z[2 * i];
    double e_in = 0.0; /* the padded signal's rms: its new_size - n zeros add nothing to the sum */
    for (int i = 0; i < n; i++) e_in += (double)x[i] * (double)x[i];
    const float rms = (float)sqrt(e_in / (double)new_size), rms2 = rms_of(out, new_size);
    float g = 1.0f;
    if (rms > 0.0f && rms2 > 0.0f) g = rms / rms2;
    for (int i = 0; i < new_size; i++) out[i] *= g; /* cudaPart.cu:165 */
    free(a);
    free(b);
    free(A);
    free(Bs);
    free(work);
    free(z);
    plan_free(&ph);
    plan_free(&pf);
    return g;
}

/* One filter set: Y = sum_i ((X * H_i) * w_i) * D, then both c2r.
 * GPUSoundSource.cu:118-292 (operand order), CPUSoundSource.cpp:244-253 (sum
 * order), kernels.cu:199-205 (complex multiply). */
static void filter_set(const jfo_engine *e, const float *X, const float *D, int nterms,
                       const int *rows, const float *w, float *Y /* 2*Nc*2 */,
                       float *z /* 2N */) {
    const int Nc = e->Nc;
    for (int ear = 0; ear < 2; ear++) {
        float *Yo = Y + (size_t)ear * Nc * 2;
        for (int k = 0; k < Nc; k++) {
            float accr = 0.0f, acci = 0.0f;
            const float ar = X[2 * k], ai = X[2 * k + 1];
            const float dr = D[2 * k], di = D[2 * k + 1];
            for (int t = 0; t < nterms; t++) {
                const float *H = e->table + ((size_t)rows[t] * 2 + ear) * Nc * 2;
                float br = H[2 * k], bi = H[2 * k + 1];
                float cr = ar * br - ai * bi;
                float ci = ar * bi + ai * br;
                if (nterms > 1) { /* case 1 has no scale kernel */
                    cr = w[t] * cr;
                    ci = w[t] * ci;
                }
                float pr = cr * dr - ci * di;
                float pi = cr * di + ci * dr;
                accr += pr;
                acci += pi;
            }
            Yo[2 * k] = accr;
            Yo[2 * k + 1] = acci;
        }
    }
    irfft2_plan(&e->pf, Y, Y + (size_t)Nc * 2, z);
}

/* One source, one block: Audio.cu:118-157 around CPUSoundSource.cpp:274-339 /
 * GPUSoundSource.cu:320-385.  Returns -1 for a position the reference cannot
 * interpolate (missing ring); the block is then silence. */
static int source_block(const jfo_engine *e, jfo_source *q, float ele, float azi,
                        const float coords[3], float *blk, float *scratch) {
    const int N = e->N, B = e->B, Nc = e->Nc;
    float *X = scratch;                 /* Nc*2 */
    float *D = X + (size_t)Nc * 2;      /* Nc*2 */
    float *Y = D + (size_t)Nc * 2;      /* 2*Nc*2 */
    float *z1 = Y + (size_t)Nc * 4;     /* 2N */
    float *z2 = z1 + (size_t)2 * N;     /* 2N */
    float *work = z2 + (size_t)2 * N;   /* N */
    int rc = 0;

    /* Audio.cu:121-139 feed with wrap */
    float *dst = q->x + (N - B);
    if (q->length <= 0) {
        memset(dst, 0, sizeof(float) * (size_t)B);
    } else {
        int n = 0;
        while (n < B) {
            int chunk = q->length - q->count;
            if (chunk > B - n) chunk = B - n;
            memcpy(dst + n, q->buf + q->count, sizeof(float) * (size_t)chunk);
            q->count += chunk;
            if (q->count >= q->length) q->count = 0;
            n += chunk;
        }
    }
    /* reverb on: the B samples just fed are the DRY block; the window gets the reverberated block instead */
    if (e->rv_P > 0) reverb_block(e, q, dst, scratch + scratch_floats_spat(e));

    /* CPUSoundSource.cpp:279-280 / GPUSoundSource.cu:344-346 */
    rfft_plan(&e->ph, &e->pf, q->x, X, work);
    const float scale = 1.0f / (float)N;
    for (int i = 0; i < 2 * Nc; i++) X[i] *= scale;

    int idx[4], rows[4], oidx[4], orows[4];
    float om[6], w[4], oom[6], ow[4];
    int nt = 0, ont = 0;
    if (e->mode & 1) {
        /* CPU_FD_BASIC (CPUSoundSource.cpp:50-52,113-142): nearest table row, no interpolation,
         * no distance factor, no crossfade */
        rows[0] = e->has_grid ? grid_pick(&e->grid, ele, azi) : jfo_pick_hrtf(ele, azi);
        w[0] = 1.0f;
        for (int k = 0; k < Nc; k++) {
            D[2 * k] = 1.0f;
            D[2 * k + 1] = 0.0f;
        }
        filter_set(e, X, D, 1, rows, w, Y, z1);
        memcpy(blk, z1 + 2 * (N - B), sizeof(float) * 2 * (size_t)B);
        q->old_azi = azi;
        q->old_ele = ele;
        memmove(q->x, q->x + B, sizeof(float) * (size_t)(N - B));
        return 0;
    }
    int (*rule)(float, float, int *, float *) = (e->mode & 2) ? jfo_interp_corrected : jfo_interp;
    int xfade = (q->old_azi != azi || q->old_ele != ele);
    if (e->has_grid) {
        if (grid_interp(&e->grid, ele, azi, idx, om)) rc = -1;
        if (xfade && grid_interp(&e->grid, q->old_ele, q->old_azi, oidx, oom)) rc = -1;
    } else {
        if (rule(ele, azi, idx, om)) rc = -1;
        if (xfade && rule(q->old_ele, q->old_azi, oidx, oom)) rc = -1;
    }
    if (rc == 0) {
        nt = jfo_terms(idx, om, rows, w);
        if (xfade) ont = jfo_terms(oidx, oom, orows, ow);
        jfo_distance_factor(coords[0], coords[1], coords[2], Nc, D);
        if (!xfade) {
            filter_set(e, X, D, nt, rows, w, Y, z1);
        } else {
            filter_set(e, X, D, ont, orows, ow, Y, z1);
            filter_set(e, X, D, nt, rows, w, Y, z2);
            /* kernels.cu:132-137 */
            float *o1 = z1 + 2 * (N - B), *o2 = z2 + 2 * (N - B);
            for (int i = 0; i < B; i++) {
                float fn = (float)i / (B - 1.0f);
                o1[2 * i] = o1[2 * i] * (1.0f - fn) + o2[2 * i] * fn;
                o1[2 * i + 1] = o1[2 * i + 1] * (1.0f - fn) + o2[2 * i + 1] * fn;
            }
        }
        memcpy(blk, z1 + 2 * (N - B), sizeof(float) * 2 * (size_t)B);
    } else {
        memset(blk, 0, sizeof(float) * 2 * (size_t)B);
    }
    q->old_azi = azi;
    q->old_ele = ele;
    /* Audio.cu:153-157 */
    memmove(q->x, q->x + B, sizeof(float) * (size_t)(N - B));
    return rc;
}

static size_t scratch_floats(const jfo_engine *e) {
    /* the spatialiser's part + the reverb stage's: input 2B, spectrum 2(B+1), sums 2(B+1), c2r output 4B, work 2B */
    return scratch_floats_spat(e) + (size_t)12 * (size_t)e->B + 16;
}

/* Audio.cu:94-163 */
void jfo_process_block(jfo_engine *e, float *out) {
    const int B = e->B;
    float *scratch = (float *)malloc(sizeof(float) * scratch_floats(e));
    for (int i = 0; i < 2 * B; i++) out[i] = 0.0f;
    for (int s = 0; s < e->n_sources; s++) {
        jfo_source *q = &e->src[s];
        source_block(e, q, q->ele, q->azi, q->coords, q->last, scratch);
        for (int i = 0; i < 2 * B; i++) out[i] += q->last[i];
    }
    free(scratch);
}

const float *jfo_source_last_block(const jfo_engine *e, int s) { return e->src[s].last; }

int jfo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void jfo_process_batch(jfo_engine *e, int n_blocks, const float *pos, float *out_mix,
                       float *out_partial, int n_threads) {
    const int B = e->B, S = e->n_sources;
    const size_t blk = (size_t)2 * B;
    float *partial = out_partial;
    if (!partial) partial = (float *)malloc(sizeof(float) * (size_t)S * n_blocks * blk);
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#pragma omp parallel num_threads(n_threads)
#endif
    {
        float *scratch = (float *)malloc(sizeof(float) * scratch_floats(e));
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int s = 0; s < S; s++) {
            jfo_source *q = &e->src[s];
            for (int b = 0; b < n_blocks; b++) {
                const float *p = pos + ((size_t)b * S + s) * 5;
                q->ele = p[0];
                q->azi = p[1];
                q->coords[0] = p[2];
                q->coords[1] = p[3];
                q->coords[2] = p[4];
                source_block(e, q, q->ele, q->azi, q->coords,
                             partial + ((size_t)s * n_blocks + b) * blk, scratch);
            }
            if (n_blocks > 0)
                memcpy(q->last, partial + ((size_t)s * n_blocks + n_blocks - 1) * blk,
                       sizeof(float) * blk);
        }
        free(scratch);
    }
    (void)n_threads;
    /* Audio.cu:109-110: plain sum in source order */
    for (int b = 0; b < n_blocks; b++) {
        float *o = out_mix + (size_t)b * blk;
        for (size_t i = 0; i < blk; i++) o[i] = 0.0f;
        for (int s = 0; s < S; s++) {
            const float *p = partial + ((size_t)s * n_blocks + b) * blk;
            for (size_t i = 0; i < blk; i++) o[i] += p[i];
        }
    }
    if (!out_partial) free(partial);
}
