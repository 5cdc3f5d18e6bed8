// jf_sofa.cpp -- HRTF sets in SOFA files (AES69; include/jefferson.h: jf_sofa_*): the variables of a SimpleFreeFieldHRIR-shaped
// file out of its HDF5 container (jf_hdf5.c) and the set as a table on elevation rings for jf_engine_create_grid.
// The reference has no such reader (its TODO, FuturePlans.md:21); the checks on receivers and sampling rate are its loader's
// own (hrtf_signals.cu:68-75).  Host code, runs once per engine.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <memory>
#include <string>
#include <vector>

#include "../../include/jefferson.h"
#include "jf_hdf5.h"
#include "jf_host.h"

namespace jf {

namespace {

struct H5Closer {
    void operator()(jf_h5 *f) const { jf_h5_close(f); }
};
struct Freer {
    void operator()(void *p) const { free(p); }
};
using DoubleBuf = std::unique_ptr<double, Freer>;

// a numeric variable of the root group; dims[0..rank); -1 with *err, 1 when absent
int read_var(jf_h5 *f, const char *name, DoubleBuf *data, int *rank, uint64_t *dims, std::string *err) {
    uint64_t addr = 0;
    const int q = jf_h5_lookup(f, name, &addr);
    if (q == 1) return 1;
    double *p = nullptr;
    if (q < 0 || jf_h5_read_f64(f, addr, rank, dims, &p)) {
        *err = std::string(name) + ": " + jf_h5_error(f);
        return -1;
    }
    data->reset(p);
    return 0;
}

template <typename T>
T *alloc_n(size_t n) {
    return static_cast<T *>(calloc(n ? n : 1, sizeof(T)));
}

}  // namespace

void sofa_release(jf_sofa_set *s) {
    if (!s) return;
    free(s->ir);
    free(s->azimuth);
    free(s->elevation);
    free(s->distance);
    free(s->delay);
    memset(s, 0, sizeof *s);
}

int sofa_read(const char *path, jf_sofa_set *out, std::string *err) {
    memset(out, 0, sizeof *out);
    char text[256] = "";
    jf_h5 *raw = nullptr;
    if (jf_h5_open(path, &raw, text, sizeof text)) {
        *err = text;
        return JF_ERR_IO;
    }
    std::unique_ptr<jf_h5, H5Closer> f(raw);
    const uint64_t root = jf_h5_root(f.get());
    char val[64] = "";
    int q = jf_h5_attr_str(f.get(), root, "DataType", val, sizeof val);
    if (q < 0) {
        *err = jf_h5_error(f.get());
        return JF_ERR_IO;
    }
    if (q == 0 && strcmp(val, "FIR") != 0) {
        *err = std::string(path) + ": DataType \"" + val + "\" (impulse responses, \"FIR\", are what the engine takes)";
        return JF_ERR_IO;
    }
    if (jf_h5_attr_str(f.get(), root, "SOFAConventions", out->conventions, sizeof out->conventions) != 0) out->conventions[0] = 0;

    DoubleBuf ir, pos, rate, delay;
    int rank = 0;
    uint64_t d[JF_H5_MAXRANK] = {0};
    q = read_var(f.get(), "Data.IR", &ir, &rank, d, err);
    if (q) {
        if (q == 1) *err = std::string(path) + ": no Data.IR";
        return JF_ERR_IO;
    }
    if (rank != 3 || d[0] < 1 || d[1] < 1 || d[2] < 1 || d[0] > (1u << 20) || d[1] > 64 || d[2] > (1u << 20)) {
        *err = std::string(path) + ": Data.IR is not [M][R][N]";
        return JF_ERR_IO;
    }
    const size_t M = d[0], R = d[1], N = d[2];
    q = read_var(f.get(), "SourcePosition", &pos, &rank, d, err);
    if (q) {
        if (q == 1) *err = std::string(path) + ": no SourcePosition";
        return JF_ERR_IO;
    }
    if (rank != 2 || d[0] != M || d[1] != 3) {
        *err = std::string(path) + ": SourcePosition is not [M][3] for Data.IR's M";
        return JF_ERR_IO;
    }
    bool cartesian = false;
    uint64_t pa = 0;
    if (jf_h5_lookup(f.get(), "SourcePosition", &pa) == 0) {
        if (jf_h5_attr_str(f.get(), pa, "Type", val, sizeof val) == 0) {
            if (strcmp(val, "cartesian") == 0) cartesian = true;
            else if (strcmp(val, "spherical") != 0) {
                *err = std::string(path) + ": SourcePosition of Type \"" + val + "\"";
                return JF_ERR_IO;
            }
        }
        if (!cartesian && jf_h5_attr_str(f.get(), pa, "Units", val, sizeof val) == 0 && strncmp(val, "degree", 6) != 0) {
            *err = std::string(path) + ": spherical SourcePosition in \"" + val + "\" (degrees are what SOFA prescribes)";
            return JF_ERR_IO;
        }
    }
    q = read_var(f.get(), "Data.SamplingRate", &rate, &rank, d, err);
    if (q) {
        if (q == 1) *err = std::string(path) + ": no Data.SamplingRate";
        return JF_ERR_IO;
    }
    size_t n_rate = 1;
    for (int i = 0; i < rank; i++) n_rate *= d[i];
    if (rank > 1) {
        *err = std::string(path) + ": Data.SamplingRate is neither a scalar nor [I] / [M]";
        return JF_ERR_IO;
    }
    if (n_rate < 1) {
        *err = std::string(path) + ": empty Data.SamplingRate";
        return JF_ERR_IO;
    }
    for (size_t i = 1; i < n_rate; i++)
        if (rate.get()[i] != rate.get()[0]) {
            *err = std::string(path) + ": measurements at different sampling rates";
            return JF_ERR_IO;
        }
    q = read_var(f.get(), "Data.Delay", &delay, &rank, d, err);
    if (q < 0) return JF_ERR_IO;
    size_t delay_rows = 0;
    if (q == 0) {
        if (rank != 2 || d[1] != R || (d[0] != 1 && d[0] != M)) {
            *err = std::string(path) + ": Data.Delay is neither [1][R] nor [M][R]";
            return JF_ERR_IO;
        }
        delay_rows = d[0];
    }

    out->n_measurements = (int)M;
    out->n_receivers = (int)R;
    out->n_samples = (int)N;
    out->sample_rate = rate.get()[0];
    out->ir = alloc_n<float>(M * R * N);
    out->azimuth = alloc_n<float>(M);
    out->elevation = alloc_n<float>(M);
    out->distance = alloc_n<float>(M);
    out->delay = alloc_n<float>(M * R);
    if (!out->ir || !out->azimuth || !out->elevation || !out->distance || !out->delay) {
        sofa_release(out);
        *err = "out of host memory";
        return JF_ERR_NOMEM;
    }
    for (size_t i = 0; i < M * R * N; i++) out->ir[i] = (float)ir.get()[i];
    for (size_t i = 0; i < M; i++) {
        const double *p = pos.get() + 3 * i;
        double az = p[0], el = p[1], r = p[2];
        if (cartesian) {  // x to the front, y to the left, z up
            r = sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
            az = atan2(p[1], p[0]) * (180.0 / M_PI);
            el = atan2(p[2], sqrt(p[0] * p[0] + p[1] * p[1])) * (180.0 / M_PI);
        }
        az = fmod(az, 360.0);
        if (az < 0) az += 360.0;
        out->azimuth[i] = (float)az;
        out->elevation[i] = (float)el;
        out->distance[i] = (float)r;
        for (size_t rcv = 0; rcv < R; rcv++)
            out->delay[i * R + rcv] = delay_rows ? (float)delay.get()[(delay_rows == 1 ? 0 : i) * R + rcv] : 0.0f;
    }
    return JF_OK;
}

static bool sofa_ok(const jf_sofa_set *s) {
    return s && s->n_measurements > 0 && s->n_receivers > 0 && s->n_samples > 0 && s->ir && s->azimuth && s->elevation && s->delay;
}

// N + the largest whole delay; a JF_ERR_* code for delays that are negative or fractional
int sofa_taps(const jf_sofa_set *s, std::string *err) {
    if (!sofa_ok(s)) return JF_ERR_ARG;
    float worst = 0;
    const size_t n = (size_t)s->n_measurements * s->n_receivers;
    for (size_t i = 0; i < n; i++) {
        const float v = s->delay[i];
        if (!(v >= 0) || v > 65536 || fabsf(v - rintf(v)) > 1e-3f) {
            if (err) *err = "Data.Delay holds negative or fractional delays: only whole samples are applied";
            return JF_ERR_IO;
        }
        if (v > worst) worst = v;
    }
    return s->n_samples + (int)rintf(worst);
}

int sofa_table(const jf_sofa_set *s, float tol_deg, jf_grid_layout *layout, float *hrir, int taps, std::string *err) {
    if (!sofa_ok(s) || !layout || !hrir) {
        *err = "null set, layout or table";
        return JF_ERR_ARG;
    }
    if (s->n_receivers != 2) {
        *err = "a set of " + std::to_string(s->n_receivers) + " receivers (two ears are what the engine renders)";
        return JF_ERR_IO;
    }
    if (s->sample_rate != 44100.0) {
        *err = "a set sampled at " + std::to_string(s->sample_rate) + " Hz (44100 is what the engine is written for: hrtf_signals.cu:68-75)";
        return JF_ERR_IO;
    }
    const int need = sofa_taps(s, err);
    if (need < 0) return need;
    if (taps < need) {
        *err = "the set needs " + std::to_string(need) + " taps per impulse response, the table has " + std::to_string(taps);
        return JF_ERR_ARG;
    }
    const size_t M = (size_t)s->n_measurements, N = (size_t)s->n_samples;
    std::vector<float> az(M);
    std::vector<int> row_of(M);
    for (size_t i = 0; i < M; i++) {
        float a = 360.0f - s->azimuth[i];  // counter-clockwise -> the table's clockwise sense
        if (a >= 360.0f) a -= 360.0f;
        az[i] = a;
    }
    const int rc = host_grid_from_positions(M, az.data(), s->elevation, tol_deg, &layout->n_rings, layout->ring_elevation,
                                            layout->ring_count, layout->ring_step, row_of.data(), err);
    if (rc) return rc;
    // KEMAR's own rings (-40 .. 90 in tens, 56 60 72 ... 1 measurements): the reference's description of them, its ROUNDED
    // steps included (hrtf_signals.cu:8), so that jf_engine_create_grid recognises the grid and the engine IS
    // jf_engine_create -- the reference's rule, the reference's blocks -- whether the set came as WAV files or as a SOFA file
    const RingTable &k = ring_table();
    bool kemar = layout->n_rings == k.n_rings;
    for (int r = 0; kemar && r < k.n_rings; r++)
        kemar = layout->ring_elevation[r] == k.ele[r] && layout->ring_count[r] == k.offset[r + 1] - k.offset[r];
    if (kemar)
        for (int r = 0; r < k.n_rings; r++) layout->ring_step[r] = kemar_ring_steps()[r];
    memset(hrir, 0, sizeof(float) * M * 2 * (size_t)taps);
    for (size_t i = 0; i < M; i++)
        for (int ear = 0; ear < 2; ear++) {
            const size_t shift = (size_t)rintf(s->delay[i * 2 + ear]);
            memcpy(hrir + ((size_t)row_of[i] * 2 + ear) * (size_t)taps + shift, s->ir + (i * 2 + ear) * N, sizeof(float) * N);
        }
    return JF_OK;
}

int hdf5_read(const char *path, const char *dataset, double **out, int *rank, unsigned long long *dims, std::string *err) {
    char text[256] = "";
    jf_h5 *raw = nullptr;
    if (jf_h5_open(path, &raw, text, sizeof text)) {
        *err = text;
        return JF_ERR_IO;
    }
    std::unique_ptr<jf_h5, H5Closer> f(raw);
    uint64_t addr = 0, d[JF_H5_MAXRANK] = {0};
    const int q = jf_h5_lookup(f.get(), dataset, &addr);
    if (q == 1) {
        *err = std::string(dataset) + ": no such object";
        return JF_ERR_ARG;
    }
    if (q < 0 || jf_h5_read_f64(f.get(), addr, rank, d, out)) {
        *err = jf_h5_error(f.get());
        return JF_ERR_IO;
    }
    for (int i = 0; i < JF_H5_MAXRANK; i++) dims[i] = i < *rank ? d[i] : 0;
    return JF_OK;
}

int hdf5_attr(const char *path, const char *object, const char *attr, char *out, size_t cap, std::string *err) {
    char text[256] = "";
    jf_h5 *raw = nullptr;
    if (jf_h5_open(path, &raw, text, sizeof text)) {
        *err = text;
        return JF_ERR_IO;
    }
    std::unique_ptr<jf_h5, H5Closer> f(raw);
    uint64_t addr = 0;
    int q = jf_h5_lookup(f.get(), object, &addr);
    if (q == 0) q = jf_h5_attr_str(f.get(), addr, attr, out, cap);
    if (q < 0) {
        *err = jf_h5_error(f.get());
        return JF_ERR_IO;
    }
    if (q == 1) {
        *err = std::string(object) + ": no string attribute " + attr;
        return JF_ERR_ARG;
    }
    return JF_OK;
}

}  // namespace jf
