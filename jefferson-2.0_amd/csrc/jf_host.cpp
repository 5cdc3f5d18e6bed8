// jf_host.cpp -- geometry / index rules / file I/O of the engine (host, no GPU).
// Float32 arithmetic is kept operation-for-operation as the reference performs
// it, because azimuth/elevation rounding and the interpolation weights are part
// of the audible result.  Compile with -ffp-contract=off.
#include "jf_host.h"

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/jefferson.h"

namespace jf {

namespace {
constexpr double kPi = 3.14159265358979323846264338327950288;  // Universal.cuh:14-16
const int kElevPos[kNumElev] = {-40, -30, -20, -10, 0, 10, 20, 30, 40, 50, 60, 70, 80, 90};
const float kAziInc[kNumElev] = {6.43f, 6.00f, 5.00f, 5.00f, 5.00f, 5.00f, 5.00f,
                                 6.00f, 6.43f, 8.00f, 10.00f, 15.00f, 30.00f, 361.0f};

struct Rings {
    RingTable rt;
    int pos_ele[kNumHrtf];
    int pos_azi[kNumHrtf];
    Rings() {
        int j = 0;
        rt = RingTable{};
        rt.n_rings = kNumElev;
        rt.kemar = 1;
        rt.offset[0] = 0;
        for (int e = 0; e < kNumElev; e++) {
            rt.inc[e] = kAziInc[e];
            rt.ele[e] = (float)kElevPos[e];
            // the reference steps a float azimuth; ring sizes follow from that
            for (float azi = 0; azi < 360; azi += kAziInc[e]) {
                if (j < kNumHrtf) {
                    pos_ele[j] = kElevPos[e];
                    pos_azi[j] = (int)round(azi);
                }
                j++;
            }
            rt.offset[e + 1] = j;
        }
        for (int e = kNumElev; e < kMaxRings; e++) rt.offset[e + 1] = j;
        rt.n_rows = j;
        rt.pick = nullptr;
    }
};
const Rings &rings() {
    static const Rings r;
    return r;
}
}  // namespace

const RingTable &ring_table() { return rings().rt; }
const float *kemar_ring_steps() { return kAziInc; }

// include/jefferson.h: jf_hrtf_grid -> the table the kernels and the host rules work from.  The reference's own grid
// (14 rings at -40 .. 90, its counts, its rounded steps) is recognised and becomes ring_table() itself (kemar = 1: the
// reference's rule applies by default); anything else is a general grid.
int host_grid_table(int n_rings, const float *ring_ele, const int *ring_count, const float *ring_step, RingTable *out,
                    std::string *err) {
    auto bad = [&](const char *m) {
        if (err) *err = m;
        return JF_ERR_ARG;
    };
    if (!ring_ele || !ring_count || !out) return bad("null grid description");
    if (n_rings < 1 || n_rings > kMaxRings) return bad("a grid has 1 .. JF_MAX_RINGS elevation rings");
    RingTable rt{};
    rt.n_rings = n_rings;
    long long rows = 0;
    for (int r = 0; r < n_rings; r++) {
        const float el = ring_ele[r];
        if (!(el >= -90.0f && el <= 90.0f)) return bad("ring elevations lie in [-90, 90] degrees");
        if (r > 0 && !(el > ring_ele[r - 1])) return bad("ring elevations must ascend");
        const int n = ring_count[r];
        if (n < 1 || n > 3600) return bad("a ring has 1 .. 3600 measurements");
        float step = ring_step ? ring_step[r] : 360.0f / (float)n;
        if (n == 1 && !(step >= 360.0f)) step = 361.0f;  // one measurement: it is the ring (hrtf_signals.cu:8 writes 361 too)
        // n steps must cover the circle and n - 1 must not: the ring's last measurement lies below 360 degrees
        if (!(step > 0.0f) || (n > 1 && (!((float)(n - 1) * step < 360.0f) || !((float)n * step >= 359.0f))))
            return bad("a ring's azimuth step does not fit its count (measurement i sits at i * step, i < count, below 360)");
        rt.ele[r] = el;
        rt.inc[r] = step;
        rt.offset[r] = (int)rows;
        rows += n;
    }
    if (rows > 32000) return bad("more than 32000 table rows");
    for (int r = n_rings; r <= kMaxRings; r++) rt.offset[r] = (int)rows;
    rt.n_rows = (int)rows;
    const RingTable &k = ring_table();
    bool same = n_rings == k.n_rings;
    for (int r = 0; same && r < n_rings; r++)
        same = rt.ele[r] == k.ele[r] && rt.inc[r] == k.inc[r] && rt.offset[r + 1] == k.offset[r + 1];
    *out = same ? k : rt;
    return JF_OK;
}

// include/jefferson.h: jf_grid_from_positions -- the rings of a set from the directions of its measurements
int host_grid_from_positions(size_t n, const float *azi, const float *ele, float tol, int *n_rings, float *ring_ele,
                             int *ring_count, float *ring_step, int *row_of, std::string *err) {
    auto bad = [&](const std::string &m) {
        if (err) *err = m;
        return JF_ERR_ARG;
    };
    if (!azi || !ele || !n_rings || !ring_ele || !ring_count || !ring_step || !row_of) return bad("null argument");
    if (n == 0 || n > 32000) return bad("1 .. 32000 measurements");
    if (!(tol >= 0.0f && tol <= 5.0f)) return bad("tolerance outside [0, 5] degrees");
    std::vector<float> a(n);
    std::vector<int> idx(n);
    for (size_t i = 0; i < n; i++) {
        if (!(ele[i] >= -90.0f && ele[i] <= 90.0f) || !(azi[i] > -1.0e6f && azi[i] < 1.0e6f))
            return bad("measurement " + std::to_string(i) + ": elevation outside [-90, 90] or azimuth not finite");
        float f = azi[i] - 360.0f * floorf(azi[i] / 360.0f);
        if (!(f < 360.0f)) f = 0.0f;
        if (360.0f - f <= tol) f -= 360.0f;  // 359.99 is azimuth 0 of its ring
        a[i] = f;
        idx[i] = (int)i;
    }
    std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) { return ele[x] < ele[y]; });
    int rings = 0, rows = 0;
    for (size_t lo = 0; lo < n;) {
        size_t hi = lo + 1;
        while (hi < n && ele[idx[hi]] - ele[idx[lo]] <= tol) hi++;
        if (rings == kMaxRings) return bad("more than JF_MAX_RINGS elevation rings");
        const int cnt = (int)(hi - lo);
        std::stable_sort(idx.begin() + lo, idx.begin() + hi, [&](int x, int y) { return a[x] < a[y]; });
        const float step = 360.0f / (float)cnt;
        for (int i = 0; i < cnt; i++) {
            const float want = (float)i * step, got = a[idx[lo + i]];
            if (cnt > 1 && !(fabsf(got - want) <= tol))
                return bad("ring at elevation " + std::to_string(ele[idx[lo]]) + ": measurement " + std::to_string(idx[lo + i]) +
                           " at azimuth " + std::to_string(got) + " where a ring of " + std::to_string(cnt) + " from azimuth 0 has " +
                           std::to_string(want));
            row_of[idx[lo + i]] = rows + i;
        }
        ring_ele[rings] = ele[idx[lo]];
        ring_count[rings] = cnt;
        ring_step[rings] = cnt == 1 ? 361.0f : step;
        rows += cnt;
        rings++;
        lo = hi;
    }
    for (int r = 1; r < rings; r++)
        if (!(ring_ele[r] > ring_ele[r - 1])) return bad("two rings at one elevation");
    *n_rings = rings;
    return JF_OK;
}

// The nearest measurement of a general grid (twin of dev_grid_pick).
int host_grid_pick(const RingTable &rt, float ele, float azi) {
    if (rt.kemar) return host_pick_hrtf(ele, azi);
    float dmin = 1e37f;
    int ring = 0;
    for (int r = 0; r < rt.n_rings; r++) {
        float d = ele - rt.ele[r];
        if (d < 0) d = -d;
        if (d < dmin) {
            dmin = d;
            ring = r;
        }
    }
    const int n = rt.offset[ring + 1] - rt.offset[ring];
    float a = azi - 360.0f * floorf(azi / 360.0f);
    if (!(a < 360.0f)) a = 0.0f;
    int i = (int)floorf(a / rt.inc[ring] + 0.5f);
    if (i >= n) i = 0;
    return rt.offset[ring] + i;
}

void table_position(int j, int *ele, int *azi) {
    *ele = rings().pos_ele[j];
    *azi = rings().pos_azi[j];
}

int host_pick_hrtf(float obj_ele, float obj_azi) {
    const RingTable &rt = ring_table();
    obj_ele = roundf(obj_ele / 10) * 10;
    int ring = 0;
    float best = 1e37f;
    for (int e = 0; e < kNumElev; e++) {
        float d = obj_ele - kElevPos[e];
        if (d < 0) d = -d;
        if (d < best) {
            best = d;
            ring = e;
        }
    }
    obj_azi = roundf(obj_azi);
    best = 1e37f;
    int pick = 0;
    const int n = rt.offset[ring + 1] - rt.offset[ring];
    for (int i = 0; i < n; i++) {
        float d = obj_azi - i * rt.inc[ring];
        if (d < 0) d = -d;
        if (d < best) {
            best = d;
            pick = rt.offset[ring] + i;
        }
    }
    return pick;
}

int host_interpolation(float ele, float azi, int idx[4], float omegas[6]) {
    // (-50, 91): where both truncated elevations name a measured ring.  The setters' whole degrees end at 90; a latched record
    // may carry 90.x, which the reference's statements place on the 90-degree ring twice (weights 0.x and -0.x)
    if (!(ele > -50.0f && ele < 91.0f) || !(azi > -1.0e6f && azi < 1.0e6f)) return JF_ERR_RANGE;
    const int phi0 = (int)ele / 10 * 10;
    const int phi1 = (int)(ele + 9) / 10 * 10;
    int r0 = -1, r1 = -1;
    for (int e = 0; e < kNumElev; e++) {
        if (kElevPos[e] == phi0) r0 = e;
        if (kElevPos[e] == phi1) r1 = e;
    }
    if (r0 < 0 || r1 < 0) return JF_ERR_RANGE;
    const float dt1 = kAziInc[r0], dt2 = kAziInc[r1];
    const int th0 = (int)((int)(azi / dt1) * dt1);
    const int th1 = (int)((int)((azi + dt1 - 1) / dt1) * dt1);
    const int th2 = (int)((int)(azi / dt2) * dt2);
    const int th3 = (int)((int)((azi + dt2 - 1) / dt2) * dt2);
    omegas[0] = (azi - th0) / dt1;
    omegas[1] = (th1 - azi) / dt1;
    omegas[2] = (azi - th2) / dt2;
    omegas[3] = (th3 - azi) / dt2;
    omegas[4] = (ele - phi0) / 10.0f;
    omegas[5] = (phi1 - ele) / 10.0f;
    idx[0] = host_pick_hrtf((float)phi0, (float)th0);
    idx[1] = host_pick_hrtf((float)phi0, (float)th1);
    idx[2] = host_pick_hrtf((float)phi1, (float)th2);
    idx[3] = host_pick_hrtf((float)phi1, (float)th3);
    return JF_OK;
}

// The corrected rule (include/jefferson.h JF_FLAG_CORRECTED_INTERPOLATION); twin of dev_interp_corrected.
int host_interpolation_corrected(float ele, float azi, int idx[4], float omegas[6]) {
    return host_grid_interpolation(ring_table(), ele, azi, idx, omegas);
}

// ... in its general form, for any grid of rings (twin of dev_interp_corrected, which says why the reference's grid keeps
// its closed form: the same bits either way)
int host_grid_interpolation(const RingTable &rt, float ele, float azi, int idx[4], float omegas[6]) {
    if (!(ele <= 90.0f) || !(ele > -1.0e6f) || !(azi > -1.0e6f && azi < 1.0e6f)) return JF_ERR_RANGE;
    float a = azi - 360.0f * floorf(azi / 360.0f);
    if (!(a < 360.0f)) a = 0.0f;
    int r0;
    float phi0, span;
    if (rt.kemar) {
        if (ele < -40.0f) ele = -40.0f;
        const float q = floorf(ele / 10.0f);
        phi0 = 10.0f * q;
        r0 = (int)q + 4;
        span = 10.0f;
    } else {
        const int last = rt.n_rings - 1;
        if (ele < rt.ele[0]) ele = rt.ele[0];
        if (ele > rt.ele[last]) ele = rt.ele[last];
        r0 = 0;
        for (int r = 1; r <= last; r++) r0 = rt.ele[r] <= ele ? r : r0;
        phi0 = rt.ele[r0];
        span = rt.ele[r0 < last ? r0 + 1 : r0] - phi0;
    }
    const bool on_ring = ele == phi0;
    const int ring[2] = {r0, on_ring ? r0 : r0 + 1};
    const float omE = on_ring ? 0.0f : (ele - phi0) / span;
    for (int j = 0; j < 2; j++) {
        const int r = ring[j];
        const float d = rt.inc[r];
        const int n = rt.offset[r + 1] - rt.offset[r];
        int i0 = (int)floorf(a / d);
        if (i0 > n - 1) i0 = n - 1;
        float wa = (a - (float)i0 * d) / d;
        if (wa < 0.0f) wa = 0.0f;
        if (wa > 1.0f) wa = 1.0f;
        if (n == 1) wa = 0.0f;
        int i1 = i0 + 1 == n ? 0 : i0 + 1;
        if (wa == 0.0f) i1 = i0;
        idx[2 * j] = rt.offset[r] + i0;
        idx[2 * j + 1] = rt.offset[r] + i1;
        omegas[2 * j] = wa;
        omegas[2 * j + 1] = 1.0f - wa;
    }
    omegas[4] = omE;
    omegas[5] = 1.0f - omE;
    return JF_OK;
}

void host_from_spherical(float ele, float azi, float r, float out[5]) {
    ele = roundf(ele);
    azi = roundf(azi);
    out[0] = ele;
    out[1] = azi;
    out[2] = (float)(r * sin(azi * kPi / 180.0f));
    out[3] = (float)(r * sin(ele * kPi / 180.0f));
    out[4] = (float)(r * -cos(azi * kPi / 180.0f));
}

int host_from_cartesian(float x, float y, float z, float out[5], float *r_out) {
    const float r = sqrtf(x * x + z * z + y * y);
    const float horiz = sqrtf(x * x + z * z);
    if (!(r > 0.0f) || !(r < 3.0e38f)) return JF_ERR_RANGE;
    float ele = (float)(atan2f(y, horiz) * 180.0f / kPi);
    float azi = (float)(atan2f(-x / r, -z / r) * 180.0f / kPi);
    if (azi < 0.0f) azi += 360;
    out[0] = roundf(ele);
    out[1] = roundf(azi);
    out[2] = x;
    out[3] = y;
    out[4] = z;
    if (r_out) *r_out = r;
    return JF_OK;
}

// ------------------------------------------------------------------ WAV ---
namespace {
struct WavInfo {
    int format = 0;  // 1 PCM, 3 float
    int channels = 0;
    int rate = 0;
    int bits = 0;
    long data_off = 0;
    size_t data_bytes = 0;
};

uint32_t rd32(const unsigned char *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t rd16(const unsigned char *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

int wav_open(const char *path, FILE **fp, WavInfo *wi, std::string *err) {
    FILE *f = fopen(path, "rb");
    if (!f) {
        *err = std::string("cannot open ") + path;
        return JF_ERR_IO;
    }
    unsigned char hd[12];
    if (fread(hd, 1, 12, f) != 12 || memcmp(hd, "RIFF", 4) || memcmp(hd + 8, "WAVE", 4)) {
        fclose(f);
        *err = std::string("not a RIFF/WAVE file: ") + path;
        return JF_ERR_IO;
    }
    bool have_fmt = false;
    for (;;) {
        unsigned char ch[8];
        if (fread(ch, 1, 8, f) != 8) break;
        const uint32_t sz = rd32(ch + 4);
        if (!memcmp(ch, "fmt ", 4)) {
            unsigned char fm[40] = {0};
            const size_t n = sz < sizeof(fm) ? sz : sizeof(fm);
            if (fread(fm, 1, n, f) != n) break;
            wi->format = rd16(fm);
            wi->channels = rd16(fm + 2);
            wi->rate = (int)rd32(fm + 4);
            wi->bits = rd16(fm + 14);
            if (wi->format == 0xFFFE && n >= 26) wi->format = rd16(fm + 24);  // WAVE_FORMAT_EXTENSIBLE
            if (sz > n) fseek(f, (long)(sz - n), SEEK_CUR);
            if (sz & 1) fseek(f, 1, SEEK_CUR);
            have_fmt = true;
        } else if (!memcmp(ch, "data", 4)) {
            wi->data_off = ftell(f);
            wi->data_bytes = sz;
            // never trust the header for an allocation: a truncated (or hostile) file yields what is present
            if (wi->data_off >= 0 && fseek(f, 0, SEEK_END) == 0) {
                const long end = ftell(f);
                const size_t present = end > wi->data_off ? (size_t)(end - wi->data_off) : 0;
                if (wi->data_bytes > present) wi->data_bytes = present;
                fseek(f, wi->data_off, SEEK_SET);
            }
            if (!have_fmt) break;
            *fp = f;
            return JF_OK;
        } else {
            fseek(f, (long)(sz + (sz & 1)), SEEK_CUR);
        }
    }
    fclose(f);
    *err = std::string("malformed WAV: ") + path;
    return JF_ERR_IO;
}

// libsndfile's sf_read_float scaling: integer PCM / 2^(bits-1); 8-bit is unsigned.
int wav_read_all(const char *path, std::vector<float> *samples, WavInfo *wi, std::string *err) {
    FILE *f = nullptr;
    int rc = wav_open(path, &f, wi, err);
    if (rc) return rc;
    const int bps = wi->bits / 8;
    if (wi->channels < 1 || bps < 1 || bps > 4 || !((wi->format == 1) || (wi->format == 3 && bps == 4))) {
        fclose(f);
        *err = std::string("unsupported WAV encoding: ") + path;
        return JF_ERR_IO;
    }
    // a truncated file (header claims more than is there) yields what is present
    std::vector<unsigned char> raw(wi->data_bytes);
    const size_t got = fread(raw.data(), 1, raw.size(), f);
    fclose(f);
    const size_t n = got / (size_t)bps;
    samples->resize(n);
    for (size_t i = 0; i < n; i++) {
        const unsigned char *p = raw.data() + i * bps;
        float v;
        if (wi->format == 3) {
            memcpy(&v, p, 4);
        } else if (bps == 1) {
            v = ((int)p[0] - 128) / 128.0f;
        } else if (bps == 2) {
            v = (int16_t)rd16(p) / 32768.0f;
        } else if (bps == 3) {
            int32_t x = p[0] | (p[1] << 8) | (p[2] << 16);
            if (x & 0x800000) x -= 0x1000000;
            v = x / 8388608.0f;
        } else {
            v = (int32_t)rd32(p) / 2147483648.0f;
        }
        (*samples)[i] = v;
    }
    return JF_OK;
}
}  // namespace

int wav_read_mono(const char *path, float **out, size_t *n_frames, int *sample_rate, std::string *err) {
    std::vector<float> s;
    WavInfo wi;
    int rc = wav_read_all(path, &s, &wi, err);
    if (rc) return rc;
    if (wi.channels > 2) {  // cudaPart.cu:57-60
        *err = std::string(path) + ": only mono or stereo accepted";
        return JF_ERR_IO;
    }
    const size_t frames = s.size() / (size_t)wi.channels;
    float *buf = (float *)malloc(sizeof(float) * (frames ? frames : 1));
    if (!buf) return JF_ERR_NOMEM;
    if (wi.channels == 1) {
        memcpy(buf, s.data(), sizeof(float) * frames);
    } else {
        // cudaPart.cu:50-52: L/2 + R/2 (double arithmetic, stored to float)
        for (size_t i = 0; i < frames; i++) buf[i] = (float)(s[2 * i] / 2.0 + s[2 * i + 1] / 2.0);
    }
    *out = buf;
    *n_frames = frames;
    if (sample_rate) *sample_rate = wi.rate;
    return JF_OK;
}

int wav_write_stereo24(const char *path, const float *il, size_t n_frames, int rate, std::string *err) {
    FILE *f = fopen(path, "wb");
    if (!f) {
        *err = std::string("cannot create ") + path;
        return JF_ERR_IO;
    }
    const uint32_t data = (uint32_t)(n_frames * 2 * 3);
    unsigned char h[44];
    auto w32 = [](unsigned char *p, uint32_t v) { p[0] = v; p[1] = v >> 8; p[2] = v >> 16; p[3] = v >> 24; };
    auto w16 = [](unsigned char *p, uint16_t v) { p[0] = (unsigned char)v; p[1] = (unsigned char)(v >> 8); };
    memcpy(h, "RIFF", 4);
    w32(h + 4, 36 + data);
    memcpy(h + 8, "WAVEfmt ", 8);
    w32(h + 16, 16);
    w16(h + 20, 1);
    w16(h + 22, 2);
    w32(h + 24, (uint32_t)rate);
    w32(h + 28, (uint32_t)rate * 6);
    w16(h + 32, 6);
    w16(h + 34, 24);
    memcpy(h + 36, "data", 4);
    w32(h + 40, data);
    fwrite(h, 1, 44, f);
    std::vector<unsigned char> buf(n_frames * 6);
    for (size_t i = 0; i < n_frames * 2; i++) {
        // libsndfile float -> PCM_24: scale by 0x7FFFFF and round; clipped here
        // (libsndfile wraps unless SFC_SET_CLIPPING is on)
        float v = il[i] * 8388607.0f;
        if (v > 8388607.0f) v = 8388607.0f;
        if (v < -8388608.0f) v = -8388608.0f;
        const int32_t x = (int32_t)lrintf(v);
        buf[3 * i] = (unsigned char)x;
        buf[3 * i + 1] = (unsigned char)(x >> 8);
        buf[3 * i + 2] = (unsigned char)(x >> 16);
    }
    const bool ok = fwrite(buf.data(), 1, buf.size(), f) == buf.size();
    fclose(f);
    if (!ok) {
        *err = std::string("short write: ") + path;
        return JF_ERR_IO;
    }
    return JF_OK;
}

// ------------------------------------------------------------ reverb gain ---
namespace {
// in-place radix-2 complex FFT in double (n a power of two), sign = -1 forward / +1 inverse
void dfft(std::vector<double> &re, std::vector<double> &im, int sign) {
    const size_t n = re.size();
    for (size_t i = 1, j = 0; i < n; i++) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            std::swap(re[i], re[j]);
            std::swap(im[i], im[j]);
        }
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const double ang = sign * 2.0 * kPi / (double)len;
        for (size_t base = 0; base < n; base += len)
            for (size_t k = 0; k < len / 2; k++) {
                const double wr = cos(ang * (double)k), wi = sin(ang * (double)k);
                const size_t a = base + k, b = a + len / 2;
                const double tr = re[b] * wr - im[b] * wi, ti = re[b] * wi + im[b] * wr;
                re[b] = re[a] - tr;
                im[b] = im[a] - ti;
                re[a] += tr;
                im[a] += ti;
            }
    }
}
}  // namespace

ReverbSchedule host_reverb_schedule(long long j0, int K, int M, long long fut_m) {
    ReverbSchedule s{};
    const long long j1 = j0 + K;
    // X_m is formed in the call that takes in block M m - 1:  j0 < M m <= j1
    s.m_lo = j0 / M + 1;
    const long long m_hi = j1 / M;
    s.n_tr = m_hi >= s.m_lo ? (int)(m_hi - s.m_lo + 1) : 0;
    // big blocks that lie inside the call: m = ma .. m_hi - 1, their wet signal is FULL(m), anchored at X_{m+1}
    s.ma = (j0 + M - 1) / M;
    s.n_mid = m_hi > s.ma ? (int)(m_hi - s.ma) : 0;
    // the other blocks go through the uniform stage (head) + TAIL of their big block
    if (s.n_mid > 0) {
        s.n_ranges = 2;
        s.kb[0] = 0;
        s.kn[0] = (int)(s.ma * M - j0);
        s.kb[1] = (int)(m_hi * M - j0);
        s.kn[1] = K - s.kb[1];
        // of the middle's blocks only the last 2 M - 1 are transformed (the state the next blocks read: the head has 2 M
        // partitions) ...
        s.copy_lo = s.kn[0];
        s.copy_hi = s.kb[1] - (2 * M - 1);
        if (s.copy_hi < s.copy_lo) s.copy_hi = s.copy_lo;
        // ... and only the last whole big block is copied to the dry ring (later calls' transforms reach back two big blocks)
        // (with the head's 2 M - 1 blocks transformed, which writes them to the dry ring as well, that is all of them)
        s.skip_lo = s.copy_lo;
        s.skip_hi = s.copy_lo > s.kb[1] - M ? s.copy_lo : s.kb[1] - M;
        if (s.skip_hi > s.copy_hi) s.skip_hi = s.copy_hi;
    } else {
        s.n_ranges = 1;
        s.kb[0] = 0;
        s.kn[0] = K;
    }
    s.tail_early = s.tail_late = -1;
    s.fut_m = fut_m;
    // TAIL of the big block the call starts in, if one of its blocks goes through the uniform stage and nobody has formed it
    // yet (its X_m are all there -- a whole big block ago, since the head covers two big blocks of taps)
    const long long mb = j0 / M;
    if (s.kn[0] > 0 && s.fut_m < mb) {
        s.tail_early = mb;
        s.fut_m = mb;
    }
    // ... and of the big block the call ends in, if the call reaches into it behind a boundary it has passed itself
    const bool late = s.n_mid > 0 ? s.kn[1] > 0 : (m_hi > mb && j1 > m_hi * M);
    if (late && s.fut_m < m_hi) {
        s.tail_late = m_hi;
        s.fut_m = m_hi;
    }
    return s;
}

float host_reverb_rms_gain(const float *x, size_t n, const float *ir, size_t n_ir) {
    // cudaPart.cu:170-186 PadData: both padded to new_size = n + (n_ir - n_ir/2); the product of
    // the two new_size-point spectra is a CIRCULAR convolution of that length (the tail wraps).
    const size_t new_size = n + (n_ir - n_ir / 2);
    size_t m = 1;
    while (m < n + n_ir) m <<= 1;
    std::vector<double> ar(m, 0.0), ai(m, 0.0), br(m, 0.0), bi(m, 0.0);
    double e_in = 0.0;
    for (size_t i = 0; i < n; i++) {
        ar[i] = x[i];
        e_in += (double)x[i] * x[i];
    }
    for (size_t i = 0; i < n_ir; i++) br[i] = ir[i];
    dfft(ar, ai, -1);
    dfft(br, bi, -1);
    for (size_t i = 0; i < m; i++) {
        const double r = ar[i] * br[i] - ai[i] * bi[i], q = ar[i] * bi[i] + ai[i] * br[i];
        ar[i] = r;
        ai[i] = q;
    }
    dfft(ar, ai, +1);
    std::vector<double> y(new_size, 0.0);
    for (size_t i = 0; i < n + n_ir - 1; i++) y[i % new_size] += ar[i] / (double)m;  // fold the linear result
    double e_out = 0.0;
    for (size_t i = 0; i < new_size; i++) e_out += y[i] * y[i];
    if (!(e_in > 0.0) || !(e_out > 0.0)) return 1.0f;
    return (float)sqrt(e_in / e_out);
}

// ------------------------------------------------------- KEMAR directory ---
int load_hrir_dir(const char *dir, std::vector<float> *hrir, int *taps_out, std::string *err) {
    char path[1024];
    snprintf(path, sizeof(path), "%s/elev0/H0e000a.wav", dir);
    FILE *probe = fopen(path, "rb");
    const bool compact = probe != nullptr;
    if (probe) fclose(probe);
    int taps = 0;
    for (int j = 0; j < kNumHrtf; j++) {
        int ele, azi;
        table_position(j, &ele, &azi);
        for (int ear = 0; ear < 2; ear++) {
            std::vector<float> s;
            WavInfo wi;
            int want_ch, take_ch;
            if (compact) {
                // right half-sphere measured; left half = mirror with ears exchanged
                const int a = azi <= 180 ? azi : 360 - azi;
                snprintf(path, sizeof(path), "%s/elev%d/H%de%03da.wav", dir, ele, ele, a);
                want_ch = 2;
                take_ch = azi <= 180 ? ear : 1 - ear;
            } else {
                snprintf(path, sizeof(path), "%s/elev%d/%c%de%03da.wav", dir, ele, ear ? 'R' : 'L', ele, azi);
                want_ch = 1;  // hrtf_signals.cu:68-71
                take_ch = 0;
            }
            int rc = wav_read_all(path, &s, &wi, err);
            if (rc) return rc;
            if (wi.channels != want_ch) {
                *err = std::string("incorrect number of channels in HRTF ") + path;
                return JF_ERR_IO;
            }
            if (wi.rate != 44100) {  // hrtf_signals.cu:72-75
                *err = std::string("incorrect sampling rate in ") + path;
                return JF_ERR_IO;
            }
            const int frames = (int)(s.size() / (size_t)want_ch);
            if (taps == 0) {
                taps = frames;
                hrir->assign((size_t)kNumHrtf * 2 * taps, 0.0f);
            }
            if (frames != taps || taps <= 0) {
                *err = std::string("HRIR length differs in ") + path;
                return JF_ERR_IO;
            }
            float *dst = hrir->data() + ((size_t)j * 2 + ear) * taps;
            for (int n = 0; n < taps; n++) dst[n] = s[(size_t)n * want_ch + take_ch];
        }
    }
    *taps_out = taps;
    return JF_OK;
}

}  // namespace jf
