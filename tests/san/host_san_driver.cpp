// Sanitizer driver for the product's host-side code (jefferson-2.0_amd/csrc/jf_host.cpp: geometry, index/weight rules, WAV
// I/O, the KEMAR directory loader, the reverb's schedule and gain) -- built by tests/test_sanitizers.py with
// -fsanitize=address,undefined and run on the CPU.  Test infrastructure: calls product code, checks only that it neither
// faults nor leaks and that a few invariants hold; the numbers themselves are the parity tests' business.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <random>
#include <string>
#include <vector>

#include "../../include/jefferson.h"
#include "../../jefferson-2.0_amd/csrc/jf_host.h"

using namespace jf;

static int fails = 0;
#define CHECK(c)                                                       \
    do {                                                               \
        if (!(c)) {                                                    \
            fprintf(stderr, "CHECK failed line %d: %s\n", __LINE__, #c); \
            fails++;                                                   \
        }                                                              \
    } while (0)

static void put16(std::vector<uint8_t> &v, unsigned x) { v.push_back(x & 255), v.push_back((x >> 8) & 255); }
static void put32(std::vector<uint8_t> &v, unsigned x) { put16(v, x & 65535), put16(v, x >> 16); }
// a minimal PCM16 WAV (the loader's input format)
static std::vector<uint8_t> wav16(int channels, int rate, const std::vector<int16_t> &samples) {
    std::vector<uint8_t> v;
    const unsigned data = (unsigned)samples.size() * 2;
    v.insert(v.end(), {'R', 'I', 'F', 'F'});
    put32(v, 36 + data);
    v.insert(v.end(), {'W', 'A', 'V', 'E', 'f', 'm', 't', ' '});
    put32(v, 16), put16(v, 1), put16(v, channels), put32(v, rate), put32(v, rate * channels * 2), put16(v, channels * 2), put16(v, 16);
    v.insert(v.end(), {'d', 'a', 't', 'a'});
    put32(v, data);
    for (int16_t s : samples) put16(v, (uint16_t)s);
    return v;
}
static void write_file(const std::string &p, const std::vector<uint8_t> &b, size_t n) {
    FILE *f = fopen(p.c_str(), "wb");
    if (!f) return;
    fwrite(b.data(), 1, n, f);
    fclose(f);
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const std::string tmp = argv[1];  // an empty scratch directory
    std::mt19937 rng(12345);

    // ---- tables, pick, index/weight rules (both), far outside the measured range too
    const RingTable &rt = ring_table();
    CHECK(rt.offset[kNumElev] == kNumHrtf);
    for (int j = 0; j < kNumHrtf; j++) {
        int e, a;
        table_position(j, &e, &a);
        CHECK(e >= -40 && e <= 90 && a >= 0 && a < 360);
        CHECK(host_pick_hrtf((float)e, (float)a) == j);
    }
    long n_ok = 0;
    for (int e2 = -140; e2 <= 220; e2++)
        for (int a2 = -40; a2 <= 760; a2++) {
            const float ele = 0.5f * e2, azi = 0.5f * a2;
            int idx[4];
            float om[6];
            for (int rule = 0; rule < 2; rule++) {
                const int rc = rule ? host_interpolation_corrected(ele, azi, idx, om) : host_interpolation(ele, azi, idx, om);
                if (rc == JF_OK) {
                    n_ok++;
                    for (int t = 0; t < 4; t++) CHECK(idx[t] >= 0 && idx[t] < kNumHrtf);
                }
            }
            const int p = host_pick_hrtf(ele, azi);
            CHECK(p >= -1 && p < kNumHrtf);
        }
    CHECK(n_ok > 100000);
    const float odd[] = {0.f, -0.f, 1e-30f, 1e30f, -1e30f, INFINITY, -INFINITY, NAN, 359.999f, 360.f, -40.f, 90.f, 90.0001f};
    for (float e : odd)
        for (float a : odd) {
            int idx[4];
            float om[6], rec[5], r;
            (void)host_interpolation(e, a, idx, om);
            (void)host_interpolation_corrected(e, a, idx, om);
            (void)host_pick_hrtf(e, a);
            host_from_spherical(e, a, 0.5f, rec);
            (void)host_from_cartesian(e, a, 1.f, rec, &r);
        }
    {
        float rec[5], r;
        CHECK(host_from_cartesian(0.f, 0.f, 0.f, rec, &r) != JF_OK);
        CHECK(host_from_cartesian(0.f, 0.f, -1.f, rec, &r) == JF_OK);
        host_from_spherical(5.f, 3.f, 0.5f, rec);
        CHECK(rec[0] == 5.f && rec[1] == 3.f);
    }

    // ---- the reverb's schedule and gain
    for (int i = 0; i < 200000; i++) {
        const int M = (rng() & 1) ? 16 : 8;
        const long long j0 = rng() % 100000;
        const int K = 1 + (int)(rng() % 300);
        const long long fut = (long long)(rng() % 7000) - 10;
        const ReverbSchedule s = host_reverb_schedule(j0, K, M, fut);
        CHECK(s.n_tr >= 0 && s.n_mid >= 0 && s.n_ranges >= 1 && s.n_ranges <= 2);
        CHECK(s.kn[0] >= 0 && s.kb[0] == 0 && s.kn[0] <= K);
        if (s.n_ranges == 2) CHECK(s.kb[1] >= s.kn[0] && s.kb[1] + s.kn[1] == K && s.kn[0] + s.n_mid * M + s.kn[1] == K);
        else CHECK(s.kn[0] == K);
        if (s.n_ranges == 2) CHECK(s.copy_lo <= s.copy_hi && s.skip_lo <= s.skip_hi && s.skip_lo >= s.copy_lo && s.skip_hi <= s.copy_hi && s.copy_hi <= s.kb[1]);
        CHECK(s.fut_m >= fut);
    }
    {
        std::vector<float> x(777), ir(1), ir2(300);
        for (auto &v : x) v = (float)((int)(rng() % 2001) - 1000) / 1000.f;
        ir[0] = 0.5f;
        for (size_t i = 0; i < ir2.size(); i++) ir2[i] = expf(-0.02f * (float)i) * ((rng() & 1) ? 1.f : -1.f);
        const float g1 = host_reverb_rms_gain(x.data(), x.size(), ir.data(), ir.size());
        CHECK(fabsf(g1 - 2.f) < 1e-3f);
        const float g2 = host_reverb_rms_gain(x.data(), x.size(), ir2.data(), ir2.size());
        CHECK(g2 > 0.f && isfinite(g2));
        (void)host_reverb_rms_gain(x.data(), 1, ir2.data(), ir2.size());
    }

    // ---- WAV: a round trip, then every truncation and 3000 random corruptions of a valid file
    {
        std::string err;
        std::vector<float> st(2 * 500);
        for (auto &v : st) v = (float)((int)(rng() % 2001) - 1000) / 1000.f;
        const std::string p = tmp + "/rt.wav";
        CHECK(wav_write_stereo24(p.c_str(), st.data(), 500, 44100, &err) == JF_OK);
        float *mono = nullptr;
        size_t n = 0;
        int rate = 0;
        CHECK(wav_read_mono(p.c_str(), &mono, &n, &rate, &err) == JF_OK && n == 500 && rate == 44100);
        if (mono) {
            for (size_t i = 0; i < n; i++) CHECK(fabsf(mono[i] - (st[2 * i] / 2 + st[2 * i + 1] / 2)) < 3e-7f);
            free(mono);
        }
        CHECK(wav_write_stereo24((tmp + "/no/such/dir.wav").c_str(), st.data(), 500, 44100, &err) != JF_OK);
        CHECK(wav_read_mono((tmp + "/absent.wav").c_str(), &mono, &n, &rate, &err) != JF_OK);
        std::vector<int16_t> s16(2 * 64);
        for (auto &v : s16) v = (int16_t)(rng() % 65536 - 32768);
        const std::vector<uint8_t> good = wav16(2, 44100, s16);
        const std::string q = tmp + "/fuzz.wav";
        for (size_t len = 0; len <= good.size(); len++) {
            write_file(q, good, len);
            mono = nullptr;
            if (wav_read_mono(q.c_str(), &mono, &n, &rate, &err) == JF_OK) free(mono);
        }
        for (int i = 0; i < 3000; i++) {
            std::vector<uint8_t> bad = good;
            const int flips = 1 + (int)(rng() % 4);
            for (int f = 0; f < flips; f++) bad[rng() % 60 % bad.size()] = (uint8_t)rng();   // the header region
            write_file(q, bad, bad.size());
            mono = nullptr;
            if (wav_read_mono(q.c_str(), &mono, &n, &rate, &err) == JF_OK) free(mono);
        }
    }

    // ---- the KEMAR directory loader: nothing there, a complete synthetic compact set, one file short, one file of another length
    {
        std::vector<float> hrir;
        int taps = 0;
        std::string err;
        CHECK(load_hrir_dir((tmp + "/nothing").c_str(), &hrir, &taps, &err) != JF_OK);
        const std::string root = tmp + "/compact";
        mkdir(root.c_str(), 0755);
        std::string last;
        for (int j = 0; j < kNumHrtf; j++) {
            int e, a;
            table_position(j, &e, &a);
            if (a > 180) continue;
            char d[64], f[96];
            snprintf(d, sizeof(d), "/elev%d", e);
            mkdir((root + d).c_str(), 0755);
            snprintf(f, sizeof(f), "/elev%d/H%de%03da.wav", e, e, a);
            std::vector<int16_t> s(2 * 128);
            for (size_t i = 0; i < s.size(); i++) s[i] = (int16_t)((j * 131 + (int)i * 7) % 2000 - 1000);
            const std::vector<uint8_t> w = wav16(2, 44100, s);
            write_file(root + f, w, w.size());
            last = root + f;
        }
        CHECK(load_hrir_dir(root.c_str(), &hrir, &taps, &err) == JF_OK && taps == 128 && hrir.size() == (size_t)kNumHrtf * 2 * 128);
        // mirrored rows exchange the ears
        for (int j = 0; j < kNumHrtf; j++) {
            int e, a;
            table_position(j, &e, &a);
            if (a == 0 || a >= 180) continue;
            const int m = host_pick_hrtf((float)e, (float)(360 - a));
            int e2, a2;
            table_position(m, &e2, &a2);
            if (e2 != e || a2 != 360 - a) continue;
            CHECK(memcmp(&hrir[((size_t)j * 2 + 0) * 128], &hrir[((size_t)m * 2 + 1) * 128], sizeof(float) * 128) == 0);
        }
        std::vector<int16_t> shorter(2 * 100, 1);
        const std::vector<uint8_t> w = wav16(2, 44100, shorter);
        write_file(last, w, w.size());
        CHECK(load_hrir_dir(root.c_str(), &hrir, &taps, &err) != JF_OK);
        unlink(last.c_str());
        CHECK(load_hrir_dir(root.c_str(), &hrir, &taps, &err) != JF_OK);
    }
    printf("host sanitizer driver: %d failed checks\n", fails);
    return fails ? 1 : 0;
}
