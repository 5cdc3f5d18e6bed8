// jf_host.h -- host-side pieces of the engine that do not touch the GPU:
// geometry and index/weight rules of the reference's SoundSource, the KEMAR
// directory loader and WAV I/O.  Product code (not the oracle).
#pragma once
#include <stddef.h>

#include <string>
#include <vector>

#include "jf_device.h"

namespace jf {

// hrtf_signals.cu:7-12 + the loader loop :119-140 (float-accumulated azimuth).
const RingTable &ring_table();
// (elevation, (int)round(azimuth)) of table row j (hrtf_signals.cu:121-124).
void table_position(int j, int *ele, int *azi);

int host_pick_hrtf(float obj_ele, float obj_azi);                             // hrtf_signals.cu:20-51
int host_interpolation(float ele, float azi, int idx[4], float omegas[6]);    // SoundSource.cu:65-105
int host_interpolation_corrected(float ele, float azi, int idx[4], float omegas[6]);  // JF_FLAG_CORRECTED_INTERPOLATION
// any grid of elevation rings (include/jefferson.h: jf_hrtf_grid): its table, the corrected rule in its general form, the
// nearest measurement
const float *kemar_ring_steps();  // hrtf_signals.cu:8
int host_grid_table(int n_rings, const float *ring_ele, const int *ring_count, const float *ring_step, RingTable *out,
                    std::string *err);
int host_grid_interpolation(const RingTable &rt, float ele, float azi, int idx[4], float omegas[6]);
int host_grid_pick(const RingTable &rt, float ele, float azi);
int host_grid_from_positions(size_t n, const float *azi, const float *ele, float tol, int *n_rings, float *ring_ele,
                             int *ring_count, float *ring_step, int *row_of, std::string *err);
void host_from_spherical(float ele, float azi, float r, float out[5]);        // SoundSource.cu:41-54
int host_from_cartesian(float x, float y, float z, float out[5], float *r);   // SoundSource.cu:20-36

// KEMAR directory -> [710][2][taps] float32 (int16 / 32768).
int load_hrir_dir(const char *dir, std::vector<float> *hrir, int *taps, std::string *err);

// rms(x) / rms(x (*) ir) with the padding and circular length of cudaPart.cu:170-186, in double.
float host_reverb_rms_gain(const float *x, size_t n, const float *ir, size_t n_ir);

// What the non-uniformly partitioned reverb does in a call of K blocks that starts at absolute block j0 (big blocks of M blocks;
// fut_m: TAIL has been formed for every big block up to this one) -- the schedule only, no GPU state (ReverbBigParams in
// jf_device.h explains X_m, TAIL, FULL).  The engine turns it into kernel parameters; tests/test_reverb_plan.py replays runs
// of calls against a model of the rings.
struct ReverbSchedule {
    long long m_lo;        // transforms X_m for m = m_lo .. m_lo + n_tr - 1 (those with j0 < M m <= j0 + K)
    int n_tr;
    long long ma;          // whole big blocks inside the call: ma .. ma + n_mid - 1, formed by FULL
    int n_mid;
    int n_ranges;          // block ranges that go through the uniform stage (head) + TAIL: [kb, kb + kn)
    int kb[2], kn[2];
    int copy_lo, copy_hi;  // blocks of the call that only copy their samples to the dry ring
    int skip_lo, skip_hi;  // blocks the transform kernel does not visit at all
    long long tail_early;  // TAIL(m) to form before anything else (-1: none) ...
    long long tail_late;   // ... and behind the transforms (-1: none)
    long long fut_m;       // the new value of fut_m
};
ReverbSchedule host_reverb_schedule(long long j0, int K, int M, long long fut_m);

// SOFA files (jf_sofa.cpp; include/jefferson.h: jf_sofa_*)
}  // namespace jf
struct jf_sofa_set;
struct jf_grid_layout;
namespace jf {
int sofa_read(const char *path, ::jf_sofa_set *out, std::string *err);
void sofa_release(::jf_sofa_set *s);
int sofa_taps(const ::jf_sofa_set *s, std::string *err);
int sofa_table(const ::jf_sofa_set *s, float tol_deg, ::jf_grid_layout *layout, float *hrir, int taps, std::string *err);
int hdf5_read(const char *path, const char *dataset, double **out, int *rank, unsigned long long *dims, std::string *err);
int hdf5_attr(const char *path, const char *object, const char *attr, char *out, size_t cap, std::string *err);

int wav_read_mono(const char *path, float **out, size_t *n_frames, int *sample_rate, std::string *err);
int wav_write_stereo24(const char *path, const float *interleaved, size_t n_frames, int sample_rate,
                       std::string *err);

}  // namespace jf
