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
void host_from_spherical(float ele, float azi, float r, float out[5]);        // SoundSource.cu:41-54
int host_from_cartesian(float x, float y, float z, float out[5], float *r);   // SoundSource.cu:20-36

// KEMAR directory -> [710][2][taps] float32 (int16 / 32768).
int load_hrir_dir(const char *dir, std::vector<float> *hrir, int *taps, std::string *err);

// rms(x) / rms(x (*) ir) with the padding and circular length of cudaPart.cu:170-186, in double.
float host_reverb_rms_gain(const float *x, size_t n, const float *ir, size_t n_ir);

int wav_read_mono(const char *path, float **out, size_t *n_frames, int *sample_rate, std::string *err);
int wav_write_stereo24(const char *path, const float *interleaved, size_t n_frames, int sample_rate,
                       std::string *err);

}  // namespace jf
