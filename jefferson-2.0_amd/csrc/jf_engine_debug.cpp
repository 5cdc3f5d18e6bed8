// jf_engine_debug.cpp -- every entry point of include/jefferson_debug.h: parity taps for the tests, timing hooks for bench.py,
// tuning switches of the A/B scripts, and the accessors through which jf_group.c and bench.py reach the engine's stream and
// device buffers.  None of them is part of the drop-in boundary (include/jefferson.h: jf_engine.cpp, jf_engine_reverb.cpp).
#include "jf_engine_internal.h"

extern "C" {

int jf_debug_hdf5_read(const char *path, const char *dataset, double **out, int *rank, unsigned long long *dims) {
    return jf_guard([&]() -> int {
    if (!path || !dataset || !out || !rank || !dims) return fail(nullptr, JF_ERR_ARG, "null argument");
    std::string err;
    const int rc = hdf5_read(path, dataset, out, rank, dims, &err);
    return rc ? fail(nullptr, rc, err) : JF_OK;
    });
}

int jf_debug_hdf5_attr(const char *path, const char *object, const char *attr, char *out, size_t cap) {
    return jf_guard([&]() -> int {
    if (!path || !object || !attr || !out || !cap) return fail(nullptr, JF_ERR_ARG, "null argument");
    std::string err;
    const int rc = hdf5_attr(path, object, attr, out, cap, &err);
    return rc ? fail(nullptr, rc, err) : JF_OK;
    });
}

int jf_profile_read_reverb(jf_engine *e, double *reverb_ms) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!e || !reverb_ms) return JF_ERR_ARG;
    JF_HIP(e, hipStreamSynchronize(e->stream));
    double r = 0;
    for (size_t i = 0; e->profiling >= 2 && e->rv_P > 0 && i < e->ev_used && i < e->ev_reverb.size(); i++) {
        float ms = 0;
        JF_HIP(e, hipEventElapsedTime(&ms, e->ev_reverb[i].a, e->ev_reverb[i].b));
        r += ms;
    }
    *reverb_ms = r;
    return JF_OK;
    });
}

int jf_debug_set_reverb_side_workgroups(jf_engine *e, int workgroups) {
    if (!e || workgroups < 8 || workgroups > 65536) return JF_ERR_ARG;
    e->rv_side_wgs = workgroups;
    return JF_OK;
}

float *jf_batch_mix_device(jf_engine *e) { return e ? e->d_mix : nullptr; }

float *jf_batch_partial_device(jf_engine *e) { return e ? e->d_partial : nullptr; }

void *jf_engine_stream(jf_engine *e) { return e ? (void *)e->stream : nullptr; }

int jf_profile_enable(jf_engine *e, int enable) {
    return jf_guard([&]() -> int {
    if (e) {
        DeviceGuard bind_(e);
        const int rc_ = rv_ahead_discard(e);  // (the next block's stage may have gone ahead in the old form)
        if (rc_) return rc_;
    }
    DeviceGuard bind(e);
    if (!e) return JF_ERR_ARG;
    JF_HIP(e, hipStreamSynchronize(e->stream));
    e->profiling = enable < 0 ? 0 : (enable > 2 ? 2 : enable);
    e->ev_used = 0;
    e->profile_calls = 0;
    return JF_OK;
    });
}

int jf_profile_set_stride(jf_engine *e, int every) {
    return jf_guard([&]() -> int {
    if (!e || every < 1) return JF_ERR_ARG;
    e->profile_stride = every;
    e->profile_calls = 0;
    return JF_OK;
    });
}

int jf_profile_read(jf_engine *e, double *fused_ms, double *prep_ms, double *mix_ms, long *launches) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!e) return JF_ERR_ARG;
    JF_HIP(e, hipStreamSynchronize(e->stream));
    double f = 0, p = 0, m = 0;
    for (size_t i = 0; i < e->ev_used; i++) {
        float ms = 0;
        JF_HIP(e, hipEventElapsedTime(&ms, e->ev_fused[i].a, e->ev_fused[i].b));
        f += ms;
        if (e->profiling >= 2) {
            JF_HIP(e, hipEventElapsedTime(&ms, e->ev_prep[i].a, e->ev_prep[i].b));
            p += ms;
            JF_HIP(e, hipEventElapsedTime(&ms, e->ev_mix[i].a, e->ev_mix[i].b));
            m += ms;
        }
    }
    if (fused_ms) *fused_ms = f;
    if (prep_ms) *prep_ms = p;
    if (mix_ms) *mix_ms = m;
    if (launches) *launches = (long)e->ev_used;
    return JF_OK;
    });
}

// ---- debugging taps -----------------------------------------------------------
int jf_debug_copy_from_device(jf_engine *e, const void *device_ptr, void *host, size_t bytes) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!e || !device_ptr || !host) return JF_ERR_ARG;
    JF_HIP(e, hipStreamSynchronize(e->stream));
    JF_HIP(e, hipMemcpy(host, device_ptr, bytes, hipMemcpyDeviceToHost));
    return JF_OK;
    });
}

int jf_debug_set_rt_max_sources(jf_engine *e, int n) {
    return jf_guard([&]() -> int {
    if (e) {
        DeviceGuard bind_(e);
        const int rc_ = rv_ahead_discard(e);  // (the next block's stage may have gone ahead in the old form)
        if (rc_) return rc_;
    }
    if (!e || n < 0) return JF_ERR_ARG;
    e->rt_max_sources = n;
    return JF_OK;
    });
}

int jf_debug_set_source_group(jf_engine *e, int group) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!e || group < 0 || (group > 0 && e->S % group)) return JF_ERR_ARG;
    e->src_group = group;
    if (group > 0 && e->sorted_order) {  // a pinned group size means consecutive sources
        JF_HIP(e, hipStreamSynchronize(e->stream));
        for (int s = 0; s < e->S; s++) e->order[s] = s;
        JF_HIP(e, h2d(e, e->d_order, e->order.data(), sizeof(int) * e->S));
        e->sorted_order = false;
    }
    return JF_OK;
    });
}

int jf_debug_source_order(const jf_engine *e, int *order) {
    if (!e || !order) return JF_ERR_ARG;
    // the per-source kernel (a run that resolved to G = 1) does not go through the order: its block u is source u
    for (int s = 0; s < e->S; s++) order[s] = e->last_group == 1 ? s : e->order[s];  // (no run yet: what a grouped run takes)
    return JF_OK;
}

int jf_debug_set_reverb_form(jf_engine *e, int form) {
    return jf_guard([&]() -> int {
    if (e) {
        DeviceGuard bind_(e);
        const int rc_ = rv_ahead_discard(e);  // (the next block's stage may have gone ahead in the old form)
        if (rc_) return rc_;
    }
    if (!e || form < 0 || form > 3) return JF_ERR_ARG;
    e->rv_form = form;
    return JF_OK;
    });
}

int jf_debug_set_interp_table(jf_engine *e, int on) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!e) return JF_ERR_ARG;
    if (on < 0 || on > 2) return fail(e, JF_ERR_ARG, "0 = never, 1 = always, 2 = decided per run");
    if (on && !e->interp_avail) return fail(e, JF_ERR_STATE, "this engine was created without the pre-interpolated rows");
    e->interp_use = on;  // the mode word of the next run changes with it: descriptors prepared ahead no longer match
    if (on == 1) {       // "always" builds them now (a run under "per run" builds them when it first takes them)
        const int rc = ensure_interp_rows(e);
        if (rc) return rc;
        if (!e->interp_built) return fail(e, JF_ERR_NOMEM, "no device memory for the pre-interpolated rows");
    }
    return JF_OK;
    });
}

int jf_debug_interp_table(const jf_engine *e) { return e && e->interp_avail ? e->interp_use : 0; }

int jf_debug_interp_table_built(const jf_engine *e) { return e && e->interp_built ? 1 : 0; }

int jf_debug_last_run_used_rows(const jf_engine *e) { return e && e->last_rows ? 1 : 0; }

int jf_debug_count_desc_flags(jf_engine *e, int n_items, int mask) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!e || n_items <= 0 || (size_t)n_items > (size_t)e->S * e->maxK) return JF_ERR_ARG;
    std::vector<ItemDesc> d((size_t)n_items);
    JF_HIP(e, hipStreamSynchronize(e->stream));
    JF_HIP(e, hipMemcpy(d.data(), e->d_desc, sizeof(ItemDesc) * d.size(), hipMemcpyDeviceToHost));
    int n = 0;
    for (const ItemDesc &x : d) n += (x.flags & mask) != 0;
    return n;
    });
}

int jf_debug_read_table_rows(jf_engine *e, int first_row, int n, float *out) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (e && out && n > 0 && first_row >= 0 && first_row + (long long)n > e->rt.n_rows) {  // pre-interpolated rows: built on demand
        const int rc = ensure_interp_rows(e);
        if (rc) return rc;
    }
    const int total = e ? e->rt.n_rows + (e->interp_built ? kInterpRows : 0) : 0;
    if (!e || !out || n <= 0 || first_row < 0 || first_row > total - n) return fail(e, JF_ERR_ARG, "rows outside the table");
    JF_HIP(e, hipStreamSynchronize(e->stream));
    JF_HIP(e, hipMemcpy(out, e->d_htab + (size_t)first_row * 512, sizeof(float4) * 512 * (size_t)n, hipMemcpyDeviceToHost));
    return JF_OK;
    });
}

int jf_debug_set_reverb_partitioning(jf_engine *e, int how) {
    return jf_guard([&]() -> int {
    if (!e || how < 0 || how > 2) return JF_ERR_ARG;
    e->rv_partitioning = how;  // in effect from the next jf_reverb_set_ir
    return JF_OK;
    });
}

int jf_debug_set_reverb_ahead(jf_engine *e, int on) {
    return jf_guard([&]() -> int {
    if (!e) return JF_ERR_ARG;
    DeviceGuard bind(e);
    const int rc = rv_ahead_discard(e);
    if (rc) return rc;
    e->rv_ahead_on = on != 0;
    return JF_OK;
    });
}

int jf_debug_set_reverb_lazy_state(jf_engine *e, int on) {
    return jf_guard([&]() -> int {
    if (e) {
        DeviceGuard bind_(e);
        const int rc_ = rv_ahead_discard(e);  // (the next block's stage may have gone ahead in the old form)
        if (rc_) return rc_;
    }
    if (!e) return JF_ERR_ARG;
    e->rv_lazy_small = on != 0;  // (transforms already put off are still formed by the call that needs them)
    return JF_OK;
    });
}

int jf_debug_set_reverb_head_fused(jf_engine *e, int on) {
    return jf_guard([&]() -> int {
    if (e) {
        DeviceGuard bind_(e);
        const int rc_ = rv_ahead_discard(e);  // (the next block's stage may have gone ahead in the old form)
        if (rc_) return rc_;
    }
    if (!e) return JF_ERR_ARG;
    e->rv_head_fused = on != 0;
    return JF_OK;
    });
}

int jf_debug_set_reverb_async(jf_engine *e, int on) {
    return jf_guard([&]() -> int {
    if (e) {
        DeviceGuard bind_(e);
        const int rc_ = rv_ahead_discard(e);  // (the next block's stage may have gone ahead in the old form)
        if (rc_) return rc_;
    }
    if (!e) return JF_ERR_ARG;
    e->rv_async = on != 0;  // what the side stream has in flight is waited for by the next call's stage (run_reverb_stage)
    return JF_OK;
    });
}

int jf_debug_reverb_schedule(long long j0, int K, int M, long long fut_m, long long out[16]) {
    if (!out || K <= 0 || M <= 0 || j0 < 0) return JF_ERR_ARG;
    const ReverbSchedule s = host_reverb_schedule(j0, K, M, fut_m);
    const long long v[16] = {s.m_lo, s.n_tr, s.ma, s.n_mid, s.n_ranges, s.kb[0], s.kn[0], s.kb[1], s.kn[1], s.copy_lo, s.copy_hi,
                             s.skip_lo, s.skip_hi, s.tail_early, s.tail_late, s.fut_m};
    for (int i = 0; i < 16; i++) out[i] = v[i];
    return JF_OK;
}

int jf_debug_reverb_partitions(const jf_engine *e, int *head, int *big, int *big_taps) {
    if (!e) return JF_ERR_ARG;
    if (head) *head = e->rv_P;
    if (big) *big = e->rv_P1 > 0 ? e->rv_P1 - 1 : 0;  // H'_2 .. H'_P1 (H'_0 and H'_1 are the head's taps; FULL uses all)
    if (big_taps) *big_taps = e->rv_B1;
    return e->rv_P_total;
}

int jf_debug_read_table(jf_engine *e, float *out) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!e || !out) return JF_ERR_ARG;
    std::vector<float4> h((size_t)e->rt.n_rows * 512);
    JF_HIP(e, hipStreamSynchronize(e->stream));
    JF_HIP(e, hipMemcpy(h.data(), e->d_htab, sizeof(float4) * h.size(), hipMemcpyDeviceToHost));
    for (int j = 0; j < e->rt.n_rows; j++) {
        float *L = out + ((size_t)j * 2 + 0) * kNc * 2;
        float *R = out + ((size_t)j * 2 + 1) * kNc * 2;
        const float4 *row = h.data() + (size_t)j * 512;
        L[0] = row[0].x;
        L[1] = 0.0f;
        L[1024] = row[0].y;
        L[1025] = 0.0f;
        R[0] = row[0].z;
        R[1] = 0.0f;
        R[1024] = row[0].w;
        R[1025] = 0.0f;
        for (int k = 1; k < 512; k++) {
            L[2 * k] = row[k].x;
            L[2 * k + 1] = row[k].y;
            R[2 * k] = row[k].z;
            R[2 * k + 1] = row[k].w;
        }
    }
    return JF_OK;
    });
}

int jf_debug_interp_device(jf_engine *e, int n, const float *ele, const float *azi, int *rows, float *weights,
                           int *nterms) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!e || n <= 0 || !ele || !azi || !rows || !weights || !nterms) return JF_ERR_ARG;
    float *d_e = nullptr, *d_a = nullptr, *d_w = nullptr;
    int *d_r = nullptr, *d_n = nullptr;
    auto body = [&]() -> int {
        JF_HIP(e, hipMalloc(&d_e, sizeof(float) * n));
        JF_HIP(e, hipMalloc(&d_a, sizeof(float) * n));
        JF_HIP(e, hipMalloc(&d_w, sizeof(float) * 4 * n));
        JF_HIP(e, hipMalloc(&d_r, sizeof(int) * 4 * n));
        JF_HIP(e, hipMalloc(&d_n, sizeof(int) * n));
        JF_HIP(e, h2d(e, d_e, ele, sizeof(float) * n));
        JF_HIP(e, h2d(e, d_a, azi, sizeof(float) * n));
        JF_HIP(e, launch_interp_debug(e->rt, d_e, d_a, d_r, d_w, d_n, n, corrected_rule(e) ? 1 : 0, e->stream));
        JF_HIP(e, hipStreamSynchronize(e->stream));
        JF_HIP(e, hipMemcpy(rows, d_r, sizeof(int) * 4 * n, hipMemcpyDeviceToHost));
        JF_HIP(e, hipMemcpy(weights, d_w, sizeof(float) * 4 * n, hipMemcpyDeviceToHost));
        JF_HIP(e, hipMemcpy(nterms, d_n, sizeof(int) * n, hipMemcpyDeviceToHost));
        return JF_OK;
    };
    int rc = body();
    (void)hipFree(d_e);
    (void)hipFree(d_a);
    (void)hipFree(d_w);
    (void)hipFree(d_r);
    (void)hipFree(d_n);
    return rc;
    });
}

int jf_debug_rfft_device(jf_engine *e, int n, const float *windows, float *spectra) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!e || n <= 0 || !windows || !spectra) return JF_ERR_ARG;
    float *d_w = nullptr;
    float2 *d_s = nullptr;
    auto body = [&]() -> int {
        JF_HIP(e, hipMalloc(&d_w, sizeof(float) * (size_t)n * kN));
        JF_HIP(e, hipMalloc(&d_s, sizeof(float2) * (size_t)n * kNc));
        JF_HIP(e, h2d(e, d_w, windows, sizeof(float) * (size_t)n * kN));
        JF_HIP(e, launch_rfft_debug(d_w, n, e->d_twpack, d_s, e->stream));
        JF_HIP(e, hipStreamSynchronize(e->stream));
        JF_HIP(e, hipMemcpy(spectra, d_s, sizeof(float2) * (size_t)n * kNc, hipMemcpyDeviceToHost));
        return JF_OK;
    };
    int rc = body();
    (void)hipFree(d_w);
    (void)hipFree(d_s);
    return rc;
    });
}

int jf_debug_last_source_group(const jf_engine *e) { return e ? e->last_group : JF_ERR_ARG; }

const char *jf_debug_last_kernels(jf_engine *e) {
    if (!e) return "";
    if (e->kernels_use_frozen) return e->kernels_frozen.c_str();  // (the stage's fields describe the block launched ahead)
    try {
        const std::string nb = std::to_string(e->B / 64), bs = std::to_string(e->B);
        std::string k;
        if (!e->last_rt && !e->last_prep_skipped) k = "prep_kernel;";
        if (e->rv_P > 0) {
            if (e->last_catchup) k += "reverb_fft_kernel<" + bs + ">@ring;";
            const ReverbPlan &pl = e->last_plan;
            const std::string b1 = std::to_string(e->rv_B1);
            auto per_wg = [&](int) { return std::string(",1>;"); };  // transforms per workgroup and turn (persistent since round 5)
            auto products = [&](const ReverbBigParams &g) {
                if (g.n_prod <= 0) return std::string();
                const std::string mac = g.n_prod >= 4 ? "reverb_big_mac_kernel<" + b1 + ",16>"
                                        : JF_RV_BIG_MAC1_SHARED && g.mac_wgs == 0 ? "reverb_big_mac1_kernel<" + b1 + ">"
                                                                                  : "reverb_big_mac_kernel<" + b1 + ",1>";
                return mac + ";reverb_big_ifft_kernel<" + b1 + per_wg(g.n_prod);
            };
            auto transforms = [&](const ReverbBigParams &g) {
                return g.n_tr > 0 ? "reverb_big_fft_kernel<" + b1 + per_wg(g.n_tr) : std::string();
            };
            const int tile = e->B == 256 ? 8 : 16, grp = e->B == 256 ? 2 : 4;
            auto stage_b = [&](int form) {
                if (form == 3) return "reverb_mac_tiled_kernel<" + bs + "," + std::to_string(tile) + ">;";
                if (form == 4) return "reverb_mac_kernel<" + bs + ",1,true>;";
                if (form == 0) return std::string();
                return "reverb_mac_kernel<" + bs + "," + std::to_string(form == 2 ? grp : 1) + ">;";
            };
            if (pl.big) k += products(pl.tail_early);
            if (e->last_rv_form == 5) {
                // (the head ran inside the real-time kernel, named below; transforms left in line follow it)
            } else if (e->last_rv_form == 4) {
                k += stage_b(4);
                if (pl.big) k += transforms(pl.transforms);
            } else {
                if (e->last_small_fft) k += "reverb_fft_kernel<" + bs + ">;";
                if (pl.big) {
                    if (pl.n_ranges > 1) k += stage_b(pl.forms[0]);
                    k += transforms(pl.transforms);
                    k += products(pl.middle) + products(pl.tail_late);
                    k += stage_b(pl.forms[pl.n_ranges > 1 ? 1 : 0]);
                } else {
                    k += stage_b(e->last_rv_form);
                }
            }
        }
        if (e->rv_P > 0) k += e->last_side;
        // launch_mix: few partial blocks per audio block (16, 32 or 64 groups) take the one-thread-per-float form
        const int n_part = e->last_group > 0 ? e->S / e->last_group : e->S;
        const std::string mix_name = (n_part == 16 || n_part == 32 || n_part == 64)
                                         ? ";mix_few_kernel<" + std::to_string(n_part / 16) + ">" : std::string(";mix_kernel");
        if (e->last_rt) {
            const bool fused = e->rv_P > 0 && e->last_rv_form == 5;
            k += "rt_block_kernel<" + nb + "," + std::to_string(rt_waves_per_wg(e->S)) + (fused ? ",reverb>" : ">");
            if (fused && e->last_plan.big && e->last_plan.transforms.n_tr > 0) {
                const std::string b1 = std::to_string(e->rv_B1);
                k += ";reverb_big_fft_kernel<" + b1 + ",1>";
            }
        }
        else k += std::string(e->last_group > 1 ? "fused_pair_kernel<" : "fused_block_kernel<") + nb +
                  (e->last_fused_prep ? ">+prep" : ">") + (e->last_mix_prep ? ";mix_prep_kernel" : mix_name);
        e->kernels = k;
        return e->kernels.c_str();
    } catch (...) {
        return "";
    }
}

int jf_debug_set_prep_ahead(jf_engine *e, int on) {
    return jf_guard([&]() -> int {
    if (!e) return JF_ERR_ARG;
    e->prep_ahead = on != 0;
    e->ahead.valid = false;
    return JF_OK;
    });
}

int jf_debug_set_grid_limit(jf_engine *e, int workgroups) {
    return jf_guard([&]() -> int {
    if (!e || workgroups < 0) return JF_ERR_ARG;
    e->grid_limit = workgroups;
    return JF_OK;
    });
}

int jf_debug_stage_taps(jf_engine *e, int n, const float *positions, const float *windows, float *dist,
                        float *spectra) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!e || n <= 0 || !positions || !dist || (spectra && !windows)) return JF_ERR_ARG;
    float *d_p = nullptr, *d_w = nullptr;
    float2 *d_d = nullptr, *d_s = nullptr;
    auto body = [&]() -> int {
        JF_HIP(e, hipMalloc(&d_p, sizeof(float) * 5 * (size_t)n));
        JF_HIP(e, hipMalloc(&d_d, sizeof(float2) * (size_t)n * kNc));
        JF_HIP(e, h2d(e, d_p, positions, sizeof(float) * 5 * (size_t)n));
        if (spectra) {
            JF_HIP(e, hipMalloc(&d_w, sizeof(float) * (size_t)n * kN));
            JF_HIP(e, hipMalloc(&d_s, sizeof(float2) * (size_t)n * 2 * kNc));
            JF_HIP(e, h2d(e, d_w, windows, sizeof(float) * (size_t)n * kN));
        }
        JF_HIP(e, launch_stage_debug(e->rt, kernel_mode(e), d_p, d_w, n, e->d_htab, e->d_twpack, d_d, d_s,
                                     e->stream));
        JF_HIP(e, hipStreamSynchronize(e->stream));
        JF_HIP(e, hipMemcpy(dist, d_d, sizeof(float2) * (size_t)n * kNc, hipMemcpyDeviceToHost));
        if (spectra) JF_HIP(e, hipMemcpy(spectra, d_s, sizeof(float2) * (size_t)n * 2 * kNc, hipMemcpyDeviceToHost));
        return JF_OK;
    };
    int rc = body();
    (void)hipFree(d_p);
    (void)hipFree(d_w);
    (void)hipFree(d_d);
    (void)hipFree(d_s);
    return rc;
    });
}

int jf_debug_read_stamps(jf_engine *e, unsigned long long *out, int n) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!e || !out || n < 0 || n > 8192) return JF_ERR_ARG;
    JF_HIP(e, hipStreamSynchronize(e->stream));
    memcpy(out, (const char *)e->h_err + 16, sizeof(unsigned long long) * (size_t)n);
    return JF_OK;
    });
}

}  // extern "C"
