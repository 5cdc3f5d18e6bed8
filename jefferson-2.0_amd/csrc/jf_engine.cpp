// jf_engine.cpp -- implementation of the C ABI (include/jefferson.h) on HIP: creation, the batch pipeline, the per-block calls.
// One engine = one GPU (one process per GPU in multi-GPU runs).  No CPU
// fallback: every processing entry point runs the HIP kernels or fails.
// (The convolution reverb's schedule: jf_engine_reverb.cpp; include/jefferson_debug.h's entry points: jf_engine_debug.cpp;
// the engine's state and what the three units share: jf_engine_internal.h.)
#include "jf_engine_internal.h"

// The pre-interpolated rows, built on first use (jf_engine::interp_avail): a table of 710 + kInterpRows rows takes the place of
// the 710-row one -- the measured rows copied, the weighted sums formed behind them on the engine's stream.  Without room
// for the 386 MB the engine goes on without rows (per-block weighting), for good.
int ensure_interp_rows(jf_engine *e) {
    if (e->interp_built || !e->interp_avail) return JF_OK;
    float4 *big = nullptr;
    const size_t n_rows = (size_t)e->rt.n_rows;
    if (hipMalloc(&big, sizeof(float4) * (n_rows + kInterpRows) * 512) != hipSuccess) {
        (void)hipGetLastError();
        e->interp_avail = false;
        e->interp_use = 0;
        return JF_OK;
    }
    hipError_t q = hipMemcpyAsync(big, e->d_htab, sizeof(float4) * n_rows * 512, hipMemcpyDeviceToDevice, e->stream);
    if (q == hipSuccess) q = launch_table_interp_build(e->rt, corrected_rule(e) ? 1 : 0, big, e->stream);
    if (q == hipSuccess) q = hipStreamSynchronize(e->stream);  // (everything that reads the old table has finished as well)
    if (q != hipSuccess) {
        (void)hipFree(big);
        JF_HIP(e, q);
    }
    (void)hipFree(e->d_htab);
    e->d_htab = big;
    e->interp_built = true;
    return JF_OK;
}



// prep -> [reverb] -> fused -> mix on the engine stream, K blocks starting at d_pos.
// first_block: index of d_pos's first block in the uploaded trajectory (jf_batch_run), -1 for positions from elsewhere.
int run_blocks(jf_engine *e, const float *d_pos, int K, float *d_mix_out, int first_block) {
    if (device_fault(e)) return fail(e, JF_ERR_DEVICE, kHandOffMsg);  // fatal: see device_fault
    {
        const int rc = rv_ahead_discard(e);  // (a stage launched ahead by a one-block call: this call does its own)
        if (rc) return rc;
    }
    const int p = e->cur;
    EventPair *ep = nullptr, *ef = nullptr, *em = nullptr;
    // a pair of event records costs ~7 us of stream time: they may be put around every n-th run only (the runs in
    // between launch the same kernels, untimed)
    const bool timed = e->profiling && (e->profile_stride <= 1 || e->profile_calls++ % e->profile_stride == 0);
    e->timed_now = timed;
    if (timed) {
        ef = next_events(e, e->ev_fused);
        if (!ef) return fail(e, JF_ERR_DEVICE, "hipEventCreate failed");
    }
    if (e->profiling >= 2 && timed) {
        ep = next_events(e, e->ev_prep);
        em = next_events(e, e->ev_mix);
        if (!ep || !em) return fail(e, JF_ERR_DEVICE, "hipEventCreate failed");
    }
    // sources a pair of wavefronts sums before it stores a stereo block: as many as leave about two units for every
    // resident pair (2048 on MI355X): larger groups mean fewer inverse transforms and fewer partial blocks for the
    // mix (profiles/group_sweep.py times every size against this choice)
    const long long n_items = (long long)K * e->S;
    const int G = e->src_group > 0 ? e->src_group
                  : (e->S % 32 == 0 && n_items >= 131072) ? 32
                  : (e->S % 16 == 0 && n_items >= 32768) ? 16
                  : (e->S % 8 == 0 && n_items >= 16384) ? 8
                  : (e->S % 4 == 0 && n_items >= 8192) ? 4
                  : (e->S % 2 == 0 && (n_items >= 4096 || e->S >= 1024)) ? 2
                                                         : 1;
    FusedParams P;
    P.G = (e->S % G == 0) ? G : 1;
    const int canon = P.G > 1;  // descriptors in the pair-kernel layout
    // whole-degree positions as pre-interpolated rows: the pair kernel's descriptors only
    bool rows = canon && e->interp_avail && e->interp_use != 0;
    if (rows && e->interp_use == 2 && first_block >= 0 && (size_t)(first_block + K) < e->traj_moved.size()) {
        const double moved = (double)(e->traj_moved[first_block + K] - e->traj_moved[first_block]) / (double)n_items;
        rows = moved <= kInterpMovedMax;
    }
    // a one-block call is the audio callback's (jf_submit_block with more sources than the one-launch kernel takes, or while
    // profiling): it never pays the 386 MB allocation, the build and the stream synchronisation -- it weights per block
    // (bit-identical) until a batch run, or the pre-warm call jf_debug_set_interp_table(e, 1), has built the rows
    if (rows && !e->interp_built && K == 1 && first_block < 0) rows = false;
    if (rows && !e->interp_built) {  // the first batch run that takes them builds them
        const int rc = ensure_interp_rows(e);
        if (rc) return rc;
        rows = e->interp_built;
    }
    e->last_rows = rows;
    const int mode_now = kernel_mode(e) | (rows ? kModeInterpRows : 0);
    // per-kernel timing (profiling >= 2) keeps prep and mix as launches of their own
    const bool have = e->ahead.valid && e->profiling < 2 && first_block >= 0 && e->ahead.first == first_block &&
                      e->ahead.K == K && e->ahead.mode == mode_now && e->ahead.canon == canon &&
                      e->ahead.traj_gen == e->traj_gen;
    e->ahead.valid = false;
    e->last_prep_skipped = have;
    if (have) std::swap(e->d_desc, e->d_desc_ahead);
    if (ep) JF_HIP(e, hipEventRecord(ep->a, e->stream));
    if (!have) JF_HIP(e, launch_prep(e->rt, mode_now, d_pos, e->d_state[p], e->d_desc, e->S, K, canon, e->stream));
    if (ep) JF_HIP(e, hipEventRecord(ep->b, e->stream));
    {
        const int rc = run_reverb_stage(e, p, K);
        if (rc) return rc;
    }
    P.htab = e->d_htab;
    P.tw = e->d_twpack;
    P.desc = e->d_desc;
    P.sigs = e->rv_P > 0 ? e->d_sigs_wet : e->d_sigs;
    P.st_in = e->d_state[p];
    P.st_out = e->d_state[p ^ 1];
    P.hist_in = e->d_hist[p];
    P.hist_out = e->d_hist[p ^ 1];
    P.pos = d_pos;
    P.partial = e->d_partial;
    P.S = e->S;
    P.K = K;
    P.B = e->B;
    e->last_group = P.G;
    P.mode = mode_now;
    P.err = e->hd_err;
    P.order = e->d_order;
    // the window that follows in the trajectory, if there is a whole one: its descriptors are prepared by this run --
    // inside the pair kernel's own launch (trailing workgroups, in the kernel's tail), else inside the mix launch
    const bool ahead_ok = e->prep_ahead && e->profiling < 2 && first_block >= 0 && first_block + 2 * K <= e->traj_blocks;
    const bool ahead_in_fused = ahead_ok && P.G > 1;
    P.n_pair_wgs = 0;
    P.prep_pos = ahead_in_fused ? d_pos + (size_t)K * e->S * 5 : nullptr;
    P.prep_desc = e->d_desc_ahead;
    P.prep_K = K;
    P.prep_canon = canon;
    P.rt = e->rt;
    int max_wgs = e->resident_wgs[P.G > 1 ? (rows ? 2 : 1) : 0];
    if (e->grid_limit > 0 && e->grid_limit < max_wgs) max_wgs = e->grid_limit;
    if (ef) JF_HIP(e, hipEventRecord(ef->a, e->stream));
    JF_HIP(e, launch_fused(P, max_wgs, e->stream));
    if (ef) JF_HIP(e, hipEventRecord(ef->b, e->stream));
    if (em) JF_HIP(e, hipEventRecord(em->a, e->stream));
    e->last_mix_prep = ahead_ok && !ahead_in_fused;
    e->last_fused_prep = ahead_in_fused;
    if (ahead_ok) {
        if (ahead_in_fused)
            JF_HIP(e, launch_mix(e->d_partial, d_mix_out, e->S / P.G, K, e->B, e->stream));
        else
            JF_HIP(e, launch_mix_prep(e->d_partial, d_mix_out, e->S / P.G, K, e->B, e->rt, mode_now,
                                      d_pos + (size_t)K * e->S * 5, e->d_desc_ahead, e->S, K, canon, e->stream));
        e->ahead.valid = true;
        e->ahead.first = first_block + K;
        e->ahead.K = K;
        e->ahead.mode = mode_now;
        e->ahead.canon = canon;
        e->ahead.traj_gen = e->traj_gen;
    } else {
        JF_HIP(e, launch_mix(e->d_partial, d_mix_out, e->S / P.G, K, e->B, e->stream));
    }
    if (em) JF_HIP(e, hipEventRecord(em->b, e->stream));
    if (timed) e->ev_used++;
    e->cur = p ^ 1;
    e->last_rt = false;
    return submit_side(e);
}

static void snapshot_positions(jf_engine *e, float *dst /* [S][5] */) {
    std::lock_guard<std::mutex> lk(e->pos_mu);
    for (int s = 0; s < e->S; s++) {
        const HostPos &q = e->pos[s];
        float *d = dst + 5 * s;
        d[0] = q.ele;
        d[1] = q.azi;
        d[2] = q.x;
        d[3] = q.y;
        d[4] = q.z;
    }
}


// zero one source's (or every source's, src < 0) window, counters and reverb state
int reset_sources(jf_engine *e, int src) {
    e->ahead.valid = false;  // the old position of the next block changes
    const size_t s0 = src < 0 ? 0 : (size_t)src, ns = src < 0 ? (size_t)e->S : 1;
    const int p = e->cur;
    quiesce_side(e);  // (what it has left in the fut ring is zeroed below with the rest)
    JF_HIP(e, hipMemsetAsync(e->d_hist[p] + s0 * kN, 0, sizeof(float) * kN * ns, e->stream));
    JF_HIP(e, hipMemsetAsync(e->d_state[p] + s0, 0, sizeof(SrcState) * ns, e->stream));
    if (e->rv_P > 0) {
        const size_t B = (size_t)e->B;
        JF_HIP(e, hipMemsetAsync(e->d_rv_fdl + s0 * e->rv_Rg * B, 0, sizeof(float2) * e->rv_Rg * B * ns, e->stream));
        JF_HIP(e, hipMemsetAsync(e->d_rv_fdl + (size_t)e->S * e->rv_Rg * B + s0 * e->rv_Rg, 0, sizeof(float2) * e->rv_Rg * ns, e->stream));
        JF_HIP(e, hipMemsetAsync(e->d_rv_wet + s0 * e->rv_Wr, 0, sizeof(float) * e->rv_Wr * ns, e->stream));
        JF_HIP(e, hipMemsetAsync(e->d_rv_prev[p] + s0 * B, 0, sizeof(float) * B * ns, e->stream));
        JF_HIP(e, hipMemsetAsync(e->d_rv_count[p] + s0, 0, sizeof(int) * ns, e->stream));
        if (e->rv_P1 > 0) {  // the big partitions' delay line, the dry ring they read and what they have promised the next blocks
            const size_t B1 = (size_t)e->rv_B1;
            JF_HIP(e, hipMemsetAsync(e->d_rv_fdl1 + s0 * e->rv_R1 * B1, 0, sizeof(float2) * e->rv_R1 * B1 * ns, e->stream));
            JF_HIP(e, hipMemsetAsync(e->d_rv_fdl1 + (size_t)e->S * e->rv_R1 * B1 + s0 * e->rv_R1, 0, sizeof(float2) * e->rv_R1 * ns, e->stream));
            JF_HIP(e, hipMemsetAsync(e->d_rv_dryring + s0 * e->rv_Rn * B1, 0, sizeof(float) * e->rv_Rn * B1 * ns, e->stream));
            JF_HIP(e, hipMemsetAsync(e->d_rv_fut + s0 * e->rv_Fn * B1, 0, sizeof(float) * e->rv_Fn * B1 * ns, e->stream));
        }
    }
    return JF_OK;
}

namespace {

void destroy_engine(jf_engine *e) {
    if (!e) return;
    DeviceGuard bind(e);
    if (e->rv_side) (void)hipStreamSynchronize(e->rv_side);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    free_reverb(e);
    for (float *p : e->d_signal)
        if (p) (void)hipFree(p);
    (void)hipFree(e->d_htab);
    (void)hipFree(e->d_tw);
    (void)hipFree(e->d_twpack);
    (void)hipFree(e->d_sigs);
    (void)hipFree(e->d_zero);
    for (int i = 0; i < 2; i++) {
        (void)hipFree(e->d_state[i]);
        (void)hipFree(e->d_hist[i]);
    }
    (void)hipFree(e->d_desc);
    (void)hipFree(e->d_desc_ahead);
    (void)hipFree(e->d_partial);
    (void)hipFree(e->d_mix);
    (void)hipFree(e->d_pos_rt);
    (void)hipFree(e->d_traj);
    (void)hipFree(e->d_order);
    (void)hipFree(e->d_pick);
    if (e->h_pos_pinned) (void)hipHostFree(e->h_pos_pinned);
    if (e->h_out_pinned) (void)hipHostFree(e->h_out_pinned);
    if (e->h_done) (void)hipHostFree(e->h_done);
    if (e->h_err) (void)hipHostFree(e->h_err);
    for (auto *pool : {&e->ev_prep, &e->ev_fused, &e->ev_mix, &e->ev_reverb})
        for (auto &p : *pool) {
            (void)hipEventDestroy(p.a);
            (void)hipEventDestroy(p.b);
        }
    if (e->rv_ev_main) (void)hipEventDestroy(e->rv_ev_main);
    if (e->rv_ev_side) (void)hipEventDestroy(e->rv_ev_side);
    if (e->rv_side) (void)hipStreamDestroy(e->rv_side);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

// grid: the table of the HRTF set's measurement grid (null: the reference's KEMAR grid, 710 rows)
int create_engine(const jf_config *cfg, const RingTable *grid, const float *hrir, int taps, jf_engine **out) {
    if (!cfg || !hrir || !out) return fail(nullptr, JF_ERR_ARG, "null argument");
    *out = nullptr;
    const int B = cfg->frames_per_buffer;
    if (B < 64 || B > 256 || B % 64) return fail(nullptr, JF_ERR_ARG, "frames_per_buffer must be 64, 128, 192 or 256");
    if (cfg->hrtf_len <= 0 || taps <= 0 || taps > cfg->hrtf_len)
        return fail(nullptr, JF_ERR_ARG, "need 0 < taps <= hrtf_len");
    // PAD_LEN = 2^ceil(log2(B + L - 1)) (Universal.cuh:12); the kernels are built for 1024
    const int pad = (int)pow(2, ceil(log2((double)(B + cfg->hrtf_len - 1))));
    if (pad != kN) return fail(nullptr, JF_ERR_ARG, "frames_per_buffer + hrtf_len - 1 must pad to 1024");
    if (cfg->n_sources <= 0) return fail(nullptr, JF_ERR_ARG, "n_sources must be positive");
    if (cfg->max_batch_blocks <= 0) return fail(nullptr, JF_ERR_ARG, "max_batch_blocks must be positive");
    if (cfg->flags & ~(JF_FLAG_CORRECTED_INTERPOLATION | JF_FLAG_NO_INTERP_TABLE))
        return fail(nullptr, JF_ERR_ARG, "unknown bits in flags");
    if (kernels_build_kind() != 0) {
        // a fault-injection or timing-only build of the kernels (jf_experiments.h): wrong results by design
        const char *allow = getenv("JF_ALLOW_EXPERIMENT");
        if (!allow || strcmp(allow, "1") != 0)
            return fail(nullptr, JF_ERR_STATE,
                        "this library was built with an experiment switch that gives wrong results by design "
                        "(set JF_ALLOW_EXPERIMENT=1 to use it in a test)");
    }

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, JF_ERR_DEVICE, "no HIP device available (this library has no CPU path)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, JF_ERR_ARG, "device ordinal out of range");

    jf_engine *e = new jf_engine();
    e->cfg = *cfg;
    e->B = B;
    e->S = cfg->n_sources;
    e->maxK = cfg->max_batch_blocks;
    const size_t S = (size_t)e->S, K = (size_t)e->maxK;
    int rc = JF_OK;
    int prev_dev = -1;
    (void)hipGetDevice(&prev_dev);
    auto body = [&]() -> int {
        JF_HIP(e, hipSetDevice(cfg->device));
        {
            // the engine's stream at the highest priority the device offers, the side stream (the reverb's work ahead of time,
            // run_reverb_stage) at the lowest: where the two meet, the block in hand goes first
            int lo = 0, hi = 0;
            JF_HIP(e, hipDeviceGetStreamPriorityRange(&lo, &hi));
            JF_HIP(e, hipStreamCreateWithPriority(&e->stream, hipStreamNonBlocking, hi));
            JF_HIP(e, hipStreamCreateWithPriority(&e->rv_side, hipStreamNonBlocking, lo));
        }
        JF_HIP(e, hipEventCreateWithFlags(&e->rv_ev_main, hipEventDisableTiming));
        JF_HIP(e, hipEventCreateWithFlags(&e->rv_ev_side, hipEventDisableTiming));
        for (int kind = 0; kind < 3; kind++) JF_HIP(e, fused_resident_workgroups(B / 64, kind, &e->resident_wgs[kind]));
        {
            int cus = 0;
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess && cus >= 16)
                e->rv_side_wgs = 3 * cus / 4;
            else
                (void)hipGetLastError();
        }
        // (nothing of the engine's behaviour is read from the environment: jefferson_debug.h's setters are the overrides)
        e->interp_avail = !(cfg->flags & JF_FLAG_NO_INTERP_TABLE);
        e->interp_use = e->interp_avail ? 2 : 0;
        // the 710 measured rows only; the pre-interpolated ones come with the first run that takes them (ensure_interp_rows)
        e->rt = grid ? *grid : ring_table();
        JF_HIP(e, hipMalloc(&e->d_htab, sizeof(float4) * (size_t)e->rt.n_rows * 512));
        JF_HIP(e, hipMalloc(&e->d_tw, sizeof(float2) * 1024));
        JF_HIP(e, hipMalloc(&e->d_sigs, sizeof(SrcSignal) * S));
        for (int i = 0; i < 2; i++) {
            JF_HIP(e, hipMalloc(&e->d_state[i], sizeof(SrcState) * S));
            JF_HIP(e, hipMalloc(&e->d_hist[i], sizeof(float) * S * kN));
            JF_HIP(e, hipMemsetAsync(e->d_state[i], 0, sizeof(SrcState) * S, e->stream));
            JF_HIP(e, hipMemsetAsync(e->d_hist[i], 0, sizeof(float) * S * kN, e->stream));
        }
        JF_HIP(e, hipMalloc(&e->d_desc, sizeof(ItemDesc) * S * K));
        JF_HIP(e, hipMalloc(&e->d_desc_ahead, sizeof(ItemDesc) * S * K));
        JF_HIP(e, hipMalloc(&e->d_partial, sizeof(float) * S * K * 2 * B));
        JF_HIP(e, hipMalloc(&e->d_mix, sizeof(float) * K * 2 * B));
        JF_HIP(e, hipMalloc(&e->d_pos_rt, sizeof(float) * S * 5));
        e->rt.pick = nullptr;
        if (e->rt.kemar) {
            // nearest table row per (ring, integer azimuth), by the search itself (host_pick_hrtf = hrtf_signals.cu:20-51)
            static const int elev[kNumElev] = {-40, -30, -20, -10, 0, 10, 20, 30, 40, 50, 60, 70, 80, 90};
            std::vector<short> pick((size_t)kNumElev * kPickAzi);
            for (int r = 0; r < kNumElev; r++)
                for (int a = 0; a < kPickAzi; a++) pick[(size_t)r * kPickAzi + a] = (short)host_pick_hrtf((float)elev[r], (float)a);
            JF_HIP(e, hipMalloc(&e->d_pick, sizeof(short) * pick.size()));
            JF_HIP(e, h2d(e, e->d_pick, pick.data(), sizeof(short) * pick.size()));
            e->rt.pick = e->d_pick;
        }
        JF_HIP(e, hipMalloc(&e->d_order, sizeof(int) * S));
        e->order.resize(S);
        for (size_t s = 0; s < S; s++) e->order[s] = (int)s;
        JF_HIP(e, h2d(e, e->d_order, e->order.data(), sizeof(int) * S));
        // host memory the kernels read and write in place, and whose words the host polls while a kernel runs: mapped AND
        // coherent (fine-grained) explicitly -- not left to the runtime's default or to HIP_HOST_COHERENT
        const unsigned kHostFlags = hipHostMallocMapped | hipHostMallocCoherent;
        JF_HIP(e, hipHostMalloc(&e->h_pos_pinned, sizeof(float) * S * 5, kHostFlags));
        JF_HIP(e, hipHostMalloc(&e->h_out_pinned, sizeof(float) * 2 * B * kRtMaxWgs, kHostFlags));
        JF_HIP(e, hipHostGetDevicePointer((void **)&e->hd_pos, e->h_pos_pinned, 0));
        JF_HIP(e, hipHostGetDevicePointer((void **)&e->hd_out, e->h_out_pinned, 0));
        JF_HIP(e, hipHostMalloc(&e->h_done, sizeof(int) * kRtMaxWgs, kHostFlags));
        memset(e->h_done, 0, sizeof(int) * kRtMaxWgs);
        JF_HIP(e, hipHostGetDevicePointer((void **)&e->hd_done, e->h_done, 0));
        // the error word, followed by 64 KB that timing experiments of the kernels may fill (JF_EXP_STAMPS)
        JF_HIP(e, hipHostMalloc(&e->h_err, sizeof(int) * 4 + 65536, kHostFlags));
        memset(e->h_err, 0, sizeof(int) * 4 + 65536);
        JF_HIP(e, hipHostGetDevicePointer((void **)&e->hd_err, e->h_err, 0));
        e->d_signal.assign(S, nullptr);
        JF_HIP(e, hipMalloc(&e->d_zero, sizeof(float) * kN));
        JF_HIP(e, hipMemsetAsync(e->d_zero, 0, sizeof(float) * kN, e->stream));
        e->h_sigs.assign(S, SrcSignal{e->d_zero, kN, 0});
        JF_HIP(e, h2d(e, e->d_sigs, e->h_sigs.data(), sizeof(SrcSignal) * S));
        // SoundSource::SoundSource() defaults (SoundSource.cu:3-16)
        e->pos.assign(S, HostPos{0.0f, 0.0f, 0.5f, 0.0f, 0.0f, 0.5f});

        // twiddles exp(+2 pi i j / 1024) from double
        std::vector<float2> tw(1024);
        for (int j = 0; j < 1024; j++) {
            const double a = 2.0 * 3.14159265358979323846264338327950288 * j / 1024.0;
            tw[j] = make_float2((float)cos(a), (float)sin(a));
        }
        JF_HIP(e, h2d(e, e->d_tw, tw.data(), sizeof(float2) * 1024));
        // the same values re-laid per FFT pass (jf_device.h kTw*)
        std::vector<float2> pack(kTwPack);
        for (int lane = 0; lane < 64; lane++) {
            const int a = lane & 3, i = lane >> 2;
            for (int t = 0; t < 16; t++) pack[kTwW3 + 64 * t + lane] = tw[(a * (i + 16 * t) + 768 * a) & 1023];
            for (int q = 0; q < 8; q++) pack[kTwU + 64 * q + lane] = tw[lane + 64 * q];
            for (int r = 0; r < 8; r++) pack[kTwWC + 64 * r + lane] = tw[(2 * r * lane) & 1023];
        }
        for (int m = 0; m < 16; m++)
            for (int i = 0; i < 16; i++) pack[kTwW2 + 16 * m + i] = tw[(4 * i * m) & 1023];
        for (int r = 0; r < 8; r++)
            for (int k = 0; k < 8; k++) pack[kTwWB + 8 * r + k] = tw[(16 * r * k) & 1023];
        JF_HIP(e, hipMalloc(&e->d_twpack, sizeof(float2) * kTwPack));
        JF_HIP(e, h2d(e, e->d_twpack, pack.data(), sizeof(float2) * kTwPack));

        // HRTF spectra on the GPU (read_hrtf_signals + transform_hrtfs)
        float *d_hrir = nullptr;
        const size_t hb = sizeof(float) * (size_t)e->rt.n_rows * 2 * (size_t)taps;
        JF_HIP(e, hipMalloc(&d_hrir, hb));
        hipError_t s1 = h2d(e, d_hrir, hrir, hb);
        hipError_t s2 = s1 == hipSuccess ? launch_table_build(d_hrir, e->rt.n_rows, taps, e->d_twpack, e->d_htab, e->stream) : s1;
        hipError_t s3 = s2 == hipSuccess ? hipStreamSynchronize(e->stream) : s2;
        (void)hipFree(d_hrir);
        JF_HIP(e, s3);
        return JF_OK;
    };
    rc = body();
    if (rc != JF_OK) {
        g_create_error = e->err;
        destroy_engine(e);
        e = nullptr;
    }
    if (prev_dev >= 0 && prev_dev != cfg->device) (void)hipSetDevice(prev_dev);  // leave the caller's device as it was
    *out = e;
    return rc;
}

}  // namespace

// =============================================================== C ABI ====
extern "C" {

int jf_engine_create(const jf_config *cfg, const float *hrir, int taps, jf_engine **out) {
    return jf_guard([&]() -> int {
    return create_engine(cfg, nullptr, hrir, taps, out);
    });
}

// ---- any grid of elevation rings ("SOFA / other HRTF sets", FuturePlans.md:21) ----
static const float kKemarEle[kNumElev] = {-40, -30, -20, -10, 0, 10, 20, 30, 40, 50, 60, 70, 80, 90};
static int g_kemar_count[kNumElev];

int jf_kemar_grid(jf_hrtf_grid *out) {
    if (!out) return JF_ERR_ARG;
    const RingTable &k = ring_table();
    for (int r = 0; r < kNumElev; r++) g_kemar_count[r] = k.offset[r + 1] - k.offset[r];  // (the same values whoever writes them)
    out->n_rings = kNumElev;
    out->ring_elevation = kKemarEle;
    out->ring_count = g_kemar_count;
    out->ring_step = kemar_ring_steps();
    return JF_OK;
}

static int grid_table(const jf_hrtf_grid *grid, RingTable *rt) {
    if (!grid) return fail(nullptr, JF_ERR_ARG, "null grid");
    std::string err;
    const int rc = host_grid_table(grid->n_rings, grid->ring_elevation, grid->ring_count, grid->ring_step, rt, &err);
    return rc ? fail(nullptr, rc, err) : JF_OK;
}

int jf_grid_from_positions(size_t n, const float *azimuth_deg, const float *elevation_deg, float tol_deg, jf_grid_layout *layout,
                           int *row_of) {
    return jf_guard([&]() -> int {
    if (!layout) return fail(nullptr, JF_ERR_ARG, "null layout");
    std::string err;
    const int rc = host_grid_from_positions(n, azimuth_deg, elevation_deg, tol_deg, &layout->n_rings, layout->ring_elevation,
                                            layout->ring_count, layout->ring_step, row_of, &err);
    return rc ? fail(nullptr, rc, err) : JF_OK;
    });
}

int jf_grid_rows(const jf_hrtf_grid *grid) {
    return jf_guard([&]() -> int {
    RingTable rt;
    const int rc = grid_table(grid, &rt);
    return rc ? rc : rt.n_rows;
    });
}

int jf_grid_interpolation(const jf_hrtf_grid *grid, float ele, float azi, int idx[4], float omegas[6]) {
    return jf_guard([&]() -> int {
    if (!idx || !omegas) return JF_ERR_ARG;
    RingTable rt;
    const int rc = grid_table(grid, &rt);
    return rc ? rc : host_grid_interpolation(rt, ele, azi, idx, omegas);
    });
}

int jf_grid_pick(const jf_hrtf_grid *grid, float ele, float azi) {
    return jf_guard([&]() -> int {
    RingTable rt;
    const int rc = grid_table(grid, &rt);
    if (rc) return rc;
    if (!(ele >= -1.0e6f && ele <= 1.0e6f) || !(azi > -1.0e6f && azi < 1.0e6f)) return JF_ERR_RANGE;
    return host_grid_pick(rt, ele, azi);
    });
}

int jf_engine_create_grid(const jf_config *cfg, const jf_hrtf_grid *grid, const float *hrir, int taps, jf_engine **out) {
    return jf_guard([&]() -> int {
    if (out) *out = nullptr;
    RingTable rt;
    const int rc = grid_table(grid, &rt);
    if (rc) return rc;
    return create_engine(cfg, &rt, hrir, taps, out);
    });
}

// ---- SOFA files (jf_sofa.cpp over jf_hdf5.c) ----
int jf_sofa_read(const char *path, jf_sofa_set *out) {
    return jf_guard([&]() -> int {
    if (!path || !out) return fail(nullptr, JF_ERR_ARG, "null argument");
    std::string err;
    const int rc = sofa_read(path, out, &err);
    return rc ? fail(nullptr, rc, err) : JF_OK;
    });
}

void jf_sofa_release(jf_sofa_set *set) { sofa_release(set); }

int jf_sofa_taps(const jf_sofa_set *set) {
    return jf_guard([&]() -> int {
    std::string err;
    const int rc = sofa_taps(set, &err);
    return rc < 0 ? fail(nullptr, rc, err.empty() ? "not a set read by jf_sofa_read" : err) : rc;
    });
}

int jf_sofa_table(const jf_sofa_set *set, float tol_deg, jf_grid_layout *layout, float *hrir, int taps) {
    return jf_guard([&]() -> int {
    std::string err;
    const int rc = sofa_table(set, tol_deg, layout, hrir, taps, &err);
    return rc ? fail(nullptr, rc, err) : JF_OK;
    });
}

int jf_engine_create_sofa(const jf_config *cfg, const char *path, float tol_deg, jf_engine **out) {
    return jf_guard([&]() -> int {
    if (out) *out = nullptr;
    if (!cfg || !path || !out) return fail(nullptr, JF_ERR_ARG, "null argument");
    jf_sofa_set set;
    std::string err;
    int rc = sofa_read(path, &set, &err);
    if (rc) return fail(nullptr, rc, err);
    struct Release {
        jf_sofa_set *s;
        ~Release() { sofa_release(s); }
    } release{&set};
    const int taps = sofa_taps(&set, &err);
    if (taps < 0) return fail(nullptr, taps, std::string(path) + ": " + err);
    if (taps > cfg->hrtf_len)
        return fail(nullptr, JF_ERR_ARG, std::string(path) + ": impulse responses of " + std::to_string(taps) + " taps, hrtf_len is " + std::to_string(cfg->hrtf_len));
    std::vector<float> hrir((size_t)set.n_measurements * 2 * (size_t)taps);
    jf_grid_layout lay;
    rc = sofa_table(&set, tol_deg, &lay, hrir.data(), taps, &err);
    if (rc) return fail(nullptr, rc, std::string(path) + ": " + err);
    RingTable rt;
    rc = host_grid_table(lay.n_rings, lay.ring_elevation, lay.ring_count, lay.ring_step, &rt, &err);
    if (rc) return fail(nullptr, rc, std::string(path) + ": " + err);
    return create_engine(cfg, &rt, hrir.data(), taps, out);
    });
}



int jf_engine_create_from_dir(const jf_config *cfg, const char *hrir_dir, jf_engine **out) {
    return jf_guard([&]() -> int {
    if (!cfg || !hrir_dir || !out) return fail(nullptr, JF_ERR_ARG, "null argument");
    std::vector<float> hrir;
    int taps = 0;
    std::string err;
    int rc = load_hrir_dir(hrir_dir, &hrir, &taps, &err);
    if (rc) return fail(nullptr, rc, err);
    return create_engine(cfg, nullptr, hrir.data(), taps, out);
    });
}

void jf_engine_destroy(jf_engine *e) { destroy_engine(e); }

const char *jf_last_error(const jf_engine *e) { return e ? e->err.c_str() : g_create_error.c_str(); }

int jf_frames_per_buffer(const jf_engine *e) { return e ? e->B : JF_ERR_ARG; }
int jf_pad_len(const jf_engine *e) { return e ? kN : JF_ERR_ARG; }
int jf_num_sources(const jf_engine *e) { return e ? e->S : JF_ERR_ARG; }
int jf_table_rows(const jf_engine *e) { return e ? e->rt.n_rows : JF_ERR_ARG; }

int jf_source_set_signal(jf_engine *e, int src, const float *mono, size_t n) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!valid_src(e, src) || (n && !mono) || n > 0x7fffffffu) return fail(e, JF_ERR_ARG, "bad source or signal");
    {
        const int rc = rv_ahead_discard(e);  // (the stage launched ahead read the old signal)
        if (rc) return rc;
    }
    JF_HIP(e, hipStreamSynchronize(e->stream));
    if (e->rv_side && e->rv_side_busy) JF_HIP(e, hipStreamSynchronize(e->rv_side));  // its transforms read the signals
    // The device copy always has length >= PAD_LEN so that the kernel wraps the loop with
    // one conditional subtract: a shorter signal is stored as whole repetitions of itself
    // (the looped stream is identical), an empty one as the shared zero buffer.
    float *d_new = nullptr;
    size_t n_dev = n;
    if (n) {
        const float *src_host = mono;
        std::vector<float> tiled;
        if (n < (size_t)kN) {
            const size_t reps = ((size_t)kN + n - 1) / n;
            tiled.resize(reps * n);
            for (size_t r = 0; r < reps; r++) memcpy(tiled.data() + r * n, mono, sizeof(float) * n);
            src_host = tiled.data();
            n_dev = reps * n;
        }
        JF_HIP(e, hipMalloc(&d_new, sizeof(float) * n_dev));
        hipError_t st = h2d(e, d_new, src_host, sizeof(float) * n_dev);
        if (st != hipSuccess) {
            (void)hipFree(d_new);
            JF_HIP(e, st);
        }
    }
    if (e->d_signal[src]) (void)hipFree(e->d_signal[src]);
    e->d_signal[src] = d_new;
    e->h_sigs[src] = n ? SrcSignal{d_new, (int)n_dev, 0} : SrcSignal{e->d_zero, kN, 0};
    JF_HIP(e, h2d(e, e->d_sigs + src, &e->h_sigs[src], sizeof(SrcSignal)));
    const int zero = 0;  // count = 0 (cudaPart.cu:198-199 run with a fresh source)
    if (e->rv_P > 0)  // the play position of the dry signal lives in the reverb stage
        JF_HIP(e, h2d(e, e->d_rv_count[e->cur] + src, &zero, sizeof(int)));
    else
        JF_HIP(e, h2d(e, &e->d_state[e->cur][src].count, &zero, sizeof(int)));
    return JF_OK;
    });
}

int jf_source_set_cartesian(jf_engine *e, int src, float x, float y, float z) {
    return jf_guard([&]() -> int {
    if (!valid_src(e, src)) return fail(e, JF_ERR_ARG, "bad source index");
    float rec[5], r;
    int rc = host_from_cartesian(x, y, z, rec, &r);
    if (rc) return fail(e, rc, "zero or non-finite coordinates");
    if (!elevation_ok(e, rec[0])) return fail(e, JF_ERR_RANGE, elevation_msg(e));
    std::lock_guard<std::mutex> lk(e->pos_mu);
    e->pos[src] = HostPos{rec[0], rec[1], r, x, y, z};
    return JF_OK;
    });
}

int jf_source_set_spherical(jf_engine *e, int src, float ele, float azi, float r) {
    return jf_guard([&]() -> int {
    if (!valid_src(e, src)) return fail(e, JF_ERR_ARG, "bad source index");
    float rec[5];
    host_from_spherical(ele, azi, r, rec);
    if (!elevation_ok(e, rec[0])) return fail(e, JF_ERR_RANGE, elevation_msg(e));
    if (!(fabsf(rec[1]) < 1.0e6f) || !(fabsf(r) < 3.0e38f)) return fail(e, JF_ERR_RANGE, "non-finite azimuth or radius");
    std::lock_guard<std::mutex> lk(e->pos_mu);
    e->pos[src] = HostPos{rec[0], rec[1], r, rec[2], rec[3], rec[4]};
    return JF_OK;
    });
}

int jf_source_get_position(const jf_engine *e, int src, float out[6]) {
    return jf_guard([&]() -> int {
    if (!valid_src(e, src) || !out) return JF_ERR_ARG;
    jf_engine *m = const_cast<jf_engine *>(e);
    std::lock_guard<std::mutex> lk(m->pos_mu);
    const HostPos &q = e->pos[src];
    out[0] = q.ele;
    out[1] = q.azi;
    out[2] = q.r;
    out[3] = q.x;
    out[4] = q.y;
    out[5] = q.z;
    return JF_OK;
    });
}

int jf_source_reset(jf_engine *e, int src) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!valid_src(e, src)) return fail(e, JF_ERR_ARG, "bad source index");
    {
        const int rc = rv_ahead_discard(e);
        if (rc) return rc;
    }
    JF_HIP(e, hipStreamSynchronize(e->stream));
    return reset_sources(e, src);
    });
}

int jf_position_from_spherical(float ele, float azi, float r, float out[JF_POS_FLOATS]) {
    return jf_guard([&]() -> int {
    if (!out) return JF_ERR_ARG;
    host_from_spherical(ele, azi, r, out);
    return JF_OK;
    });
}

int jf_position_from_cartesian(float x, float y, float z, float out[JF_POS_FLOATS]) {
    return jf_guard([&]() -> int {
    if (!out) return JF_ERR_ARG;
    return host_from_cartesian(x, y, z, out, nullptr);
    });
}

int jf_positions_from_spherical(size_t n, const float *ele, const float *azi, const float *r, float *out) {
    return jf_guard([&]() -> int {
    if (n && (!ele || !azi || !r || !out)) return JF_ERR_ARG;
    for (size_t i = 0; i < n; i++) host_from_spherical(ele[i], azi[i], r[i], out + 5 * i);
    return JF_OK;
    });
}

int jf_interpolation(float ele, float azi, int idx[4], float omegas[6]) {
    return jf_guard([&]() -> int {
    if (!idx || !omegas) return JF_ERR_ARG;
    return host_interpolation(ele, azi, idx, omegas);
    });
}

int jf_interpolation_ex(float ele, float azi, unsigned flags, int idx[4], float omegas[6]) {
    return jf_guard([&]() -> int {
    if (!idx || !omegas) return JF_ERR_ARG;
    return (flags & JF_FLAG_CORRECTED_INTERPOLATION) ? host_interpolation_corrected(ele, azi, idx, omegas)
                                                     : host_interpolation(ele, azi, idx, omegas);
    });
}

int jf_pick_hrtf(float ele, float azi) { return host_pick_hrtf(ele, azi); }

// ---- per-block -----------------------------------------------------------
int jf_submit_block(jf_engine *e) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!e) return JF_ERR_ARG;
    if (e->in_flight) return fail(e, JF_ERR_STATE, "a block is already in flight");
    if (device_fault(e)) return fail(e, JF_ERR_DEVICE, kHandOffMsg);
    if (e->paused.load(std::memory_order_relaxed)) {  // Audio.cu:101: nothing is consumed, output is silence
        JF_HIP(e, hipMemsetAsync(e->d_mix, 0, sizeof(float) * 2 * e->B, e->stream));
    } else {
        snapshot_positions(e, e->h_pos_pinned);
        if (e->S <= e->rt_max_sources && !e->profiling) {
            // few sources: ONE launch does descriptors, spatialisation and mix, reading the positions
            // from and writing the stereo block to pinned host memory -- no copies, one sync
            const int p = e->cur;
            e->ahead.valid = false;  // this block moves every source's old position
            ReverbParams head;
            bool head_fused = false;
            e->kernels_use_frozen = false;
            if (e->rv_ahead) {
                e->rv_ahead = false;  // the stage of this block was launched behind the last block's spatialiser: rv_ahead
            } else {
                // the wet ring is then this block's signal (written by the stage's own kernel, or by the real-time kernel's
                // waves themselves: head_fused)
                const int rc = run_reverb_stage(e, p, 1, &head, &head_fused);
                if (rc) return rc;
            }
            FusedParams P;
            P.htab = e->d_htab;
            P.tw = e->d_twpack;
            P.desc = nullptr;
            P.sigs = e->rv_P > 0 ? e->d_sigs_wet : e->d_sigs;
            P.st_in = e->d_state[p];
            P.st_out = e->d_state[p ^ 1];
            P.hist_in = e->d_hist[p];
            P.hist_out = e->d_hist[p ^ 1];
            P.pos = e->hd_pos;
            P.partial = nullptr;
            P.S = e->S;
            P.K = 1;
            P.B = e->B;
            P.G = 1;
            P.err = e->hd_err;
            P.order = e->d_order;
            P.mode = kernel_mode(e);
            // a wave per source, 8 or 16 waves to the workgroup (jf_kernels.hip: rt_block_kernel); at most kRtMaxWgs workgroups
            // = 2048 sources: beyond that a wave takes several
            const int rtw = rt_waves_per_wg(e->S);
            int wgs = (e->S + rtw - 1) / rtw;
            if (wgs > kRtMaxWgs) wgs = kRtMaxWgs;
            e->rt_seq = e->rt_seq == 0x7fffffff ? 1 : e->rt_seq + 1;
            JF_HIP(e, launch_rt_block(P, e->rt, e->hd_pos, e->hd_out, e->hd_done, e->rt_seq, wgs, head_fused ? &head : nullptr,
                                      e->stream));
            if (e->post_tr) {  // X_m of the big block this block completed, behind the kernel that wrote the block's samples
                JF_HIP(e, launch_reverb_big_side(&e->post_tr_p, nullptr, e->stream));
                e->post_tr = false;
            }
            {
                const int rc = submit_side(e);
                if (rc) return rc;
            }
            e->cur = p ^ 1;
            e->last_rt = true;
            e->rt_wgs = wgs;
            e->in_flight = true;
            if (rv_ahead_possible(e)) {
                // the NEXT block's stage, behind this block's spatialiser (jf_engine::rv_ahead)
                e->kernels_frozen = jf_debug_last_kernels(e);
                e->rv_book.rv_head = e->rv_head;
                e->rv_book.rv_blocks = e->rv_blocks;
                e->rv_book.rv_fut_m = e->rv_fut_m;
                e->rv_book.last_rv_form = e->last_rv_form;
                e->rv_book.last_plan = e->last_plan;
                e->rv_book.last_side = e->last_side;
                e->rv_book.last_catchup = e->last_catchup;
                e->rv_book.last_small_fft = e->last_small_fft;
                const int rc = run_reverb_stage(e, e->cur, 1);
                if (rc) return rc;
                e->rv_ahead = true;
                e->kernels_use_frozen = true;
            }
            return JF_OK;
        }
        JF_HIP(e, hipMemcpyAsync(e->d_pos_rt, e->h_pos_pinned, sizeof(float) * 5 * e->S, hipMemcpyHostToDevice,
                                 e->stream));
        int rc = run_blocks(e, e->d_pos_rt, 1, e->d_mix);
        if (rc) return rc;
    }
    JF_HIP(e, hipMemcpyAsync(e->h_out_pinned, e->d_mix, sizeof(float) * 2 * e->B, hipMemcpyDeviceToHost, e->stream));
    e->rt_wgs = 0;
    e->in_flight = true;
    return JF_OK;
    });
}

int jf_collect_block(jf_engine *e, float *out) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!e || !out) return JF_ERR_ARG;
    if (!e->in_flight) return fail(e, JF_ERR_STATE, "no block in flight");
    bool landed = false;
    if (e->rt_wgs > 0) {
        // The real-time kernel says when its blocks are in host memory: poll its words (a few microseconds of spinning on the
        // audio thread, as cudaStreamSynchronize does in the reference, Audio.cu:107) -- and fall back to the stream if they
        // do not come (a faulting kernel must surface as an error, not as a spin).
        // The spin is bounded by TIME (kRtPollNs: two milliseconds, a third of a 256-sample block's real time), read every
        // 256 polls; after that the stream synchronisation below takes over.
        const volatile int *done = e->h_done;
        timespec t0;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (long spins = 0; !landed; spins++) {
            landed = true;
            for (int g = 0; g < e->rt_wgs; g++) landed = landed && done[g] == e->rt_seq;
            if (landed) break;
            __builtin_ia32_pause();
            if ((spins & 255) == 255) {
                timespec t1;
                clock_gettime(CLOCK_MONOTONIC, &t1);
                if ((t1.tv_sec - t0.tv_sec) * 1000000000L + (t1.tv_nsec - t0.tv_nsec) > kRtPollNs) break;
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (!landed) JF_HIP(e, hipStreamSynchronize(e->stream));
    if (device_fault(e)) {  // per-block calls with more than rt_max_sources sources run the pair kernel too
        e->in_flight = false;
        return fail(e, JF_ERR_DEVICE, kHandOffMsg);
    }
    memcpy(out, e->h_out_pinned, sizeof(float) * 2 * e->B);
    // the real-time kernel's workgroups each left the sum of their sources: add them in workgroup order
    for (int g = 1; g < e->rt_wgs; g++) {
        const float *pg = e->h_out_pinned + (size_t)g * 2 * e->B;
        for (int n = 0; n < 2 * e->B; n++) out[n] += pg[n];
    }
    float peak = 0.0f;
    for (int n = 0; n < 2 * e->B; n++) peak = fmaxf(peak, fabsf(out[n]));
    e->last_peak = peak;
    e->in_flight = false;
    return JF_OK;
    });
}

int jf_process_block(jf_engine *e, float *out) {
    return jf_guard([&]() -> int {
    int rc = jf_submit_block(e);
    if (rc) return rc;
    return jf_collect_block(e, out);
    });
}

int jf_callback(jf_engine *e, float *out) {
    return jf_guard([&]() -> int {
    if (!e || !out) return JF_ERR_ARG;
    int rc;
    if (e->have_prev) {
        rc = jf_collect_block(e, out);
        if (rc) return rc;
    } else {
        memset(out, 0, sizeof(float) * 2 * e->B);  // intermediate[] before the first block
    }
    rc = jf_submit_block(e);
    if (rc) return rc;
    e->have_prev = true;
    return JF_OK;
    });
}

int jf_pa_callback(const void *, void *output, unsigned long frames, const void *, unsigned long, void *user) {
    return jf_guard([&]() -> int {
    jf_engine *e = (jf_engine *)user;
    if (!output) return 0;
    // a stream opened with another buffer size, or an engine error: hand PortAudio silence, never garbage
    if (!e || frames != (unsigned long)e->B || jf_callback(e, (float *)output) != JF_OK)
        memset(output, 0, sizeof(float) * 2 * frames);
    return 0;
    });
}

int jf_set_mode(jf_engine *e, int mode) {
    return jf_guard([&]() -> int {
    if (!e || (mode != JF_MODE_FD_COMPLEX && mode != JF_MODE_FD_BASIC)) return fail(e, JF_ERR_ARG, "unknown mode");
    e->mode.store(mode, std::memory_order_relaxed);  // read at the next block, like Data::type (Audio.cu:104)
    return JF_OK;
    });
}

int jf_set_pause(jf_engine *e, int paused) {
    return jf_guard([&]() -> int {
    if (!e) return JF_ERR_ARG;
    e->paused.store(paused != 0, std::memory_order_relaxed);
    return JF_OK;
    });
}




// ---- batch -----------------------------------------------------------------
int jf_batch_upload_positions(jf_engine *e, int total_blocks, const float *positions) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!e || total_blocks <= 0 || !positions) return fail(e, JF_ERR_ARG, "bad trajectory");
    JF_HIP(e, hipStreamSynchronize(e->stream));
    e->traj_gen++;
    e->ahead.valid = false;
    const size_t bytes = sizeof(float) * 5 * (size_t)e->S * (size_t)total_blocks;
    if (total_blocks > e->traj_blocks) {
        (void)hipFree(e->d_traj);
        e->d_traj = nullptr;
        e->traj_blocks = 0;
        JF_HIP(e, hipMalloc(&e->d_traj, bytes));
    }
    e->traj_blocks = total_blocks;
    JF_HIP(e, h2d(e, e->d_traj, positions, bytes));
    // how many items of every block move (their (ele, azi) differ from the block before; block 0 counts as staying):
    // what decides whether a run reads pre-interpolated rows (jf_engine::interp_use)
    e->traj_moved.assign((size_t)total_blocks + 1, 0u);
    if (e->interp_avail)
        for (int b = 1; b < total_blocks; b++) {
            const float *p1 = positions + (size_t)b * e->S * 5, *p0 = p1 - (size_t)e->S * 5;
            unsigned n = 0;
            for (int s = 0; s < e->S; s++) n += p1[5 * s] != p0[5 * s] || p1[5 * s + 1] != p0[5 * s + 1];
            e->traj_moved[(size_t)b + 1] = e->traj_moved[b] + n;
        }
    // Processing order of the pair kernel: a unit sums G sources that are next to each other in this order.  With
    // automatic grouping the sources are ordered by the table row nearest to their first position, so that the units a
    // compute unit works on at a time read neighbouring rows of the 5.8 MB table (the L2 of an XCD holds 4 MB); the mix is
    // the same sum in another association.  jf_debug_set_source_group pins consecutive sources (identity order).
    const bool want_sorted = e->src_group == 0 && e->S > 1;
    if (want_sorted || e->sorted_order) {
        std::vector<std::pair<int, int>> key((size_t)e->S);
        for (int s = 0; s < e->S; s++) {
            const float *p = positions + 5 * (size_t)s;
            const bool ok = p[0] > -1.0e6f && p[0] < 1.0e6f && p[1] > -1.0e6f && p[1] < 1.0e6f;
            key[s] = {want_sorted && ok ? host_grid_pick(e->rt, p[0], p[1]) : 0, s};
        }
        std::stable_sort(key.begin(), key.end());
        for (int s = 0; s < e->S; s++) e->order[s] = key[s].second;
        JF_HIP(e, h2d(e, e->d_order, e->order.data(), sizeof(int) * e->S));
        e->sorted_order = want_sorted;
    }
    return JF_OK;
    });
}

int jf_batch_run(jf_engine *e, int first_block, int n_blocks, float *d_out_mix) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!e) return JF_ERR_ARG;
    if (n_blocks <= 0 || n_blocks > e->maxK) return fail(e, JF_ERR_ARG, "n_blocks exceeds max_batch_blocks");
    if (first_block < 0 || first_block + n_blocks > e->traj_blocks)
        return fail(e, JF_ERR_ARG, "window outside the uploaded trajectory");
    if (e->in_flight) return fail(e, JF_ERR_STATE, "a per-block call is in flight");
    const int rc = run_blocks(e, e->d_traj + (size_t)first_block * e->S * 5, n_blocks, d_out_mix ? d_out_mix : e->d_mix,
                              first_block);
    e->own_mix_blocks = rc == JF_OK && !d_out_mix ? n_blocks : 0;  // what jf_batch_fetch may hand out
    return rc;
    });
}

int jf_device_numa_node(int device, int *node) {
    return jf_guard([&]() -> int {
    if (!node) return JF_ERR_ARG;
    *node = -1;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) {
        (void)hipGetLastError();
        return fail(nullptr, JF_ERR_DEVICE, "no such HIP device");
    }
    for (char *c = bus; *c; c++) *c = (char)tolower((unsigned char)*c);  // sysfs spells the address in lower case
    const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/numa_node";
    if (FILE *f = fopen(path.c_str(), "r")) {
        int n = -1;
        if (fscanf(f, "%d", &n) == 1) *node = n;
        fclose(f);
    }
    return JF_OK;
    });
}

int jf_pin_thread_to_device(int device) {
    return jf_guard([&]() -> int {
    int node = -1;
    const int rc = jf_device_numa_node(device, &node);
    if (rc != JF_OK) return rc;
    if (node < 0) return fail(nullptr, JF_ERR_STATE, "the system does not say which NUMA node the device is on");
    // /sys/devices/system/node/node<N>/cpulist: "0-63,128-191"
    const std::string path = "/sys/devices/system/node/node" + std::to_string(node) + "/cpulist";
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return fail(nullptr, JF_ERR_STATE, "no CPU list for the device's NUMA node");
    char line[4096] = {0};
    const bool got = fgets(line, sizeof(line), f) != nullptr;
    fclose(f);
    if (!got) return fail(nullptr, JF_ERR_STATE, "no CPU list for the device's NUMA node");
    cpu_set_t allowed, want;
    CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return fail(nullptr, JF_ERR_STATE, "sched_getaffinity failed");
    int n_set = 0;
    for (const char *p = line; *p;) {
        char *end = nullptr;
        const long a = strtol(p, &end, 10);
        if (end == p) break;
        long b = a;
        p = end;
        if (*p == '-') {
            b = strtol(p + 1, &end, 10);
            p = end;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++)
            if (c >= 0 && CPU_ISSET((int)c, &allowed)) {
                CPU_SET((int)c, &want);
                n_set++;
            }
        while (*p == ',' || *p == ' ' || *p == '\n') p++;
    }
    if (n_set == 0) return fail(nullptr, JF_ERR_STATE, "none of the CPUs of the device's NUMA node is allowed to this process");
    if (sched_setaffinity(0, sizeof(want), &want) != 0) return fail(nullptr, JF_ERR_STATE, "sched_setaffinity failed");
    return JF_OK;
    });
}

int jf_synchronize(jf_engine *e) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!e) return JF_ERR_ARG;
    if (e->rv_side && e->rv_side_busy) JF_HIP(e, hipStreamSynchronize(e->rv_side));
    JF_HIP(e, hipStreamSynchronize(e->stream));
    if (device_fault(e)) return fail(e, JF_ERR_DEVICE, kHandOffMsg);
    return JF_OK;
    });
}

int jf_process_batch(jf_engine *e, int n_blocks, const float *positions, float *out_mix) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!e || !positions || !out_mix || n_blocks <= 0) return fail(e, JF_ERR_ARG, "bad batch arguments");
    int rc = jf_batch_upload_positions(e, n_blocks, positions);
    if (rc) return rc;
    const size_t blk = (size_t)2 * e->B;
    for (int b0 = 0; b0 < n_blocks; b0 += e->maxK) {
        const int k = n_blocks - b0 < e->maxK ? n_blocks - b0 : e->maxK;
        rc = jf_batch_run(e, b0, k, nullptr);
        if (rc) return rc;
        JF_HIP(e, hipMemcpyAsync(out_mix + (size_t)b0 * blk, e->d_mix, sizeof(float) * blk * k, hipMemcpyDeviceToHost,
                                 e->stream));
        JF_HIP(e, hipStreamSynchronize(e->stream));
        if (device_fault(e)) return fail(e, JF_ERR_DEVICE, kHandOffMsg);
    }
    // n_blocks callbacks have run: the sources stand where the last of them read them
    return jf_sources_set_latched(e, positions + (size_t)(n_blocks - 1) * e->S * JF_POS_FLOATS);
    });
}

int jf_sources_set_latched(jf_engine *e, const float *records) {
    return jf_guard([&]() -> int {
    if (!e || !records) return JF_ERR_ARG;
    std::lock_guard<std::mutex> lk(e->pos_mu);
    for (int s = 0; s < e->S; s++) {
        const float *r = records + (size_t)s * JF_POS_FLOATS;
        e->pos[s] = HostPos{r[0], r[1], sqrtf(r[2] * r[2] + r[3] * r[3] + r[4] * r[4]), r[2], r[3], r[4]};
    }
    return JF_OK;
    });
}

int jf_batch_fetch(jf_engine *e, int n_blocks, float *out_mix) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!e || !out_mix || n_blocks <= 0 || n_blocks > e->maxK) return fail(e, JF_ERR_ARG, "bad fetch arguments");
    if (n_blocks > e->own_mix_blocks)
        return fail(e, JF_ERR_STATE, "jf_batch_fetch: the last jf_batch_run did not leave that many blocks in the engine's own buffer "
                                     "(it was given a device pointer, failed, or has not run)");
    JF_HIP(e, hipMemcpyAsync(out_mix, e->d_mix, sizeof(float) * 2 * e->B * (size_t)n_blocks, hipMemcpyDeviceToHost, e->stream));
    JF_HIP(e, hipStreamSynchronize(e->stream));
    if (device_fault(e)) return fail(e, JF_ERR_DEVICE, kHandOffMsg);
    return JF_OK;
    });
}






























float jf_last_block_peak(const jf_engine *e) { return e ? e->last_peak : 0.0f; }


// ---- WAV -----------------------------------------------------------------------
int jf_wav_read_mono(const char *path, float **out, size_t *n_frames, int *sample_rate) {
    return jf_guard([&]() -> int {
    if (!path || !out || !n_frames) return JF_ERR_ARG;
    std::string err;
    int rc = wav_read_mono(path, out, n_frames, sample_rate, &err);
    if (rc) g_create_error = err;
    return rc;
    });
}

int jf_wav_write_stereo24(const char *path, const float *interleaved, size_t n_frames, int sample_rate) {
    return jf_guard([&]() -> int {
    if (!path || (!interleaved && n_frames)) return JF_ERR_ARG;
    std::string err;
    int rc = wav_write_stereo24(path, interleaved, n_frames, sample_rate, &err);
    if (rc) g_create_error = err;
    return rc;
    });
}

void jf_free(void *p) { free(p); }

}  // extern "C"
