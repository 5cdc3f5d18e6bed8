// jf_engine_reverb.cpp -- the host side of the convolution reverb: what a call's stage consists of
// (run_reverb_stage: the plan of transforms, products and tails; the work that goes to the side stream; the stage of the next
// block launched ahead and taken back), and the two public entry points jf_reverb_set_ir / jf_reverb_rms_gain (cudaPart.cu:65-205).
#include "jf_engine_internal.h"

// head_out (one-block calls through the real-time kernel; may be null): if the stage's head can run inside that kernel, it is
// NOT launched here -- *head_out receives its parameters, *head_fused says so, and e->post_tr holds what must follow the kernel
int run_reverb_stage(jf_engine *e, int p, int K, ReverbParams *head_out, bool *head_fused) {
    if (head_fused) *head_fused = false;
    e->post_tr = false;
    if (e->rv_P <= 0) return JF_OK;
    if (e->side_tr) {
        // the last stage's work for the side stream was never submitted (a launch between that stage and submit_side failed
        // and the caller went on): it goes first -- the transform it holds is of samples the dry ring still has
        const int rc = submit_side(e);
        if (rc) return rc;
    }
    EventPair *er = nullptr;
    if (e->profiling >= 2 && e->timed_now) {
        er = next_events(e, e->ev_reverb);
        if (!er) return fail(e, JF_ERR_DEVICE, "hipEventCreate failed");
        JF_HIP(e, hipEventRecord(er->a, e->stream));
    }
    ReverbParams R;
    R.tw = e->d_tw;
    R.dry = e->d_sigs;
    R.dry_count_in = e->d_rv_count[p];
    R.dry_count_out = e->d_rv_count[p ^ 1];
    R.prev_in = e->d_rv_prev[p];
    R.prev_out = e->d_rv_prev[p ^ 1];
    R.fdl = e->d_rv_fdl;
    R.hspec = e->d_rv_hspec;
    R.wet = e->d_rv_wet;
    R.st_in = e->d_state[p];
    R.S = e->S;
    R.K = K;
    R.B = e->B;
    R.P = e->rv_P;
    R.Rg = e->rv_Rg;
    R.Wr = e->rv_Wr;
    R.head = e->rv_head;
    R.mac_form = e->rv_form;
    ReverbPlan plan;
    plan.big = e->rv_P1 > 0;
    bool defer_small = false, need_small = true;
    // One-block calls -- the real-time shape -- keep the big partitions' kernels off the block's critical path.  The head
    // covers TWO big blocks of taps (2 M partitions of B), so TAIL(m) = sum_{q >= 2} X_{m+1-q} H'_q needs nothing newer than
    // X_{m-1}, which exists a whole big block before big block m begins.  When a one-block call completes big block mb, the
    // transform X_{mb+1}, the products of TAIL(mb + 2) and their inverse transform go to a second stream BEHIND the block's
    // spatialiser (submit_side); the first block to read the result is seventeen blocks away, and the stage of the next call
    // that is not such a one-block call -- or the next one that puts work there -- makes the engine's stream wait for that
    // stream (an event).  (With a head of M partitions TAIL(mb + 1) needed X_{mb+1} and was needed by the very next block: in
    // line, that block and the one before it cost 40 and 9 us more than the other fourteen at configs[4], 256 sources.)
    // Calls that pin a form, batch calls and profiled calls do everything in line on the engine's stream.
    const bool async_ok = plan.big && K == 1 && e->rv_async && e->rv_form == 0 && e->profiling < 2 && e->rv_side != nullptr;
    const bool completes = plan.big && (e->rv_blocks + K) / e->rv_M > e->rv_blocks / e->rv_M;  // transforms in this call
    if (e->rv_side_busy && (!async_ok || completes || e->rv_side_urgent)) {
        JF_HIP(e, hipStreamWaitEvent(e->stream, e->rv_ev_side, 0));
        e->rv_side_busy = e->rv_side_urgent = false;
    }
    ReverbBigParams &s_tr = e->side_p[0], &s_prod = e->side_p[1];
    e->last_side.clear();
    const long long fut_m_before = e->rv_fut_m;
    bool side_wanted = false;
    if (plan.big) {
        // Absolute block indices j0 .. j1 - 1; big block m = blocks 16 m .. 16 m + 15.
        const long long j0 = e->rv_blocks;
        const int B1 = e->rv_B1, R1 = e->rv_R1, Rn = e->rv_Rn, Fn = e->rv_Fn, M = e->rv_M;
        R.dryring = e->d_rv_dryring;
        R.Rd = Rn * B1;
        R.dry_pos0 = (int)((j0 * e->B) % R.Rd);
        R.fut = e->d_rv_fut;
        R.F = Fn * B1;
        R.fut_pos0 = (int)((j0 * e->B) % R.F);
        ReverbBigParams G;
        G.tw1 = e->d_rv_tw1;
        G.dryring = e->d_rv_dryring;
        G.dry = e->d_sigs;
        G.dry_count_in = e->d_rv_count[p];
        G.dry_pos0 = R.dry_pos0;
        G.fdl1 = e->d_rv_fdl1;
        G.hspec1 = e->d_rv_hspec1;
        G.ybig = e->d_rv_ybig;
        G.fut = e->d_rv_fut;
        G.wet = e->d_rv_wet;
        G.st_in = e->d_state[p];
        G.S = e->S;
        G.B = e->B;
        G.B1 = B1;
        G.P1 = e->rv_P1;
        G.R1 = R1;
        G.Rn = Rn;
        G.Fn = Fn;
        G.Wr = e->rv_Wr;
        G.M = M;
        G.NP = e->rv_P1 + 17;
        auto mod = [](long long a, int n) { return (int)(((a % n) + n) % n); };
        const ReverbSchedule sc = host_reverb_schedule(j0, K, M, e->rv_fut_m);  // which X_m, FULL, TAIL and ranges: jf_host.cpp
        e->rv_fut_m = sc.fut_m;
        // whole big blocks up to the call's end: the small transforms of its last blocks are put off (rv_small_stale) ...
        defer_small = e->rv_lazy_small && sc.n_mid > 0 && sc.kn[1] == 0 && sc.n_tr > 0;
        // ... and a call that takes a block through the head needs the ones an earlier call put off, first
        need_small = sc.n_mid == 0 || sc.kn[0] > 0 || sc.kn[1] > 0;
        plan.transforms = G;
        if (defer_small) {
            plan.transforms.state_out = 1;
            plan.transforms.dryring_out = e->d_rv_dryring;
            plan.transforms.prev_out = e->d_rv_prev[p ^ 1];
            plan.transforms.dry_count_out = e->d_rv_count[p ^ 1];
            plan.transforms.call_samples = K * e->B;
        }
        plan.transforms.n_tr = sc.n_tr;
        plan.transforms.tr_slot_first = mod(sc.m_lo, R1);
        plan.transforms.tr_rel_first = (int)((sc.m_lo - 2) * B1 - j0 * e->B);
        const int n_mid = sc.n_mid;
        plan.middle = G;
        plan.middle.n_prod = n_mid;
        plan.middle.anchor_slot_first = mod(sc.ma + 1, R1);  // FULL(m) is anchored at X_{m+1}
        plan.middle.h_first = 0;
        plan.middle.n_part = e->rv_P1 + 1;
        plan.middle.to_wet = 1;
        plan.middle.wet_k0 = (int)(sc.ma * M - j0);
        plan.n_ranges = sc.n_ranges;
        for (int r = 0; r < 2; r++) {
            plan.kb[r] = sc.kb[r];
            plan.kn[r] = sc.kn[r];
        }
        R.copy_lo = sc.copy_lo;
        R.copy_hi = sc.copy_hi;
        R.skip_lo = sc.skip_lo;
        R.skip_hi = sc.skip_hi;
        if (defer_small) {
            R.copy_hi = R.copy_lo;   // nothing is copied, nothing behind the front blocks is transformed
            R.skip_hi = K;
        }
        auto tail_for = [&](long long m) {  // TAIL(m) = sum_{q = 2 .. P1} X_{m+1-q} H'_q: the newest spectrum is X_{m-1}
            ReverbBigParams T = G;
            T.n_prod = 1;
            T.anchor_slot_first = mod(m - 1, R1);
            T.h_first = 2;
            T.n_part = e->rv_P1 - 1;
            T.to_wet = 0;
            T.fut_first = mod(m, Fn);
            return T;
        };
        plan.tail_early = G;
        plan.tail_late = G;
        if (sc.tail_early >= 0) plan.tail_early = tail_for(sc.tail_early);
        if (sc.tail_late >= 0) plan.tail_late = tail_for(sc.tail_late);
        if (async_ok && sc.n_tr > 0) {
            // the block completes big block mb: X_{mb+1} and, with it, TAIL(mb + 2) -- which the block after the next
            // sixteen is the first to read
            const long long mb = j0 / M;
            const std::string b1 = std::to_string(B1);
            s_tr = plan.transforms;
            plan.transforms.n_tr = 0;
            // all 2 B1 samples from the dry ring -- the head kernel has just written this block's there -- and none from the
            // signal at the play position, which the next call moves on while the side stream may still be reading
            s_tr.dry_pos0 = (R.dry_pos0 + e->B) % R.Rd;
            s_tr.tr_rel_first -= e->B;
            // (if nobody has formed TAIL(mb + 1) -- the run of one-block calls began inside this big block -- both, and the
            // next call waits for them)
            const bool both = e->rv_fut_m < mb + 1;
            s_prod = tail_for(both ? mb + 1 : mb + 2);
            s_prod.n_prod = both ? 2 : 1;
            s_prod.ybig = e->d_rv_yacc;
            s_prod.mac_wgs = both ? 0 : e->rv_side_wgs;
            e->side_urgent = both;
            e->side_fut_m = mb + 2;
            side_wanted = true;
            e->last_side = "reverb_big_fft_kernel<" + b1 + ",1>@side;reverb_big_mac_kernel<" + b1 + ",1>@side;reverb_big_ifft_kernel<" +
                           b1 + ",1>@side;";
        }
        if (plan.transforms.n_tr > e->rv_steps_max || n_mid > e->rv_steps_max)
            return fail(e, JF_ERR_STATE, "reverb: more big-partition steps in a call than buffers");
    }
    e->last_small_fft = K - (R.skip_hi - R.skip_lo) > 0;
    e->last_catchup = false;
    if (e->rv_small_stale && need_small) {
        // the last 2 M - 1 blocks before this call, from the dry ring: block rv_blocks - n .. rv_blocks - 1, slots rv_head - n ..
        ReverbParams C = R;
        const int n = 2 * e->rv_M - 1;
        C.K = n;
        C.catchup = 1;
        C.head = (int)((((long long)e->rv_head - n) % e->rv_Rg + e->rv_Rg) % e->rv_Rg);
        C.dry_pos0 = (int)((((e->rv_blocks - n) * e->B) % R.Rd + R.Rd) % R.Rd);
        C.copy_lo = C.copy_hi = C.skip_lo = C.skip_hi = 0;
        JF_HIP(e, launch_reverb_catchup(C, e->stream));
        e->rv_small_stale = false;
        e->last_catchup = true;
    }
    plan.head_fused = head_out != nullptr && K == 1 && e->rv_head_fused && e->rv_form == 0 && e->profiling < 2 &&
                      e->rv_P <= kRvFusedHeadMax && rt_waves_per_wg(e->S) == 8 && (e->B == 64 || e->B == 128 || e->B == 256);
    {
        const hipError_t q = launch_reverb(R, &plan, e->stream, &e->last_rv_form);
        if (q != hipSuccess) {
            e->rv_fut_m = fut_m_before;  // nothing of this call's schedule has been formed
            JF_HIP(e, q);
        }
    }
    if (plan.head_fused) {
        R.kb = 0;
        R.kn = 1;
        *head_out = R;
        *head_fused = true;
        if (plan.transforms.n_tr > 0) {  // (in line: the block completed a big block and the side stream is not used)
            e->post_tr = true;
            e->post_tr_p = plan.transforms;
        }
    }
    e->last_plan = plan;
    e->side_tr = side_wanted;

    if (er) JF_HIP(e, hipEventRecord(er->b, e->stream));
    e->rv_head = (e->rv_head + K) % e->rv_Rg;
    e->rv_blocks += K;
    if (defer_small) e->rv_small_stale = true;  // (a stale state from before is obsolete now: older than the head reaches)
    return JF_OK;
}

// What run_reverb_stage left for the side stream, submitted once the block's own kernels (the spatialiser's too) are in the
// engine's stream: the side stream waits for them -- it then works beside what FOLLOWS the block (in real time: nothing; in a
// run of calls back to back: the next blocks, which find room because its long kernel is launched narrow) -- and the block's
// own kernels are not held up by it.
int submit_side(jf_engine *e) {
    if (!e->side_tr) return JF_OK;
    JF_HIP(e, hipEventRecord(e->rv_ev_main, e->stream));
    JF_HIP(e, hipStreamWaitEvent(e->rv_side, e->rv_ev_main, 0));
    JF_HIP(e, launch_reverb_big_side(&e->side_p[0], &e->side_p[1], e->rv_side));
    // launched: TAIL up to side_fut_m will be there (nothing before this line may claim so -- a stage whose launches failed
    // must leave the schedule asking for them again)
    e->side_tr = false;
    if (e->rv_fut_m < e->side_fut_m) e->rv_fut_m = e->side_fut_m;
    e->rv_side_urgent = e->side_urgent;
    e->rv_side_busy = true;
    JF_HIP(e, hipEventRecord(e->rv_ev_side, e->rv_side));
    return JF_OK;
}

// The next block's stage launched ahead (jf_engine::rv_ahead) is taken back: see there.
int rv_ahead_discard(jf_engine *e) {
    if (!e->rv_ahead) return JF_OK;
    JF_HIP(e, hipStreamSynchronize(e->stream));  // nothing of it is still being written
    e->rv_head = e->rv_book.rv_head;
    e->rv_blocks = e->rv_book.rv_blocks;
    e->rv_fut_m = e->rv_book.rv_fut_m;
    e->last_rv_form = e->rv_book.last_rv_form;
    e->last_plan = e->rv_book.last_plan;
    e->last_side = e->rv_book.last_side;
    e->last_catchup = e->rv_book.last_catchup;
    e->last_small_fft = e->rv_book.last_small_fft;
    e->side_tr = false;  // (what the stage wanted on the side stream had not been submitted yet)
    e->post_tr = false;
    // the stage may have made the engine's stream wait for the side stream (and cleared these): waited it has, so leave them
    e->rv_ahead = false;
    e->kernels_use_frozen = false;
    return JF_OK;
}

// May the stage of the block after the one just launched go ahead?  A plain head only: the block completes no big block (its
// transforms would have to follow its spatialiser), owes no TAIL, the side stream has nothing urgent, nothing is put off.
bool rv_ahead_possible(const jf_engine *e) {
    if (e->rv_P <= 0 || !e->rv_ahead_on || e->rv_form != 0 || e->profiling || e->rv_head_fused || e->rv_small_stale) return false;
    if (e->S > e->rt_max_sources || e->S >= 2048) return false;  // (the one-launch path; the one-block head kernel's range)
    if (e->paused.load(std::memory_order_relaxed)) return false;
    if (e->rv_P1 > 0) {
        if (e->rv_side_urgent || e->side_tr) return false;
        const ReverbSchedule sc = host_reverb_schedule(e->rv_blocks, 1, e->rv_M, e->rv_fut_m);
        if (sc.tail_early >= 0 || sc.tail_late >= 0) return false;
        // a block that completes a big block: only if its transforms and products go to the side stream (they are submitted
        // behind ITS spatialiser, by the call that consumes the stage: side_tr stays pending till then)
        if (sc.n_tr > 0 && !(e->rv_async && e->rv_side != nullptr)) return false;
    }
    return true;
}

void free_reverb(jf_engine *e) {
    quiesce_side(e);
    e->side_tr = false;
    e->rv_small_stale = false;
    e->last_side.clear();
    (void)hipFree(e->d_rv_yacc);
    e->d_rv_yacc = nullptr;
    (void)hipFree(e->d_rv_hspec);
    (void)hipFree(e->d_rv_fdl);
    (void)hipFree(e->d_rv_wet);
    (void)hipFree(e->d_sigs_wet);
    for (int i = 0; i < 2; i++) {
        (void)hipFree(e->d_rv_prev[i]);
        (void)hipFree(e->d_rv_count[i]);
        e->d_rv_prev[i] = nullptr;
        e->d_rv_count[i] = nullptr;
    }
    (void)hipFree(e->d_rv_tw1);
    (void)hipFree(e->d_rv_hspec1);
    (void)hipFree(e->d_rv_fdl1);
    (void)hipFree(e->d_rv_ybig);
    (void)hipFree(e->d_rv_dryring);
    (void)hipFree(e->d_rv_fut);
    e->d_rv_tw1 = e->d_rv_hspec1 = e->d_rv_fdl1 = e->d_rv_ybig = nullptr;
    e->d_rv_dryring = e->d_rv_fut = nullptr;
    e->d_rv_hspec = nullptr;
    e->d_rv_fdl = nullptr;
    e->d_rv_wet = nullptr;
    e->d_sigs_wet = nullptr;
    e->rv_P = e->rv_Rg = e->rv_Wr = e->rv_head = 0;
    e->rv_P_total = e->rv_P1 = e->rv_B1 = e->rv_M = e->rv_R1 = e->rv_Rn = e->rv_Fn = e->rv_steps_max = 0;
    e->rv_blocks = e->rv_fut_m = 0;
    e->last_plan = ReverbPlan();
}

extern "C" {

// ---- convolution reverb ----------------------------------------------------
int jf_reverb_set_ir(jf_engine *e, const float *ir, size_t n_ir, float gain) {
    return jf_guard([&]() -> int {
    DeviceGuard bind(e);
    if (!e || (n_ir && !ir) || n_ir > (size_t)1 << 26) return fail(e, JF_ERR_ARG, "bad impulse response");
    if (e->in_flight) return fail(e, JF_ERR_STATE, "a per-block call is in flight");
    {
        const int rc = rv_ahead_discard(e);
        if (rc) return rc;
    }
    JF_HIP(e, hipStreamSynchronize(e->stream));
    const bool was_on = e->rv_P > 0;
    free_reverb(e);
    if (n_ir == 0) {
        if (was_on) return reset_sources(e, -1);
        return JF_OK;
    }
    const int B = e->B;
    if (B != 64 && B != 128 && B != 256)
        return fail(e, JF_ERR_ARG, "reverb needs frames_per_buffer of 64, 128 or 256 (FFT of 2 blocks)");
    if ((long long)e->maxK * B >= (1LL << 30))  // the stage's play positions are 32-bit sums of a position and K B samples
        return fail(e, JF_ERR_ARG, "max_batch_blocks too large for the reverb stage");
    const size_t S = (size_t)e->S;
    const int P_total = (int)((n_ir + B - 1) / B);
    // Non-uniform partitioning for a response of at least three big partitions (unless a uniform form is pinned, or
    // jf_debug_set_reverb_partitioning says otherwise): the stage below is then the head of rv_big_blocks(B) partitions of B
    const bool nonuniform = e->rv_partitioning == 2 ||
                            (e->rv_partitioning == 0 && e->rv_form == 0 && P_total >= 3 * rv_big_blocks(B));
    const int M = rv_big_blocks(B);
    const int P = nonuniform ? 2 * M : P_total;  // the head: two big partitions' worth of taps (run_reverb_stage says why)
    const int B1 = M * B;
    const int P1 = nonuniform ? (int)((n_ir > (size_t)B1 ? n_ir - B1 : 0) + B1 - 1) / B1 : 0;
    const int steps_max = e->maxK / M + 1;               // big blocks one call can complete
    const int R1 = P1 + 16 + steps_max + 4, Rn = steps_max + 3, Fn = 4;  // (+ 16: the product kernel reads whole groups of 16 slots)
    const int Rg = P + e->maxK;                          // slots a call may still read + the ones it writes
    const int Wr = (e->maxK + kN / B + 1) * B;           // >= PAD_LEN, multiple of B
    float *d_ir = nullptr;
    auto body = [&]() -> int {
        // each followed by the compact copies of its packed bin-0 pairs: h0[P], fdl0[S][Rg]
        JF_HIP(e, hipMalloc(&e->d_rv_hspec, sizeof(float2) * ((size_t)P * B + P)));
        JF_HIP(e, hipMalloc(&e->d_rv_fdl, sizeof(float2) * (S * Rg * B + S * Rg)));
        JF_HIP(e, hipMalloc(&e->d_rv_wet, sizeof(float) * S * Wr));
        JF_HIP(e, hipMalloc(&e->d_sigs_wet, sizeof(SrcSignal) * S));
        for (int i = 0; i < 2; i++) {
            JF_HIP(e, hipMalloc(&e->d_rv_prev[i], sizeof(float) * S * B));
            JF_HIP(e, hipMalloc(&e->d_rv_count[i], sizeof(int) * S));
            JF_HIP(e, hipMemsetAsync(e->d_rv_prev[i], 0, sizeof(float) * S * B, e->stream));
            JF_HIP(e, hipMemsetAsync(e->d_rv_count[i], 0, sizeof(int) * S, e->stream));
        }
        std::vector<SrcSignal> wet(S);
        for (size_t s = 0; s < S; s++) wet[s] = SrcSignal{e->d_rv_wet + s * Wr, Wr, 0};
        JF_HIP(e, h2d(e, e->d_sigs_wet, wet.data(), sizeof(SrcSignal) * S));
        JF_HIP(e, hipMalloc(&d_ir, sizeof(float) * n_ir));
        JF_HIP(e, h2d(e, d_ir, ir, sizeof(float) * n_ir));
        // 1/B: normalisation of the B-point inverse used for the 2B-point real transform
        JF_HIP(e, launch_reverb_ir(d_ir, (int)n_ir, P, B, gain / (float)B, e->d_tw, e->d_rv_hspec, e->stream));
        if (P1 > 0) {
            // twiddles exp(+2 pi i j / (2 B1)), j < 2 B1 (a full circle), from double
            std::vector<float2> tw1((size_t)2 * B1);
            for (int j = 0; j < 2 * B1; j++) {
                const double a = 3.14159265358979323846264338327950288 * j / (double)B1;
                tw1[j] = make_float2((float)cos(a), (float)sin(a));
            }
            // ... followed by the transforms' own selection of them, laid out the way their lanes read them (jf_reverb.hip:
            // BigTwiddles::load)
            const int n_pack = big_twiddle_pack_len(B1);
            for (int k = 0; k < n_pack; k++) tw1.push_back(tw1[(size_t)big_twiddle_pack_index(B1, k)]);
            JF_HIP(e, hipMalloc(&e->d_rv_tw1, sizeof(float2) * tw1.size()));
            JF_HIP(e, h2d(e, e->d_rv_tw1, tw1.data(), sizeof(float2) * tw1.size()));
            const size_t NP = (size_t)P1 + 17;  // H'_0 .. H'_P1 and 16 partitions of zeros
            JF_HIP(e, hipMalloc(&e->d_rv_hspec1, sizeof(float2) * (NP * B1 + NP)));
            JF_HIP(e, hipMemsetAsync(e->d_rv_hspec1, 0, sizeof(float2) * (NP * B1 + NP), e->stream));
            JF_HIP(e, hipMalloc(&e->d_rv_fdl1, sizeof(float2) * (S * R1 * B1 + S * R1)));
            JF_HIP(e, hipMalloc(&e->d_rv_ybig, sizeof(float2) * S * steps_max * B1));
            JF_HIP(e, hipMalloc(&e->d_rv_dryring, sizeof(float) * S * Rn * B1));
            JF_HIP(e, hipMalloc(&e->d_rv_fut, sizeof(float) * S * Fn * B1));
            JF_HIP(e, hipMalloc(&e->d_rv_yacc, sizeof(float2) * S * 2 * B1));
            // 1/B1: normalisation of the B1-point inverse used for the 2 B1-point real transform
            // H'_0 .. H'_P1: the response from its first tap on in partitions of B1 (ReverbBigParams)
            JF_HIP(e, launch_reverb_big_ir(d_ir, (int)n_ir, 0, P1 + 1, B1, gain / (float)B1, e->d_rv_tw1, e->d_rv_hspec1, e->stream));
        }
        JF_HIP(e, hipStreamSynchronize(e->stream));
        return JF_OK;
    };
    int rc = body();
    (void)hipFree(d_ir);
    if (rc != JF_OK) {
        const std::string msg = e->err;
        free_reverb(e);
        return fail(e, rc, msg);
    }
    e->rv_P = P;
    e->rv_Rg = Rg;
    e->rv_Wr = Wr;
    e->rv_head = 0;
    e->rv_P_total = P_total;
    e->rv_P1 = P1;
    e->rv_B1 = P1 > 0 ? B1 : 0;
    e->rv_M = P1 > 0 ? M : 0;
    e->rv_R1 = R1;
    e->rv_Rn = Rn;
    e->rv_Fn = Fn;
    e->rv_steps_max = steps_max;
    e->rv_blocks = 0;
    e->rv_fut_m = 1;  // TAIL(0) and TAIL(1) are sums over spectra of the time before the start: the zeros of the reset
    return reset_sources(e, -1);
    });
}

float jf_reverb_rms_gain(const float *signal, size_t n, const float *ir, size_t n_ir) {
    if (!signal || !ir || n == 0 || n_ir == 0) return 1.0f;
    try {
        return host_reverb_rms_gain(signal, n, ir, n_ir);
    } catch (...) {
        return 1.0f;
    }
}

}  // extern "C"
