// One outer iteration (c2r_iterate) and the evolve3D loop (evolve.F90:83-281): device-resident entries, the host-pointer
// entries with their copies, restart, the iteration hook.
#include "ctx.hpp"

using namespace c2r;

extern "C" {

// One outer iteration on a single rank: set_rates_to_zero, pass_all_sources, global_pass (evolve.F90:243-269).  With few
// sources in one batch the whole iteration is ONE replayed hipGraph and ONE host wait (FusedIter); otherwise the three
// steps in turn.  stats_host (or null): receives the photon-statistics sums of the pass (evolve.F90:570).
static int iterate_impl(Ctx *ctx, double dt, double *stats_host, double *loss, int64_t *nb, int64_t *vis, int64_t *conv,
                        double *sum1)
{
    int rc;
    ctx->step_dt = dt;
    if ((rc = sync_step(ctx))) return rc;
    const int nloc = n_local_sources(ctx);
    bool can_fuse = ctx->fused_iter && !ctx->allfrac && !ctx->queue_next && ctx->nranks == 1 && !ctx->balance && ctx->use_graph && ctx->sched_hint && ctx->prof == 0 && nloc > 0 &&
                    nloc <= kFewSources && ctx->box_hint >= 1 &&
                    !(ctx->thermal && ctx->tprm.cosmological && !ctx->have_zred);
    if (can_fuse) {
        if ((rc = ensure_sweep_scratch(ctx, nloc))) return rc;
        can_fuse = nloc <= ctx->sc[0].cap;
    }
    auto zero_rates = [ctx]() -> int {
        HIP_TRY(hipMemsetAsync(ctx->grid[4], 0, grid_bytes(ctx, 4), ctx->stream));
        if (ctx->thermal) HIP_TRY(hipMemsetAsync(ctx->grid[5], 0, grid_bytes(ctx, 5), ctx->stream));     // evolve.F90:435
        ctx->rates_clean = true; ctx->sparse_valid = false;
        return C2R_OK;
    };
    double *four = stats_host ? ctx->d_hsc->four : nullptr;
    if (!can_fuse) {
        // the three steps, with one wait behind the global pass instead of one behind each of the last two -- and where the pass
        // runs as replayed chains (64 - 768 sources: sweep.hip run_chains) the fold and the global pass are enqueued behind them
        // gated on the device, so that the whole iteration is one host wait
        FusedIter tail;
        tail.dt = dt; tail.stats = stats_host != nullptr;
        tail.post = [ctx, dt, four](const int *gate) -> int {
            const int r = sweep_finish(ctx, gate);
            return r ? r : global_pass_enqueue(ctx, dt, four, 0, ctx->ncell, gate, gate != nullptr);
        };
        const bool use_tail = !ctx->allfrac && ctx->nranks == 1 && !ctx->balance && ctx->use_graph && ctx->chain_graph && ctx->chain_tail && ctx->prof == 0 &&
                              !(ctx->thermal && ctx->tprm.cosmological && !ctx->have_zred);
        if ((rc = zero_rates())) return rc;
        if ((rc = pass_sources_impl(ctx, nullptr, nullptr, nullptr, vis, ctx->nranks == 1 && !ctx->balance, use_tail ? &tail : nullptr))) return rc;
        if (tail.tail_done) {
            if (conv) *conv = (int64_t)ctx->h_sc->conv;
            if (sum1) *sum1 = ctx->h_sc->sum;
        } else if ((rc = global_pass_impl(ctx, dt, conv, sum1, four, 0, (size_t)-1))) return rc;
        if (loss) *loss = ctx->h_sc->photon_loss;
        if (nb) *nb = ctx->h_sc->sum_nbox;
    } else {
        FusedIter fz;
        fz.dt = dt; fz.stats = stats_host != nullptr;
        fz.pre = [ctx, &fz]() -> int { return sweep_prepare(ctx, true, fz.batch_in_prepare); };
        fz.post = [ctx, dt, four](const int *gate) -> int {
            const int r = sweep_finish(ctx, gate);
            return r ? r : global_pass_enqueue(ctx, dt, four, 0, ctx->ncell, gate, gate != nullptr);
        };
        if ((rc = pass_sources_impl(ctx, &fz, loss, nb, vis))) return rc;
        if (!fz.tail_done) {               // no graph, or a source went on beyond the sub-box the graph ends at
            if ((rc = fz.post(nullptr))) return rc;
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            if (loss) *loss = ctx->h_sc->photon_loss;
            if (nb) *nb = ctx->h_sc->sum_nbox;
        }
        if (conv) *conv = (int64_t)ctx->h_sc->conv;
        if (sum1) *sum1 = ctx->h_sc->sum;
    }
    if (stats_host) for (int m = 0; m < 4; ++m) stats_host[m] = ctx->h_sc->four[m];
    return C2R_OK;
}

int c2r_iterate(c2r_ctx *c, double dt, double *photon_loss, int64_t *sum_nbox, int64_t *visited, int64_t *conv_flag,
                double *sum_xh1)
{
    if (!c) return C2R_EINVAL;
    Ctx *ctx = C(c);
    int rc = check_ready(ctx);
    if (rc) return rc;
    if (ctx->nranks > 1) FAIL(C2R_ESTATE, "c2r_iterate is the single-rank iteration: with several ranks call c2r_zero_rates, "
                                          "c2r_pass_sources, the collective and c2r_global_pass in turn (or c2r_evolve3d)");
    return iterate_impl(ctx, dt, nullptr, photon_loss, sum_nbox, visited, conv_flag, sum_xh1);
}

// tail (or null): enqueued on the context's stream once the step's last kernel has been -- the host-pointer entries put their
// device-to-host copies there, so that the step ends with ONE host wait behind results and copies alike
static int evolve3d_worker(c2r_ctx *c, double dt, int restart_niter, double restart_loss, c2r_report *rep,
                           const std::function<int()> &tail = nullptr)
{
    if (!c) return C2R_EINVAL;
    Ctx *ctx = C(c);
    int rc = check_ready(ctx);
    if (rc) return rc;
    const c2r_params &p = ctx->prm;
    c2r_report local;
    if (!rep) rep = &local;
    memset(rep, 0, sizeof *rep);
    using clk = std::chrono::steady_clock;
    int niter = 0;
    int64_t conv_flag = (int64_t)ctx->ncell;                                           // :149
    double prev1 = (double)(((2.0f * (float)p.mesh[0]) * (float)p.mesh[1]) * (float)p.mesh[2]);   // :150-151
    double prev0 = prev1;
    if (restart_niter < 0) {
        // evolve.F90:145-146  xh_av = xh ; xh_intermed = xh
        HIP_TRY(hipMemcpyAsync(ctx->grid[2], ctx->grid[1], grid_bytes(ctx, 1), hipMemcpyDeviceToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(ctx->grid[3], ctx->grid[1], grid_bytes(ctx, 1), hipMemcpyDeviceToDevice, ctx->stream));
        if (ctx->allfrac) {      // :142-143 (ALLFRAC): all of (:,:,:,:)
            HIP_TRY(hipMemcpyAsync(ctx->grid[8], ctx->grid[7], grid_bytes(ctx, 7), hipMemcpyDeviceToDevice, ctx->stream));
            HIP_TRY(hipMemcpyAsync(ctx->grid[9], ctx->grid[7], grid_bytes(ctx, 7), hipMemcpyDeviceToDevice, ctx->stream));
        }
    }
    const int64_t c1 = (int64_t)(p.convergence_fraction * p.mesh[0] * p.mesh[1] * p.mesh[2]);    // :162
    const int64_t c2 = (ctx->nsrc - 1) / 3;
    const int64_t conv_criterion = std::min(c1, c2);
    rep->conv_criterion = conv_criterion;
    rep->timing_split = ctx->nranks > 1 ? 1 : 0;
    double totalsrc = 0.0;
    // :136 state_before(xh): the four sums land in pinned memory; they are read when the step has ended (no host wait here)
    if ((rc = photon_sums_launch(ctx, 1, 1, ctx->d_hsc->before))) return rc;
    for (int i = 0; i < ctx->nsrc; ++i) totalsrc += ctx->nflux[i];                      // photonstatistics.F90:266
    totalsrc = totalsrc * p.S_star * dt;
    double sum1 = 0.0;
    if (restart_niter >= 0) {
        // evolve.F90:153-157: start_from_dump loaded niter, photon_loss_all, phih_grid, xh_av and
        // xh_intermed (the caller put them in the device arrays); one global pass; the saved
        // previous-sum variables (evolve.F90:67-74) are zero in a freshly started process
        niter = restart_niter;
        prev1 = prev0 = 0.0;
        rep->photon_loss_all = restart_loss;
        rc = c2r_global_pass(c, dt, &conv_flag, &sum1);
        // logged in the slot of the iteration whose global pass this repeats
        if (niter >= 1 && niter <= C2R_MAX_ITER_LOG) rep->it_conv_flag[niter - 1] = conv_flag;
    } else {
        rc = c2r_sum(c, 3, &sum1);                                                     // :183
    }
    if (rc) return rc;
    for (;;) {
        double sum0 = (double)(float)ctx->ncell - sum1;                                // :184
        if (ctx->allfrac && (rc = c2r_sum(c, 9, &sum0))) return rc;                    // :180-181 (ALLFRAC): sum(xh_intermed(:,:,:,0))
        const double rel1 = sum1 > 0.0 ? fabs(sum1 - prev1) / sum1 : 1.0;
        const double rel0 = sum0 > 0.0 ? fabs(sum0 - prev0) / sum0 : 1.0;
        if (niter > 0 && niter <= C2R_MAX_ITER_LOG) {
            rep->it_rel_change_xh1[niter - 1] = rel1; rep->it_rel_change_xh0[niter - 1] = rel0;
            rep->it_sum_xh1[niter - 1] = sum1;
        }
        if (conv_flag < conv_criterion || (rel1 < p.convergence_fraction && rel0 < p.convergence_fraction)) {   // :212
            HIP_TRY(hipMemcpyAsync(ctx->grid[1], ctx->grid[3], grid_bytes(ctx, 1), hipMemcpyDeviceToDevice, ctx->stream));   // :218
            if (ctx->allfrac) HIP_TRY(hipMemcpyAsync(ctx->grid[7], ctx->grid[9], grid_bytes(ctx, 7), hipMemcpyDeviceToDevice, ctx->stream));   // :216
            if (ctx->thermal && (rc = final_temperature_enqueue(ctx))) return rc;       // :220 set_final_temperature_point
            rep->converged = 1;
            break;
        } else if (niter > p.max_outer_iter) {                                         // :228
            rep->converged = 0;
            break;
        }
        prev1 = sum1; prev0 = sum0;
        niter++;
        double loss = 0; int64_t nb = 0, vis = 0;
        if (ctx->nranks == 1) {
            // :243-269 in one piece (iterate_impl): nothing happens between the pass and the global pass on one rank
            auto t0 = clk::now();
            rc = iterate_impl(ctx, dt, niter <= C2R_MAX_ITER_LOG ? ctx->h_it4 + 4 * (size_t)(niter - 1) : nullptr, &loss, &nb, &vis,
                              &conv_flag, &sum1);
            if (rc) return rc;
            rep->seconds_sweep += std::chrono::duration<double>(clk::now() - t0).count();     // (sweep and chemistry: one wait)
            rep->photon_loss_all = loss; rep->sum_nbox_all = nb; rep->visited += vis;
            rep->chem_not_converged = (int32_t)ctx->h_sc->chemfail;
            if (niter <= C2R_MAX_ITER_LOG) { rep->it_conv_flag[niter - 1] = conv_flag; rep->it_sum_nbox[niter - 1] = nb; }
            if (ctx->iter_hook && ctx->iter_hook(ctx->iter_user, niter, rep->photon_loss_all) != 0) FAIL(C2R_ECALLBACK, "iteration hook failed");
            continue;
        }
        rc = c2r_zero_rates(c);                                                        // :243
        if (rc) return rc;
        auto t0 = clk::now();
        rc = c2r_pass_sources(c, &loss, &nb, &vis);                                    // :246
        if (rc) return rc;
        const bool slab = ctx->nranks > 1 && ctx->rs && ctx->ag && ctx->ar;
        size_t so[kMaxSlabRanks], sc[kMaxSlabRanks];                                                         // slabs of all ranks (cells)
        if (slab) {
            // (nranks <= kMaxSlabRanks: c2r_set_slab_chemistry / c2r_set_rank refuse anything else)
            for (int r = 0; r < ctx->nranks; ++r) slab_of(ctx, r, ctx->nranks, &so[r], &sc[r]);
            // reduce-scatter instead of evolve.F90:599's all-reduce: this rank gets the summed rates of its z-slab
            if (ctx->rs(ctx->slab_user, ctx->grid[4], so, sc, ctx->nranks, (void *)ctx->stream) != 0) FAIL(C2R_ECALLBACK, "reduce-scatter callback failed");
            if (ctx->thermal && ctx->rs(ctx->slab_user, ctx->grid[5], so, sc, ctx->nranks, (void *)ctx->stream) != 0)
                FAIL(C2R_ECALLBACK, "reduce-scatter callback failed");
        }
        if (ctx->nranks > 1) {
            if (!slab) rc = c2r_allreduce_rates(c);                                    // evolve.F90:599
            if (rc) return rc;
            // evolve.F90:587,612: photon_loss and sum_nbox ride along as a 2-element f64 vector
            // (sum_nbox is exact in f64)
            ctx->h_sc->pair[0] = loss; ctx->h_sc->pair[1] = (double)nb;
            HIP_TRY(hipMemcpyAsync(ctx->d_pair, ctx->h_sc->pair, 2 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            if (ctx->ar(ctx->ar_user, ctx->d_pair, 2, (void *)ctx->stream) != 0) FAIL(C2R_ECALLBACK, "all-reduce callback failed");
            HIP_TRY(hipMemcpyAsync(ctx->h_sc->pair, ctx->d_pair, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            loss = ctx->h_sc->pair[0]; nb = (int64_t)llround(ctx->h_sc->pair[1]);
        }
        auto t1 = clk::now();
        rep->photon_loss_all = loss; rep->sum_nbox_all = nb; rep->visited += vis;
        // :269 global_pass; evolve.F90:570 calculate_photon_statistics(dt,xh_intermed,xh_av) + report (the conservation
        // line): the sums come out of the same kernel into this iteration's pinned slot
        if (!slab) {
            rc = global_pass_impl(ctx, dt, &conv_flag, &sum1, niter <= C2R_MAX_ITER_LOG ? ctx->d_hit4 + 4 * (size_t)(niter - 1) : nullptr, 0, (size_t)-1);
            if (rc) return rc;
        } else {
            // evolve0D_global on the own slab only (evolve.F90:548-555 visits every cell on every rank), the counts summed
            // over the ranks, the pass's outputs gathered: xh_av (the next sweep reads all of it), xh_intermed (Test 2 and the
            // accepted state), the temperatures.  The sums that feed Test 2 and the photon statistics are then taken over
            // the whole arrays exactly as the replicated pass takes them: bit-identical decisions on every rank.
            const size_t mo = so[ctx->rank], mc = sc[ctx->rank];
            int64_t conv_local = 0;
            rc = global_pass_impl(ctx, dt, &conv_local, nullptr, nullptr, mo, mc);
            if (rc) return rc;
            ctx->h_sc->pair[0] = (double)conv_local; ctx->h_sc->pair[1] = (double)ctx->h_sc->chemfail;
            HIP_TRY(hipMemcpyAsync(ctx->d_pair, ctx->h_sc->pair, 2 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            if (ctx->ar(ctx->ar_user, ctx->d_pair, 2, (void *)ctx->stream) != 0) FAIL(C2R_ECALLBACK, "all-reduce callback failed");
            HIP_TRY(hipMemcpyAsync(ctx->h_sc->pair, ctx->d_pair, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            size_t bo[kMaxSlabRanks], bc[kMaxSlabRanks];
            for (int w = 2; w <= 3; ++w) {
                for (int r = 0; r < ctx->nranks; ++r) { bo[r] = so[r] * sizeof(double); bc[r] = sc[r] * sizeof(double); }
                if (ctx->ag(ctx->slab_user, ctx->grid[w], bo, bc, ctx->nranks, (void *)ctx->stream) != 0) FAIL(C2R_ECALLBACK, "all-gather callback failed");
                // (-DALLFRAC drivers: the pass also wrote the stored neutral halves of its slab, arrays 8 and 9)
                if (ctx->allfrac && ctx->ag(ctx->slab_user, ctx->grid[w + 6], bo, bc, ctx->nranks, (void *)ctx->stream) != 0) FAIL(C2R_ECALLBACK, "all-gather callback failed");
            }
            if (ctx->thermal) {
                for (int r = 0; r < ctx->nranks; ++r) { bo[r] = so[r] * 3 * sizeof(float); bc[r] = sc[r] * 3 * sizeof(float); }
                if (ctx->ag(ctx->slab_user, ctx->grid[6], bo, bc, ctx->nranks, (void *)ctx->stream) != 0) FAIL(C2R_ECALLBACK, "all-gather callback failed");
            }
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            conv_flag = (int64_t)llround(ctx->h_sc->pair[0]); ctx->h_sc->chemfail = (unsigned int)llround(ctx->h_sc->pair[1]);
            if ((rc = c2r_sum(c, 3, &sum1))) return rc;
            if (niter <= C2R_MAX_ITER_LOG && (rc = photon_sums_launch(ctx, 3, 2, ctx->d_hit4 + 4 * (size_t)(niter - 1)))) return rc;
        }
        auto t2 = clk::now();
        rep->seconds_sweep += std::chrono::duration<double>(t1 - t0).count();
        rep->seconds_chem += std::chrono::duration<double>(t2 - t1).count();
        rep->chem_not_converged = (int32_t)ctx->h_sc->chemfail;
        if (niter <= C2R_MAX_ITER_LOG) { rep->it_conv_flag[niter - 1] = conv_flag; rep->it_sum_nbox[niter - 1] = nb; }
        // evolve.F90:271-275: the place where the reference decides on an iteration dump
        if (ctx->iter_hook) {
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            if (ctx->iter_hook(ctx->iter_user, niter, rep->photon_loss_all) != 0) FAIL(C2R_ECALLBACK, "iteration hook failed");
        }
    }
    if (ctx->nranks > 1 && ctx->rs && ctx->ag && ctx->ar && niter > (restart_niter > 0 ? restart_niter : 0)) {
        // the step leaves phih_grid (phiheat_grid) complete on every rank, as the all-reduce does (output.F90 writes them)
        size_t bo[kMaxSlabRanks], bc[kMaxSlabRanks];
        for (int r = 0; r < ctx->nranks; ++r) { size_t o, n; slab_of(ctx, r, ctx->nranks, &o, &n); bo[r] = o * sizeof(double); bc[r] = n * sizeof(double); }
        if (ctx->ag(ctx->slab_user, ctx->grid[4], bo, bc, ctx->nranks, (void *)ctx->stream) != 0) FAIL(C2R_ECALLBACK, "all-gather callback failed");
        if (ctx->thermal && ctx->ag(ctx->slab_user, ctx->grid[5], bo, bc, ctx->nranks, (void *)ctx->stream) != 0) FAIL(C2R_ECALLBACK, "all-gather callback failed");
    }
    // evolve.F90:277-279 calculate_photon_statistics(dt,xh,xh_av): enqueued, then whatever the caller wants behind the step
    // (the host-pointer entries: their downloads), then the step's one final wait
    if ((rc = photon_sums_launch(ctx, 1, 2, ctx->d_hsc->after))) return rc;
    if (tail && (rc = tail())) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const double *before = ctx->h_sc->before, *after = ctx->h_sc->after;
    rep->niter = niter; rep->conv_flag = conv_flag;
    for (int k = (restart_niter > 0 ? restart_niter : 0); k < niter && k < C2R_MAX_ITER_LOG; ++k) {
        const double *a4 = ctx->h_it4 + 4 * (size_t)k;
        const double trec = a4[2] * ctx->vol * dt, tcol = a4[3] * ctx->vol * dt;
        const double tion = trec + (before[0] * ctx->vol - a4[0] * ctx->vol);
        rep->it_photcons[k] = totalsrc > 0.0 ? (tion - tcol) / totalsrc : 0.0;
    }
    rep->h0_before = before[0] * ctx->vol; rep->h1_before = before[1] * ctx->vol;
    rep->h0_after = after[0] * ctx->vol;   rep->h1_after = after[1] * ctx->vol;
    rep->totrec = after[2] * ctx->vol * dt; rep->totcollisions = after[3] * ctx->vol * dt;
    rep->dh0 = rep->h0_before - rep->h0_after;                                        // photonstatistics.F90:225
    rep->total_ion = rep->totrec + rep->dh0;
    rep->totalsrc = totalsrc;
    rep->photcons = totalsrc > 0.0 ? (rep->total_ion - rep->totcollisions) / totalsrc : 0.0;   // :268 (LLS_loss = 0)
    return C2R_OK;
}

int c2r_evolve3d_dev(c2r_ctx *c, double dt, c2r_report *rep)
{
    return evolve3d_worker(c, dt, -1, 0.0, rep);
}

int c2r_evolve3d_restart_dev(c2r_ctx *c, double dt, int32_t niter, double photon_loss_all, c2r_report *rep)
{
    if (niter < 0) return C2R_EINVAL;
    return evolve3d_worker(c, dt, niter, photon_loss_all, rep);
}

int c2r_set_iteration_hook(c2r_ctx *c, c2r_iteration_fn fn, void *user)
{
    if (!c) return C2R_EINVAL;
    C(c)->iter_hook = fn; C(c)->iter_user = user;
    return C2R_OK;
}

static void pin_host_array(Ctx *ctx, const void *ptr, size_t bytes)
{
    if (!ptr) return;
    auto it = ctx->pinned.find(ptr);
    if (it != ctx->pinned.end() && it->second >= bytes) return;
    if (it != ctx->pinned.end()) { hipHostUnregister(const_cast<void *>(ptr)); ctx->pinned.erase(it); }
    // best effort: an array that cannot be registered is simply copied as pageable memory
    if (hipHostRegister(const_cast<void *>(ptr), bytes, hipHostRegisterDefault) == hipSuccess) ctx->pinned[ptr] = bytes;
    else (void)hipGetLastError();
}

// The host-pointer entries: every array the caller hands over is page-locked once (the driver allocates them once per run,
// evolve_data.F90:75-90), the uploads are enqueued without a host wait in front of the step, the downloads behind its last
// kernel (evolve3d_worker's tail), and the call waits ONCE, at the end.  The two groups of copies are timed with HIP events.
namespace {
struct HostCopies {
    Ctx *ctx; hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit HostCopies(Ctx *c) : ctx(c) { for (auto &e : ev) hipEventCreate(&e); }
    ~HostCopies() { for (auto &e : ev) if (e) hipEventDestroy(e); }
    int up(int which, const void *host)
    {
        // (-DALLFRAC drivers: xh / xh_av / xh_intermed are (mesh,0:1) arrays -- copy_in / copy_out move both halves)
        pin_host_array(ctx, host, grid_bytes(ctx, which) * ((ctx->allfrac && which >= 1 && which <= 3) ? 2 : 1));
        const int rc = copy_in(ctx, which, host);
        if (rc) return rc;
        if (which == 4 || which == 5) { ctx->rates_clean = false; ctx->sparse_valid = false; }
        return C2R_OK;
    }
    int down(int which, void *host)
    {
        if (!host) return C2R_OK;
        pin_host_array(ctx, host, grid_bytes(ctx, which) * ((ctx->allfrac && which >= 1 && which <= 3) ? 2 : 1));
        return copy_out(ctx, which, host);
    }
    int mark(int i) { HIP_TRY(hipEventRecord(ev[i], ctx->stream)); return C2R_OK; }
    void finish(c2r_report *rep)       // after the step's final wait
    {
        if (!rep) return;
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, ev[0], ev[1]) == hipSuccess) rep->seconds_upload = 1e-3 * ms;
        if (hipEventElapsedTime(&ms, ev[2], ev[3]) == hipSuccess) rep->seconds_download = 1e-3 * ms;
        (void)hipGetLastError();
        rep->seconds_total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
};
}  // namespace

int c2r_evolve3d(c2r_ctx *c, double dt, const float *ndens, double *xh, double *xh_av, double *xh_int,
                 double *phih, c2r_report *rep)
{
    if (!c || !ndens || !xh) return C2R_EINVAL;
    Ctx *ctx = C(c);
    int rc;
    HIP_TRY(hipSetDevice(ctx->prm.device));
    c2r_report local;
    if (!rep) rep = &local;
    HostCopies hc(ctx);
    if ((rc = hc.mark(0)) || (rc = hc.up(0, ndens)) || (rc = hc.up(1, xh)) || (rc = hc.mark(1))) return rc;
    auto tail = [&]() -> int {
        int r;
        if ((r = hc.mark(2)) || (r = hc.down(1, xh)) || (r = hc.down(2, xh_av)) || (r = hc.down(3, xh_int)) || (r = hc.down(4, phih))) return r;
        return hc.mark(3);
    };
    if ((rc = evolve3d_worker(c, dt, -1, 0.0, rep, tail))) return rc;
    hc.finish(rep);
    return C2R_OK;
}

int c2r_evolve3d_restart(c2r_ctx *c, double dt, int32_t niter, double photon_loss_all, const float *ndens,
                         double *xh, double *xh_av, double *xh_int, double *phih, c2r_report *rep)
{
    if (!c || !ndens || !xh || !xh_av || !xh_int || !phih || niter < 0) return C2R_EINVAL;
    Ctx *ctx = C(c);
    int rc;
    HIP_TRY(hipSetDevice(ctx->prm.device));
    c2r_report local;
    if (!rep) rep = &local;
    HostCopies hc(ctx);
    if ((rc = hc.mark(0)) || (rc = hc.up(0, ndens)) || (rc = hc.up(1, xh)) || (rc = hc.up(2, xh_av)) || (rc = hc.up(3, xh_int)) ||
        (rc = hc.up(4, phih)) || (rc = hc.mark(1))) return rc;
    auto tail = [&]() -> int {
        int r;
        if ((r = hc.mark(2)) || (r = hc.down(1, xh)) || (r = hc.down(2, xh_av)) || (r = hc.down(3, xh_int)) || (r = hc.down(4, phih))) return r;
        return hc.mark(3);
    };
    if ((rc = evolve3d_worker(c, dt, niter, photon_loss_all, rep, tail))) return rc;
    hc.finish(rep);
    return C2R_OK;
}

int c2r_evolve3d_thermal(c2r_ctx *c, double dt, int32_t restart_niter, double photon_loss_all, const float *ndens,
                         double *xh, double *xh_av, double *xh_int, double *phih, double *phiheat, float *temperature_grid,
                         c2r_report *rep)
{
    if (!c || !ndens || !xh || !temperature_grid) return C2R_EINVAL;
    Ctx *ctx = C(c);
    if (!ctx->thermal) FAIL(C2R_ESTATE, "c2r_set_thermal has not been called");
    const bool restart = restart_niter >= 0;
    if (restart && (!xh_av || !xh_int || !phih || !phiheat)) return C2R_EINVAL;
    int rc;
    HIP_TRY(hipSetDevice(ctx->prm.device));
    c2r_report local;
    if (!rep) rep = &local;
    HostCopies hc(ctx);
    if ((rc = hc.mark(0)) || (rc = hc.up(0, ndens)) || (rc = hc.up(1, xh)) || (rc = hc.up(6, temperature_grid))) return rc;
    if (restart) {       // start_from_dump (evolve.F90:328-426) read these, phiheat_grid and temperature_grid (:372-375)
        if ((rc = hc.up(2, xh_av)) || (rc = hc.up(3, xh_int)) || (rc = hc.up(4, phih)) || (rc = hc.up(5, phiheat))) return rc;
    }
    if ((rc = hc.mark(1))) return rc;
    auto tail = [&]() -> int {
        int r;
        if ((r = hc.mark(2)) || (r = hc.down(1, xh)) || (r = hc.down(2, xh_av)) || (r = hc.down(3, xh_int)) || (r = hc.down(4, phih)) ||
            (r = hc.down(5, phiheat)) || (r = hc.down(6, temperature_grid))) return r;
        return hc.mark(3);
    };
    if ((rc = evolve3d_worker(c, dt, restart ? restart_niter : -1, photon_loss_all, rep, tail))) return rc;
    hc.finish(rep);
    return C2R_OK;
}

}  // extern "C"
