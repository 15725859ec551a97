// The source sweep on the host side: sweep scratch, the step block, the per-shell launch schedule of a batch of sources
// (BatchSweep), the pass over all of a rank's sources, one source, one cell.  Kernels: kernels_sweep.hpp.
#include "ctx.hpp"
#include "kernels_sweep.hpp"
#include <climits>

namespace c2r {

namespace {
void free_chain(SweepScratch &sc)
{
    hipFree(sc.d_planes); hipFree(sc.d_gbox); hipFree(sc.d_gbox_h); hipFree(sc.d_batch); hipFree(sc.d_batch_init); hipFree(sc.d_loss_partial);
    hipFree(sc.d_perm);
    if (sc.h_perm) hipHostFree(sc.h_perm);
    if (sc.h_batch) hipHostFree(sc.h_batch);
    if (sc.h_nactive) hipHostFree(sc.h_nactive);
    for (auto &e : sc.ev_box) hipEventDestroy(e);
    if (sc.ev_done) hipEventDestroy(sc.ev_done);
    const hipStream_t st = sc.stream; const bool own = sc.own_stream;     // the stream outlives the arrays
    sc = SweepScratch();
    sc.stream = st; sc.own_stream = own;
}
}  // namespace

void free_sweep_scratch(Ctx *ctx)
{
    for (int c = 0; c < kMaxChains; ++c) free_chain(ctx->sc[c]);
    ctx->batch_cap = 0; ctx->batch_want = 0; ctx->chain_cap = 0; ctx->nchains = 1;
}

int n_local_sources(const Ctx *ctx)
{
    if (ctx->explicit_share) return (int)ctx->share.size();
    return ctx->nsrc > ctx->rank ? (ctx->nsrc - ctx->rank + ctx->nranks - 1) / ctx->nranks : 0;
}

namespace {
// How many chains a pass over `want` sources is split into (SweepScratch).  One, unless the pass is long enough to be
// worth overlapping and short enough to need it: with 64 - 768 sources a shell launch takes 10 - 400 us, of which the
// drain of its last workgroups and the gap to the next launch are a tenth (one GPU's share of the 8-GPU bench, 125 sources:
// +13 % per source against 1000 at once); two or three interleaved chains hide both.  With more sources the launches are long
// and one chain is best (and per-launch timing stays meaningful); with fewer, the few-source schedule (one hipGraph) applies.
// Deterministic rates keep one chain: k_gamma_reduce sums a batch's per-source grids in source order.
int choose_chains(const Ctx *ctx, int want)
{
    if (ctx->chains_env > 0) return std::min(ctx->chains_env, std::max(1, want));
    if (ctx->prm.deterministic_rates || want < 64 || want > 768) return 1;
    // meshes that outgrow the L2s (n_HI >= 64 MB): from 1.5 sources per mesh plane ONE chain whose far shells run under the
    // plane-ordered block mapping (stage_perm) is at least as fast as two chains of half the sources (256^3, 400 / 500 / 640 / 768
    // sources: 57.3 / 71.6 / 89.3 / 106.8 ms against 58.1 / 72.2 / 92.0 / 112.9; at 128^3 the chains win: profiles/r05_xcd/ab_mid*.txt)
    if (ctx->stream_hint && ctx->xcd_order != 0 &&
        (double)want >= ctx->xcd_min_per_plane * (double)std::min(ctx->prm.mesh[0], std::min(ctx->prm.mesh[1], ctx->prm.mesh[2]))) return 1;
    return (want >= 192 && want <= 384) ? 3 : 2;
}

int alloc_chain(Ctx *ctx, SweepScratch &sc, int cap)
{
    if (!sc.stream) { HIP_TRY(hipStreamCreateWithFlags(&sc.stream, hipStreamNonBlocking)); sc.own_stream = true; }
    HIP_TRY(hipMalloc(&sc.d_planes, (size_t)cap * 2 * 6 * ctx->PP * sizeof(double)));
    if (ctx->prm.deterministic_rates) HIP_TRY(hipMalloc(&sc.d_gbox, (size_t)cap * 2 * ctx->ncell * sizeof(double)));
    if (ctx->prm.deterministic_rates && ctx->thermal) HIP_TRY(hipMalloc(&sc.d_gbox_h, (size_t)cap * 2 * ctx->ncell * sizeof(double)));
    HIP_TRY(hipMalloc(&sc.d_loss_partial, (size_t)cap * 6 * ctx->tiles_cap * sizeof(double)));
    // small per-batch arrays: doubles first, then ints
    //   nflux[cap] final_loss[cap] loss_acc[cap] nflux_xray[cap] | srcpos[3cap] srcw[3cap] active0[cap] active1[cap] final_nbox[cap] nactive[2]
    sc.batch_bytes = (size_t)cap * 4 * sizeof(double) + ((size_t)cap * 9 + 2) * sizeof(int);
    HIP_TRY(hipMalloc(&sc.d_batch, sc.batch_bytes));
    HIP_TRY(hipMalloc(&sc.d_batch_init, sc.batch_bytes));
    sc.batch_image.clear();
    HIP_TRY(hipHostMalloc((void **)&sc.h_batch, sc.batch_bytes, hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer((void **)&sc.d_hbatch, sc.h_batch, 0));
    {
        double *d = reinterpret_cast<double *>(sc.d_batch);
        sc.d_nflux_b = d; sc.d_final_loss = d + cap; sc.d_loss_acc = d + 2 * (size_t)cap; sc.d_nflux_x = d + 3 * (size_t)cap;
        int *i = reinterpret_cast<int *>(d + 4 * (size_t)cap);
        sc.d_srcpos_b = i; sc.d_srcw_b = i + 3 * (size_t)cap; sc.d_active[0] = i + 6 * (size_t)cap;
        sc.d_active[1] = i + 7 * (size_t)cap; sc.d_final_nbox = i + 8 * (size_t)cap; sc.d_nactive = i + 9 * (size_t)cap;
    }
    HIP_TRY(hipMalloc(&sc.d_perm, (size_t)3 * cap * sizeof(int)));
    HIP_TRY(hipHostMalloc((void **)&sc.h_perm, (size_t)3 * cap * sizeof(int)));
    HIP_TRY(hipHostMalloc((void **)&sc.h_nactive, (size_t)(ctx->nbox_max + 2) * sizeof(int), hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer((void **)&sc.d_hnactive, sc.h_nactive, 0));
    sc.ev_box.resize(ctx->nbox_max + 2);
    for (auto &e : sc.ev_box) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&sc.ev_done, hipEventDisableTiming));
    sc.cap = cap;
    return C2R_OK;
}
}  // namespace

// Per-source scratch: two shells x six face planes of (2R+1)^2 f64.  Size the batch so that it
// fits the budget; 288 GB of HBM normally holds every source of a rank at once.
int ensure_sweep_scratch(Ctx *ctx, int want)
{
    // batch_want: the request the current allocation was sized for (it may have been capped by the budget)
    if (want <= ctx->batch_cap || want <= ctx->batch_want) return C2R_OK;
    free_sweep_scratch(ctx);
    const size_t per_src = 2 * 6 * ctx->PP * sizeof(double) + (size_t)6 * ctx->tiles_cap * sizeof(double) + 64 +
                           (ctx->prm.deterministic_rates ? (ctx->thermal ? 4 : 2) * ctx->ncell * sizeof(double) : 0);
    size_t budget = ctx->prm.scratch_bytes;
    if (budget == 0) {
        size_t fr = 0, tot = 0;
        HIP_TRY(hipMemGetInfo(&fr, &tot));
        budget = fr / 4;
    }
    int cap = (int)std::min<size_t>((size_t)want, std::max<size_t>(1, budget / per_src));
    if (ctx->batch_cap_opt > 0) cap = std::max(1, std::min(cap, ctx->batch_cap_opt));          // (option batch_cap: tests of several rounds)
    cap = std::min(cap, 65535);                 // grid.z of k_sweep_shell
    // the sources of a round, split over the chains (each chain's arrays hold its share)
    const int nch = choose_chains(ctx, cap);
    const int per = (cap + nch - 1) / nch;
    for (int c = 0; c < nch; ++c) { const int rc = alloc_chain(ctx, ctx->sc[c], per); if (rc) return rc; }
    ctx->nchains = nch; ctx->chain_cap = per;
    ctx->batch_cap = cap;
    ctx->batch_want = want;
    ++ctx->gen;                                   // every captured launch points into the old scratch
    return C2R_OK;
}

namespace {
// udiv() precondition: the divisor's significand must not be all ones (Markstein's exception)
bool udiv_ok(double d)
{
    uint64_t u; memcpy(&u, &d, sizeof u);
    return std::isnormal(d) && (u & 0xFFFFFFFFFFFFFULL) != 0xFFFFFFFFFFFFFULL;
}

// The cells face f owns in shell q, clipped to the trace limits (see FaceRect)
FaceRect face_rect(const Ctx *ctx, int f, int q, int rows = kRows)
{
    FaceRect r{};
    const int axis = 2 - (f >> 1), pd = (f & 1) ? -q : q;
    if (pd < -ctx->hl[axis] || pd > ctx->hr[axis]) return r;
    const int ua = (axis == 0) ? 1 : 0, va = (axis == 2) ? 1 : 2;
    const int qa = (axis == 0) ? q - 1 : q;          // x faces own |a| < q
    const int qb = (axis == 2) ? q : q - 1;          // y and x faces own |b| < q
    const int a_lo = std::max(-qa, -ctx->hl[ua]), a_hi = std::min(qa, ctx->hr[ua]);
    const int b_lo = std::max(-qb, -ctx->hl[va]), b_hi = std::min(qb, ctx->hr[va]);
    if (a_hi < a_lo || b_hi < b_lo) return r;
    r.a_lo = a_lo; r.wa = a_hi - a_lo + 1; r.b_lo = b_lo; r.wb = b_hi - b_lo + 1;
    r.magic = r.wa > 1 ? (unsigned)((1ULL << 32) / (unsigned)r.wa + 1ULL) : 0u;
    // k_sweep_shell gives a thread two rows of the same sign: (0,1),(2,3),... and (-1,-2),(-3,-4),...
    r.pp = (b_hi + rows) / rows;                // groups of the rows 0..b_hi
    r.npr = r.pp + (-b_lo + rows - 1) / rows;   // + groups of the rows -1..b_lo
    r.ntiles = (int)(((long long)r.wa * r.npr + kBlock - 1) / kBlock);
    return r;
}

KParams make_kparams(const Ctx *ctx, const SweepScratch &sc)
{
    KParams k{};
    const c2r_params &p = ctx->prm;
    for (int d = 0; d < 3; ++d) { k.n[d] = p.mesh[d]; k.hl[d] = ctx->hl[d]; k.hr[d] = ctx->hr[d]; }
    // dr, vol, coldensh_LLS, inv_dr0, dr2 stay zero here: the kernels read them from the device-resident step block
    // (load_step), so that captured launches do not depend on the time step
    k.step = reinterpret_cast<const StepBlock *>(ctx->d_step);
    k.shell_step = reinterpret_cast<const ShellStep *>(ctx->d_step + sizeof(StepBlock));
    k.sigma = p.sigma_HI; k.wfloor = p.weight_floor; k.sqrt2 = p.sqrt2; k.sqrt3 = p.sqrt3;
    k.fourpi = 4.0 * p.pi;                      // evolve_point.F90:177: 4.0*pi*dist2*path, left to right
    k.max_coldensh = p.max_coldensh; k.tau_limit = p.tau_photo_limit;
    k.minlogtau = p.minlogtau; k.dlogtau = p.dlogtau; k.numtau = p.numtau; k.numtau_d = (double)p.numtau;
    k.eps = p.epsilon;
    k.inv_dlogtau = 1.0 / p.dlogtau;
    k.exact_udiv = udiv_ok(p.dlogtau);               // (load_step adds udiv_ok(dr[0]))
    k.R = ctx->R; k.P = ctx->P; k.PP = ctx->PP;
    k.nhi = ctx->d_nhi; k.nhi_T = ctx->d_nhi_T;
    // the accumulators of the rates: phih_grid and its transposed companion -- or, for the second half of an overlapped
    // pass (pass_sources_impl), the second pair
    k.phih = ctx->acc_phih ? ctx->acc_phih : (double *)ctx->grid[4]; k.phih_T = ctx->acc_phih_T ? ctx->acc_phih_T : ctx->d_phih_T;
    k.gbox = sc.d_gbox; k.gbox_h = ctx->thermal ? sc.d_gbox_h : nullptr;
    k.lls_type = ctx->lls_type; k.R_max2 = ctx->R_max_LLS * ctx->R_max_LLS; k.lls = ctx->d_lls; k.lls_T = ctx->d_lls_T;
    k.thick = ctx->d_thick; k.thin = ctx->d_thin; k.logtab = ctx->d_logtab;
    k.hthick = ctx->d_hthick; k.hthin = ctx->d_hthin; k.heat = (double *)ctx->grid[5]; k.heat_T = ctx->d_heat_T;
    k.tau_heat_limit = ctx->tprm.tau_heat_limit;
    k.odtab = ctx->d_odtab;
    k.od_per_e = (double)(0.301029995663981195213738894724493027L / (long double)p.dlogtau);
    k.od_per_ln = (double)(0.434294481903251827651128918916605082L / (long double)p.dlogtau);
    k.srcpos = sc.d_srcpos_b; k.srcw = sc.d_srcw_b; k.normflux = sc.d_nflux_b; k.planes = sc.d_planes;
    k.xthick = ctx->d_xthick; k.xthin = ctx->d_xthin; k.xhthick = ctx->d_xhthick; k.xhthin = ctx->d_xhthin; k.normflux_x = sc.d_nflux_x;
    return k;
}

}  // namespace

// The step block as the device should hold it now: sent only when it differs from what was sent last (once per time step, or
// when a setter changed something).  Ordered on the context's stream before whatever is enqueued next; never inside a capture.
int sync_step(Ctx *ctx)
{
    const c2r_params &p = ctx->prm;
    const int nsh = ctx->Qmax + 1;
    std::vector<char> img(sizeof(StepBlock) + (size_t)nsh * sizeof(ShellStep), 0);
    StepBlock *st = reinterpret_cast<StepBlock *>(img.data());
    ShellStep *sh = reinterpret_cast<ShellStep *>(img.data() + sizeof(StepBlock));
    for (int d = 0; d < 3; ++d) { st->dr[d] = ctx->dr[d]; st->dr2[d] = ctx->dr[d] * ctx->dr[d]; }
    st->vol = ctx->vol; st->coldensh_LLS = ctx->lls; st->inv_dr0 = 1.0 / ctx->dr[0];
    st->exact_udiv_dr0 = udiv_ok(ctx->dr[0]) ? 1 : 0; st->n_shell = nsh;
    for (int q = 1; q < nsh; ++q) {
        for (int d = 0; d < 3; ++d) { const double t = ctx->dr[d] * (double)q; sh[q].d2axis[d] = t * t; }   // sign drops out
        sh[q].path_scale = ctx->dr[0] / (double)q;
        sh[q].lls_scale = ctx->lls_type == 2 ? 1.0 / (double)q : ctx->lls / (double)q;
    }
    // doric.f90:73,78 -- temperature is uniform (isothermal), so both rate coefficients are per-step constants; evaluated
    // with the host libm like the reference does at run time
    ChemStep &c = st->chem;
    c.dt = ctx->step_dt;
    c.recpow = pow(ctx->temper / 1e4, p.albpow);
    c.brech0 = (double)ctx->clumping * p.bh00 * c.recpow;
    c.acolh0 = p.colh0 * sqrt(ctx->temper) * exp(-p.temph0 / ctx->temper);
    c.clumping = (double)ctx->clumping; c.sqrtt = sqrt(ctx->temper); c.expt = exp(-p.temph0 / ctx->temper);
    // cosmology.F90:220: dzdt = H0*(1.+zred)*sqrt(Omega0*(1.+zred)**3+1.-Omega0)
    c.zp = 1.0 + ctx->zred;
    c.dzdt = (ctx->thermal && ctx->tprm.cosmological) ? ctx->tprm.H0 * c.zp * sqrt(ctx->tprm.Omega0 * (c.zp * c.zp * c.zp) + 1.0 - ctx->tprm.Omega0) : 0.0;
    if (img == ctx->step_image) return C2R_OK;
    // through the pinned staging block; the previous copy (a time step ago) has long read it, but say so
    if (ctx->ev_step_recorded) HIP_TRY(hipEventSynchronize(ctx->ev_step));
    memcpy(ctx->h_step, img.data(), img.size());
    HIP_TRY(hipMemcpyAsync(ctx->d_step, ctx->h_step, img.size(), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipEventRecord(ctx->ev_step, ctx->stream));
    ctx->ev_step_recorded = true;
    ctx->step_image.swap(img);
    return C2R_OK;
}

void prof_begin(Ctx *ctx, std::vector<std::pair<hipEvent_t, hipEvent_t>> &pool, size_t &used)
{
    if (!ctx->prof) return;
    if (used == pool.size()) { hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); pool.emplace_back(a, b); }
    hipEventRecord(pool[used].first, ctx->stream);
}
void prof_end(Ctx *ctx, std::vector<std::pair<hipEvent_t, hipEvent_t>> &pool, size_t &used)
{
    if (!ctx->prof) return;
    hipEventRecord(pool[used].second, ctx->stream);
    ++used;
}
void prof_collect(Ctx *ctx)
{
    if (!ctx->prof) return;
    for (size_t i = 0; i < ctx->ev_sweep_used; ++i) { float ms = 0; hipEventElapsedTime(&ms, ctx->ev_sweep[i].first, ctx->ev_sweep[i].second); ctx->prof_sweep_ms += ms; }
    for (size_t i = 0; i < ctx->ev_chem_used; ++i) { float ms = 0; hipEventElapsedTime(&ms, ctx->ev_chem[i].first, ctx->ev_chem[i].second); ctx->prof_chem_ms += ms; }
    for (size_t i = 0; i < ctx->ev_sweep_used; ++i) ctx->prof_sweep_n += i < ctx->ev_sweep_cnt.size() ? ctx->ev_sweep_cnt[i] : 1;
    ctx->ev_sweep_cnt.clear();
    ctx->prof_chem_n += (long long)ctx->ev_chem_used;
    ctx->ev_sweep_used = ctx->ev_chem_used = 0;
}

namespace {

// Sub-boxes ending at q <= kFusedQmax run in k_sweep_box_fused (one workgroup per source walks the shells).  With few
// sources and look-ahead pairs only the first sub-box does: beyond it three pair launches (22 us at 128^3 x 1 source) beat
// the single workgroup's five shells (36 us).
bool box_is_fused(const Ctx *ctx, int nbox, bool pair_ok)
{
    const c2r_params &p = ctx->prm;
    const int q0 = p.subboxsize * (nbox - 1) + 1, q1 = std::min(p.subboxsize * nbox, ctx->Qmax);
    return ctx->fuse_small && q1 <= kFusedQmax && q1 - q0 + 1 <= kMaxFused && !(pair_ok && nbox > 1);
}

// Shells q and q + 1 of sub-box nb as one look-ahead launch (k_sweep_pair_fast)?  Neither shell has cells on the sub-box
// surface (their loss partials and the order of the loss sums stay those of single launches), both have cells, and the
// second shell's threads -- one per cell and source, each redoing the arithmetic of ~5 cells -- fit the GPU at once: the
// pair trades arithmetic for a dependent launch, which pays only while a launch is latency (measured, profiles/
// r03_launch_bound: 128^3 x 1 source 0.385 -> 0.305 ms per iteration, 256^3 x 32 sources 6.4 -> 17.9 ms without this limit).
constexpr long long kPairMaxCells = 160000;
long long shell_cells(const Ctx *ctx, int q)
{
    long long c = 0;
    for (int f = 0; f < 6; ++f) { const FaceRect r = face_rect(ctx, f, q); if (r.ntiles > 0) c += (long long)r.wa * r.wb; }
    return c;
}
bool shell_on_surface(const Ctx *ctx, int nb, int q)
{
    for (int d = 0; d < 3; ++d)
        if (std::min(ctx->prm.subboxsize * nb, ctx->hr[d]) <= q || std::min(ctx->prm.subboxsize * nb, ctx->hl[d]) <= q) return true;
    return false;
}
bool pair_here(const Ctx *ctx, int nb, int q, int q1, int n_active, bool pair_ok)
{
    if (!pair_ok || q + 1 > q1 || shell_on_surface(ctx, nb, q) || shell_on_surface(ctx, nb, q + 1)) return false;
    const long long c0 = shell_cells(ctx, q), c1 = shell_cells(ctx, q + 1);
    return c0 > 0 && c1 > 0 && (long long)n_active * c1 <= kPairMaxCells;
}

// Which of a source's two plane sets holds the last shell of sub-box nbox - 1 (shell 0, the source cell, is in set 0).
// Every launch that stores planes reads one set and writes the other: a single shell, a shell of the fused first
// sub-boxes, or a look-ahead pair (two shells, one alternation) -- the rules of sweep_batch's enqueue_box
// (box_is_fused, pair_here), replayed for the sub-boxes before nbox.
int plane_set_before(const Ctx *ctx, int nbox, int n_active, bool pair_ok)
{
    const c2r_params &p = ctx->prm;
    int set = 0;
    for (int nb = 1; nb < nbox; ++nb) {
        const int q0 = p.subboxsize * (nb - 1) + 1, q1 = std::min(p.subboxsize * nb, ctx->Qmax);
        const bool fused = box_is_fused(ctx, nb, pair_ok);
        for (int q = q0; q <= q1; ++q) {
            if (shell_cells(ctx, q) == 0) continue;
            if (!fused && pair_here(ctx, nb, q, q1, n_active, pair_ok)) ++q;
            set ^= 1;
        }
    }
    return set;
}

// Wait for an event by polling it first: the wake-up of a blocking event wait is 20 - 50 us, which is what a sub-box of a few
// dozen sources lasts; after half a millisecond the ordinary wait takes over.
int wait_polling(Ctx *ctx, hipEvent_t ev)
{
    if (!ctx->poll_wait) { HIP_TRY(hipEventSynchronize(ev)); return C2R_OK; }
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0; hipEventQuery(ev) != hipSuccess; ++spins) {
        if ((spins & 63u) == 63u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(500)) {
            (void)hipGetLastError();
            HIP_TRY(hipEventSynchronize(ev));
            return C2R_OK;
        }
        cpu_relax();
    }
    (void)hipGetLastError();                 // (hipErrorNotReady of the polls)
    return C2R_OK;
}

// One batch of sources through the sweep -- local sources [first, first+count) of this rank's list: the staging block, the
// launches of a sub-box (source cells, fused first sub-boxes, shells and look-ahead pairs, loss sums, the decision), the
// captured launch sequence of a small batch and the wait behind a fused iteration, the run-ahead schedule.  sweep_batch()
// below is its only user.  dbg: optional device N^3 array receiving coldensh_out (single-source test path).
struct BatchSweep {
    Ctx *ctx; SweepScratch &sc; const c2r_params &p;
    const int first, count; const bool first_of_pass; double *const dbg; FusedIter *const fz;
    const size_t cap; hipStream_t st; KParams k;
    // the pinned staging block (layout of ensure_sweep_scratch)
    double *h_nf, *h_fl, *h_nfx; int *h_pos, *h_posw, *h_act, *h_na, *h_fnb;
    int n_active = 0;
    int cur = 0, last_bps = 0;     // which active list is current; size of the last shell's loss partials per source (0: none), for k_box_decide
    int totals_at_box = 0;         // fused iteration: the sub-box whose decision also writes the batch's totals (0: none)

    const int shape_count;         // sources of the whole pass (this rank): what the shape of the loss sums is chosen by
    bool chained = false;          // one of several chains in flight (run_chains): no per-launch timing events, no host waits of its own
    int launches = 0;              // shell launches enqueued so far (k_sweep_shell*, pairs)
    int bound = 0, known = 0;      // run_chains: upper bound of the device's active count; sub-boxes whose count has been read back
    int next = 1, last = 0;        // run_chains: the sub-box to enqueue next, the one enqueued last
    bool perm_ready = false;       // the batch's sources sorted along each axis are on the device (stage_perm): far shells may run plane-ordered

    BatchSweep(Ctx *c, SweepScratch &sc_, int first_, int count_, bool first_of_pass_, double *dbg_, FusedIter *fz_)
        : ctx(c), sc(sc_), p(c->prm), first(first_), count(count_), first_of_pass(first_of_pass_), dbg(dbg_), fz(fz_),
          cap((size_t)sc_.cap), st(sc_.stream), k(make_kparams(c, sc_)), shape_count(c->queue_next ? c->nsrc : n_local_sources(c))
    {
        h_nf = reinterpret_cast<double *>(sc.h_batch); h_fl = h_nf + cap; h_nfx = h_nf + 3 * cap;
        h_pos = reinterpret_cast<int *>(h_nf + 4 * cap); h_posw = h_pos + 3 * cap; h_act = h_pos + 6 * cap; h_na = h_pos + 9 * cap;
        h_fnb = h_pos + 8 * cap;   // the batch's results travel back through the same block (same layout as the device block)
    }

    // ---- 1. the staging block: sources, wrapped positions, the initial active list ---------------------------------
    // (it is next written by the next sweep_batch, after this one's final synchronize; it is uploaded by the graph's copy
    // node, by k_prepare_nhi from its device image, or directly)
    void stage()
    {
        memset(sc.h_batch, 0, sc.batch_bytes);                     // loss_acc = 0, final_nbox = 0, active lists
        const bool can_trace = ctx->hr[2] > 0 && ctx->hl[2] > 0;       // while condition, evolve_source.F90:130-131
        n_active = 0;
        for (int i = 0; i < count; ++i) {
            const int g = ctx->explicit_share ? ctx->share[first + i]
                                              : ctx->rank + (first + i) * ctx->nranks;      // master_slave.F90:85
            for (int d = 0; d < 3; ++d) {
                h_pos[3 * i + d] = ctx->srcpos[3 * (size_t)g + d];
                const int m = (h_pos[3 * i + d] - 1) % p.mesh[d];
                h_posw[3 * i + d] = m < 0 ? m + p.mesh[d] : m;         // evolve_point.F90:122 for the source cell
            }
            h_nf[i] = ctx->nflux[g];
            h_nfx[i] = (ctx->xray && g < (int)ctx->nflux_x.size()) ? ctx->nflux_x[g] : 0.0;       // NormFlux_xray(ns), sourceprops.F90:381
            const double flux = h_nf[i] * p.S_star;
            if (flux > p.loss_fraction * flux && can_trace) h_act[n_active++] = i;
            else h_fl[i] = flux;                                       // loop never entered: loss = initial value
        }
        h_na[0] = n_active; h_na[1] = 0;
    }

    // Plane-ordered far shells (k_sweep_shell_xcd): no debug array, and enough sources that
    // several share a mesh plane (n_active / planes per face sign); uploaded behind the staging block
    void stage_perm()
    {
        perm_ready = false;
        if (dbg || !sc.d_perm || ctx->xcd_order == 0 || n_active < ctx->xcd_min_sources) return;
        const int nmin = std::min(p.mesh[0], std::min(p.mesh[1], p.mesh[2]));
        // (with ordered rates from half that count: there the mapping pays by keeping a source's per-source grid writes together --
        // 256^3, batches of 250 sources: -7 %, 1000 sources in four batches -7 ... -20 %: profiles/r05_xcd/ab_det_counts.txt)
        if (ctx->xcd_order < 0 && (double)n_active < ctx->xcd_min_per_plane * (double)nmin * (sc.d_gbox ? 0.5 : 1.0)) return;
        for (int d = 0; d < 3; ++d) {
            int *perm = sc.h_perm + (size_t)d * cap;
            for (int t = 0; t < n_active; ++t) perm[t] = h_act[t];
            std::stable_sort(perm, perm + n_active, [&](int a, int b) { return h_posw[3 * a + d] < h_posw[3 * b + d]; });
        }
        perm_ready = true;
    }

    // ---- 2. the launches of one sub-box -----------------------------------------------------------------------------
    // what the launches of sub-box nbox share
    struct Box {
        int nbox, bound;               // the sub-box; upper bound of the device's active count (sizes the grids)
        int next_bound;                // what the NEXT sub-box's launches are sized for, where that is fixed already (a captured sequence); else INT_MAX
        int boxR[3], boxL[3];          // last_r / last_l - srcpos (evolve_source.F90:135-136)
        bool pair_ok, fused_box, det;
        int pbuf;                      // which plane set holds shell q0 - 1
        int q0, q1;
    };

    ShellArgs shell_args(const Box &bx, int q) const
    {
        const int (&boxR)[3] = bx.boxR, (&boxL)[3] = bx.boxL;
        ShellArgs sa{};
        sa.q = q;
        sa.buf_prev = (q - 1) & 1; sa.buf_cur = q & 1;       // (a look-ahead pair sets its own, below)
        sa.tiles_max = 0;
        for (int f = 0; f < 6; ++f) { sa.face[f] = face_rect(ctx, f, q); sa.tiles_max = std::max(sa.tiles_max, sa.face[f].ntiles); }
        sa.has_boundary = 0;
        for (int d = 0; d < 3; ++d) {
            sa.boxR[d] = boxR[d]; sa.boxL[d] = boxL[d];
            if (boxR[d] <= q || boxL[d] <= q) sa.has_boundary = 1;
        }
        sa.alam = (double)((float)(q - 1) + 0.5f) / (double)(float)q;
        sa.dp2 = (double)q * (double)q; sa.inv_dp2 = 1.0 / sa.dp2;
        sa.inv_q = 1.0 / (double)q;                  // ((dr_d q)^2, dr[0]/q, coldensh_LLS/q: the step block, sync_step)
        sa.active = sc.d_active[cur]; sa.n_active = sc.d_nactive + cur;
        sa.loss_partial = sc.d_loss_partial; sa.dbg_cdout = dbg;
        return sa;
    }

    void launch_source_cells(const Box &bx)
    {
        const int (&boxR)[3] = bx.boxR, (&boxL)[3] = bx.boxL;
        const int bound = bx.bound;
        {
            if (ctx->thermal && ctx->xray)
                hipLaunchKernelGGL(k_source_cells<3>, dim3((bound + 63) / 64), dim3(64), 0, st, k, bound,
                                   sc.d_active[cur], boxR[0], boxR[1], boxR[2], boxL[0], boxL[1], boxL[2],
                                   sc.d_loss_acc, dbg);
            else if (ctx->thermal)
                hipLaunchKernelGGL(k_source_cells<1>, dim3((bound + 63) / 64), dim3(64), 0, st, k, bound,
                                   sc.d_active[cur], boxR[0], boxR[1], boxR[2], boxL[0], boxL[1], boxL[2],
                                   sc.d_loss_acc, dbg);
            else if (ctx->xray)
                hipLaunchKernelGGL(k_source_cells<2>, dim3((bound + 63) / 64), dim3(64), 0, st, k, bound,
                                   sc.d_active[cur], boxR[0], boxR[1], boxR[2], boxL[0], boxL[1], boxL[2],
                                   sc.d_loss_acc, dbg);
            else
                hipLaunchKernelGGL(k_source_cells<0>, dim3((bound + 63) / 64), dim3(64), 0, st, k, bound,
                                   sc.d_active[cur], boxR[0], boxR[1], boxR[2], boxL[0], boxL[1], boxL[2],
                                   sc.d_loss_acc, dbg);
        }
    }

    // near the source: the whole sub-box of every active source in ONE launch (k_sweep_box_fused)
    void launch_fused_box(const Box &bx)
    {
        const int (&boxR)[3] = bx.boxR, (&boxL)[3] = bx.boxL;
        const int nbox = bx.nbox, bound = bx.bound, q0 = bx.q0, q1 = bx.q1;
        const bool det = bx.det;
        // near the source: the whole sub-box of every active source in ONE launch (k_sweep_box_fused)
        BoxArgs ba{};
        int most = 0;
        for (int q = q0; q <= q1; ++q) {
            ShellArgs sa = shell_args(bx, q);
            if (sa.tiles_max == 0) continue;
            const int k = ba.nshell++;
            int off = 0;
            for (int f = 0; f < 6; ++f) { ba.face_off[k][f] = off; off += sa.face[f].ntiles ? sa.face[f].wa * sa.face[f].wb : 0; }
            ba.face_off[k][6] = ba.face_off[k][7] = off;
            ba.ncell[k] = off; most = std::max(most, off);
            ba.sh[k] = sa;
        }
        if (nbox == 1 && ctx->fold_source_cell) {
            if (ba.nshell > 0) ba.source_cell = 1;
            else {      // no shell at all to walk (degenerate limits): the plain kernel after all
                if (ctx->thermal && ctx->xray)
                    hipLaunchKernelGGL(k_source_cells<3>, dim3((bound + 63) / 64), dim3(64), 0, st, k, bound, sc.d_active[cur],
                                       boxR[0], boxR[1], boxR[2], boxL[0], boxL[1], boxL[2], sc.d_loss_acc, dbg);
                else if (ctx->thermal)
                    hipLaunchKernelGGL(k_source_cells<1>, dim3((bound + 63) / 64), dim3(64), 0, st, k, bound, sc.d_active[cur],
                                       boxR[0], boxR[1], boxR[2], boxL[0], boxL[1], boxL[2], sc.d_loss_acc, dbg);
                else if (ctx->xray)
                    hipLaunchKernelGGL(k_source_cells<2>, dim3((bound + 63) / 64), dim3(64), 0, st, k, bound, sc.d_active[cur],
                                       boxR[0], boxR[1], boxR[2], boxL[0], boxL[1], boxL[2], sc.d_loss_acc, dbg);
                else
                    hipLaunchKernelGGL(k_source_cells<0>, dim3((bound + 63) / 64), dim3(64), 0, st, k, bound, sc.d_active[cur],
                                       boxR[0], boxR[1], boxR[2], boxL[0], boxL[1], boxL[2], sc.d_loss_acc, dbg);
            }
        }
        if (ba.nshell > 0) {
            ba.active = sc.d_active[cur]; ba.n_active = sc.d_nactive + cur; ba.loss_acc = sc.d_loss_acc;
            // one workgroup per source: 256 / 512 / 1024 threads by the largest shell; with many sources 512 at most (two
            // workgroups per CU hide each other's shell-to-shell latency: cold 256^3 x 1000 0.973 -> 0.939 ms per
            // iteration).  By the PASS's source count: the block size shapes the loss sums, which must depend neither on timing
            // nor on how the pass was cut into batches and chains.
            int bt = most <= 256 ? 256 : (most <= 512 ? 512 : 1024);
            if (shape_count >= 256) bt = std::min(bt, 512);
            const dim3 grid(bound), blk(bt);
            // (not in the k_sweep_shell launch timing of c2r_profile: a different kernel, 21^3 cells per source)
#define C2R_LAUNCH_FUSED_H(D, L, H) do { if (ctx->fast) hipLaunchKernelGGL((k_sweep_box_fused<D, L, true, H>), grid, blk, 0, st, k, ba); \
                                else hipLaunchKernelGGL((k_sweep_box_fused<D, L, false, H>), grid, blk, 0, st, k, ba); } while (0)
#define C2R_LAUNCH_FUSED(D, L) do { if (ctx->thermal && ctx->xray) C2R_LAUNCH_FUSED_H(D, L, 3); else if (ctx->thermal) C2R_LAUNCH_FUSED_H(D, L, 1); else if (ctx->xray) C2R_LAUNCH_FUSED_H(D, L, 2); else C2R_LAUNCH_FUSED_H(D, L, 0); } while (0)
            switch (ctx->lls_type * 2 + (det ? 1 : 0)) {
                case 2: C2R_LAUNCH_FUSED(false, 1); break;
                case 3: C2R_LAUNCH_FUSED(true, 1); break;
                case 4: C2R_LAUNCH_FUSED(false, 2); break;
                case 5: C2R_LAUNCH_FUSED(true, 2); break;
                case 6: C2R_LAUNCH_FUSED(false, 3); break;
                default: C2R_LAUNCH_FUSED(true, 3); break;
            }
#undef C2R_LAUNCH_FUSED
#undef C2R_LAUNCH_FUSED_H
        }
    }

    // one launch per shell (or per look-ahead pair), the loss sums of shells that touch the sub-box surface
    void launch_shells(Box &bx)
    {
        const int nbox = bx.nbox, bound = bx.bound, q0 = bx.q0, q1 = bx.q1;
        const bool det = bx.det, pair_ok = bx.pair_ok;
        int &pbuf = bx.pbuf;
        int in_box = 0;                         // k_sweep_shell launches of this sub-box (coarse timing)
        last_bps = 0;
        const int prof = chained ? 0 : ctx->prof;     // (chains in flight overlap: run_chains times the round as a whole)
        if (prof == 2) prof_begin(ctx, ctx->ev_sweep, ctx->ev_sweep_used);
        for (int q = q0; q <= q1; ++q) {
            ShellArgs sa = shell_args(bx, q);
            if (sa.tiles_max == 0) continue;
            sa.buf_prev = pbuf; sa.buf_cur = 1 - pbuf;
            // Few sources, fast mode: shells q and q+1 in ONE launch, both from the planes of shell q-1 (k_sweep_pair_fast:
            // the second recomputes the first's column densities) -- half the dependent launches where a launch is nothing
            // but latency.  Not where either shell has cells on the sub-box surface (their loss partials and the order of
            // the loss sums stay those of the single launches).
            if (pair_here(ctx, nbox, q, q1, n_active, pair_ok)) {
                ShellArgs sb = shell_args(bx, q + 1);
                {
                    // the second shell's threads take kPairRows rows each (its per-thread work is the recompute of
                    // 2 (rows + 1) cells of the first shell: short chains on more threads, the GPU is empty anyway)
                    sb.tiles_max = 0;
                    for (int f = 0; f < 6; ++f) { sb.face[f] = face_rect(ctx, f, q + 1, kPairRows); sb.tiles_max = std::max(sb.tiles_max, sb.face[f].ntiles); }
                    sb.buf_prev = pbuf; sb.buf_cur = 1 - pbuf;        // (buf_prev of the second shell is never read)
                    ++in_box;
                    const dim3 grid(std::max(sa.tiles_max, sb.tiles_max), 12, bound), blk(kBlock);
#define C2R_LAUNCH_PAIR_H(D, L, H) do { \
    if (ctx->fast) { if (ctx->stream_hint) hipLaunchKernelGGL((k_sweep_pair_fast<D, L, true, H>), grid, blk, 0, st, k, sa, sb); \
                     else hipLaunchKernelGGL((k_sweep_pair_fast<D, L, false, H>), grid, blk, 0, st, k, sa, sb); } \
    else if (ctx->stream_hint) hipLaunchKernelGGL((k_sweep_pair<D, L, true, H>), grid, blk, 0, st, k, sa, sb); \
    else hipLaunchKernelGGL((k_sweep_pair<D, L, false, H>), grid, blk, 0, st, k, sa, sb); } while (0)
#define C2R_LAUNCH_PAIR(D, L) do { if (ctx->thermal && ctx->xray) C2R_LAUNCH_PAIR_H(D, L, 3); else if (ctx->thermal) C2R_LAUNCH_PAIR_H(D, L, 1); else if (ctx->xray) C2R_LAUNCH_PAIR_H(D, L, 2); else C2R_LAUNCH_PAIR_H(D, L, 0); } while (0)
                    switch (ctx->lls_type * 2 + (det ? 1 : 0)) {
                        case 2: C2R_LAUNCH_PAIR(false, 1); break;
                        case 3: C2R_LAUNCH_PAIR(true, 1); break;
                        case 4: C2R_LAUNCH_PAIR(false, 2); break;
                        case 5: C2R_LAUNCH_PAIR(true, 2); break;
                        case 6: C2R_LAUNCH_PAIR(false, 3); break;
                        default: C2R_LAUNCH_PAIR(true, 3); break;
                    }
#undef C2R_LAUNCH_PAIR
#undef C2R_LAUNCH_PAIR_H
                    pbuf = 1 - pbuf;
                    ++q;
                    continue;
                }
            }
            pbuf = 1 - pbuf;
            ++in_box;
            if (prof == 1) prof_begin(ctx, ctx->ev_sweep, ctx->ev_sweep_used);
            if (perm_ready && !sa.has_boundary && q >= ctx->xcd_qmin && (double)bound >= ctx->xcd_min_alive * (double)n_active) {
                // the plane-ordered mapping: 8 XCD groups x (an eighth of the sources) x (the six faces' tiles)
                XcdArgs xa{};
                xa.perm = sc.d_perm; xa.n = n_active; xa.cap = (int)cap; xa.final_nbox = sc.d_final_nbox;
                int tiles6 = 0;
                for (int f = 0; f < 6; ++f) tiles6 += sa.face[f].ntiles;
                const dim3 grid(8u * (unsigned)((n_active + 7) / 8) * (unsigned)tiles6), blk(kBlock);
#define C2R_LAUNCH_XCD_D(D, L, H, F) do { if (ctx->stream_hint) hipLaunchKernelGGL((k_sweep_shell_xcd<D, L, true, H, F>), grid, blk, 0, st, k, sa, xa); \
                                          else hipLaunchKernelGGL((k_sweep_shell_xcd<D, L, false, H, F>), grid, blk, 0, st, k, sa, xa); } while (0)
#define C2R_LAUNCH_XCD_F(L, H, F) do { if (det) C2R_LAUNCH_XCD_D(true, L, H, F); else C2R_LAUNCH_XCD_D(false, L, H, F); } while (0)
#define C2R_LAUNCH_XCD_H(L, H) do { if (ctx->fast) C2R_LAUNCH_XCD_F(L, H, true); else C2R_LAUNCH_XCD_F(L, H, false); } while (0)
#define C2R_LAUNCH_XCD(L) do { if (ctx->thermal && ctx->xray) C2R_LAUNCH_XCD_H(L, 3); else if (ctx->thermal) C2R_LAUNCH_XCD_H(L, 1); else if (ctx->xray) C2R_LAUNCH_XCD_H(L, 2); else C2R_LAUNCH_XCD_H(L, 0); } while (0)
                switch (ctx->lls_type) {
                    case 1: C2R_LAUNCH_XCD(1); break;
                    case 2: C2R_LAUNCH_XCD(2); break;
                    default: C2R_LAUNCH_XCD(3); break;
                }
#undef C2R_LAUNCH_XCD
#undef C2R_LAUNCH_XCD_H
#undef C2R_LAUNCH_XCD_F
#undef C2R_LAUNCH_XCD_D
                ++ctx->xcd_launches;
            } else
            {
                const dim3 grid(sa.tiles_max, 6, bound), blk(kBlock);
#define C2R_LAUNCH_SWEEP_H(D, L, H) do { \
    if (ctx->fast) { if (ctx->stream_hint) hipLaunchKernelGGL((k_sweep_shell_fast<D, L, true, H>), grid, blk, 0, st, k, sa); \
                     else hipLaunchKernelGGL((k_sweep_shell_fast<D, L, false, H>), grid, blk, 0, st, k, sa); } \
    else if (ctx->stream_hint) hipLaunchKernelGGL((k_sweep_shell<D, L, true, H>), grid, blk, 0, st, k, sa); \
    else hipLaunchKernelGGL((k_sweep_shell<D, L, false, H>), grid, blk, 0, st, k, sa); } while (0)
#define C2R_LAUNCH_SWEEP(D, L) do { if (ctx->thermal && ctx->xray) C2R_LAUNCH_SWEEP_H(D, L, 3); else if (ctx->thermal) C2R_LAUNCH_SWEEP_H(D, L, 1); else if (ctx->xray) C2R_LAUNCH_SWEEP_H(D, L, 2); else C2R_LAUNCH_SWEEP_H(D, L, 0); } while (0)
                switch (ctx->lls_type * 2 + (det ? 1 : 0)) {
                    case 2: C2R_LAUNCH_SWEEP(false, 1); break;
                    case 3: C2R_LAUNCH_SWEEP(true, 1); break;
                    case 4: C2R_LAUNCH_SWEEP(false, 2); break;
                    case 5: C2R_LAUNCH_SWEEP(true, 2); break;
                    case 6: C2R_LAUNCH_SWEEP(false, 3); break;
                    default: C2R_LAUNCH_SWEEP(true, 3); break;
                }
#undef C2R_LAUNCH_SWEEP
#undef C2R_LAUNCH_SWEEP_H
            }
            if (prof == 1) { prof_end(ctx, ctx->ev_sweep, ctx->ev_sweep_used); ctx->ev_sweep_cnt.push_back(1); }
            // the partials of the sub-box's last shell are summed by k_box_decide itself when few sources are active (one
            // launch less where launches are all there is); with many, its single workgroup would read 6 x tiles partials
            // for every source (2.5 MB through one CU: 89 us per sub-box, 1.6 % of the bench step) -- one block per source then
            // (decided by the PASS's source count: `bound` depends on when the host happens to see a count arrive, the batch's
            // size on the scratch budget and the chains -- and the two paths round differently: the photon loss, which feeds
            // the keep/retire decision, must not)
            const bool fold = shape_count <= kFoldLossMax;
            if (sa.has_boundary && (q < q1 || !fold))
                hipLaunchKernelGGL(k_loss_reduce, dim3(bound), dim3(256), 0, st, sc.d_active[cur], sc.d_nactive + cur,
                                   sc.d_loss_partial, 6 * sa.tiles_max, sc.d_loss_acc);
            if (q == q1 && sa.has_boundary && fold) last_bps = 6 * sa.tiles_max;
        }
        if (prof == 2) { prof_end(ctx, ctx->ev_sweep, ctx->ev_sweep_used); ctx->ev_sweep_cnt.push_back(in_box); }
        launches += in_box;
    }

    // k_box_decide: which sources go on to the next sub-box (evolve_source.F90:128-131)
    void launch_decision(const Box &bx)
    {
        const int nbox = bx.nbox;
        int *const halt = sc.d_hnactive;                  // slot 0 of the pinned counts: the sub-box a replayed sequence halted at (0: none)
        const int can_grow = (p.subboxsize * nbox < ctx->hr[2]) && (p.subboxsize * nbox < ctx->hl[2]);
        if (n_active <= 64) {
            // one wave decides; at the sub-box a fused iteration's graph ends with it also leaves the batch's totals and
            // results (SmallTotals) -- host_final_*: the staging block's final_nbox / final_loss through its mapped alias
            SmallTotals tot{};
            if (nbox == totals_at_box) {
                tot.on = 1; tot.nsrc = count; tot.photon_loss = ctx->d_photon_loss; tot.sum_nbox = ctx->d_sum_nbox;
                tot.host_loss = &ctx->d_hsc->photon_loss; tot.host_nbox = &ctx->d_hsc->sum_nbox;
                tot.host_final_loss = reinterpret_cast<double *>(sc.d_hbatch) + cap;
                tot.host_final_nbox = reinterpret_cast<int *>(reinterpret_cast<double *>(sc.d_hbatch) + 4 * cap) + 8 * cap;
            }
            hipLaunchKernelGGL(k_box_decide_small, dim3(1), dim3(64), 0, st, sc.d_active[cur], sc.d_nactive + cur,
                               sc.d_active[1 - cur], sc.d_nactive + (1 - cur), sc.d_hnactive + nbox, sc.d_nflux_b,
                               p.S_star, p.loss_fraction, can_grow, nbox, sc.d_loss_acc, sc.d_final_loss, sc.d_final_nbox,
                               (const double *)sc.d_loss_partial, last_bps, tot, bx.next_bound, halt);
        } else
        hipLaunchKernelGGL(k_box_decide, dim3(1), dim3(1024), 0, st, sc.d_active[cur], sc.d_nactive + cur,
                           sc.d_active[1 - cur], sc.d_nactive + (1 - cur), sc.d_hnactive + nbox, sc.d_nflux_b,
                           p.S_star, p.loss_fraction, can_grow, nbox, sc.d_loss_acc, sc.d_final_loss, sc.d_final_nbox,
                           (const double *)sc.d_loss_partial, last_bps, bx.next_bound, halt);
    }

    // every launch of sub-box nbox for `bound` sources at most (no host wait, no event); flips `cur`
    int enqueue_box(const int nbox, const int bound, const int next_bound = INT_MAX)
    {
        Box bx{};
        bx.nbox = nbox; bx.bound = bound; bx.next_bound = next_bound;
        for (int d = 0; d < 3; ++d) {
            bx.boxR[d] = std::min(p.subboxsize * nbox, ctx->hr[d]);
            bx.boxL[d] = std::min(p.subboxsize * nbox, ctx->hl[d]);
        }
        // which plane set holds shell q0 - 1: the shells alternate between the two sets, a look-ahead pair advances two
        // shells in one alternation -- a pure function of the schedule up to this sub-box (a replayed graph does not run
        // this code), see plane_set_before
        bx.pair_ok = ctx->pair_shells && n_active <= kFewSources && !dbg && ctx->prof != 1;
        bx.pbuf = plane_set_before(ctx, nbox, n_active, bx.pair_ok);
        bx.fused_box = box_is_fused(ctx, nbox, bx.pair_ok);
        bx.q0 = p.subboxsize * (nbox - 1) + 1; bx.q1 = std::min(p.subboxsize * nbox, ctx->Qmax);
        bx.det = sc.d_gbox != nullptr;
        // (the fused first sub-box does the source cells itself: one launch less)
        if (nbox == 1 && !(bx.fused_box && ctx->fold_source_cell)) launch_source_cells(bx);
        last_bps = 0;
        if (bx.fused_box) launch_fused_box(bx); else launch_shells(bx);
        launch_decision(bx);
        cur = 1 - cur;
        return C2R_OK;
    }

    // the batch's share of photon_loss / sum_nbox, added to the pass's running totals in source order -- on stream `on`: the
    // chain's own, or the context's when several chains were in flight (run_chains: chain after chain, i.e. in source order)
    int enqueue_totals(std::vector<int> *nbox_out, std::vector<double> *loss_out, hipStream_t on, bool first_batch, const int *gate = nullptr)
    {
        hipLaunchKernelGGL(k_batch_totals, dim3(1), dim3(1024), 0, on, count, sc.d_final_loss, sc.d_final_nbox,
                           ctx->d_photon_loss, ctx->d_sum_nbox, first_batch ? 1 : 0, &ctx->d_hsc->photon_loss,
                           &ctx->d_hsc->sum_nbox, gate);
        HIP_TRY(hipGetLastError());
        if (nbox_out) HIP_TRY(hipMemcpyAsync(h_fnb, sc.d_final_nbox, (size_t)count * sizeof(int), hipMemcpyDeviceToHost, on));
        if (loss_out) HIP_TRY(hipMemcpyAsync(h_fl, sc.d_final_loss, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, on));
        return C2R_OK;
    }
    // deterministic rates: the per-source grids are summed in source order once every source has its final sub-box
    void gamma_reduce(const int *gate)
    {
        if (sc.d_gbox)
            hipLaunchKernelGGL(k_gamma_reduce, dim3((p.mesh[0] + 7) / 8, (p.mesh[1] + 31) / 32, p.mesh[2]), dim3(256), 0, st, k, count,
                               sc.d_final_nbox, p.subboxsize, (double *)ctx->grid[4], ctx->thermal ? (double *)ctx->grid[5] : nullptr, gate);
    }

    // ---- 3. a small batch's launch sequence up to sub-box `hint` as ONE hipGraph ---------------------------------------
    // (re)capture into bg: the batch upload (plain pass) or what precedes the pass (fused iteration), sub-boxes 1..hint, and --
    // fused iteration -- the gated rest of the iteration.  On failure the context falls back to eager launches for good.
    void capture(Ctx::BatchGraph &bg, const bool fuse_iter, const int hint)
    {
        if (bg.exec) { hipGraphExecDestroy(bg.exec); bg.exec = nullptr; }
        if (bg.graph) { hipGraphDestroy(bg.graph); bg.graph = nullptr; }
        if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            ++ctx->captures;
            // fused iteration: no upload node -- k_prepare_nhi restores the batch's state block from its device image
            if (fuse_iter) fz->batch_in_prepare = true;
            int rc = fuse_iter ? fz->pre() : C2R_OK;
            if (fuse_iter) fz->batch_in_prepare = false;
            totals_at_box = fuse_iter ? std::min(hint, ctx->nbox_max) : 0;
            if (rc == C2R_OK && !fuse_iter) rc = (int)hipMemcpyAsync(sc.d_batch, sc.h_batch, sc.batch_bytes, hipMemcpyHostToDevice, st);
            cur = 0;
            for (int nbox = 1; nbox <= hint && nbox <= ctx->nbox_max && rc == C2R_OK; ++nbox) rc = enqueue_box(nbox, n_active);
            if (fuse_iter && rc == C2R_OK) {
                // the batch's totals and results, then the gated rest of the iteration: d_nactive[cur] is the count the
                // last decision left (cur has been flipped by it)
                gamma_reduce(sc.d_nactive + cur);
                rc = fz->post(sc.d_nactive + cur);
            }
            totals_at_box = 0;
            const hipError_t e = hipStreamEndCapture(st, &bg.graph);
            if (rc == C2R_OK && e == hipSuccess && hipGraphInstantiate(&bg.exec, bg.graph, nullptr, nullptr, 0) == hipSuccess) {
                bg.gen = ctx->gen; bg.count = count; bg.n_active = n_active; bg.hint = hint; bg.acc = (const void *)k.phih;
                bg.fused = fuse_iter; bg.stats = fuse_iter && fz->stats;
            } else {
                if (bg.graph) { hipGraphDestroy(bg.graph); bg.graph = nullptr; }
                bg.exec = nullptr;
                (void)hipGetLastError();
                ctx->use_graph = false;            // this runtime / stream cannot capture: eager from now on
            }
        } else { (void)hipGetLastError(); ctx->use_graph = false; }
    }

    // A chain's whole pass as one launch sequence (run_chains): the batch upload and sub-boxes 1..H of the chain's last pass, every
    // launch sized by bounds_from_profile, every decision guarding the next sub-box's size.  false: not captured (the context falls
    // back to launch-by-launch for good where the runtime cannot capture at all).
    bool capture_chain(Ctx::ChainGraph &cg);

    // behind a fused iteration's graph: the last kernel of the gated tail stores the count of completed passes to pinned
    // memory as its final act -- poll it instead of blocking (bounded); true: the gate was open, the whole iteration has run
    int wait_fused(const Ctx::BatchGraph &bg, const int done, bool &arrived_out)
    {
        bool arrived = false;
        if (bg.fused && ctx->spin_wait) {
            // The last kernel of the gated tail stores the count of completed passes to pinned memory as its final
            // act: poll it (and the sub-box count, which tells a shut gate) instead of blocking -- the wake-up of a
            // stream synchronize is a tenth of a 0.26 ms iteration.  Bounded: after 2 ms the ordinary wait takes over.
            const unsigned long long want = ctx->seq_seen + 1;
            const auto t0 = std::chrono::steady_clock::now();
            for (unsigned spins = 0;; ++spins) {
                if (__atomic_load_n(&ctx->h_sc->seq, __ATOMIC_ACQUIRE) == want) { arrived = true; break; }
                if (__atomic_load_n(&sc.h_nactive[done], __ATOMIC_ACQUIRE) > 0) break;
                if ((spins & 255u) == 255u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
                cpu_relax();
            }
        }
        if (!arrived) {
            HIP_TRY(hipStreamSynchronize(st));
            arrived = bg.fused && __atomic_load_n(&ctx->h_sc->seq, __ATOMIC_ACQUIRE) == ctx->seq_seen + 1;
        }
        arrived_out = arrived;
        return C2R_OK;
    }

    // ---- 4. the schedule ---------------------------------------------------------------------------------------------
    int run(std::vector<int> *nbox_out, std::vector<double> *loss_out);
};

int BatchSweep::run(std::vector<int> *nbox_out, std::vector<double> *loss_out)
{
    stage();
    stage_perm();
    // The active count lives on the device (d_nactive[cur]); the host only needs an upper bound to
    // size the grids.  It runs ONE sub-box ahead: box n+1 is enqueued (sized by the count known
    // after box n-1) before the count after box n is read back, so the GPU never drains while the
    // host waits; blocks of sources that retired in between return at once.
    int bound = n_active;          // upper bound of the device count for the launches being enqueued
    int known = 0;                 // sub-boxes whose resulting count has been read back
    // How far ahead of the device the host runs.  Normally ONE sub-box: box n+1 is enqueued, sized by the count known
    // after box n-1, before the count after box n is read back -- the GPU never drains while the host waits, and
    // blocks of sources that retired in between return at once.  A batch of FEW sources (<= kFewSources) is nothing but
    // launch latency, and every wait is a host round trip with the GPU idle: there the host does not wait at all up to
    // the sub-box the previous pass ended at (box_hint: in the steady state of an outer iteration the sources retire
    // where they did last time), only picking up counts that have already arrived; at that sub-box it waits for the
    // box's own count (normally zero: done).  Measured (profiles/r02_launch_bound/): 128^3 x 1 source 0.80 -> 0.73 ms per
    // outer iteration; with 1000 sources the same rule costs 5-20 % (stale large grids), hence the limit.
    const bool few = ctx->sched_hint && n_active <= kFewSources;
    const int hint = few ? std::max(1, ctx->box_hint) : 1;
    int first_box = 1;
    // A batch of few sources whose previous pass ended at sub-box `hint` replays that whole launch sequence (the batch
    // upload, the source cells, sub-boxes 1..hint) as ONE hipGraph: the arguments of every launch are the same from
    // outer iteration to outer iteration (the batch's data travel in the pinned staging block, read when the copy node
    // runs), a replay costs one host call instead of ~8 us per launch, and dependent nodes follow each other in ~2 us.
    const bool graph_ok = ctx->use_graph && ctx->sched_hint && n_active > 0 && n_active <= kFewSources && ctx->box_hint >= 1 &&
                          !dbg && ctx->prof == 0;
    bool uploaded = false;
    const bool fuse_iter = fz && graph_ok && first_of_pass;
    bool pre_run = false;
    if (graph_ok) {
        // (one slot for the batch's plain pass, one for the fused iteration around it: a host that alternates between
        // c2r_pass_sources and c2r_iterate does not re-capture every time)
        Ctx::BatchGraph &bg = ctx->graphs[2 * first + (fuse_iter ? 1 : 0)];
        if (!(bg.exec && bg.gen == ctx->gen && bg.count == count && bg.n_active == n_active && bg.hint == hint &&
              bg.fused == fuse_iter && (!fuse_iter || bg.stats == fz->stats) && bg.acc == (const void *)k.phih))
            capture(bg, fuse_iter, hint);
        if (bg.exec) {
            const int done = std::min(hint, ctx->nbox_max);
            if (bg.fused) {
                sc.h_nactive[done] = -1;                        // (so that a stale zero is not taken for this launch's count)
                // the device image of the state block: sent only when it differs from what was sent last (steady state: never)
                if (sc.batch_image.size() != sc.batch_bytes || memcmp(sc.batch_image.data(), sc.h_batch, sc.batch_bytes) != 0) {
                    sc.batch_image.assign(sc.h_batch, sc.h_batch + sc.batch_bytes);
                    // (from the pinned block itself: it is not touched again before this iteration's kernels have run)
                    HIP_TRY(hipMemcpyAsync(sc.d_batch_init, sc.h_batch, sc.batch_bytes, hipMemcpyHostToDevice, st));
                }
            }
            HIP_TRY(hipGraphLaunch(bg.exec, st));
            uploaded = true;
            pre_run = bg.fused;
            cur = done & 1;
            bool arrived = false;
            { const int rc = wait_fused(bg, done, arrived); if (rc) return rc; }
            known = done; bound = sc.h_nactive[done];
            first_box = done + 1;
            if (bg.fused && arrived) { ctx->seq_seen += 1; bound = 0; fz->tail_done = true; }   // the gate was open: the whole iteration has run
        } else cur = 0;
    }
    if (fz && !pre_run) { const int rc = fz->pre(); if (rc) return rc; }
    if (!uploaded) HIP_TRY(hipMemcpyAsync(sc.d_batch, sc.h_batch, sc.batch_bytes, hipMemcpyHostToDevice, st));
    // (h_perm, like the staging block, is rewritten by the next batch only after this one's final synchronize)
    if (perm_ready) HIP_TRY(hipMemcpyAsync(sc.d_perm, sc.h_perm, (size_t)3 * cap * sizeof(int), hipMemcpyHostToDevice, st));
    for (int nbox = first_box; nbox <= ctx->nbox_max && bound > 0; ++nbox) {
        { const int rc = enqueue_box(nbox, bound); if (rc) return rc; }
        HIP_TRY(hipEventRecord(sc.ev_box[nbox], st));
        // counts that have already arrived (never blocks)
        while (known < nbox && hipEventQuery(sc.ev_box[known + 1]) == hipSuccess) bound = sc.h_nactive[++known];
        // blocking read-back: the box's own count where the previous pass ended, the previous box's beyond
        // (many sources: always the previous box's -- one sub-box stays in flight from the first box on)
        const int need = !few ? nbox - 1 : (nbox == hint ? nbox : (nbox > hint ? nbox - 1 : 0));
        if (need > known) {
            { const int rc = wait_polling(ctx, sc.ev_box[need]); if (rc) return rc; }
            known = need; bound = sc.h_nactive[need];
        }
    }
    if (!(fz && fz->tail_done)) {          // (the fused iteration's graph has done this already)
        gamma_reduce(nullptr);
        { const int rc = enqueue_totals(nbox_out, loss_out, st, first_of_pass); if (rc) return rc; }
        HIP_TRY(hipStreamSynchronize(st));
    }
    if (nbox_out) nbox_out->assign(h_fnb, h_fnb + count);
    if (loss_out) loss_out->assign(h_fl, h_fl + count);
    return C2R_OK;
}

static void bounds_from_profile(const std::vector<int> &prof, int n_active, std::vector<int> &bounds, int &H);

bool BatchSweep::capture_chain(Ctx::ChainGraph &cg)
{
    if (cg.exec) { hipGraphExecDestroy(cg.exec); cg.exec = nullptr; }
    if (cg.graph) { hipGraphDestroy(cg.graph); cg.graph = nullptr; }
    bounds_from_profile(cg.profile, n_active, cg.bounds, cg.H);
    if (cg.H < 1) return false;
    if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); ctx->use_graph = false; return false; }
    ++ctx->captures;
    int rc = (int)hipMemcpyAsync(sc.d_batch, sc.h_batch, sc.batch_bytes, hipMemcpyHostToDevice, st);
    if (rc == C2R_OK && perm_ready) rc = (int)hipMemcpyAsync(sc.d_perm, sc.h_perm, (size_t)3 * cap * sizeof(int), hipMemcpyHostToDevice, st);
    const int cur0 = cur, launches0 = launches;
    cur = 0;
    for (int nbox = 1; nbox <= cg.H && rc == C2R_OK; ++nbox) rc = enqueue_box(nbox, cg.bounds[nbox], nbox < cg.H ? cg.bounds[nbox + 1] : INT_MAX);
    cg.launches = launches - launches0;
    cur = cur0; launches = launches0;
    const hipError_t e = hipStreamEndCapture(st, &cg.graph);
    if (rc == C2R_OK && e == hipSuccess && hipGetLastError() == hipSuccess && hipGraphInstantiate(&cg.exec, cg.graph, nullptr, nullptr, 0) == hipSuccess) {
        cg.gen = ctx->gen; cg.count = count; cg.n_active = n_active; cg.shape_count = shape_count; cg.acc = (const void *)k.phih;
        cg.perm = perm_ready;
        cg.passes_since_capture = 0;
        return true;
    }
    if (cg.graph) { hipGraphDestroy(cg.graph); cg.graph = nullptr; }
    cg.exec = nullptr;
    (void)hipGetLastError();
    ctx->use_graph = false;            // this runtime / stream cannot capture: launch by launch from now on
    return false;
}

int sweep_batch(Ctx *ctx, int first, int count, bool first_of_pass, double *dbg, std::vector<int> *nbox_out,
                std::vector<double> *loss_out, FusedIter *fz = nullptr)
{
    BatchSweep bs(ctx, ctx->sc[0], first, count, first_of_pass, dbg, fz);
    return bs.run(nbox_out, loss_out);
}

// One round of a pass as SEVERAL chains in flight: local sources [first, first + count) in nch contiguous shares, each with
// its own scratch and stream (SweepScratch).  Same launches, same arguments, same per-source results as one chain after
// the other; only the order in which the Gamma atomics of different sources land can differ (as it may within one launch).
//
// Two ways to drive them:
// * REPLAYED (round 6; the steady state of an outer iteration): each chain's launch sequence -- the batch upload and every
//   launch of the sub-boxes its previous pass went through, sized for the counts that pass left plus a margin -- is ONE
//   hipGraph on the chain's stream.  The host launches nch graphs and waits once per chain: no host round trip per sub-box,
//   dependent launches follow each other at the command processor's pace whatever the host is doing.  The sizes are guarded
//   on the device (k_box_decide's next_bound: a decision that keeps more sources than the next launches were sized for
//   leaves the device count at zero and reports the sub-box), and whatever the sequence did not cover -- sources still active
//   behind its last sub-box, or a halt -- the host finishes launch by launch as below.
// * LAUNCH BY LAUNCH (first pass of a source list, changing counts): this thread drives the chains in lock-step, sub-box n of
//   every chain enqueued before any count of sub-box n - kChainAhead is waited for, so the streams always hold work of every
//   chain and the GPU overlaps one chain's launch with the tail of another's.
#ifndef C2R_CHAIN_AHEAD
#define C2R_CHAIN_AHEAD 2          // (3 measured the same: profiles/r05_chains)
#endif
constexpr int kChainAhead = C2R_CHAIN_AHEAD;

// The launch sizes a replayed sequence is captured with, from the chain's last pass: sub-box nbox for the count sub-box nbox - 1
// left, plus a margin of an eighth (at least 2) so that a source that goes one sub-box further than last time does not halt the
// sequence -- a surplus source costs a few hundred workgroups that return at once.  H: the sub-boxes that pass went through.
static void bounds_from_profile(const std::vector<int> &prof, int n_active, std::vector<int> &bounds, int &H)
{
    H = 0;
    while (H + 1 < (int)prof.size() && prof[H] > 0) ++H;         // prof[k] > 0 for k < H: sub-box k + 1 had sources to trace
    bounds.assign((size_t)H + 1, 0);
    for (int nb = 1; nb <= H; ++nb) {
        const int c = prof[nb - 1];
        bounds[nb] = nb == 1 ? n_active : std::min(n_active, c + std::max(2, c / 8));
    }
}

// May the chain's pass be replayed from cg as it is?  The same batch and launch shapes, the same number of sub-boxes as the last
// pass went through, every launch at least as large as that pass needed and the whole not much larger.
static bool chain_graph_fits(const Ctx *ctx, const Ctx::ChainGraph &cg, const BatchSweep &b, const void *acc)
{
    if (!cg.exec || cg.gen != ctx->gen || cg.count != b.count || cg.n_active != b.n_active || cg.shape_count != b.shape_count || cg.acc != acc ||
        cg.perm != b.perm_ready) return false;
    std::vector<int> want; int H = 0;
    bounds_from_profile(cg.profile, b.n_active, want, H);
    if (H != cg.H) return false;
    long long have = 0, need = 0;
    for (int nb = 1; nb <= H; ++nb) {
        if (cg.bounds[nb] < cg.profile[nb - 1]) return false;
        have += cg.bounds[nb]; need += cg.profile[nb - 1];
    }
    return have <= need + need / 2 + 4LL * H;
}

static_assert(kFusedQmaxK == kFusedQmax, "the fused kernel's LDS planes are sized for kFusedQmax");
static_assert(kGateChains >= kMaxChains, "k_chain_gate holds every chain's pointers");
int run_chains(Ctx *ctx, int first, int count, bool first_of_pass, std::vector<int> *nbox_out, FusedIter *tail)
{
    const int nch = std::min(ctx->nchains, count);
    const int per = (count + nch - 1) / nch;
    std::vector<BatchSweep> ch;
    ch.reserve(nch);
    for (int c = 0, f = first; c < nch && f < first + count; ++c, f += per)
        ch.emplace_back(ctx, ctx->sc[c], f, std::min(per, first + count - f), first_of_pass && c == 0, nullptr, nullptr);
    // the pass's inputs (n_HI, the zeroed transposed accumulators: sweep_prepare on the context's stream) before any chain starts
    HIP_TRY(hipEventRecord(ctx->ev_prepared, ctx->stream));
    if (ctx->prof) prof_begin(ctx, ctx->ev_sweep, ctx->ev_sweep_used);
    std::vector<Ctx::ChainGraph *> graph_of(ch.size(), nullptr);
    for (size_t c = 0; c < ch.size(); ++c) {
        BatchSweep &b = ch[c];
        b.chained = true;
        b.stage();
        if (ctx->xcd_order > 0) b.stage_perm();       // (chains take the plane-ordered mapping only where it is forced: measured, not adopted -- DESIGN 3d)
        if (b.st != ctx->stream) HIP_TRY(hipStreamWaitEvent(b.st, ctx->ev_prepared, 0));
        b.bound = b.n_active; b.known = 0; b.cur = 0; b.next = 1;
        Ctx::ChainGraph &cg = ctx->chain_graphs[b.first];
        ++cg.passes_since_capture;
        bool replay = false;
        if (ctx->use_graph && ctx->chain_graph && b.n_active > 0 && !cg.profile.empty()) {
            replay = chain_graph_fits(ctx, cg, b, (const void *)b.k.phih);
            // (re)capture: nothing to replay yet, or the counts have settled (two passes alike), or the last capture is a while
            // ago -- a capture costs about what it saves in one pass, so counts that move every pass are driven launch by launch
            if (!replay && (!cg.exec || cg.gen != ctx->gen || cg.profile == cg.profile_prev || cg.passes_since_capture > 3))
                replay = b.capture_chain(cg);
        }
        if (replay) {
            b.sc.h_nactive[0] = 0;                                                  // (the halt slot)
            HIP_TRY(hipGraphLaunch(cg.exec, b.st));
            HIP_TRY(hipEventRecord(b.sc.ev_done, b.st));
            graph_of[c] = &cg;
            b.bound = 0;                                                            // (not driven below until the replay has been looked at)
            ++ctx->chain_replays;
        } else {
            HIP_TRY(hipMemcpyAsync(b.sc.d_batch, b.sc.h_batch, b.sc.batch_bytes, hipMemcpyHostToDevice, b.st));
            if (b.perm_ready) HIP_TRY(hipMemcpyAsync(b.sc.d_perm, b.sc.h_perm, (size_t)3 * b.cap * sizeof(int), hipMemcpyHostToDevice, b.st));
            ++ctx->chain_eager;
        }
    }
    // launch by launch, in lock-step: every chain that has sources to trace gets its next sub-box, then the counts are looked at
    auto lockstep = [&]() -> int {
        for (;;) {
            bool any = false;
            for (auto &b : ch) {
                if (b.bound <= 0 || b.next > ctx->nbox_max) continue;
                any = true;
                { const int rc = b.enqueue_box(b.next, b.bound); if (rc) return rc; }
                HIP_TRY(hipEventRecord(b.sc.ev_box[b.next], b.st));
                b.last = b.next++;
            }
            if (!any) return C2R_OK;
            // kChainAhead sub-boxes stay in flight per chain: the count after sub-box last - kChainAhead sizes (and ends) the next
            // round of launches.  (One ahead, as a single chain of many sources runs, leaves the streams empty while this thread
            // enqueues the next sub-box of every chain -- near the source a sub-box is five launches of 10 - 20 us, about what
            // enqueueing it costs; blocks of sources that retired in between return at once: at most 512 sources here.)
            for (auto &b : ch) {
                if (b.bound <= 0) continue;
                while (b.known < b.last && hipEventQuery(b.sc.ev_box[b.known + 1]) == hipSuccess) b.bound = b.sc.h_nactive[++b.known];
                const int need = b.last - kChainAhead;
                if (need > b.known) {
                    // (polling before blocking: a sub-box near the source lasts about as long as the wake-up of an event wait)
                    { const int rc = wait_polling(ctx, b.sc.ev_box[need]); if (rc) return rc; }
                    b.known = need; b.bound = b.sc.h_nactive[need];
                }
            }
        }
    };
    { const int rc = lockstep(); if (rc) return rc; }
    // Every chain replays and the caller has handed over what follows the pass (iterate_impl: the fold of the transposed rates, the
    // global pass): the chains join the context's stream, k_chain_gate says on the device whether every sequence ran to its end, the
    // totals and the tail follow GATED by it, and the host waits ONCE for the whole iteration.  A closed gate (a halt, or sources
    // still active behind a sequence's last sub-box) leaves the gated launches undone: the chains are finished launch by launch
    // below and the caller runs the tail the ordinary way.
    bool all_replay = !ch.empty();
    for (size_t c = 0; c < ch.size(); ++c) all_replay = all_replay && graph_of[c] != nullptr;
    if (tail && all_replay && !ctx->prof) {
        ChainGateArgs ga{};
        ga.nch = (int)ch.size();
        for (size_t c = 0; c < ch.size(); ++c) {
            BatchSweep &b = ch[c];
            ga.n_active[c] = b.sc.d_nactive + (graph_of[c]->H & 1); ga.halt[c] = b.sc.d_hnactive;
            if (b.st != ctx->stream) HIP_TRY(hipStreamWaitEvent(ctx->stream, b.sc.ev_done, 0));
        }
        hipLaunchKernelGGL(k_chain_gate, dim3(1), dim3(64), 0, ctx->stream, ga, ctx->d_gate);
        for (auto &b : ch) { const int rc = b.enqueue_totals(nbox_out, nullptr, ctx->stream, b.first_of_pass, ctx->d_gate); if (rc) return rc; }
        { const int rc = tail->post(ctx->d_gate); if (rc) return rc; }
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (__atomic_load_n(&ctx->h_sc->seq, __ATOMIC_ACQUIRE) == ctx->seq_seen + 1) {       // the gate was open: the whole iteration has run
            ctx->seq_seen += 1; tail->tail_done = true; ++ctx->chain_tails;
        }
    }
    // the replayed chains: one wait each; where the sequence ended with sources still active, or halted, the rest launch by launch
    bool more = false;
    for (size_t c = 0; c < ch.size(); ++c) {
        if (!graph_of[c]) continue;
        BatchSweep &b = ch[c];
        b.launches += graph_of[c]->launches;
        if (tail && tail->tail_done) { b.next = graph_of[c]->H + 1; continue; }
        { const int rc = wait_polling(ctx, b.sc.ev_done); if (rc) return rc; }
        const int halt = b.sc.h_nactive[0];
        const int at = halt > 0 ? halt : graph_of[c]->H;           // the last sub-box whose decision stands
        b.known = at; b.cur = at & 1; b.next = at + 1; b.bound = b.sc.h_nactive[at];
        if (halt > 0) {
            // the decision of sub-box `halt` kept more sources than the next launches were sized for: it left the device count at
            // zero (the rest of the sequence did nothing) and the true count in the host's slot -- put it back, go on from there
            HIP_TRY(hipMemcpyAsync(b.sc.d_nactive + b.cur, &b.sc.h_nactive[at], sizeof(int), hipMemcpyHostToDevice, b.st));
            ++ctx->chain_halts;
        }
        if (b.bound > 0 && b.next <= ctx->nbox_max) more = true;
    }
    if (more) { const int rc = lockstep(); if (rc) return rc; }
    // the chains join the context's stream; their totals follow in chain order = source order
    int launches = 0;
    for (auto &b : ch) launches += b.launches;
    if (!(tail && tail->tail_done)) {
        for (auto &b : ch) {
            if (b.st != ctx->stream) {
                HIP_TRY(hipEventRecord(b.sc.ev_done, b.st));
                HIP_TRY(hipStreamWaitEvent(ctx->stream, b.sc.ev_done, 0));
            }
        }
        if (ctx->prof) { prof_end(ctx, ctx->ev_sweep, ctx->ev_sweep_used); ctx->ev_sweep_cnt.push_back(launches); }
        for (auto &b : ch) { const int rc = b.enqueue_totals(nbox_out ? nbox_out : nullptr, nullptr, ctx->stream, b.first_of_pass); if (rc) return rc; }
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    // what the next pass's sequence is sized by: the counts this pass left, sub-box by sub-box (every slot up to the last
    // sub-box enqueued has been written; beyond the first zero nothing is looked at)
    for (auto &b : ch) {
        Ctx::ChainGraph &cg = ctx->chain_graphs[b.first];
        cg.profile_prev.swap(cg.profile);
        cg.profile.assign(1, b.n_active);
        for (int nb = 1; nb < b.next && cg.profile.back() > 0; ++nb) cg.profile.push_back(b.sc.h_nactive[nb]);
        if (cg.profile.back() > 0) cg.profile.push_back(0);       // (ended at the last sub-box there is)
    }
    if (nbox_out) {
        nbox_out->clear();
        for (auto &b : ch) nbox_out->insert(nbox_out->end(), b.h_fnb, b.h_fnb + b.count);
    }
    return C2R_OK;
}
}  // namespace

// +-x faces read (x,y)-transposed replicas so that their waves, which run along y, touch unit
// stride: refresh the replicas before a pass, fold their Gamma back after it.
// zero_rates: also set_rates_to_zero (evolve.F90:430-440) -- and every clearing inside the one kernel instead of memsets
// (the fused iteration, where a launch more or less is what counts)
int sweep_prepare(Ctx *ctx, bool zero_rates, bool copy_batch)
{
    const c2r_params &p = ctx->prm;
    const dim3 g((p.mesh[0] + 31) / 32, (p.mesh[1] + 31) / 32, p.mesh[2]);
    ZeroGrids z{};
    if (zero_rates) {
        z.g[0] = (double *)ctx->grid[4]; z.g[1] = ctx->d_phih_T;
        if (ctx->thermal) { z.g[2] = (double *)ctx->grid[5]; z.g[3] = ctx->d_heat_T; }
    }
    // copy_batch (fused iteration): the batch's pristine state block (d_batch_init, kept current by sweep_batch) over the working one
    WordCopy wc{};
    if (copy_batch) { const SweepScratch &sc = ctx->sc[0]; wc.src = (const unsigned *)sc.d_batch_init; wc.dst = (unsigned *)sc.d_batch; wc.n = (unsigned)(sc.batch_bytes / 4); }
    hipLaunchKernelGGL(k_prepare_nhi, g, dim3(256), 0, ctx->stream, p.mesh[0], p.mesh[1], p.mesh[2], p.epsilon,
                       (const float *)ctx->grid[0], (const double *)ctx->grid[2], ctx->d_nhi, ctx->d_nhi_T, z, wc,
                       ctx->allfrac ? (const double *)ctx->grid[8] : nullptr);
    HIP_TRY(hipGetLastError());
    if (!zero_rates) {
        HIP_TRY(hipMemsetAsync(ctx->d_phih_T, 0, grid_bytes(ctx, 4), ctx->stream));
        if (ctx->thermal) HIP_TRY(hipMemsetAsync(ctx->d_heat_T, 0, grid_bytes(ctx, 5), ctx->stream));
    }
    return C2R_OK;
}

int sweep_finish(Ctx *ctx, const int *gate)
{
    const c2r_params &p = ctx->prm;
    // phih_T is [k][i][j]: transposing it back swaps the roles of the two mesh extents
    const dim3 g((p.mesh[1] + 31) / 32, (p.mesh[0] + 31) / 32, p.mesh[2]);
    hipLaunchKernelGGL((k_transpose_xy<double, true>), g, dim3(256), 0, ctx->stream, p.mesh[1], p.mesh[0], p.mesh[2],
                       (const double *)(ctx->acc_phih_T ? ctx->acc_phih_T : ctx->d_phih_T),
                       ctx->acc_phih ? ctx->acc_phih : (double *)ctx->grid[4], gate);
    if (ctx->thermal)
        hipLaunchKernelGGL((k_transpose_xy<double, true>), g, dim3(256), 0, ctx->stream, p.mesh[1], p.mesh[0], p.mesh[2],
                           (const double *)ctx->d_heat_T, (double *)ctx->grid[5], gate);
    HIP_TRY(hipGetLastError());
    return C2R_OK;
}

long long visited_for_nbox(const Ctx *ctx, int nbox)
{
    if (nbox <= 0) return 0;
    long long v = 1;
    for (int d = 0; d < 3; ++d) {
        const int r = std::min(ctx->prm.subboxsize * nbox, ctx->hr[d]), l = std::min(ctx->prm.subboxsize * nbox, ctx->hl[d]);
        v *= (long long)(r + l + 1);
    }
    return v;
}


int upload_lls_grid(Ctx *ctx, const float *lls_grid)
{
    const c2r_params &p = ctx->prm;
    if (!ctx->d_lls) { HIP_TRY(hipMalloc(&ctx->d_lls, grid_bytes(ctx, 0))); HIP_TRY(hipMalloc(&ctx->d_lls_T, grid_bytes(ctx, 0))); }
    HIP_TRY(hipMemcpyAsync(ctx->d_lls, lls_grid, grid_bytes(ctx, 0), hipMemcpyHostToDevice, ctx->stream));
    const dim3 g((p.mesh[0] + 31) / 32, (p.mesh[1] + 31) / 32, p.mesh[2]);
    hipLaunchKernelGGL((k_transpose_xy<float, false>), g, dim3(256), 0, ctx->stream, p.mesh[0], p.mesh[1], p.mesh[2],
                       (const float *)ctx->d_lls, ctx->d_lls_T);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return C2R_OK;
}

// do_grid over this rank's sources.  fz (c2r_iterate, one small batch): the batch's graph also carries what precedes and
// follows the pass (sweep_batch); sweep_prepare / sweep_finish are then fz->pre / fz->post, not called here.
// no_wait (iterate_impl): return with sweep_finish enqueued and not waited for -- the caller enqueues the global pass behind
// it, waits once and reads the totals itself (they are in h_sc after that wait).
namespace {
// ---- the all-reduce of the rates overlapped with the sweep (c2r_set_exchange_overlap) ------------------------------------
// (every rank must come to the same answer: nothing below depends on the rank's own share)
bool exchange_overlap_applies(Ctx *ctx, const FusedIter *fz)
{
    if (!ctx->exchange_overlap || ctx->nranks <= 1 || !ctx->ar || (ctx->rs && ctx->ag) || fz || ctx->prm.deterministic_rates ||
        ctx->thermal || ctx->nsrc / ctx->nranks < ctx->overlap_min_sources || !ctx->sparse_valid /* = over rates the library zeroed */) return false;
    // the previous pass's sub-boxes (every rank knows them: gather_nbox_all) say whether this pass's rates will travel packed
    if (ctx->sparse_exchange && (int)ctx->nbox_all.size() == ctx->nsrc) {
        double total = 0.0;
        for (int i = 0; i < ctx->nsrc; ++i) total += (double)visited_for_nbox(ctx, ctx->nbox_all[i]);
        if (total <= ctx->sparse_fraction * (double)ctx->ncell) return false;
    } else if (ctx->sparse_exchange) return false;          // nothing known yet (first pass of a list): the plain path learns it
    return true;
}

int overlap_begin(Ctx *ctx)
{
    if (!ctx->d_phih2) {
        HIP_TRY(hipMalloc(&ctx->d_phih2, grid_bytes(ctx, 4)));
        HIP_TRY(hipMalloc(&ctx->d_phih2_T, grid_bytes(ctx, 4)));
        HIP_TRY(hipStreamCreateWithFlags(&ctx->xstream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&ctx->ev_half, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ctx->ev_xdone, hipEventDisableTiming));
    }
    HIP_TRY(hipMemsetAsync(ctx->d_phih2, 0, grid_bytes(ctx, 4), ctx->stream));
    HIP_TRY(hipMemsetAsync(ctx->d_phih2_T, 0, grid_bytes(ctx, 4), ctx->stream));
    return C2R_OK;
}

// the half's rates are complete in their accumulator (sweep_finish has folded the transposed part back, on the context's
// stream): their all-reduce goes to the exchange stream -- behind the other half's, in rank-independent order
int overlap_exchange_half(Ctx *ctx, int half)
{
    HIP_TRY(hipEventRecord(ctx->ev_half, ctx->stream));
    HIP_TRY(hipStreamWaitEvent(ctx->xstream, ctx->ev_half, 0));
    double *buf = half == 0 ? (double *)ctx->grid[4] : ctx->d_phih2;
    if (ctx->ar(ctx->ar_user, buf, ctx->ncell, (void *)ctx->xstream) != 0) FAIL(C2R_ECALLBACK, "all-reduce callback failed");
    return C2R_OK;
}

int overlap_end(Ctx *ctx)
{
    HIP_TRY(hipEventRecord(ctx->ev_xdone, ctx->xstream));
    HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_xdone, 0));
    hipLaunchKernelGGL(k_add_grid, dim3(kSumBlocks), dim3(256), 0, ctx->stream, ctx->ncell, (const double *)ctx->d_phih2, (double *)ctx->grid[4]);
    HIP_TRY(hipGetLastError());
    ctx->rates_reduced_pass = ctx->pass_id;
    ++ctx->xchg_calls; ++ctx->xchg_overlapped;
    ctx->xchg_bytes_last = 2 * (long long)ctx->ncell * (long long)sizeof(double);
    ctx->xchg_bytes_total += ctx->xchg_bytes_last;
    return C2R_OK;
}
}  // namespace

int pass_sources_impl(Ctx *ctx, FusedIter *fz, double *photon_loss, int64_t *sum_nbox, int64_t *visited, bool no_wait, FusedIter *chain_tail)
{
    int rc;
    if ((rc = sync_step(ctx))) return rc;
    // do_grid_master / do_grid_slave (master_slave.F90:124-330; c2r_set_source_queue): this rank's sources of the pass are not known
    // in advance -- it asks the host's queue for `queue_chunk` more whenever it has swept what it had
    const bool queued = ctx->queue_next != nullptr && fz == nullptr;
    if (queued) { ctx->share.clear(); ctx->explicit_share = true; ctx->auto_share = true; }
    else balance_before_pass(ctx);
    int nloc = queued ? 0 : n_local_sources(ctx);
    long long vis = 0;
    // the sparse exchange (c2r_allreduce_rates) is only right for ONE pass over rates the library itself had zeroed: everything
    // outside this pass's sub-boxes is then zero on every rank.  fz: the fused iteration zeroes them itself (fz->pre)
    ++ctx->pass_id; ctx->sparse_valid = ctx->rates_clean || fz != nullptr; ctx->rates_clean = false;
    ctx->last_nbox.clear();
    ctx->h_sc->photon_loss = 0.0; ctx->h_sc->sum_nbox = 0;      // (the stream is idle between calls)
    const bool overlap = !queued && exchange_overlap_applies(ctx, fz);
    if (queued) {
        // rounds of one chunk each: ask, sweep, ask again (the chunk's launches end with a host wait, as every round does)
        bool prepared = false;
        std::vector<int> nb;
        for (;;) {
            int32_t qf = 0, qc = 0;
            if (ctx->queue_next(ctx->queue_user, (int64_t)ctx->pass_id, ctx->queue_chunk, &qf, &qc) != 0) FAIL(C2R_ECALLBACK, "source queue callback failed");
            if (qc <= 0) break;
            if (qf < 0 || qc > ctx->queue_chunk || qf + qc > ctx->nsrc) FAIL(C2R_ECALLBACK, "source queue handed out sources beyond the list");
            const int first = (int)ctx->share.size();
            for (int i = 0; i < qc; ++i) ctx->share.push_back(qf + i);
            if ((rc = ensure_sweep_scratch(ctx, ctx->queue_chunk))) return rc;
            if (!prepared) { if ((rc = sweep_prepare(ctx))) return rc; prepared = true; }
            for (int f = first, count = 0; f < first + qc; f += count) {
                if (ctx->nchains > 1 && first + qc - f >= 2 * kFewSources) {
                    count = std::min(ctx->batch_cap, first + qc - f);
                    rc = run_chains(ctx, f, count, f == 0, &nb, nullptr);
                } else {
                    count = std::min(ctx->sc[0].cap, first + qc - f);
                    rc = sweep_batch(ctx, f, count, f == 0, nullptr, &nb, nullptr, nullptr);
                }
                if (rc) return rc;
                for (int v : nb) { vis += visited_for_nbox(ctx, v); ctx->last_nbox.push_back(v); }
            }
        }
        if (prepared && (rc = sweep_finish(ctx))) return rc;
        ctx->box_hint = 0;
        for (int v : ctx->last_nbox) ctx->box_hint = std::max(ctx->box_hint, v);
        nloc = 0;                                   // (everything below that sweeps is skipped)
    }
    if (nloc == 0 && overlap) {
        // a rank without sources (more ranks than sources in its share) still takes part in both exchanges
        if ((rc = overlap_begin(ctx)) || (rc = overlap_exchange_half(ctx, 0)) || (rc = overlap_exchange_half(ctx, 1)) || (rc = overlap_end(ctx))) return rc;
    }
    if (nloc > 0) {
        rc = ensure_sweep_scratch(ctx, nloc);
        if (rc) return rc;
        if (fz && nloc > ctx->batch_cap) FAIL(C2R_ESTATE, "fused iteration needs the sources in one batch");
        if (!fz && (rc = sweep_prepare(ctx))) return rc;
        // Several ranks, the whole grid to exchange (evolve.F90:599): the pass runs as TWO halves of this rank's sources, each
        // into its own pair of accumulators, and the all-reduce of the first half's rates travels (on a second stream) while
        // the second half is swept; the sum of the two reduced halves is what the plain pass + c2r_allreduce_rates leave
        // (re-associated: (a1+a2+..) + (b1+b2+..) over ranks and halves, 1e-16 relative).  c2r_allreduce_rates then has
        // nothing left to do for this pass (rates_reduced).  Not where the rates travel packed (cold steps: far fewer bytes
        // than half a grid), with slab chemistry, ordered rates, heating rates, or few sources.
        const int n_first = overlap ? (nloc + 1) / 2 : nloc;
        if (overlap && (rc = overlap_begin(ctx))) return rc;
        std::vector<int> nb;
        for (int half = 0; half < (overlap ? 2 : 1); ++half) {
            const int h0 = half == 0 ? 0 : n_first, h1 = half == 0 ? n_first : nloc;
            if (half == 1) { ctx->acc_phih = ctx->d_phih2; ctx->acc_phih_T = ctx->d_phih2_T; }
            for (int first = h0, count = 0; first < h1; first += count) {
                // several chains in flight where the scratch was laid out for it (ensure_sweep_scratch: 64 - 768 sources per round)
                if (ctx->nchains > 1 && !fz && h1 - first >= 2 * kFewSources) {
                    count = std::min(ctx->batch_cap, h1 - first);
                    // (chain_tail: only where this round is the whole pass -- every local source, no overlapped halves)
                    rc = run_chains(ctx, first, count, first == 0, &nb, (!overlap && first == 0 && count == nloc) ? chain_tail : nullptr);
                } else {
                    count = std::min(ctx->sc[0].cap, h1 - first);
                    rc = sweep_batch(ctx, first, count, first == 0, nullptr, &nb, nullptr, fz);
                }
                if (rc) { ctx->acc_phih = ctx->acc_phih_T = nullptr; return rc; }
                for (int v : nb) { vis += visited_for_nbox(ctx, v); ctx->last_nbox.push_back(v); }
            }
            if (!fz && !(chain_tail && chain_tail->tail_done) && (rc = sweep_finish(ctx))) { ctx->acc_phih = ctx->acc_phih_T = nullptr; return rc; }
            if (overlap && (rc = overlap_exchange_half(ctx, half))) { ctx->acc_phih = ctx->acc_phih_T = nullptr; return rc; }
        }
        ctx->acc_phih = ctx->acc_phih_T = nullptr;
        if (overlap && (rc = overlap_end(ctx))) return rc;
        ctx->box_hint = 0;
        for (int v : ctx->last_nbox) ctx->box_hint = std::max(ctx->box_hint, v);
    } else if (fz && (rc = fz->pre())) return rc;
    if (visited) *visited = vis;
    if (no_wait && !queued) return C2R_OK;
    if (!fz) HIP_TRY(hipStreamSynchronize(ctx->stream));        // k_batch_totals stored the totals in h_sc
    prof_collect(ctx);
    if (!queued && (rc = balance_after_pass(ctx))) return rc;
    if (photon_loss) *photon_loss = ctx->h_sc->photon_loss;
    if (sum_nbox) *sum_nbox = ctx->h_sc->sum_nbox;
    if (visited) *visited = vis;
    return C2R_OK;
}

}  // namespace c2r

using namespace c2r;

extern "C" {

int c2r_pass_sources(c2r_ctx *c, double *photon_loss, int64_t *sum_nbox, int64_t *visited)
{
    if (!c) return C2R_EINVAL;
    Ctx *ctx = C(c);
    int rc = check_ready(ctx);
    if (rc) return rc;
    return pass_sources_impl(ctx, nullptr, photon_loss, sum_nbox, visited);
}

int c2r_do_source(c2r_ctx *c, int32_t ns, double *cd_host, double *loss, int32_t *nbox, int64_t *visited)
{
    if (!c) return C2R_EINVAL;
    Ctx *ctx = C(c);
    int rc = check_ready(ctx);
    if (rc) return rc;
    if (ns < 1 || ns > ctx->nsrc) FAIL(C2R_EINVAL, "source number out of range");
    rc = ensure_sweep_scratch(ctx, 1);
    if (rc) return rc;
    if ((rc = sync_step(ctx))) return rc;
    ctx->sparse_valid = false; ctx->rates_clean = false;     // (one source, addressed directly: the per-rank sub-box list no longer describes phih_grid)
    double *dbg = nullptr;
    if (cd_host) {
        if (!ctx->d_dbg) HIP_TRY(hipMalloc(&ctx->d_dbg, ctx->ncell * sizeof(double)));
        HIP_TRY(hipMemsetAsync(ctx->d_dbg, 0, ctx->ncell * sizeof(double), ctx->stream));   // evolve_source.F90:91
        dbg = ctx->d_dbg;
    }
    // address the source directly, whatever the rank layout
    const int sr = ctx->rank, sn = ctx->nranks;
    const bool se = ctx->explicit_share;
    ctx->rank = 0; ctx->nranks = 1; ctx->explicit_share = false;
    std::vector<int> nb; std::vector<double> fl;
    rc = sweep_prepare(ctx);
    if (!rc) rc = sweep_batch(ctx, ns - 1, 1, true, dbg, &nb, &fl);
    if (!rc) rc = sweep_finish(ctx);
    ctx->rank = sr; ctx->nranks = sn; ctx->explicit_share = se;
    if (rc) return rc;
    prof_collect(ctx);
    if (cd_host) {
        HIP_TRY(hipMemcpyAsync(cd_host, ctx->d_dbg, ctx->ncell * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    if (loss) *loss = fl[0];
    if (nbox) *nbox = nb[0];
    if (visited) *visited = visited_for_nbox(ctx, nb[0]);
    return C2R_OK;
}

// evolve0D(dt,rtpos,ns,niter) (evolve_point.F90:83-299) for ONE cell on the caller's arrays: the reference's per-cell call
// surface (its sweep routines call it cell by cell, evolve_source.F90:227-591).  A launch and a few small copies per cell --
// slow by construction; c2r_do_source / c2r_pass_sources are the product path.
int c2r_evolve0d_host(c2r_ctx *c, int32_t ns, const int32_t rtpos[3], const int32_t last_l[3], const int32_t last_r[3],
                      const float *ndens, const double *xh_av, double *coldensh_out, double *phih_grid, double *phiheat_grid,
                      double *photon_loss_src)
{
    if (!c || !rtpos || !last_l || !last_r || !ndens || !xh_av || !coldensh_out || !phih_grid) return C2R_EINVAL;
    Ctx *ctx = C(c);
    int rc = check_ready(ctx);
    if (rc) return rc;
    if (ns < 1 || ns > ctx->nsrc) FAIL(C2R_EINVAL, "source number out of range");
    if (ctx->thermal && !phiheat_grid) FAIL(C2R_EINVAL, "non-isothermal run: evolve0D needs phiheat_grid");
    const c2r_params &p = ctx->prm;
    // :122 pos = modulo(rtpos-1,mesh)+1; :125 only cells not yet done
    int pos[3];
    for (int d = 0; d < 3; ++d) { const int m = (rtpos[d] - 1) % p.mesh[d]; pos[d] = m < 0 ? m + p.mesh[d] : m; }
    const size_t idx = (size_t)pos[0] + (size_t)p.mesh[0] * ((size_t)pos[1] + (size_t)p.mesh[1] * (size_t)pos[2]);
    if (coldensh_out[idx] != 0.0) return C2R_OK;
    // the source, and the cell's place in its sweep: checked before anything is enqueued or overwritten
    const int32_t *sp = &ctx->srcpos[3 * (size_t)(ns - 1)];
    int spw[3], del[3];
    for (int d = 0; d < 3; ++d) { const int m = (sp[d] - 1) % p.mesh[d]; spw[d] = m < 0 ? m + p.mesh[d] : m; del[d] = rtpos[d] - sp[d]; }
    // cinterp's branch (column_density.f90:108,173,226: z over y over x) as face / plane coordinates / shell
    const int ad[3] = {abs(del[0]), abs(del[1]), abs(del[2])};
    if (std::max(ad[0], std::max(ad[1], ad[2])) > ctx->Qmax) FAIL(C2R_EINVAL, "evolve0D: the cell lies beyond the trace limit of its source");
    if ((rc = ensure_sweep_scratch(ctx, 1))) return rc;
    if ((rc = sync_step(ctx))) return rc;
    ctx->sparse_valid = false; ctx->rates_clean = false;
    hipStream_t st = ctx->stream;
    // the source in slot 0 of the batch arrays; n_HI of the cell (evolve_point.F90:137-146) where the kernels read it.  The
    // small inputs travel through the context's pinned staging block (true async copies; the call ends with a stream wait)
    const double nflux = ctx->nflux[ns - 1];
    // (-DALLFRAC drivers: xh_av is the (mesh,0:1) array -- the stored neutral fraction comes first, evolve_point.F90:131-132)
    const double xav1 = std::max(xh_av[ctx->allfrac ? ctx->ncell + idx : idx], p.epsilon);
    const double xav0 = ctx->allfrac ? std::max(xh_av[idx], p.epsilon) : std::max(1.0 - xav1, p.epsilon);
    const double nhi = xav0 * (double)ndens[idx];
    const size_t idt = (size_t)pos[1] + (size_t)p.mesh[1] * ((size_t)pos[0] + (size_t)p.mesh[0] * (size_t)pos[2]);
    {
        const SweepScratch &sc = ctx->sc[0];
        double *hd = reinterpret_cast<double *>(sc.h_batch);            // >= 4 doubles + 11 ints for a batch of one
        int *hi = reinterpret_cast<int *>(hd + 4);
        hd[0] = nflux; hd[1] = nhi; hd[2] = (ctx->xray && ns - 1 < (int)ctx->nflux_x.size()) ? ctx->nflux_x[ns - 1] : 0.0;
        for (int d = 0; d < 3; ++d) { hi[d] = sp[d]; hi[3 + d] = spw[d]; }
        HIP_TRY(hipMemcpyAsync(sc.d_srcpos_b, hi, 3 * sizeof(int), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(sc.d_srcw_b, hi + 3, 3 * sizeof(int), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(sc.d_nflux_b, hd, sizeof(double), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(sc.d_nflux_x, hd + 2, sizeof(double), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(ctx->d_nhi + idx, hd + 1, sizeof(double), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(ctx->d_nhi_T + idt, hd + 1, sizeof(double), hipMemcpyHostToDevice, st));
    }
    const bool is_source = ad[0] == 0 && ad[1] == 0 && ad[2] == 0;
    int axis, a, b;
    if (ad[2] >= ad[1] && ad[2] >= ad[0]) { axis = 2; a = del[0]; b = del[1]; }
    else if (ad[1] >= ad[0]) { axis = 1; a = del[0]; b = del[2]; }
    else { axis = 0; a = del[1]; b = del[2]; }
    const int pd = del[axis], q = abs(pd), face = (2 - axis) * 2 + (pd < 0 ? 1 : 0);
    const int ua = axis == 0 ? 1 : 0, va = axis == 2 ? 1 : 2;
    double cv[4] = {0.0, 0.0, 0.0, 0.0};
    ShellArgs sa{};
    if (!is_source) {
        // the four upstream cells (column_density.f90:112-131 and the y / x counterparts): one step toward the source along
        // the face's axis, 0 or 1 along the others; sign(1,0) = +1
        const int sga = a < 0 ? -1 : 1, sgb = b < 0 ? -1 : 1, sgp = pd < 0 ? -1 : 1;
        for (int k = 0; k < 4; ++k) {
            int r[3];
            r[axis] = rtpos[axis] - sgp;
            r[ua] = rtpos[ua] - ((k & 1) ? 0 : sga);          // k = 0: (am,bm)  1: (a,bm)  2: (am,b)  3: (a,b)
            r[va] = rtpos[va] - ((k & 2) ? 0 : sgb);
            size_t id = 0, mul = 1;
            for (int d = 0; d < 3; ++d) { int m = (r[d] - 1) % p.mesh[d]; if (m < 0) m += p.mesh[d]; id += mul * (size_t)m; mul *= (size_t)p.mesh[d]; }
            cv[k] = coldensh_out[id];
        }
        sa.q = q;
        sa.alam = (double)((float)(q - 1) + 0.5f) / (double)(float)q;
        sa.dp2 = (double)q * (double)q; sa.inv_dp2 = 1.0 / sa.dp2; sa.inv_q = 1.0 / (double)q;
    }
    // :288-293 the cell lies on the surface of the current sub-box
    bool on_surface = false;
    for (int d = 0; d < 3; ++d) on_surface = on_surface || rtpos[d] == last_l[d] || rtpos[d] == last_r[d];
    KParams k = make_kparams(ctx, ctx->sc[0]);
    double *d_out = ctx->d_sum_out;                              // 4 doubles of device scratch
#define C2R_LAUNCH_CELL_H(L, H) hipLaunchKernelGGL((k_evolve0d_cell<L, H>), dim3(1), dim3(64), 0, st, k, sa, face, a, b, is_source ? 1 : 0, on_surface ? 1 : 0, cv[0], cv[1], cv[2], cv[3], d_out)
#define C2R_LAUNCH_CELL(L) do { if (ctx->thermal && ctx->xray) C2R_LAUNCH_CELL_H(L, 3); else if (ctx->thermal) C2R_LAUNCH_CELL_H(L, 1); else if (ctx->xray) C2R_LAUNCH_CELL_H(L, 2); else C2R_LAUNCH_CELL_H(L, 0); } while (0)
    switch (ctx->lls_type) { case 1: C2R_LAUNCH_CELL(1); break; case 2: C2R_LAUNCH_CELL(2); break; default: C2R_LAUNCH_CELL(3); break; }
#undef C2R_LAUNCH_CELL
#undef C2R_LAUNCH_CELL_H
    HIP_TRY(hipGetLastError());
    double *out = ctx->h_sc->four;                               // pinned
    HIP_TRY(hipMemcpyAsync(out, d_out, 4 * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    coldensh_out[idx] = out[0];                                  // :247
    phih_grid[idx] = phih_grid[idx] + out[1];                    // :283
    if (ctx->thermal) phiheat_grid[idx] = phiheat_grid[idx] + out[2];     // :285-286
    if (photon_loss_src && on_surface) *photon_loss_src = *photon_loss_src + out[3];    // :290-293
    return C2R_OK;
}

}  // extern "C"
