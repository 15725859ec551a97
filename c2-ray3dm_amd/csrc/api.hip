// Life cycle of the context, the setters of everything the driver owns, array transfer, and the host-array forms of the
// piecewise entries (do_source / do_grid on the caller's arrays).
#include "ctx.hpp"
#include "kernels_misc.hpp"

namespace c2r {

int copy_in(Ctx *ctx, int which, const void *host)
{
    if (ctx->allfrac && which >= 1 && which <= 3) {
        const double *h = static_cast<const double *>(host);
        HIP_TRY(hipMemcpyAsync(ctx->grid[which + 6], h, grid_bytes(ctx, which), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(ctx->grid[which], h + ctx->ncell, grid_bytes(ctx, which), hipMemcpyHostToDevice, ctx->stream));
        return C2R_OK;
    }
    HIP_TRY(hipMemcpyAsync(ctx->grid[which], host, grid_bytes(ctx, which), hipMemcpyHostToDevice, ctx->stream));
    return C2R_OK;
}

int copy_out(Ctx *ctx, int which, void *host)
{
    if (ctx->allfrac && which >= 1 && which <= 3) {
        double *h = static_cast<double *>(host);
        HIP_TRY(hipMemcpyAsync(h, ctx->grid[which + 6], grid_bytes(ctx, which), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipMemcpyAsync(h + ctx->ncell, ctx->grid[which], grid_bytes(ctx, which), hipMemcpyDeviceToHost, ctx->stream));
        return C2R_OK;
    }
    HIP_TRY(hipMemcpyAsync(host, ctx->grid[which], grid_bytes(ctx, which), hipMemcpyDeviceToHost, ctx->stream));
    return C2R_OK;
}

int check_ready(Ctx *ctx)
{
    // the context's allocations and launches belong to its device, whatever the caller made current since
    HIP_TRY(hipSetDevice(ctx->prm.device));
    if (!ctx->have_tables) FAIL(C2R_ESTATE, "c2r_set_tables has not been called");
    if (!ctx->have_step) FAIL(C2R_ESTATE, "c2r_set_step has not been called");
    if (ctx->thermal && ctx->xray && !ctx->have_xheat)
        FAIL(C2R_ESTATE, "non-isothermal run with the X-ray source type: c2r_set_xray_heat_tables has not been called (heat_lookuptable's \"P\" tables)");
    return C2R_OK;
}

}  // namespace c2r

using namespace c2r;

extern "C" {

int c2r_default_params(c2r_params *p)
{
    if (!p) return C2R_EINVAL;
    memset(p, 0, sizeof *p);
    p->mesh[0] = p->mesh[1] = p->mesh[2] = 0;
    p->device = 0;
    p->subboxsize = C2R_SUBBOXSIZE; p->max_subbox = C2R_MAX_SUBBOX; p->numtau = C2R_NUMTAU;
    p->max_outer_iter = C2R_MAX_OUTER_ITER; p->max_chem_iter = C2R_MAX_CHEM_ITER;
    p->epsilon = C2R_EPSILON; p->convergence_fraction = C2R_CONVERGENCE_FRACTION;
    p->minimum_fractional_change = C2R_MIN_FRACTIONAL_CHANGE;
    p->minimum_fraction_of_atoms = C2R_MIN_FRACTION_OF_ATOMS;
    p->loss_fraction = C2R_LOSS_FRACTION; p->max_coldensh = C2R_MAX_COLDENSH;
    p->tau_photo_limit = C2R_TAU_PHOTO_LIMIT; p->sigma_HI = C2R_SIGMA_HI;
    p->minlogtau = C2R_MINLOGTAU; p->dlogtau = C2R_DLOGTAU; p->weight_floor = C2R_WEIGHT_FLOOR;
    p->sqrt2 = C2R_SQRT2; p->sqrt3 = C2R_SQRT3; p->pi = C2R_PI; p->abu_c = C2R_ABU_C;
    p->bh00 = C2R_BH00; p->albpow = C2R_ALBPOW; p->colh0 = C2R_COLH0; p->temph0 = C2R_TEMPH0;
    p->S_star = C2R_S_STAR;
    p->sweep_mode = C2R_SWEEP_FAST;             // (the opt-in C2R_SWEEP_EXACT: column densities bit-identical to the Fortran)
    p->scratch_bytes = 0;
    return C2R_OK;
}

int c2r_create(c2r_ctx **out, const c2r_params *p)
{
    if (!out || !p) return C2R_EINVAL;
    *out = nullptr;
    if (p->mesh[0] < 1 || p->mesh[1] < 1 || p->mesh[2] < 1 || p->numtau < 1 || p->subboxsize < 1) return C2R_EINVAL;
    Ctx *ctx = new Ctx();
    ctx->prm = *p;
    if (p->sweep_mode != C2R_SWEEP_EXACT && p->sweep_mode != C2R_SWEEP_FAST) { delete ctx; return C2R_EINVAL; }
    ctx->fast = p->sweep_mode == C2R_SWEEP_FAST;          // the caller's choice only: no environment override
    *out = reinterpret_cast<c2r_ctx *>(ctx);     // returned even on failure so c2r_last_error works
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev < 1) FAIL(C2R_ESTATE, "no HIP device: the c2ray_hip path needs a GPU (there is no CPU fallback)");
    if (p->device < 0) {
        // C2R_DEVICE_AUTO: one process per GPU -- this process's local rank as its launcher exports it
        // (the MPI builds of the driver, mpi.F90:83-160, know only the global rank), modulo the visible devices
        static const char *const names[] = {"C2R_DEVICE", "LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", "MV2_COMM_WORLD_LOCAL_RANK",
                                            "MPI_LOCALRANKID", "PMI_LOCAL_RANK", "SLURM_LOCALID"};
        int dev = 0;
        const char *used = nullptr;
        for (const char *nm : names)
            if (const char *e = getenv(nm)) { dev = atoi(e); used = nm; break; }
        ctx->prm.device = ((dev % ndev) + ndev) % ndev;
        ctx->device_auto = true; ctx->device_var = used ? used : "";
        char b[256];
        if (used) snprintf(b, sizeof b, "device %d of %d visible (C2R_DEVICE_AUTO: %s=%d)", ctx->prm.device, ndev, used, dev);
        else snprintf(b, sizeof b, "device 0 of %d visible (C2R_DEVICE_AUTO: no local-rank variable is set)", ndev);
        ctx->info_device = b;
    } else if (p->device >= ndev) FAIL(C2R_EINVAL, "device ordinal beyond the visible HIP devices");
    else { char b[96]; snprintf(b, sizeof b, "device %d of %d visible (explicit)", p->device, ndev); ctx->info_device = b; }
    HIP_TRY(hipSetDevice(ctx->prm.device));
    HIP_TRY(hipStreamCreate(&ctx->stream));
    ctx->own_stream = true;
    ctx->ncell = (size_t)p->mesh[0] * p->mesh[1] * p->mesh[2];
    ctx->stream_hint = ctx->ncell * sizeof(double) >= ((size_t)64 << 20);      // 8 x 4 MB of L2; neutral at 128^3, +2.8 % at 256^3
    // The sweep addresses cells through buffer descriptors with 32-bit BYTE offsets (cell id * 8 and a
    // num_records of ncell * 8, kernels.hpp cell_state / shell_rows_fast): ncell * 8 must stay below 2^32,
    // i.e. ncell < 2^29 (a cubic mesh up to 812^3).  Checked before anything is allocated.
    if (ctx->ncell >= (1ULL << 29) || p->mesh[0] >= (1 << 23) || p->mesh[1] >= (1 << 23) || p->mesh[2] >= (1 << 23) ||
        (uint64_t)p->mesh[1] * p->mesh[2] >= (1ULL << 24) || (uint64_t)p->mesh[0] * p->mesh[2] >= (1ULL << 24) ||
        (uint64_t)p->mesh[0] * p->mesh[1] >= (1ULL << 24))
        FAIL(C2R_EINVAL, "mesh too large: the sweep's 32-bit byte offsets need mesh(1)*mesh(2)*mesh(3) < 2^29 cells "
                         "(812^3) and every pair product < 2^24");
    for (int w = 0; w < 5; ++w) { HIP_TRY(hipMalloc(&ctx->grid[w], grid_bytes(ctx, w))); ctx->own[w] = true; }
    HIP_TRY(hipMemset(ctx->grid[4], 0, grid_bytes(ctx, 4)));      // evolve_data.F90:76 phih_grid=0.0
    ctx->allfrac = p->allfrac != 0;
    if (ctx->allfrac)      // ionfractions_module.F90:36-38, evolve_data.F90:80-84: the (:,:,:,0) halves
        for (int w = 7; w <= 9; ++w) { HIP_TRY(hipMalloc(&ctx->grid[w], grid_bytes(ctx, w))); HIP_TRY(hipMemset(ctx->grid[w], 0, grid_bytes(ctx, w))); }
    HIP_TRY(hipMalloc(&ctx->d_nhi, grid_bytes(ctx, 2)));
    HIP_TRY(hipMalloc(&ctx->d_nhi_T, grid_bytes(ctx, 2)));
    HIP_TRY(hipMalloc(&ctx->d_phih_T, grid_bytes(ctx, 4)));
    // one spare element after each table: a copy of the last, so that tab[ip+1] exists for ip = numtau
    HIP_TRY(hipMalloc(&ctx->d_thick, (size_t)(p->numtau + 2) * sizeof(double)));
    HIP_TRY(hipMalloc(&ctx->d_thin, (size_t)(p->numtau + 2) * sizeof(double)));
    {   // log10_tab: interval i of m in [0.5,1) has centre c_i = (1 + (i+1/2)/64)/2; r_i = RN(1/c_i), T_i = RN(-log10 r_i)
        double tab[2 * kLogTab];
        for (int i = 0; i < kLogTab; ++i) {
            const long double c = 0.5L * (1.0L + ((long double)i + 0.5L) / (long double)kLogTab);
            const double r = (double)(1.0L / c);
            tab[2 * i] = r; tab[2 * i + 1] = (double)(-log10l((long double)r));
        }
        HIP_TRY(hipMalloc(&ctx->d_logtab, sizeof tab));
        HIP_TRY(hipMemcpy(ctx->d_logtab, tab, sizeof tab, hipMemcpyHostToDevice));
        // tau_od (fast mode): the same intervals, holding the table position 1 + (log10(1/r_i) - minlogtau)/dlogtau
        for (int i = 0; i < kLogTab; ++i)
            tab[2 * i + 1] = (double)(1.0L + (-log10l((long double)tab[2 * i]) - (long double)p->minlogtau) / (long double)p->dlogtau);
        HIP_TRY(hipMalloc(&ctx->d_odtab, sizeof tab));
        HIP_TRY(hipMemcpy(ctx->d_odtab, tab, sizeof tab, hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMalloc(&ctx->d_photon_loss, sizeof(double)));
    HIP_TRY(hipMalloc(&ctx->d_sum_nbox, sizeof(long long)));
    HIP_TRY(hipMalloc(&ctx->d_sum_partial, 4 * kSumBlocks * sizeof(double)));
    HIP_TRY(hipMalloc(&ctx->d_sum_out, 4 * sizeof(double)));
    HIP_TRY(hipMalloc(&ctx->d_stat_partial, 4 * kSumBlocks * sizeof(double)));
    HIP_TRY(hipMalloc(&ctx->d_conv, sizeof(unsigned long long)));
    HIP_TRY(hipMalloc(&ctx->d_chemfail, sizeof(unsigned int)));
    HIP_TRY(hipMalloc(&ctx->d_pair, 2 * sizeof(double)));
    HIP_TRY(hipMalloc(&ctx->d_gate, sizeof(int)));
    HIP_TRY(hipMalloc(&ctx->d_seq, sizeof(unsigned long long)));
    HIP_TRY(hipMemset(ctx->d_seq, 0, sizeof(unsigned long long)));
    HIP_TRY(hipMemset(ctx->d_conv, 0, sizeof(unsigned long long)));     // k_pass_final leaves them at zero again
    HIP_TRY(hipMemset(ctx->d_chemfail, 0, sizeof(unsigned int)));
    // pinned host scalars that kernels write straight through their mapped device pointers
    HIP_TRY(hipHostMalloc((void **)&ctx->h_sc, sizeof(*ctx->h_sc), hipHostMallocMapped));
    memset(ctx->h_sc, 0, sizeof(*ctx->h_sc));
    HIP_TRY(hipHostGetDevicePointer((void **)&ctx->d_hsc, ctx->h_sc, 0));
    HIP_TRY(hipHostMalloc((void **)&ctx->h_it4, (size_t)C2R_MAX_ITER_LOG * 4 * sizeof(double), hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer((void **)&ctx->d_hit4, ctx->h_it4, 0));
    // trace limits (evolve_source.F90:100-102), identical for every source
    int zlim = 0;
    for (int d = 0; d < 3; ++d) {
        ctx->hr[d] = std::min(p->max_subbox, p->mesh[d] / 2 - 1 + p->mesh[d] % 2);
        ctx->hl[d] = std::min(p->max_subbox, p->mesh[d] / 2);
    }
    zlim = std::min(ctx->hr[2], ctx->hl[2]);
    ctx->nbox_max = zlim > 0 ? (zlim + p->subboxsize - 1) / p->subboxsize : 0;
    int reach = 0;
    for (int d = 0; d < 3; ++d) reach = std::max(reach, std::max(ctx->hl[d], ctx->hr[d]));
    ctx->Qmax = std::min(ctx->nbox_max * p->subboxsize, reach);
    ctx->R = ctx->Qmax; ctx->P = 2 * ctx->R + 1; ctx->PP = (size_t)ctx->P * ctx->P;
    ctx->tiles_cap = (int)((ctx->PP + kBlock - 1) / kBlock);
    HIP_TRY(hipMalloc(&ctx->d_step, sizeof(StepBlock) + (size_t)(ctx->Qmax + 1) * sizeof(ShellStep)));
    HIP_TRY(hipHostMalloc((void **)&ctx->h_step, sizeof(StepBlock) + (size_t)(ctx->Qmax + 1) * sizeof(ShellStep)));
    HIP_TRY(hipEventCreateWithFlags(&ctx->ev_step, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&ctx->ev_prepared, hipEventDisableTiming));
    ctx->sc[0].stream = ctx->stream;                       // chain 0 of the sweep runs on the context's stream
    hipLaunchKernelGGL(k_load_code_object, dim3(1), dim3(1), 0, ctx->stream, (int *)nullptr);      // (loads the library's code object now)
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return C2R_OK;
}

void c2r_destroy(c2r_ctx *c)
{
    if (!c) return;
    Ctx *ctx = C(c);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    // (the chains' and the exchange's streams have joined the context's stream by the end of every pass; drained once more
    // before their arrays go)
    for (int k = 1; k < kMaxChains; ++k) if (ctx->sc[k].stream) hipStreamSynchronize(ctx->sc[k].stream);
    if (ctx->xstream) hipStreamSynchronize(ctx->xstream);
    for (auto &kv : ctx->graphs) { if (kv.second.exec) hipGraphExecDestroy(kv.second.exec); if (kv.second.graph) hipGraphDestroy(kv.second.graph); }
    for (auto &kv : ctx->chain_graphs) { if (kv.second.exec) hipGraphExecDestroy(kv.second.exec); if (kv.second.graph) hipGraphDestroy(kv.second.graph); }
    for (auto &kv : ctx->pinned) hipHostUnregister(const_cast<void *>(kv.first));
    free_sweep_scratch(ctx);
    for (int w = 0; w < 5; ++w) if (ctx->own[w]) hipFree(ctx->grid[w]);
    hipFree(ctx->grid[5]); hipFree(ctx->grid[6]); hipFree(ctx->grid[7]); hipFree(ctx->grid[8]); hipFree(ctx->grid[9]); hipFree(ctx->d_hthick); hipFree(ctx->d_hthin); hipFree(ctx->d_cool); hipFree(ctx->d_heat_T);
    hipFree(ctx->d_thick); hipFree(ctx->d_thin); hipFree(ctx->d_xthick); hipFree(ctx->d_xthin); hipFree(ctx->d_xhthick); hipFree(ctx->d_xhthin); hipFree(ctx->d_logtab); hipFree(ctx->d_odtab);
    hipFree(ctx->d_nhi); hipFree(ctx->d_nhi_T); hipFree(ctx->d_phih_T); hipFree(ctx->d_step); hipFree(ctx->d_pack); hipFree(ctx->d_boxdesc);
    if (ctx->h_boxdesc) hipHostFree(ctx->h_boxdesc);
    hipFree(ctx->d_lls); hipFree(ctx->d_lls_T); hipFree(ctx->d_clump);
    if (ctx->ev_prepared) hipEventDestroy(ctx->ev_prepared);
    hipFree(ctx->d_phih2); hipFree(ctx->d_phih2_T);
    if (ctx->ev_half) hipEventDestroy(ctx->ev_half);
    if (ctx->ev_xdone) hipEventDestroy(ctx->ev_xdone);
    if (ctx->xstream) hipStreamDestroy(ctx->xstream);
    for (int c = 1; c < kMaxChains; ++c) if (ctx->sc[c].own_stream && ctx->sc[c].stream) hipStreamDestroy(ctx->sc[c].stream);
    if (ctx->h_step) hipHostFree(ctx->h_step);
    if (ctx->ev_step) hipEventDestroy(ctx->ev_step);
    hipFree(ctx->d_photon_loss); hipFree(ctx->d_sum_nbox); hipFree(ctx->d_sum_partial); hipFree(ctx->d_sum_out); hipFree(ctx->d_stat_partial);
    hipFree(ctx->d_conv); hipFree(ctx->d_chemfail); hipFree(ctx->d_dbg); hipFree(ctx->d_pair); hipFree(ctx->d_seq); hipFree(ctx->d_gate); hipFree(ctx->d_nbox_all);
    if (ctx->h_nbox_all) hipHostFree(ctx->h_nbox_all);
    if (ctx->h_sc) hipHostFree(ctx->h_sc);
    if (ctx->h_it4) hipHostFree(ctx->h_it4);
    for (auto &e : ctx->ev_sweep) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    for (auto &e : ctx->ev_chem) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *c2r_last_error(const c2r_ctx *c) { return c ? C(c)->err.c_str() : "null context"; }

const char *c2r_info(c2r_ctx *c)
{
    if (!c) return "null context";
    Ctx *ctx = C(c);
    ctx->info = ctx->info_device + "; sweep_mode " + (ctx->fast ? "fast (C2R_SWEEP_FAST)" : "exact (C2R_SWEEP_EXACT)") +
                "; rates " + (ctx->prm.deterministic_rates ? "ordered per-source sums" : "f64 atomics") +
                "; rank " + std::to_string(ctx->rank) + " of " + std::to_string(ctx->nranks) +
                "; chains " + std::to_string(ctx->nchains) +
                "; chain passes replayed " + std::to_string(ctx->chain_replays) + " (halted " + std::to_string(ctx->chain_halts) + "), launch by launch " + std::to_string(ctx->chain_eager) + ", iterations with a device-gated tail " + std::to_string(ctx->chain_tails) +
                "; exchanges overlapped with the sweep " + std::to_string(ctx->xchg_overlapped) +
                "; plane-ordered launches " + std::to_string(ctx->xcd_launches) +
                "; graph captures " + std::to_string(ctx->captures);
    if (!ctx->info_warn.empty()) ctx->info += "; " + ctx->info_warn;
    return ctx->info.c_str();
}

int c2r_set_stream(c2r_ctx *c, void *s)
{
    if (!c) return C2R_EINVAL;
    Ctx *ctx = C(c);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (s == nullptr) {
        if (!ctx->own_stream) { HIP_TRY(hipStreamCreate(&ctx->stream)); ctx->own_stream = true; }
    } else {
        if (ctx->own_stream) { hipStreamDestroy(ctx->stream); ctx->own_stream = false; }
        ctx->stream = (hipStream_t)s;
    }
    ctx->sc[0].stream = ctx->stream;
    ++ctx->gen;
    return C2R_OK;
}

// The schedule's switches (include/c2ray_hip.h has the table).  None changes a result beyond the order in which the f64 atomics of
// different sources land; every captured launch sequence is dropped, and the sweep scratch where the option shapes it.
int c2r_set_option(c2r_ctx *c, const char *name, double value)
{
    if (!c || !name) return C2R_EINVAL;
    Ctx *ctx = C(c);
    HIP_TRY(hipSetDevice(ctx->prm.device));
    const std::string n(name);
    const bool on = value != 0.0;
    const int iv = (int)value;
    bool scratch = false;
    if (n == "graph") ctx->use_graph = on;
    else if (n == "chain_graph") ctx->chain_graph = on;
    else if (n == "chain_tail") ctx->chain_tail = on;
    else if (n == "fused_iter") ctx->fused_iter = on;
    else if (n == "fuse_small") ctx->fuse_small = on;
    else if (n == "fold_source_cell") ctx->fold_source_cell = on;
    else if (n == "pair_shells") ctx->pair_shells = on;
    else if (n == "sched_hint") ctx->sched_hint = on;
    else if (n == "spin_wait") ctx->spin_wait = on;
    else if (n == "poll_wait") ctx->poll_wait = on;
    else if (n == "stream_hint") { ctx->stream_hint = value < 0.0 ? ctx->ncell * sizeof(double) >= ((size_t)64 << 20) : on; scratch = true; }
    else if (n == "xcd_order") { ctx->xcd_order = iv < 0 ? -1 : (iv > 0 ? 1 : 0); scratch = true; }
    else if (n == "xcd_min_per_plane") { ctx->xcd_min_per_plane = value; scratch = true; }
    else if (n == "xcd_min_alive") ctx->xcd_min_alive = value;
    else if (n == "xcd_qmin") ctx->xcd_qmin = std::max(1, iv);
    else if (n == "xcd_min_sources") ctx->xcd_min_sources = std::max(8, iv);
    else if (n == "chains") { ctx->chains_env = std::max(0, std::min(kMaxChains, iv)); scratch = true; }
    else if (n == "batch_cap") { ctx->batch_cap_opt = std::max(0, iv); scratch = true; }
    else if (n == "sparse_exchange") ctx->sparse_exchange = on;
    else if (n == "sparse_fraction") ctx->sparse_fraction = std::max(0.0, value);
    else if (n == "exchange_overlap") ctx->exchange_overlap = on;
    else if (n == "exchange_overlap_min") ctx->overlap_min_sources = std::max(1, iv);
    else FAIL(C2R_EINVAL, "c2r_set_option: unknown option '" + n + "'");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (scratch) free_sweep_scratch(ctx);          // (laid out again by the next pass or c2r_set_sources)
    ++ctx->gen;
    return C2R_OK;
}

int c2r_set_tables(c2r_ctx *c, const double *thick, const double *thin, int32_t n)
{
    if (!c || !thick || !thin) return C2R_EINVAL;
    Ctx *ctx = C(c);
    if (n != ctx->prm.numtau + 1) FAIL(C2R_EINVAL, "table length must be numtau+1");
    HIP_TRY(hipMemcpy(ctx->d_thick, thick, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->d_thin, thin, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->d_thick + n, thick + n - 1, sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->d_thin + n, thin + n - 1, sizeof(double), hipMemcpyHostToDevice));
    ctx->have_tables = true; ++ctx->gen;
    return C2R_OK;
}

int c2r_set_xray_tables(c2r_ctx *c, const double *thick, const double *thin, int32_t n)
{
    if (!c || ((thick == nullptr) != (thin == nullptr))) return C2R_EINVAL;
    Ctx *ctx = C(c);
    HIP_TRY(hipSetDevice(ctx->prm.device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ++ctx->gen;                                   // (captured launch sequences hold the choice of kernel)
    if (!thick) { ctx->xray = false; return C2R_OK; }
    if (n != ctx->prm.numtau + 1) FAIL(C2R_EINVAL, "table length must be numtau+1");
    if (!ctx->d_xthick) HIP_TRY(hipMalloc(&ctx->d_xthick, (size_t)(n + 1) * sizeof(double)));
    if (!ctx->d_xthin) HIP_TRY(hipMalloc(&ctx->d_xthin, (size_t)(n + 1) * sizeof(double)));
    // padded like the stellar tables: tab[numtau+1] = tab[numtau]
    HIP_TRY(hipMemcpy(ctx->d_xthick, thick, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->d_xthin, thin, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->d_xthick + n, thick + n - 1, sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->d_xthin + n, thin + n - 1, sizeof(double), hipMemcpyHostToDevice));
    ctx->xray = true;
    return C2R_OK;
}

int c2r_set_xray_heat_tables(c2r_ctx *c, const double *heat_thick, const double *heat_thin, int32_t n)
{
    if (!c || !heat_thick || !heat_thin) return C2R_EINVAL;
    Ctx *ctx = C(c);
    if (n != ctx->prm.numtau + 1) FAIL(C2R_EINVAL, "table length must be numtau+1");
    HIP_TRY(hipSetDevice(ctx->prm.device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (!ctx->d_xhthick) HIP_TRY(hipMalloc(&ctx->d_xhthick, (size_t)(n + 1) * sizeof(double)));
    if (!ctx->d_xhthin) HIP_TRY(hipMalloc(&ctx->d_xhthin, (size_t)(n + 1) * sizeof(double)));
    HIP_TRY(hipMemcpy(ctx->d_xhthick, heat_thick, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->d_xhthin, heat_thin, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->d_xhthick + n, heat_thick + n - 1, sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->d_xhthin + n, heat_thin + n - 1, sizeof(double), hipMemcpyHostToDevice));
    ctx->have_xheat = true; ++ctx->gen;
    return C2R_OK;
}

int c2r_set_xray_sources(c2r_ctx *c, const double *normflux_xray, int32_t nsrc)
{
    if (!c || nsrc < 0 || (nsrc > 0 && !normflux_xray)) return C2R_EINVAL;
    Ctx *ctx = C(c);
    if (nsrc != ctx->nsrc) FAIL(C2R_EINVAL, "c2r_set_xray_sources: one value per source of the list c2r_set_sources was given");
    ctx->nflux_x.assign(normflux_xray, normflux_xray + nsrc);
    return C2R_OK;
}

int c2r_set_step(c2r_ctx *c, const double dr[3], double vol, double lls, float clumping, double temper)
{
    if (!c || !dr) return C2R_EINVAL;
    Ctx *ctx = C(c);
    if (!(dr[0] > 0) || !(dr[1] > 0) || !(dr[2] > 0) || !(vol > 0) || !(temper > 0)) FAIL(C2R_EINVAL, "dr, vol and temper must be positive");
    // (no captured launch depends on these: they reach the kernels through the device-resident step block, sync_step)
    for (int d = 0; d < 3; ++d) ctx->dr[d] = dr[d];
    ctx->vol = vol; ctx->lls = lls; ctx->clumping = clumping; ctx->temper = temper;
    ctx->have_step = true;
    return C2R_OK;
}

int c2r_set_lls(c2r_ctx *c, int32_t type, const float *lls_grid, double R_max_LLS)
{
    if (!c) return C2R_EINVAL;
    Ctx *ctx = C(c);
    if (type < 1 || type > 3) FAIL(C2R_EINVAL, "type_of_LLS must be 1, 2 or 3");
    if (type == 2 && !lls_grid) FAIL(C2R_EINVAL, "type_of_LLS=2 needs the LLS grid");
    if (type == 3 && !(R_max_LLS > 0.0)) FAIL(C2R_EINVAL, "type_of_LLS=3 needs R_max_LLS > 0");
    HIP_TRY(hipSetDevice(ctx->prm.device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (type == 2) { const int rc = upload_lls_grid(ctx, lls_grid); if (rc) return rc; }
    ctx->lls_type = type; ctx->R_max_LLS = R_max_LLS; ++ctx->gen;
    return C2R_OK;
}

int c2r_set_clumping_grid(c2r_ctx *c, const float *clump_grid)
{
    if (!c) return C2R_EINVAL;
    Ctx *ctx = C(c);
    HIP_TRY(hipSetDevice(ctx->prm.device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    // (a captured iteration holds the grid's pointer, or its absence, in the global pass's arguments: a new generation when it
    // appears or goes; a refill of the same allocation is seen by every launch)
    if (!clump_grid) { if (ctx->d_clump) ++ctx->gen; hipFree(ctx->d_clump); ctx->d_clump = nullptr; return C2R_OK; }
    if (!ctx->d_clump) { HIP_TRY(hipMalloc(&ctx->d_clump, grid_bytes(ctx, 0))); ++ctx->gen; }
    HIP_TRY(hipMemcpy(ctx->d_clump, clump_grid, grid_bytes(ctx, 0), hipMemcpyHostToDevice));
    return C2R_OK;
}

int c2r_default_thermal(c2r_thermal_params *t)
{
    if (!t) return C2R_EINVAL;
    memset(t, 0, sizeof *t);
    t->tau_heat_limit = C2R_TAU_HEAT_LIMIT;
    t->k_B = C2R_K_B; t->gamma1 = C2R_GAMMA1; t->minitemp = C2R_MINITEMP; t->relative_denergy = C2R_RELATIVE_DENERGY;
    t->thermal_rate_floor = C2R_THERMAL_RATE_FLOOR; t->thermal_time_tol = C2R_THERMAL_TIME_TOL;
    t->temp_conv_rel = C2R_TEMP_CONV_REL; t->temp_conv_abs = C2R_TEMP_CONV_ABS;
    t->H0 = C2R_H0; t->Omega0 = C2R_OMEGA0;
    t->cool_mintemp = 0.0; t->cool_dtemp = 0.0;          // from the cooling table file: temp(1), temp(2)-temp(1) (cooling.f90:78-79)
    t->cool_points = C2R_COOL_POINTS; t->thermal_max_steps = C2R_THERMAL_MAX_STEPS; t->cosmological = 1;
    return C2R_OK;
}

int c2r_set_thermal(c2r_ctx *c, const c2r_thermal_params *t, const double *heat_thick, const double *heat_thin, int32_t n,
                    const double *cie_cool)
{
    if (!c) return C2R_EINVAL;
    Ctx *ctx = C(c);
    HIP_TRY(hipSetDevice(ctx->prm.device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ++ctx->gen;
    if (ctx->prm.deterministic_rates && (t != nullptr) != ctx->thermal) free_sweep_scratch(ctx);   // per-source heating grids come and go
    if (!t) { ctx->thermal = false; return C2R_OK; }     // back to the isothermal path (the arrays stay allocated)
    if (!heat_thick || !heat_thin || !cie_cool) FAIL(C2R_EINVAL, "non-isothermal run needs the heating tables and the cooling curve");
    if (n != ctx->prm.numtau + 1) FAIL(C2R_EINVAL, "table length must be numtau+1");
    if (t->cool_points < 2 || !(t->cool_dtemp > 0.0) || !(t->gamma1 > 0.0) || !(t->k_B > 0.0) || t->thermal_max_steps < 1)
        FAIL(C2R_EINVAL, "c2r_thermal_params: cool_points >= 2, cool_dtemp > 0, gamma1 > 0, k_B > 0, thermal_max_steps >= 1");
    ctx->tprm = *t;
    // (each allocation on its own: a call that failed half-way is completed by the next one)
    if (!ctx->d_hthick) HIP_TRY(hipMalloc(&ctx->d_hthick, (size_t)(n + 1) * sizeof(double)));
    if (!ctx->d_hthin) HIP_TRY(hipMalloc(&ctx->d_hthin, (size_t)(n + 1) * sizeof(double)));
    if (!ctx->d_heat_T) HIP_TRY(hipMalloc(&ctx->d_heat_T, grid_bytes(ctx, 5)));
    if (!ctx->grid[5]) {
        HIP_TRY(hipMalloc(&ctx->grid[5], grid_bytes(ctx, 5)));
        HIP_TRY(hipMemset(ctx->grid[5], 0, grid_bytes(ctx, 5)));          // evolve_data.F90:78 phiheat_grid=0.0
    }
    if (!ctx->grid[6]) {
        HIP_TRY(hipMalloc(&ctx->grid[6], grid_bytes(ctx, 6)));
        HIP_TRY(hipMemset(ctx->grid[6], 0, grid_bytes(ctx, 6)));
    }
    hipFree(ctx->d_cool); ctx->d_cool = nullptr;
    HIP_TRY(hipMalloc(&ctx->d_cool, (size_t)t->cool_points * sizeof(double)));
    HIP_TRY(hipMemcpy(ctx->d_cool, cie_cool, (size_t)t->cool_points * sizeof(double), hipMemcpyHostToDevice));
    // padded like the photo tables: tab[numtau+1] = tab[numtau]
    HIP_TRY(hipMemcpy(ctx->d_hthick, heat_thick, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->d_hthin, heat_thin, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->d_hthick + n, heat_thick + n - 1, sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->d_hthin + n, heat_thin + n - 1, sizeof(double), hipMemcpyHostToDevice));
    ctx->thermal = true;
    return C2R_OK;
}

int c2r_set_redshift(c2r_ctx *c, double zred)
{
    if (!c) return C2R_EINVAL;
    Ctx *ctx = C(c);
    if (!(zred > -1.0)) FAIL(C2R_EINVAL, "zred must be > -1");
    ctx->zred = zred; ctx->have_zred = true;
    return C2R_OK;
}

int c2r_set_sources(c2r_ctx *c, const int32_t *srcpos, const double *normflux, int32_t nsrc)
{
    if (!c || nsrc < 0 || (nsrc > 0 && (!srcpos || !normflux))) return C2R_EINVAL;
    Ctx *ctx = C(c);
    // the same list again (the Fortran shim hands the driver's list over before every evolve3D; it changes once per redshift
    // slice, sourceprops.F90:121-167): nothing to do -- and what the last pass learnt about it (where each source ended, the
    // captured launch sequences of a small batch, the balanced shares) stays valid
    if (nsrc == ctx->nsrc && nsrc > 0 && ctx->batch_cap > 0 && !(ctx->explicit_share && !ctx->auto_share) && memcmp(ctx->srcpos.data(), srcpos, 3 * (size_t)nsrc * sizeof(int32_t)) == 0 &&
        memcmp(ctx->nflux.data(), normflux, (size_t)nsrc * sizeof(double)) == 0)
        return C2R_OK;
    ctx->srcpos.assign(srcpos, srcpos + 3 * (size_t)nsrc);
    ctx->nflux.assign(normflux, normflux + nsrc);
    ctx->nsrc = nsrc;
    ctx->sparse_valid = false;                    // (nbox_all / last_nbox no longer describe what is in phih_grid)
    ctx->nflux_x.clear();                         // (NormFlux_xray belongs to the list: c2r_set_xray_sources follows a new one)
    ctx->explicit_share = false; ctx->auto_share = false; ctx->share.clear(); ctx->last_nbox.clear(); ctx->nbox_all.clear(); ctx->box_hint = 0;
    for (auto &kv : ctx->chain_graphs) { kv.second.profile.clear(); kv.second.profile_prev.clear(); }      // (what the old list's passes left says nothing about this one)
    // set-up belongs here, not in the first evolve3D of a run (the reference allocates in evolve_ini, evolve_data.F90:75-90): the
    // sweep scratch of this rank's share -- device planes, the pinned staging block -- is a few milliseconds of allocation calls
    if (nsrc > 0 && n_local_sources(ctx) > 0) {
        HIP_TRY(hipSetDevice(ctx->prm.device));
        const int rc = ensure_sweep_scratch(ctx, n_local_sources(ctx));
        if (rc) return rc;
    }
    return C2R_OK;
}

int c2r_bind_device_buffers(c2r_ctx *c, void *ndens, void *xh, void *xh_av, void *xh_int, void *phih)
{
    if (!c) return C2R_EINVAL;
    Ctx *ctx = C(c);
    void *in[5] = {ndens, xh, xh_av, xh_int, phih};
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int w = 0; w < 5; ++w) {
        if (!in[w]) continue;
        if (ctx->own[w]) { hipFree(ctx->grid[w]); ctx->own[w] = false; }
        ctx->grid[w] = in[w];
    }
    if (phih) { ctx->rates_clean = false; ctx->sparse_valid = false; }
    ++ctx->gen;
    return C2R_OK;
}

int c2r_device_ptr(c2r_ctx *c, int32_t which, void **ptr)
{
    if (!c || !ptr || which < 0 || which > 9) return C2R_EINVAL;
    if (which > 4 && which < 7 && !C(c)->thermal) return C2R_ESTATE;
    if (which >= 7 && !C(c)->allfrac) return C2R_ESTATE;
    *ptr = C(c)->grid[which];
    return C2R_OK;
}

int c2r_upload(c2r_ctx *c, int32_t which, const void *host)
{
    if (!c || !host || which < 0 || which > 9) return C2R_EINVAL;
    Ctx *ctx = C(c);
    if (which > 4 && which < 7 && !ctx->thermal) FAIL(C2R_ESTATE, "arrays 5 and 6 exist in non-isothermal runs only (c2r_set_thermal)");
    if (which >= 7 && !ctx->allfrac) FAIL(C2R_ESTATE, "arrays 7 - 9 (the stored neutral fractions) exist with c2r_params.allfrac only");
    HIP_TRY(hipMemcpyAsync(ctx->grid[which], host, grid_bytes(ctx, which), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (which == 4 || which == 5) { ctx->rates_clean = false; ctx->sparse_valid = false; }     // the caller's rates: not a pass over zeroed ones
    return C2R_OK;
}

int c2r_download(c2r_ctx *c, int32_t which, void *host)
{
    if (!c || !host || which < 0 || which > 9) return C2R_EINVAL;
    Ctx *ctx = C(c);
    if (which > 4 && which < 7 && !ctx->thermal) FAIL(C2R_ESTATE, "arrays 5 and 6 exist in non-isothermal runs only (c2r_set_thermal)");
    if (which >= 7 && !ctx->allfrac) FAIL(C2R_ESTATE, "arrays 7 - 9 (the stored neutral fractions) exist with c2r_params.allfrac only");
    HIP_TRY(hipMemcpyAsync(host, ctx->grid[which], grid_bytes(ctx, which), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return C2R_OK;
}

int c2r_zero_rates(c2r_ctx *c)
{
    if (!c) return C2R_EINVAL;
    Ctx *ctx = C(c);
    HIP_TRY(hipMemsetAsync(ctx->grid[4], 0, grid_bytes(ctx, 4), ctx->stream));
    if (ctx->thermal) HIP_TRY(hipMemsetAsync(ctx->grid[5], 0, grid_bytes(ctx, 5), ctx->stream));     // evolve.F90:435
    ctx->rates_clean = true; ctx->sparse_valid = false;
    return C2R_OK;
}

int c2r_do_source_host(c2r_ctx *c, int32_t ns, const float *ndens, const double *xh_av, double *phih_grid,
                       double *coldensh_out, double *photon_loss_src, int32_t *nbox)
{
    if (!c || !ndens || !xh_av || !phih_grid) return C2R_EINVAL;
    Ctx *ctx = C(c);
    int rc;
    if ((rc = c2r_upload(c, 0, ndens))) return rc;
    if ((rc = copy_in(ctx, 2, xh_av))) return rc;              // (-DALLFRAC drivers: both halves of the (mesh,0:1) array)
    if ((rc = c2r_zero_rates(c))) return rc;
    if ((rc = c2r_do_source(c, ns, coldensh_out, photon_loss_src, nbox, nullptr))) return rc;
    // phih_grid(pos) = phih_grid(pos) + this source's rate (evolve_point.F90:283), on the host array
    std::vector<double> g(ctx->ncell);
    if ((rc = c2r_download(c, 4, g.data()))) return rc;
    for (size_t i = 0; i < ctx->ncell; ++i) phih_grid[i] = phih_grid[i] + g[i];
    return C2R_OK;
}

int c2r_do_grid_host(c2r_ctx *c, const float *ndens, const double *xh_av, double *phih_grid, double *phiheat_grid,
                     double *photon_loss, int64_t *sum_nbox)
{
    if (!c || !ndens || !xh_av || !phih_grid) return C2R_EINVAL;
    Ctx *ctx = C(c);
    if (ctx->thermal && !phiheat_grid) FAIL(C2R_EINVAL, "non-isothermal run: do_grid needs phiheat_grid");
    int rc;
    if ((rc = c2r_upload(c, 0, ndens))) return rc;
    if ((rc = copy_in(ctx, 2, xh_av))) return rc;
    if ((rc = c2r_zero_rates(c))) return rc;
    double loss = 0.0; int64_t nb = 0;
    if ((rc = c2r_pass_sources(c, &loss, &nb, nullptr))) return rc;
    // phih_grid(pos) = phih_grid(pos) + the rates of this rank's sources (evolve_point.F90:283-286), on the host arrays
    std::vector<double> g(ctx->ncell);
    if ((rc = c2r_download(c, 4, g.data()))) return rc;
    for (size_t i = 0; i < ctx->ncell; ++i) phih_grid[i] = phih_grid[i] + g[i];
    if (ctx->thermal) {
        if ((rc = c2r_download(c, 5, g.data()))) return rc;
        for (size_t i = 0; i < ctx->ncell; ++i) phiheat_grid[i] = phiheat_grid[i] + g[i];
    }
    if (photon_loss) *photon_loss = loss;
    if (sum_nbox) *sum_nbox = nb;
    return C2R_OK;
}

int c2r_selftest(c2r_ctx *c, int64_t *mismatches)
{
    if (!c || !mismatches) return C2R_EINVAL;
    Ctx *ctx = C(c);
    unsigned int *d_bad = nullptr;
    HIP_TRY(hipMalloc(&d_bad, sizeof(unsigned int)));
    HIP_TRY(hipMemsetAsync(d_bad, 0, sizeof(unsigned int), ctx->stream));
    const int n = 1 << 22;
    const double divisors[4] = {ctx->prm.dlogtau, ctx->have_step ? ctx->dr[0] : 1.37848875056274974e+24, 49.0, 16129.0};
    for (int i = 0; i < 4; ++i)
        hipLaunchKernelGGL(k_selftest_div, dim3(n / 256), dim3(256), 0, ctx->stream, n, divisors[i], 1.0 / divisors[i],
                           0x1234567ULL * (i + 1), d_bad);
    unsigned int bad = 0;
    HIP_TRY(hipMemcpyAsync(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    hipFree(d_bad);
    *mismatches = bad;
    return C2R_OK;
}

int c2r_profile(c2r_ctx *c, int32_t enable)
{
    if (!c) return C2R_EINVAL;
    Ctx *ctx = C(c);
    ctx->prof = enable < 0 ? 0 : (enable > 2 ? 2 : enable);
    ctx->prof_sweep_ms = ctx->prof_chem_ms = 0; ctx->prof_sweep_n = ctx->prof_chem_n = 0;
    return C2R_OK;
}

int c2r_profile_read(c2r_ctx *c, double *sweep_ms, int64_t *sweep_launches, double *chem_ms, int64_t *chem_launches)
{
    if (!c) return C2R_EINVAL;
    Ctx *ctx = C(c);
    if (sweep_ms) *sweep_ms = ctx->prof_sweep_ms;
    if (sweep_launches) *sweep_launches = ctx->prof_sweep_n;
    if (chem_ms) *chem_ms = ctx->prof_chem_ms;
    if (chem_launches) *chem_launches = ctx->prof_chem_n;
    return C2R_OK;
}

}  // extern "C"
