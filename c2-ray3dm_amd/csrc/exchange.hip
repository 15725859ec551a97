// Several ranks: which sources a rank sweeps (static stride, explicit shares, the LPT re-partition), the all-reduce of the
// rates -- whole grid or the sources' packed sub-boxes --, z-slabs of the slab chemistry.  Kernels: kernels_exchange.hpp.
#include "ctx.hpp"
#include "kernels_exchange.hpp"

namespace c2r {

// z-slab of rank r of P: whole z-planes, the first (N3 mod P) ranks one plane more; in cells
void slab_of(const Ctx *ctx, int r, int P, size_t *off, size_t *cnt)
{
    const size_t n3 = (size_t)ctx->prm.mesh[2], plane = (size_t)ctx->prm.mesh[0] * ctx->prm.mesh[1];
    const size_t base = n3 / (size_t)P, rem = n3 % (size_t)P;
    const size_t z0 = (size_t)r * base + std::min<size_t>((size_t)r, rem), nz = base + ((size_t)r < rem ? 1 : 0);
    *off = z0 * plane; *cnt = nz * plane;
}

namespace {
// Longest-processing-time partition of the sources over the ranks by cost (deterministic: ties by source
// index, then by rank); every share in ascending source order.  The role of the reference's master/worker
// scheduler (master_slave.F90:124-330) without a master: cost = cells of the sub-box the source ended with in
// the previous pass (+1: every source costs something).
void lpt_shares(const std::vector<long long> &cost, int nranks, std::vector<std::vector<int32_t>> &shares)
{
    const int n = (int)cost.size();
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost[a] > cost[b]; });
    std::vector<long long> load(nranks, 0);
    shares.assign(nranks, {});
    for (int i : order) {
        int r = 0;
        for (int k = 1; k < nranks; ++k) if (load[k] < load[r]) r = k;
        shares[r].push_back(i);
        load[r] += cost[i] + 1;
    }
    for (auto &sh : shares) std::sort(sh.begin(), sh.end());
}

}  // namespace

// Before a pass: this rank's share from the sub-box counts every rank learnt after the previous pass.
void balance_before_pass(Ctx *ctx)
{
    if (!ctx->balance || ctx->nranks <= 1 || (ctx->explicit_share && !ctx->auto_share)) return;
    if ((int)ctx->nbox_all.size() != ctx->nsrc) {            // nothing known yet (first pass, new source list): static rule
        if (ctx->auto_share) { ctx->explicit_share = false; ctx->auto_share = false; ctx->share.clear(); }
        return;
    }
    std::vector<long long> cost(ctx->nsrc);
    for (int i = 0; i < ctx->nsrc; ++i) cost[i] = visited_for_nbox(ctx, ctx->nbox_all[i]);
    std::vector<std::vector<int32_t>> shares;
    lpt_shares(cost, ctx->nranks, shares);
    ctx->share = shares[ctx->rank];
    ctx->explicit_share = true; ctx->auto_share = true;
}

namespace {
// After a pass: every rank contributes the sub-box counts of the sources it swept (zero elsewhere); the sum over
// ranks through the all-reduce callback is the full list (exact in f64).  Once per pass (nbox_all_pass).
int gather_nbox_all(Ctx *ctx)
{
    if (ctx->nbox_all_pass == ctx->pass_id && (int)ctx->nbox_all.size() == ctx->nsrc) return C2R_OK;
    if (ctx->nbox_all_cap < ctx->nsrc) {
        hipFree(ctx->d_nbox_all); ctx->d_nbox_all = nullptr; ctx->nbox_all_cap = 0;
        if (ctx->h_nbox_all) { hipHostFree(ctx->h_nbox_all); ctx->h_nbox_all = nullptr; }
        HIP_TRY(hipMalloc(&ctx->d_nbox_all, (size_t)ctx->nsrc * sizeof(double)));
        HIP_TRY(hipHostMalloc((void **)&ctx->h_nbox_all, (size_t)ctx->nsrc * sizeof(double)));
        ctx->nbox_all_cap = ctx->nsrc;
    }
    double *mine = ctx->h_nbox_all;                          // pinned: both copies below are true async DMA
    for (int i = 0; i < ctx->nsrc; ++i) mine[i] = 0.0;
    const int nloc = n_local_sources(ctx);
    for (int i = 0; i < nloc && i < (int)ctx->last_nbox.size(); ++i) {
        const int g = ctx->explicit_share ? ctx->share[i] : ctx->rank + i * ctx->nranks;
        mine[g] = (double)ctx->last_nbox[i];
    }
    const size_t bytes = (size_t)ctx->nsrc * sizeof(double);
    HIP_TRY(hipMemcpyAsync(ctx->d_nbox_all, mine, bytes, hipMemcpyHostToDevice, ctx->stream));
    if (ctx->ar(ctx->ar_user, ctx->d_nbox_all, (size_t)ctx->nsrc, (void *)ctx->stream) != 0) FAIL(C2R_ECALLBACK, "all-reduce callback failed");
    HIP_TRY(hipMemcpyAsync(mine, ctx->d_nbox_all, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->nbox_all.resize(ctx->nsrc);
    for (int i = 0; i < ctx->nsrc; ++i) ctx->nbox_all[i] = (int32_t)llround(mine[i]);
    ctx->nbox_all_pass = ctx->pass_id;
    return C2R_OK;
}

}  // namespace

int balance_after_pass(Ctx *ctx)
{
    if (!ctx->balance || ctx->nranks <= 1 || !ctx->ar || (ctx->explicit_share && !ctx->auto_share) || ctx->nsrc == 0) return C2R_OK;
    return gather_nbox_all(ctx);
}

}  // namespace c2r

using namespace c2r;

extern "C" {

int c2r_set_rank(c2r_ctx *c, int32_t rank, int32_t nranks, c2r_allreduce_fn fn, void *user)
{
    if (!c) return C2R_EINVAL;
    Ctx *ctx = C(c);
    if (nranks < 1 || rank < 0 || rank >= nranks) FAIL(C2R_EINVAL, "need 0 <= rank < nranks");
    if (nranks > 1 && !fn) FAIL(C2R_EINVAL, "nranks > 1 needs an all-reduce callback");
    if (nranks > kMaxSlabRanks && ctx->rs) FAIL(C2R_EINVAL, "slab chemistry supports up to 64 ranks (c2r_set_slab_chemistry is on)");
    ctx->rank = rank; ctx->nranks = nranks; ctx->ar = fn; ctx->ar_user = user;
    if (nranks > 1 && ctx->device_auto && ctx->device_var.empty() && ctx->info_warn.empty()) {
        // several ranks, one process per GPU, and nothing told this process which GPU is its own: every rank of the
        // node would share device 0.  Not an error (tests run several ranks on one GPU on purpose), but never silent.
        ctx->info_warn = "WARNING: C2R_DEVICE_AUTO with nranks > 1 and no local-rank variable (C2R_DEVICE, LOCAL_RANK, "
                         "OMPI_COMM_WORLD_LOCAL_RANK, MV2_COMM_WORLD_LOCAL_RANK, MPI_LOCALRANKID, PMI_LOCAL_RANK, SLURM_LOCALID): "
                         "every rank of this node runs on device 0";
        fprintf(stderr, "c2ray_hip: %s\n", ctx->info_warn.c_str());
    }
    if (ctx->auto_share) { ctx->explicit_share = false; ctx->auto_share = false; ctx->share.clear(); }
    ctx->nbox_all.clear();
    // c2r_set_sources sized the sweep scratch for the share it knew then (one rank: every source); a smaller share frees the
    // difference -- in deterministic mode that is two N^3 grids per source (re-allocated at the next pass for the new share)
    if (ctx->batch_want > 0 && n_local_sources(ctx) < ctx->batch_want) {
        HIP_TRY(hipSetDevice(ctx->prm.device));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        free_sweep_scratch(ctx);
        ++ctx->gen;
        if (n_local_sources(ctx) > 0) { const int rc = ensure_sweep_scratch(ctx, n_local_sources(ctx)); if (rc) return rc; }
    }
    return C2R_OK;
}

int c2r_set_slab_chemistry(c2r_ctx *c, c2r_reduce_scatter_fn rs, c2r_allgather_fn ag, void *user)
{
    if (!c) return C2R_EINVAL;
    Ctx *ctx = C(c);
    if ((rs == nullptr) != (ag == nullptr)) FAIL(C2R_EINVAL, "slab chemistry needs both the reduce-scatter and the all-gather callback (or neither)");
    if (rs && ctx->nranks > kMaxSlabRanks) FAIL(C2R_EINVAL, "slab chemistry supports up to 64 ranks");   // (checked here and in c2r_set_rank: a rank failing inside the loop would leave the others in their collectives)
    ctx->rs = rs; ctx->ag = ag; ctx->slab_user = user;
    return C2R_OK;
}

int c2r_set_source_queue(c2r_ctx *c, c2r_next_sources_fn next, void *user, int32_t chunk)
{
    if (!c) return C2R_EINVAL;
    Ctx *ctx = C(c);
    if (next && chunk < 1) FAIL(C2R_EINVAL, "c2r_set_source_queue: chunk must be >= 1");
    if (next && ctx->prm.deterministic_rates) FAIL(C2R_ESTATE, "sources on request and ordered rates exclude each other: the order of a rank's sources is not fixed");
    ctx->queue_next = next; ctx->queue_user = user; ctx->queue_chunk = next ? chunk : 0;
    if (ctx->auto_share) { ctx->explicit_share = false; ctx->auto_share = false; ctx->share.clear(); }
    HIP_TRY(hipSetDevice(ctx->prm.device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    free_sweep_scratch(ctx);               // (sized for a chunk from now on / for the rank's share again)
    ++ctx->gen;
    return C2R_OK;
}

int c2r_slab(const c2r_ctx *c, int32_t rank, int32_t nranks, size_t *cell_offset, size_t *cell_count)
{
    if (!c || nranks < 1 || rank < 0 || rank >= nranks || !cell_offset || !cell_count) return C2R_EINVAL;
    slab_of(C(c), rank, nranks, cell_offset, cell_count);
    return C2R_OK;
}

int c2r_set_source_share(c2r_ctx *c, const int32_t *idx, int32_t n)
{
    if (!c) return C2R_EINVAL;
    Ctx *ctx = C(c);
    if (!idx || n < 0) { ctx->explicit_share = false; ctx->auto_share = false; ctx->share.clear(); return C2R_OK; }
    for (int i = 0; i < n; ++i) if (idx[i] < 0 || idx[i] >= ctx->nsrc) FAIL(C2R_EINVAL, "source index out of range");
    ctx->share.assign(idx, idx + n);
    ctx->explicit_share = true; ctx->auto_share = false;
    return C2R_OK;
}

int c2r_last_nbox(c2r_ctx *c, int32_t *nbox, int32_t n)
{
    if (!c || (n > 0 && !nbox)) return C2R_EINVAL;
    Ctx *ctx = C(c);
    if (n != (int32_t)ctx->last_nbox.size()) FAIL(C2R_EINVAL, "length must equal the number of sources this rank swept");
    for (int i = 0; i < n; ++i) nbox[i] = ctx->last_nbox[i];
    return C2R_OK;
}

int c2r_get_device(const c2r_ctx *c, int32_t *device)
{
    if (!c || !device) return C2R_EINVAL;
    *device = C(c)->prm.device;
    return C2R_OK;
}

int c2r_set_balance(c2r_ctx *c, int32_t on)
{
    if (!c) return C2R_EINVAL;
    Ctx *ctx = C(c);
    ctx->balance = on != 0;
    if (!ctx->balance && ctx->auto_share) { ctx->explicit_share = false; ctx->auto_share = false; ctx->share.clear(); }
    ctx->nbox_all.clear();
    return C2R_OK;
}

int c2r_source_share(c2r_ctx *c, int32_t *idx, int32_t cap, int32_t *n)
{
    if (!c || !n || (cap > 0 && !idx)) return C2R_EINVAL;
    Ctx *ctx = C(c);
    const int nloc = n_local_sources(ctx);
    *n = nloc;
    for (int i = 0; i < nloc && i < cap; ++i) idx[i] = ctx->explicit_share ? ctx->share[i] : ctx->rank + i * ctx->nranks;
    return C2R_OK;
}

int c2r_balanced_shares(const int64_t *cost, int32_t nsrc, int32_t nranks, int32_t rank, int32_t *idx, int32_t *n)
{
    if (nsrc < 0 || nranks < 1 || rank < 0 || rank >= nranks || !n || (nsrc > 0 && (!cost || !idx))) return C2R_EINVAL;
    std::vector<long long> c(cost, cost + nsrc);
    std::vector<std::vector<int32_t>> shares;
    lpt_shares(c, nranks, shares);
    *n = (int32_t)shares[rank].size();
    for (size_t i = 0; i < shares[rank].size(); ++i) idx[i] = shares[rank][i];
    return C2R_OK;
}

int c2r_allreduce_rates(c2r_ctx *c)
{
    if (!c) return C2R_EINVAL;
    Ctx *ctx = C(c);
    if (ctx->nranks <= 1 || !ctx->ar) return C2R_OK;
    const c2r_params &p = ctx->prm;
    if (ctx->rates_reduced_pass == ctx->pass_id) {
        // the pass exchanged its rates itself, overlapped with its second half (c2r_set_exchange_overlap); what remains is the
        // list of sub-boxes every rank learns per pass (it decides whether the NEXT pass travels packed, and the LPT shares)
        ctx->rates_reduced_pass = -1;
        HIP_TRY(hipSetDevice(ctx->prm.device));
        return ctx->sparse_exchange ? gather_nbox_all(ctx) : C2R_OK;
    }
    ++ctx->xchg_calls;
    // Sparse form: every rank learns every source's final sub-box (one small all-reduce), so all ranks agree on the same list
    // of boxes; while their volumes add up to a fraction of the mesh, only they travel -- packed box after box in source
    // order, reduced, written back (a cell of two overlapping boxes travels twice and comes back with the same sum).  The
    // rates are zero everywhere else on every rank (set_rates_to_zero, evolve.F90:430): the result is the all-reduce's.
    if (ctx->sparse_exchange && ctx->nsrc > 0 && ctx->nsrc <= 65535 /* grid.y of k_pack_boxes */ && ctx->sparse_valid) {
        HIP_TRY(hipSetDevice(ctx->prm.device));
        int rc = gather_nbox_all(ctx);
        if (rc) return rc;
        long long total = 0;
        std::vector<BoxDesc> desc(ctx->nsrc);
        for (int i = 0; i < ctx->nsrc; ++i) {
            BoxDesc &d = desc[i];
            for (int a = 0; a < 3; ++a) { const int m = (ctx->srcpos[3 * (size_t)i + a] - 1) % p.mesh[a]; d.c[a] = m < 0 ? m + p.mesh[a] : m; }
            d.nbox = ctx->nbox_all[i]; d.off = total;
            total += visited_for_nbox(ctx, d.nbox);
        }
        if ((double)total <= ctx->sparse_fraction * (double)ctx->ncell) {
            if (total > 0) {
                if ((size_t)total > ctx->pack_cap) {
                    hipFree(ctx->d_pack); ctx->d_pack = nullptr; ctx->pack_cap = 0;
                    const size_t cap = std::max<size_t>((size_t)total, (size_t)(ctx->sparse_fraction * (double)ctx->ncell));
                    HIP_TRY(hipMalloc(&ctx->d_pack, cap * sizeof(double)));
                    ctx->pack_cap = cap;
                }
                if (ctx->nsrc > ctx->boxdesc_cap) {
                    hipFree(ctx->d_boxdesc); ctx->d_boxdesc = nullptr; ctx->boxdesc_cap = 0;
                    if (ctx->h_boxdesc) { hipHostFree(ctx->h_boxdesc); ctx->h_boxdesc = nullptr; }
                    HIP_TRY(hipMalloc(&ctx->d_boxdesc, (size_t)ctx->nsrc * sizeof(BoxDesc)));
                    HIP_TRY(hipHostMalloc((void **)&ctx->h_boxdesc, (size_t)ctx->nsrc * sizeof(BoxDesc)));
                    ctx->boxdesc_cap = ctx->nsrc;
                }
                // through the pinned staging copy (gather_nbox_all above ended with a stream wait: the previous call's copy has read it)
                memcpy(ctx->h_boxdesc, desc.data(), desc.size() * sizeof(BoxDesc));
                HIP_TRY(hipMemcpyAsync(ctx->d_boxdesc, ctx->h_boxdesc, desc.size() * sizeof(BoxDesc), hipMemcpyHostToDevice, ctx->stream));
                int nb_max = 0;
                for (const BoxDesc &d : desc) nb_max = std::max(nb_max, d.nbox);
                const long long vmax = visited_for_nbox(ctx, nb_max);
                const dim3 grid((unsigned)std::min<long long>((vmax + 255) / 256, 4096), (unsigned)ctx->nsrc), blk(256);
                for (int w = 4; w <= (ctx->thermal ? 5 : 4); ++w) {               // phih_grid, phiheat_grid (evolve.F90:599, :604-609)
                    hipLaunchKernelGGL(k_pack_boxes<false>, grid, blk, 0, ctx->stream, p.mesh[0], p.mesh[1], p.mesh[2], ctx->hl[0], ctx->hl[1],
                                       ctx->hl[2], ctx->hr[0], ctx->hr[1], ctx->hr[2], p.subboxsize, ctx->d_boxdesc, (double *)ctx->grid[w], ctx->d_pack);
                    if (ctx->ar(ctx->ar_user, ctx->d_pack, (size_t)total, (void *)ctx->stream) != 0) FAIL(C2R_ECALLBACK, "all-reduce callback failed");
                    hipLaunchKernelGGL(k_pack_boxes<true>, grid, blk, 0, ctx->stream, p.mesh[0], p.mesh[1], p.mesh[2], ctx->hl[0], ctx->hl[1],
                                       ctx->hl[2], ctx->hr[0], ctx->hr[1], ctx->hr[2], p.subboxsize, ctx->d_boxdesc, (double *)ctx->grid[w], ctx->d_pack);
                }
                HIP_TRY(hipGetLastError());
            }
            ++ctx->xchg_sparse;
            ctx->xchg_bytes_last = (total * (ctx->thermal ? 2 : 1) + ctx->nsrc) * (long long)sizeof(double);
            ctx->xchg_bytes_total += ctx->xchg_bytes_last;
            return C2R_OK;
        }
    }
    if (ctx->ar(ctx->ar_user, ctx->grid[4], ctx->ncell, (void *)ctx->stream) != 0) FAIL(C2R_ECALLBACK, "all-reduce callback failed");
    if (ctx->thermal && ctx->ar(ctx->ar_user, ctx->grid[5], ctx->ncell, (void *)ctx->stream) != 0)     // evolve.F90:604-609
        FAIL(C2R_ECALLBACK, "all-reduce callback failed");
    ctx->xchg_bytes_last = ((long long)ctx->ncell * (ctx->thermal ? 2 : 1) + (ctx->sparse_exchange ? ctx->nsrc : 0)) * (long long)sizeof(double);
    ctx->xchg_bytes_total += ctx->xchg_bytes_last;
    return C2R_OK;
}

int c2r_set_exchange_overlap(c2r_ctx *c, int32_t on)
{
    if (!c) return C2R_EINVAL;
    C(c)->exchange_overlap = on != 0;
    return C2R_OK;
}

int c2r_exchange_stats(c2r_ctx *c, int64_t *calls, int64_t *sparse_calls, int64_t *bytes_last, int64_t *bytes_total)
{
    if (!c) return C2R_EINVAL;
    Ctx *ctx = C(c);
    if (calls) *calls = ctx->xchg_calls;
    if (sparse_calls) *sparse_calls = ctx->xchg_sparse;
    if (bytes_last) *bytes_last = ctx->xchg_bytes_last;
    if (bytes_total) *bytes_total = ctx->xchg_bytes_total;
    return C2R_OK;
}

}  // extern "C"
