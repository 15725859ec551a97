// Host side of the C ABI declared in include/c2ray_hip.h: context, device buffers, the
// per-shell launch schedule of the source sweep, the global pass and the evolve3D outer loop.
#include "../../include/c2ray_hip.h"
#include "../../include/c2ray_constants.h"
#include "kernels.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

using namespace c2r;

namespace {

constexpr int kSumBlocks = 1024;      // fixed grid of every deterministic reduction
constexpr int kFoldLossMax = 64;      // up to this many active sources k_box_decide also sums the last shell's loss partials
constexpr int kFusedQmax = 10;        // sub-boxes ending at q <= this run in k_sweep_box_fused (one launch per sub-box)
constexpr int kMaxSlabRanks = 64;     // slab chemistry keeps every rank's slab offsets in fixed arrays (evolve3d_worker)
constexpr int kFewSources = 32;       // a batch of up to this many sources is nothing but launch latency (see sweep_batch)

struct Ctx {
    c2r_params prm{};
    hipStream_t stream = nullptr;
    bool own_stream = false;
    size_t ncell = 0;
    // grids: 0 ndens(f32) 1 xh 2 xh_av 3 xh_intermed 4 phih_grid; non-isothermal runs: 5 phiheat_grid 6 temperature_grid (3 x f32 per cell)
    void *grid[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    // non-isothermal runs (c2r_set_thermal): heating tables, cooling curve, transposed heating accumulator
    bool thermal = false;
    c2r_thermal_params tprm{};
    double *d_hthick = nullptr, *d_hthin = nullptr, *d_cool = nullptr, *d_heat_T = nullptr;
    double zred = 0.0; bool have_zred = false;
    // per-pass inputs of the sweep (owned): n_HI per cell and its (x,y)-transposed replica for the +-x
    // faces, and the transposed Gamma accumulator of those faces
    double *d_nhi = nullptr, *d_nhi_T = nullptr, *d_phih_T = nullptr;
    // optional per-cell inputs of the non-default physics switches
    int lls_type = 1; double R_max_LLS = 0.0;
    float *d_lls = nullptr, *d_lls_T = nullptr, *d_clump = nullptr;
    bool  own[5] = {false, false, false, false, false};
    double *d_thick = nullptr, *d_thin = nullptr;
    v2f64 *d_logtab = nullptr;                               // log10_tab's {r_i, -log10 r_i}
    v2f64 *d_odtab = nullptr;                                // fast mode: {r_i, table position of tau = 1/r_i} (tau_od)
    bool fast = false;                                       // c2r_params.sweep_mode == C2R_SWEEP_FAST
    bool have_tables = false, have_step = false;
    double dr[3] = {0, 0, 0}, vol = 0, lls = 0, temper = 0;
    float clumping = 1.0f;
    std::vector<int32_t> srcpos;   // 3 x nsrc
    std::vector<double>  nflux;
    int nsrc = 0, rank = 0, nranks = 1;
    bool explicit_share = false; std::vector<int32_t> share;   // c2r_set_source_share: this rank's sources
    std::vector<int32_t> last_nbox;                              // final sub-box count per local source, last pass
    int box_hint = 0;                                            // largest of them: how far the next pass is expected to go
    // hipGraph of a small batch's launch sequence up to box_hint (see sweep_batch); gen counts everything that
    // the captured kernel arguments depend on (tables, buffers, stream, scratch, physics switches; NOT the step's scalars: sync_step)
    struct BatchGraph { hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; unsigned long long gen = 0; int count = 0, n_active = 0, hint = 0;
                        bool fused = false; bool stats = false; };
    std::map<int, BatchGraph> graphs;                            // key: 2 x (first source of the batch) + (fused iteration ? 1 : 0)
    unsigned long long gen = 1;
    long long captures = 0;                                      // launch sequences captured so far (c2r_info; tests: a new time step must not add one)
    bool use_graph = true;                                       // C2R_GRAPH=0: never (experiments)
    bool fused_iter = true;                                      // C2R_FUSED_ITER=0: c2r_iterate always runs its three steps in turn (experiments)
    bool fold_source_cell = true;                                // C2R_FOLD_SOURCE_CELL=0: k_source_cells is always its own launch (experiments)
    bool pair_shells = true;                                     // C2R_PAIR_SHELLS=0: never two shells per launch (experiments)
    // cost-balanced distribution inside the library (c2r_set_balance): every rank learns every source's last
    // sub-box count through the all-reduce callback and computes the same LPT partition
    bool balance = false, auto_share = false;                    // auto_share: `share` was set by the balancer, not the caller
    std::vector<int32_t> nbox_all;                               // [nsrc] after a balanced pass (empty: not known yet)
    double *d_nbox_all = nullptr, *h_nbox_all = nullptr; int nbox_all_cap = 0;   // device buffer + pinned staging
    c2r_allreduce_fn ar = nullptr;
    void *ar_user = nullptr;
    // sparse exchange of the rates (c2r_allreduce_rates): while the sources' final sub-boxes cover a small part of the mesh only
    // the boxes travel (C2R_SPARSE_EXCHANGE=0: always the whole grid; C2R_SPARSE_FRACTION: the largest sum of box volumes, in
    // units of the mesh, that still goes packed)
    bool sparse_exchange = true; double sparse_fraction = 0.5;
    long long pass_id = 0, nbox_all_pass = -1;                   // passes swept so far; the pass nbox_all was gathered for
    bool sparse_valid = false;                                   // the rates in phih_grid are those of ONE c2r_pass_sources over rates the library had zeroed
    bool rates_clean = false;                                    // phih_grid (phiheat_grid) zeroed by the library and not written since
    double *d_pack = nullptr; size_t pack_cap = 0; BoxDesc *d_boxdesc = nullptr, *h_boxdesc = nullptr; int boxdesc_cap = 0;
    long long xchg_calls = 0, xchg_sparse = 0, xchg_bytes_last = 0, xchg_bytes_total = 0;
    // slab chemistry (c2r_set_slab_chemistry): reduce-scatter of the rates by z-slabs, the global pass on the own slab,
    // all-gather of its outputs -- instead of the all-reduce and a replicated global pass
    c2r_reduce_scatter_fn rs = nullptr; c2r_allgather_fn ag = nullptr; void *slab_user = nullptr;
    c2r_iteration_fn iter_hook = nullptr;
    void *iter_user = nullptr;
    // sweep geometry
    int hl[3], hr[3], nbox_max = 0, Qmax = 0, R = 0, P = 1;
    size_t PP = 1;
    int tiles_cap = 0;          // ceil(P*P/256): most tiles any face plane needs
    // sweep scratch (one batch of sources)
    int batch_cap = 0, batch_want = 0;
    bool stream_hint = false;   // non-temporal cache policy of k_sweep_shell: meshes whose n_HI array outgrows the L2s
    bool fuse_small = true;     // C2R_FUSE_SMALL=0 disables the fused first sub-boxes (experiments, A/B tests)
    bool sched_hint = true;     // C2R_SCHED_HINT=0: always one sub-box ahead (experiments, see sweep_batch)
    double *d_planes = nullptr;
    // the time step's scalars as the kernels read them (kernels.hpp StepBlock + ShellStep[Qmax + 1]): device copy, the image last sent
    char *d_step = nullptr, *h_step = nullptr; std::vector<char> step_image; double step_dt = 0.0;      // h_step: pinned staging of the copy
    hipEvent_t ev_step = nullptr; bool ev_step_recorded = false;                                          // ... and 'the copy has read it'

    int *d_srcpos_b = nullptr, *d_srcw_b = nullptr; double *d_nflux_b = nullptr;
    double *d_gbox = nullptr;   // deterministic mode: [batch_cap][2][ncell]
    double *d_gbox_h = nullptr; // ... and the per-source heating rates of a non-isothermal run
    // one device block + one pinned staging block hold the small per-batch arrays below (one copy per batch)
    char *d_batch = nullptr, *h_batch = nullptr; size_t batch_bytes = 0;
    char *d_batch_init = nullptr; std::vector<char> batch_image;   // fused iteration: the state block a small batch starts from, on the device / as last sent
    char *d_hbatch = nullptr;                                // h_batch as the device sees it (k_box_decide_small writes results there)
    int *d_active[2] = {nullptr, nullptr};
    int *d_nactive = nullptr;                                // [2]: length of d_active[0/1]
    int *h_nactive = nullptr;                                // pinned, one slot per sub-box
    int *d_hnactive = nullptr;                               // the same slots as the device sees them
    std::vector<hipEvent_t> ev_box;                          // 'slot written' events
    double *d_loss_partial = nullptr, *d_loss_acc = nullptr, *d_final_loss = nullptr;
    int *d_final_nbox = nullptr;
    double *d_photon_loss = nullptr; long long *d_sum_nbox = nullptr;
    // reductions
    double *d_sum_partial = nullptr, *d_sum_out = nullptr, *d_stat_partial = nullptr;
    unsigned long long *d_conv = nullptr; unsigned int *d_chemfail = nullptr;
    struct HostScalars { double sum; double photon_loss; long long sum_nbox; unsigned long long conv; unsigned int chemfail; double pair[2]; double four[4];
                         unsigned long long seq; double before[4], after[4]; } *h_sc = nullptr,  // pinned
      *d_hsc = nullptr;                       // ... and its device alias: kernels store results there directly
    double *d_dbg = nullptr, *d_pair = nullptr;
    unsigned long long *d_seq = nullptr, seq_seen = 0;           // passes completed by fused iterations (k_pass_final counts, the host polls h_sc->seq)
    bool spin_wait = true;                                       // C2R_SPIN_WAIT=0: always hipStreamSynchronize (experiments)
    double *h_it4 = nullptr, *d_hit4 = nullptr;   // pinned [C2R_MAX_ITER_LOG][4]: per-iteration photon-statistics sums, written by the device
    // profiling
    int prof = 0;               // 0 off; 1 an event pair around every k_sweep_shell launch; 2 one pair per sub-box
    std::vector<int> ev_sweep_cnt;   // k_sweep_shell launches covered by each pair
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_sweep, ev_chem;
    size_t ev_sweep_used = 0, ev_chem_used = 0;
    double prof_sweep_ms = 0, prof_chem_ms = 0; long long prof_sweep_n = 0, prof_chem_n = 0;
    // host arrays of the driver that c2r_evolve3d has page-locked (they are allocated once and live for
    // the whole run: evolve_data.F90:75-90), so the per-step transfers run at DMA speed
    std::map<const void *, size_t> pinned;
    std::string err;
    // c2r_info: how the device was chosen, the sweep mode, warnings (e.g. C2R_DEVICE_AUTO without a local-rank variable)
    bool device_auto = false; std::string device_var, info_device, info_warn, info;
};

// the polling loop's pause: the x86 hint, its aarch64 counterpart, nothing elsewhere
inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#else
    asm volatile("" ::: "memory");
#endif
}

inline Ctx *C(c2r_ctx *c) { return reinterpret_cast<Ctx *>(c); }
inline const Ctx *C(const c2r_ctx *c) { return reinterpret_cast<const Ctx *>(c); }

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            char buf_[512];                                                                  \
            snprintf(buf_, sizeof buf_, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr,        \
                     hipGetErrorString(e_));                                                 \
            ctx->err = buf_;                                                                 \
            return (int)e_;                                                                  \
        }                                                                                    \
    } while (0)

#define FAIL(code, msg) do { ctx->err = (msg); return (code); } while (0)

size_t grid_bytes(const Ctx *ctx, int which) { return ctx->ncell * (which == 0 ? sizeof(float) : (which == 6 ? 3 * sizeof(float) : sizeof(double))); }

void free_sweep_scratch(Ctx *ctx)
{
    hipFree(ctx->d_planes); hipFree(ctx->d_gbox); hipFree(ctx->d_gbox_h); hipFree(ctx->d_batch); hipFree(ctx->d_batch_init); hipFree(ctx->d_loss_partial);
    ctx->d_batch_init = nullptr; ctx->batch_image.clear();
    if (ctx->h_batch) hipHostFree(ctx->h_batch);
    ctx->d_batch = ctx->h_batch = nullptr; ctx->d_nactive = nullptr;
    ctx->d_planes = nullptr; ctx->d_gbox = nullptr; ctx->d_gbox_h = nullptr; ctx->d_srcpos_b = nullptr; ctx->d_srcw_b = nullptr; ctx->d_nflux_b = nullptr;
    ctx->d_active[0] = ctx->d_active[1] = nullptr;
    ctx->d_loss_partial = ctx->d_loss_acc = ctx->d_final_loss = nullptr; ctx->d_final_nbox = nullptr;
    ctx->batch_cap = 0; ctx->batch_want = 0;
}

int n_local_sources(const Ctx *ctx)
{
    if (ctx->explicit_share) return (int)ctx->share.size();
    return ctx->nsrc > ctx->rank ? (ctx->nsrc - ctx->rank + ctx->nranks - 1) / ctx->nranks : 0;
}

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
    if (const char *e = getenv("C2R_BATCH_CAP")) cap = std::max(1, std::min(cap, atoi(e)));      // experiments
    cap = std::min(cap, 65535);                 // grid.z of k_sweep_shell
    HIP_TRY(hipMalloc(&ctx->d_planes, (size_t)cap * 2 * 6 * ctx->PP * sizeof(double)));
    if (ctx->prm.deterministic_rates) HIP_TRY(hipMalloc(&ctx->d_gbox, (size_t)cap * 2 * ctx->ncell * sizeof(double)));
    if (ctx->prm.deterministic_rates && ctx->thermal) HIP_TRY(hipMalloc(&ctx->d_gbox_h, (size_t)cap * 2 * ctx->ncell * sizeof(double)));
    HIP_TRY(hipMalloc(&ctx->d_loss_partial, (size_t)cap * 6 * ctx->tiles_cap * sizeof(double)));
    // small per-batch arrays: doubles first, then ints
    //   nflux[cap] final_loss[cap] loss_acc[cap] | srcpos[3cap] srcw[3cap] active0[cap] active1[cap] final_nbox[cap] nactive[2]
    ctx->batch_bytes = (size_t)cap * 3 * sizeof(double) + ((size_t)cap * 9 + 2) * sizeof(int);
    HIP_TRY(hipMalloc(&ctx->d_batch, ctx->batch_bytes));
    HIP_TRY(hipMalloc(&ctx->d_batch_init, ctx->batch_bytes));
    ctx->batch_image.clear();
    HIP_TRY(hipHostMalloc((void **)&ctx->h_batch, ctx->batch_bytes, hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer((void **)&ctx->d_hbatch, ctx->h_batch, 0));
    {
        double *d = reinterpret_cast<double *>(ctx->d_batch);
        ctx->d_nflux_b = d; ctx->d_final_loss = d + cap; ctx->d_loss_acc = d + 2 * (size_t)cap;
        int *i = reinterpret_cast<int *>(d + 3 * (size_t)cap);
        ctx->d_srcpos_b = i; ctx->d_srcw_b = i + 3 * (size_t)cap; ctx->d_active[0] = i + 6 * (size_t)cap;
        ctx->d_active[1] = i + 7 * (size_t)cap; ctx->d_final_nbox = i + 8 * (size_t)cap; ctx->d_nactive = i + 9 * (size_t)cap;
    }
    ctx->batch_cap = cap;
    ctx->batch_want = want;
    ++ctx->gen;                                   // every captured launch points into the old scratch
    return C2R_OK;
}

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

KParams make_kparams(const Ctx *ctx)
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
    k.nhi = ctx->d_nhi; k.nhi_T = ctx->d_nhi_T; k.phih = (double *)ctx->grid[4]; k.phih_T = ctx->d_phih_T;
    k.gbox = ctx->d_gbox; k.gbox_h = ctx->thermal ? ctx->d_gbox_h : nullptr;
    k.lls_type = ctx->lls_type; k.R_max2 = ctx->R_max_LLS * ctx->R_max_LLS; k.lls = ctx->d_lls; k.lls_T = ctx->d_lls_T;
    k.thick = ctx->d_thick; k.thin = ctx->d_thin; k.logtab = ctx->d_logtab;
    k.hthick = ctx->d_hthick; k.hthin = ctx->d_hthin; k.heat = (double *)ctx->grid[5]; k.heat_T = ctx->d_heat_T;
    k.tau_heat_limit = ctx->tprm.tau_heat_limit;
    k.odtab = ctx->d_odtab;
    k.od_per_e = (double)(0.301029995663981195213738894724493027L / (long double)p.dlogtau);
    k.od_per_ln = (double)(0.434294481903251827651128918916605082L / (long double)p.dlogtau);
    k.srcpos = ctx->d_srcpos_b; k.srcw = ctx->d_srcw_b; k.normflux = ctx->d_nflux_b; k.planes = ctx->d_planes;
    return k;
}

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

// A whole outer iteration around ONE small batch (c2r_iterate): what precedes the batch's launches (the rates set to
// zero, n_HI prepared) and what follows them (Gamma of the +-x faces folded back, the global pass and its reductions)
// are recorded into the batch's hipGraph, the tail gated on the device by "no source is active after sub-box `hint`" --
// the steady state of an outer iteration.  The host then waits ONCE per iteration instead of four times (each wait is
// 15-25 us of idle GPU in a 0.4 ms iteration, profiles/r03_launch_bound).  tail_done: the gated tail has run.
struct FusedIter {
    double dt = 0.0;
    bool stats = false;
    bool batch_in_prepare = false;                // set by sweep_batch while it captures: pre() also restores the batch's state block
    std::function<int()> pre;                     // enqueue: zero rates + sweep_prepare
    std::function<int(const int *gate)> post;     // enqueue: sweep_finish + global pass, each launch a no-op unless *gate == 0 (null: unconditional)
    bool tail_done = false;
};

// One batch of sources through the sweep -- local sources [first, first+count) of this rank's list: the staging block, the
// launches of a sub-box (source cells, fused first sub-boxes, shells and look-ahead pairs, loss sums, the decision), the
// captured launch sequence of a small batch and the wait behind a fused iteration, the run-ahead schedule.  sweep_batch()
// below is its only user.  dbg: optional device N^3 array receiving coldensh_out (single-source test path).
struct BatchSweep {
    Ctx *ctx; const c2r_params &p;
    const int first, count; const bool first_of_pass; double *const dbg; FusedIter *const fz;
    const size_t cap; hipStream_t st; KParams k;
    // the pinned staging block (layout of ensure_sweep_scratch)
    double *h_nf, *h_fl; int *h_pos, *h_posw, *h_act, *h_na, *h_fnb;
    int n_active = 0;
    int cur = 0, last_bps = 0;     // which active list is current; size of the last shell's loss partials per source (0: none), for k_box_decide
    int totals_at_box = 0;         // fused iteration: the sub-box whose decision also writes the batch's totals (0: none)

    BatchSweep(Ctx *c, int first_, int count_, bool first_of_pass_, double *dbg_, FusedIter *fz_)
        : ctx(c), p(c->prm), first(first_), count(count_), first_of_pass(first_of_pass_), dbg(dbg_), fz(fz_),
          cap((size_t)c->batch_cap), st(c->stream), k(make_kparams(c))
    {
        h_nf = reinterpret_cast<double *>(ctx->h_batch); h_fl = h_nf + cap;
        h_pos = reinterpret_cast<int *>(h_nf + 3 * cap); h_posw = h_pos + 3 * cap; h_act = h_pos + 6 * cap; h_na = h_pos + 9 * cap;
        h_fnb = h_pos + 8 * cap;   // the batch's results travel back through the same block (same layout as the device block)
    }

    // ---- 1. the staging block: sources, wrapped positions, the initial active list ---------------------------------
    // (it is next written by the next sweep_batch, after this one's final synchronize; it is uploaded by the graph's copy
    // node, by k_prepare_nhi from its device image, or directly)
    void stage()
    {
        memset(ctx->h_batch, 0, ctx->batch_bytes);                     // loss_acc = 0, final_nbox = 0, active lists
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
            const double flux = h_nf[i] * p.S_star;
            if (flux > p.loss_fraction * flux && can_trace) h_act[n_active++] = i;
            else h_fl[i] = flux;                                       // loop never entered: loss = initial value
        }
        h_na[0] = n_active; h_na[1] = 0;
    }

    // ---- 2. the launches of one sub-box -----------------------------------------------------------------------------
    // what the launches of sub-box nbox share
    struct Box {
        int nbox, bound;               // the sub-box; upper bound of the device's active count (sizes the grids)
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
        sa.active = ctx->d_active[cur]; sa.n_active = ctx->d_nactive + cur;
        sa.loss_partial = ctx->d_loss_partial; sa.dbg_cdout = dbg;
        return sa;
    }

    void launch_source_cells(const Box &bx)
    {
        const int (&boxR)[3] = bx.boxR, (&boxL)[3] = bx.boxL;
        const int bound = bx.bound;
        {
            if (ctx->thermal)
                hipLaunchKernelGGL(k_source_cells<true>, dim3((bound + 63) / 64), dim3(64), 0, st, k, bound,
                                   ctx->d_active[cur], boxR[0], boxR[1], boxR[2], boxL[0], boxL[1], boxL[2],
                                   ctx->d_loss_acc, dbg);
            else
                hipLaunchKernelGGL(k_source_cells<false>, dim3((bound + 63) / 64), dim3(64), 0, st, k, bound,
                                   ctx->d_active[cur], boxR[0], boxR[1], boxR[2], boxL[0], boxL[1], boxL[2],
                                   ctx->d_loss_acc, dbg);
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
                if (ctx->thermal)
                    hipLaunchKernelGGL(k_source_cells<true>, dim3((bound + 63) / 64), dim3(64), 0, st, k, bound, ctx->d_active[cur],
                                       boxR[0], boxR[1], boxR[2], boxL[0], boxL[1], boxL[2], ctx->d_loss_acc, dbg);
                else
                    hipLaunchKernelGGL(k_source_cells<false>, dim3((bound + 63) / 64), dim3(64), 0, st, k, bound, ctx->d_active[cur],
                                       boxR[0], boxR[1], boxR[2], boxL[0], boxL[1], boxL[2], ctx->d_loss_acc, dbg);
            }
        }
        if (ba.nshell > 0) {
            ba.active = ctx->d_active[cur]; ba.n_active = ctx->d_nactive + cur; ba.loss_acc = ctx->d_loss_acc;
            // one workgroup per source: 256 / 512 / 1024 threads by the largest shell; with many sources 512 at most (two
            // workgroups per CU hide each other's shell-to-shell latency: cold 256^3 x 1000 0.973 -> 0.939 ms per
            // iteration).  By the batch's INITIAL count: the block size shapes the loss sums, which must not depend on timing.
            int bt = most <= 256 ? 256 : (most <= 512 ? 512 : 1024);
            if (n_active >= 256) bt = std::min(bt, 512);
            const dim3 grid(bound), blk(bt);
            // (not in the k_sweep_shell launch timing of c2r_profile: a different kernel, 21^3 cells per source)
#define C2R_LAUNCH_FUSED_H(D, L, H) do { if (ctx->fast) hipLaunchKernelGGL((k_sweep_box_fused<D, L, true, H>), grid, blk, 0, st, k, ba); \
                                else hipLaunchKernelGGL((k_sweep_box_fused<D, L, false, H>), grid, blk, 0, st, k, ba); } while (0)
#define C2R_LAUNCH_FUSED(D, L) do { if (ctx->thermal) C2R_LAUNCH_FUSED_H(D, L, true); else C2R_LAUNCH_FUSED_H(D, L, false); } while (0)
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
        if (ctx->prof == 2) prof_begin(ctx, ctx->ev_sweep, ctx->ev_sweep_used);
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
#define C2R_LAUNCH_PAIR(D, L) do { if (ctx->thermal) C2R_LAUNCH_PAIR_H(D, L, true); else C2R_LAUNCH_PAIR_H(D, L, false); } while (0)
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
            if (ctx->prof == 1) prof_begin(ctx, ctx->ev_sweep, ctx->ev_sweep_used);
            {
                const dim3 grid(sa.tiles_max, 6, bound), blk(kBlock);
#define C2R_LAUNCH_SWEEP_H(D, L, H) do { \
    if (ctx->fast) { if (ctx->stream_hint) hipLaunchKernelGGL((k_sweep_shell_fast<D, L, true, H>), grid, blk, 0, st, k, sa); \
                     else hipLaunchKernelGGL((k_sweep_shell_fast<D, L, false, H>), grid, blk, 0, st, k, sa); } \
    else if (ctx->stream_hint) hipLaunchKernelGGL((k_sweep_shell<D, L, true, H>), grid, blk, 0, st, k, sa); \
    else hipLaunchKernelGGL((k_sweep_shell<D, L, false, H>), grid, blk, 0, st, k, sa); } while (0)
#define C2R_LAUNCH_SWEEP(D, L) do { if (ctx->thermal) C2R_LAUNCH_SWEEP_H(D, L, true); else C2R_LAUNCH_SWEEP_H(D, L, false); } while (0)
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
            if (ctx->prof == 1) { prof_end(ctx, ctx->ev_sweep, ctx->ev_sweep_used); ctx->ev_sweep_cnt.push_back(1); }
            // the partials of the sub-box's last shell are summed by k_box_decide itself when few sources are active (one
            // launch less where launches are all there is); with many, its single workgroup would read 6 x tiles partials
            // for every source (2.5 MB through one CU: 89 us per sub-box, 1.6 % of the bench step) -- one block per source then
            // (decided by the batch's INITIAL active count: `bound` depends on when the host happens to see a count arrive,
            // and the two paths round differently -- the photon loss, which feeds the keep/retire decision, must not)
            const bool fold = n_active <= kFoldLossMax;
            if (sa.has_boundary && (q < q1 || !fold))
                hipLaunchKernelGGL(k_loss_reduce, dim3(bound), dim3(256), 0, st, ctx->d_active[cur], ctx->d_nactive + cur,
                                   ctx->d_loss_partial, 6 * sa.tiles_max, ctx->d_loss_acc);
            if (q == q1 && sa.has_boundary && fold) last_bps = 6 * sa.tiles_max;
        }
        if (ctx->prof == 2) { prof_end(ctx, ctx->ev_sweep, ctx->ev_sweep_used); ctx->ev_sweep_cnt.push_back(in_box); }
    }

    // k_box_decide: which sources go on to the next sub-box (evolve_source.F90:128-131)
    void launch_decision(const Box &bx)
    {
        const int nbox = bx.nbox;
        const int can_grow = (p.subboxsize * nbox < ctx->hr[2]) && (p.subboxsize * nbox < ctx->hl[2]);
        if (n_active <= 64) {
            // one wave decides; at the sub-box a fused iteration's graph ends with it also leaves the batch's totals and
            // results (SmallTotals) -- host_final_*: the staging block's final_nbox / final_loss through its mapped alias
            SmallTotals tot{};
            if (nbox == totals_at_box) {
                tot.on = 1; tot.nsrc = count; tot.photon_loss = ctx->d_photon_loss; tot.sum_nbox = ctx->d_sum_nbox;
                tot.host_loss = &ctx->d_hsc->photon_loss; tot.host_nbox = &ctx->d_hsc->sum_nbox;
                tot.host_final_loss = reinterpret_cast<double *>(ctx->d_hbatch) + cap;
                tot.host_final_nbox = reinterpret_cast<int *>(reinterpret_cast<double *>(ctx->d_hbatch) + 3 * cap) + 8 * cap;
            }
            hipLaunchKernelGGL(k_box_decide_small, dim3(1), dim3(64), 0, st, ctx->d_active[cur], ctx->d_nactive + cur,
                               ctx->d_active[1 - cur], ctx->d_nactive + (1 - cur), ctx->d_hnactive + nbox, ctx->d_nflux_b,
                               p.S_star, p.loss_fraction, can_grow, nbox, ctx->d_loss_acc, ctx->d_final_loss, ctx->d_final_nbox,
                               (const double *)ctx->d_loss_partial, last_bps, tot);
        } else
        hipLaunchKernelGGL(k_box_decide, dim3(1), dim3(1024), 0, st, ctx->d_active[cur], ctx->d_nactive + cur,
                           ctx->d_active[1 - cur], ctx->d_nactive + (1 - cur), ctx->d_hnactive + nbox, ctx->d_nflux_b,
                           p.S_star, p.loss_fraction, can_grow, nbox, ctx->d_loss_acc, ctx->d_final_loss, ctx->d_final_nbox,
                           (const double *)ctx->d_loss_partial, last_bps);
    }

    // every launch of sub-box nbox for `bound` sources at most (no host wait, no event); flips `cur`
    int enqueue_box(const int nbox, const int bound)
    {
        Box bx{};
        bx.nbox = nbox; bx.bound = bound;
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
        bx.det = ctx->d_gbox != nullptr;
        // (the fused first sub-box does the source cells itself: one launch less)
        if (nbox == 1 && !(bx.fused_box && ctx->fold_source_cell)) launch_source_cells(bx);
        last_bps = 0;
        if (bx.fused_box) launch_fused_box(bx); else launch_shells(bx);
        launch_decision(bx);
        cur = 1 - cur;
        return C2R_OK;
    }

    int enqueue_totals(std::vector<int> *nbox_out, std::vector<double> *loss_out)
    {
        hipLaunchKernelGGL(k_batch_totals, dim3(1), dim3(1024), 0, st, count, ctx->d_final_loss, ctx->d_final_nbox,
                           ctx->d_photon_loss, ctx->d_sum_nbox, first_of_pass ? 1 : 0, &ctx->d_hsc->photon_loss,
                           &ctx->d_hsc->sum_nbox);
        HIP_TRY(hipGetLastError());
        if (nbox_out) HIP_TRY(hipMemcpyAsync(h_fnb, ctx->d_final_nbox, (size_t)count * sizeof(int), hipMemcpyDeviceToHost, st));
        if (loss_out) HIP_TRY(hipMemcpyAsync(h_fl, ctx->d_final_loss, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, st));
        return C2R_OK;
    }
    // deterministic rates: the per-source grids are summed in source order once every source has its final sub-box
    void gamma_reduce(const int *gate)
    {
        if (ctx->d_gbox)
            hipLaunchKernelGGL(k_gamma_reduce, dim3((p.mesh[0] + 255) / 256, p.mesh[1], p.mesh[2]), dim3(256), 0, st, k, count,
                               ctx->d_final_nbox, p.subboxsize, (double *)ctx->grid[4], ctx->thermal ? (double *)ctx->grid[5] : nullptr, gate);
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
            if (rc == C2R_OK && !fuse_iter) rc = (int)hipMemcpyAsync(ctx->d_batch, ctx->h_batch, ctx->batch_bytes, hipMemcpyHostToDevice, st);
            cur = 0;
            for (int nbox = 1; nbox <= hint && nbox <= ctx->nbox_max && rc == C2R_OK; ++nbox) rc = enqueue_box(nbox, n_active);
            if (fuse_iter && rc == C2R_OK) {
                // the batch's totals and results, then the gated rest of the iteration: d_nactive[cur] is the count the
                // last decision left (cur has been flipped by it)
                gamma_reduce(ctx->d_nactive + cur);
                rc = fz->post(ctx->d_nactive + cur);
            }
            totals_at_box = 0;
            const hipError_t e = hipStreamEndCapture(st, &bg.graph);
            if (rc == C2R_OK && e == hipSuccess && hipGraphInstantiate(&bg.exec, bg.graph, nullptr, nullptr, 0) == hipSuccess) {
                bg.gen = ctx->gen; bg.count = count; bg.n_active = n_active; bg.hint = hint;
                bg.fused = fuse_iter; bg.stats = fuse_iter && fz->stats;
            } else {
                if (bg.graph) { hipGraphDestroy(bg.graph); bg.graph = nullptr; }
                bg.exec = nullptr;
                (void)hipGetLastError();
                ctx->use_graph = false;            // this runtime / stream cannot capture: eager from now on
            }
        } else { (void)hipGetLastError(); ctx->use_graph = false; }
    }

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
                if (__atomic_load_n(&ctx->h_nactive[done], __ATOMIC_ACQUIRE) > 0) break;
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
              bg.fused == fuse_iter && (!fuse_iter || bg.stats == fz->stats)))
            capture(bg, fuse_iter, hint);
        if (bg.exec) {
            const int done = std::min(hint, ctx->nbox_max);
            if (bg.fused) {
                ctx->h_nactive[done] = -1;                        // (so that a stale zero is not taken for this launch's count)
                // the device image of the state block: sent only when it differs from what was sent last (steady state: never)
                if (ctx->batch_image.size() != ctx->batch_bytes || memcmp(ctx->batch_image.data(), ctx->h_batch, ctx->batch_bytes) != 0) {
                    ctx->batch_image.assign(ctx->h_batch, ctx->h_batch + ctx->batch_bytes);
                    // (from the pinned block itself: it is not touched again before this iteration's kernels have run)
                    HIP_TRY(hipMemcpyAsync(ctx->d_batch_init, ctx->h_batch, ctx->batch_bytes, hipMemcpyHostToDevice, st));
                }
            }
            HIP_TRY(hipGraphLaunch(bg.exec, st));
            uploaded = true;
            pre_run = bg.fused;
            cur = done & 1;
            bool arrived = false;
            { const int rc = wait_fused(bg, done, arrived); if (rc) return rc; }
            known = done; bound = ctx->h_nactive[done];
            first_box = done + 1;
            if (bg.fused && arrived) { ctx->seq_seen += 1; bound = 0; fz->tail_done = true; }   // the gate was open: the whole iteration has run
        } else cur = 0;
    }
    if (fz && !pre_run) { const int rc = fz->pre(); if (rc) return rc; }
    if (!uploaded) HIP_TRY(hipMemcpyAsync(ctx->d_batch, ctx->h_batch, ctx->batch_bytes, hipMemcpyHostToDevice, st));
    for (int nbox = first_box; nbox <= ctx->nbox_max && bound > 0; ++nbox) {
        { const int rc = enqueue_box(nbox, bound); if (rc) return rc; }
        HIP_TRY(hipEventRecord(ctx->ev_box[nbox], st));
        // counts that have already arrived (never blocks)
        while (known < nbox && hipEventQuery(ctx->ev_box[known + 1]) == hipSuccess) bound = ctx->h_nactive[++known];
        // blocking read-back: the box's own count where the previous pass ended, the previous box's beyond
        // (many sources: always the previous box's -- one sub-box stays in flight from the first box on)
        const int need = !few ? nbox - 1 : (nbox == hint ? nbox : (nbox > hint ? nbox - 1 : 0));
        if (need > known) {
            HIP_TRY(hipEventSynchronize(ctx->ev_box[need]));
            known = need; bound = ctx->h_nactive[need];
        }
    }
    if (!(fz && fz->tail_done)) {          // (the fused iteration's graph has done this already)
        gamma_reduce(nullptr);
        { const int rc = enqueue_totals(nbox_out, loss_out); if (rc) return rc; }
        HIP_TRY(hipStreamSynchronize(st));
    }
    if (nbox_out) nbox_out->assign(h_fnb, h_fnb + count);
    if (loss_out) loss_out->assign(h_fl, h_fl + count);
    return C2R_OK;
}

int sweep_batch(Ctx *ctx, int first, int count, bool first_of_pass, double *dbg, std::vector<int> *nbox_out,
                std::vector<double> *loss_out, FusedIter *fz = nullptr)
{
    BatchSweep bs(ctx, first, count, first_of_pass, dbg, fz);
    return bs.run(nbox_out, loss_out);
}

// +-x faces read (x,y)-transposed replicas so that their waves, which run along y, touch unit
// stride: refresh the replicas before a pass, fold their Gamma back after it.
// zero_rates: also set_rates_to_zero (evolve.F90:430-440) -- and every clearing inside the one kernel instead of memsets
// (the fused iteration, where a launch more or less is what counts)
int sweep_prepare(Ctx *ctx, bool zero_rates = false, bool copy_batch = false)
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
    if (copy_batch) { wc.src = (const unsigned *)ctx->d_batch_init; wc.dst = (unsigned *)ctx->d_batch; wc.n = (unsigned)(ctx->batch_bytes / 4); }
    hipLaunchKernelGGL(k_prepare_nhi, g, dim3(256), 0, ctx->stream, p.mesh[0], p.mesh[1], p.mesh[2], p.epsilon,
                       (const float *)ctx->grid[0], (const double *)ctx->grid[2], ctx->d_nhi, ctx->d_nhi_T, z, wc);
    HIP_TRY(hipGetLastError());
    if (!zero_rates) {
        HIP_TRY(hipMemsetAsync(ctx->d_phih_T, 0, grid_bytes(ctx, 4), ctx->stream));
        if (ctx->thermal) HIP_TRY(hipMemsetAsync(ctx->d_heat_T, 0, grid_bytes(ctx, 5), ctx->stream));
    }
    return C2R_OK;
}

int sweep_finish(Ctx *ctx, const int *gate = nullptr)
{
    const c2r_params &p = ctx->prm;
    // phih_T is [k][i][j]: transposing it back swaps the roles of the two mesh extents
    const dim3 g((p.mesh[1] + 31) / 32, (p.mesh[0] + 31) / 32, p.mesh[2]);
    hipLaunchKernelGGL((k_transpose_xy<double, true>), g, dim3(256), 0, ctx->stream, p.mesh[1], p.mesh[0], p.mesh[2],
                       (const double *)ctx->d_phih_T, (double *)ctx->grid[4], gate);
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

// z-slab of rank r of P: whole z-planes, the first (N3 mod P) ranks one plane more; in cells
void slab_of(const Ctx *ctx, int r, int P, size_t *off, size_t *cnt)
{
    const size_t n3 = (size_t)ctx->prm.mesh[2], plane = (size_t)ctx->prm.mesh[0] * ctx->prm.mesh[1];
    const size_t base = n3 / (size_t)P, rem = n3 % (size_t)P;
    const size_t z0 = (size_t)r * base + std::min<size_t>((size_t)r, rem), nz = base + ((size_t)r < rem ? 1 : 0);
    *off = z0 * plane; *cnt = nz * plane;
}

int check_ready(Ctx *ctx)
{
    // the context's allocations and launches belong to its device, whatever the caller made current since
    HIP_TRY(hipSetDevice(ctx->prm.device));
    if (!ctx->have_tables) FAIL(C2R_ESTATE, "c2r_set_tables has not been called");
    if (!ctx->have_step) FAIL(C2R_ESTATE, "c2r_set_step has not been called");
    return C2R_OK;
}

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

int balance_after_pass(Ctx *ctx)
{
    if (!ctx->balance || ctx->nranks <= 1 || !ctx->ar || (ctx->explicit_share && !ctx->auto_share) || ctx->nsrc == 0) return C2R_OK;
    return gather_nbox_all(ctx);
}

}  // namespace

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
    p->sweep_mode = C2R_SWEEP_EXACT;
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
    if (const char *e = getenv("C2R_FUSE_SMALL")) ctx->fuse_small = atoi(e) != 0;
    if (const char *e = getenv("C2R_SCHED_HINT")) ctx->sched_hint = atoi(e) != 0;
    if (const char *e = getenv("C2R_GRAPH")) ctx->use_graph = atoi(e) != 0;
    if (const char *e = getenv("C2R_FUSED_ITER")) ctx->fused_iter = atoi(e) != 0;
    if (const char *e = getenv("C2R_PAIR_SHELLS")) ctx->pair_shells = atoi(e) != 0;
    if (const char *e = getenv("C2R_SPIN_WAIT")) ctx->spin_wait = atoi(e) != 0;
    if (const char *e = getenv("C2R_FOLD_SOURCE_CELL")) ctx->fold_source_cell = atoi(e) != 0;
    if (const char *e = getenv("C2R_SPARSE_EXCHANGE")) ctx->sparse_exchange = atoi(e) != 0;
    if (const char *e = getenv("C2R_SPARSE_FRACTION")) ctx->sparse_fraction = std::max(0.0, atof(e));
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
    if (const char *e = getenv("C2R_STREAM_HINT")) ctx->stream_hint = atoi(e) != 0;
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
    HIP_TRY(hipHostMalloc((void **)&ctx->h_nactive, (size_t)(ctx->nbox_max + 2) * sizeof(int), hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer((void **)&ctx->d_hnactive, ctx->h_nactive, 0));
    ctx->ev_box.resize(ctx->nbox_max + 2);
    for (auto &e : ctx->ev_box) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipLaunchKernelGGL(k_load_code_object, dim3(1), dim3(1), 0, ctx->stream, (int *)nullptr);      // (loads the library's code object now)
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return C2R_OK;
}

void c2r_destroy(c2r_ctx *c)
{
    if (!c) return;
    Ctx *ctx = C(c);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    for (auto &kv : ctx->graphs) { if (kv.second.exec) hipGraphExecDestroy(kv.second.exec); if (kv.second.graph) hipGraphDestroy(kv.second.graph); }
    for (auto &kv : ctx->pinned) hipHostUnregister(const_cast<void *>(kv.first));
    free_sweep_scratch(ctx);
    for (int w = 0; w < 5; ++w) if (ctx->own[w]) hipFree(ctx->grid[w]);
    hipFree(ctx->grid[5]); hipFree(ctx->grid[6]); hipFree(ctx->d_hthick); hipFree(ctx->d_hthin); hipFree(ctx->d_cool); hipFree(ctx->d_heat_T);
    hipFree(ctx->d_thick); hipFree(ctx->d_thin); hipFree(ctx->d_logtab); hipFree(ctx->d_odtab);
    hipFree(ctx->d_nhi); hipFree(ctx->d_nhi_T); hipFree(ctx->d_phih_T); hipFree(ctx->d_step); hipFree(ctx->d_pack); hipFree(ctx->d_boxdesc);
    if (ctx->h_boxdesc) hipHostFree(ctx->h_boxdesc);
    hipFree(ctx->d_lls); hipFree(ctx->d_lls_T); hipFree(ctx->d_clump);
    if (ctx->h_nactive) hipHostFree(ctx->h_nactive);
    if (ctx->h_step) hipHostFree(ctx->h_step);
    if (ctx->ev_step) hipEventDestroy(ctx->ev_step);
    hipFree(ctx->d_photon_loss); hipFree(ctx->d_sum_nbox); hipFree(ctx->d_sum_partial); hipFree(ctx->d_sum_out); hipFree(ctx->d_stat_partial);
    hipFree(ctx->d_conv); hipFree(ctx->d_chemfail); hipFree(ctx->d_dbg); hipFree(ctx->d_pair); hipFree(ctx->d_seq); hipFree(ctx->d_nbox_all);
    if (ctx->h_nbox_all) hipHostFree(ctx->h_nbox_all);
    if (ctx->h_sc) hipHostFree(ctx->h_sc);
    if (ctx->h_it4) hipHostFree(ctx->h_it4);
    for (auto &e : ctx->ev_box) hipEventDestroy(e);
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
    if (type == 2) {
        const c2r_params &p = ctx->prm;
        if (!ctx->d_lls) { HIP_TRY(hipMalloc(&ctx->d_lls, grid_bytes(ctx, 0))); HIP_TRY(hipMalloc(&ctx->d_lls_T, grid_bytes(ctx, 0))); }
        HIP_TRY(hipMemcpyAsync(ctx->d_lls, lls_grid, grid_bytes(ctx, 0), hipMemcpyHostToDevice, ctx->stream));
        const dim3 g((p.mesh[0] + 31) / 32, (p.mesh[1] + 31) / 32, p.mesh[2]);
        hipLaunchKernelGGL((k_transpose_xy<float, false>), g, dim3(256), 0, ctx->stream, p.mesh[0], p.mesh[1], p.mesh[2],
                           (const float *)ctx->d_lls, ctx->d_lls_T);
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
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

int c2r_set_final_temperature(c2r_ctx *c)
{
    if (!c) return C2R_EINVAL;
    Ctx *ctx = C(c);
    if (!ctx->thermal) return C2R_OK;                     // temperature_module.F90:181: nothing to do when isothermal
    HIP_TRY(hipSetDevice(ctx->prm.device));
    hipLaunchKernelGGL(k_final_temperature, dim3(kSumBlocks), dim3(256), 0, ctx->stream, ctx->ncell, (float *)ctx->grid[6]);
    HIP_TRY(hipGetLastError());
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
    ctx->explicit_share = false; ctx->auto_share = false; ctx->share.clear(); ctx->last_nbox.clear(); ctx->nbox_all.clear(); ctx->box_hint = 0;
    // set-up belongs here, not in the first evolve3D of a run (the reference allocates in evolve_ini, evolve_data.F90:75-90): the
    // sweep scratch of this rank's share -- device planes, the pinned staging block -- is a few milliseconds of allocation calls
    if (nsrc > 0 && n_local_sources(ctx) > 0) {
        HIP_TRY(hipSetDevice(ctx->prm.device));
        const int rc = ensure_sweep_scratch(ctx, n_local_sources(ctx));
        if (rc) return rc;
    }
    return C2R_OK;
}

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
    if (!c || !ptr || which < 0 || which > 6) return C2R_EINVAL;
    if (which > 4 && !C(c)->thermal) return C2R_ESTATE;
    *ptr = C(c)->grid[which];
    return C2R_OK;
}

int c2r_upload(c2r_ctx *c, int32_t which, const void *host)
{
    if (!c || !host || which < 0 || which > 6) return C2R_EINVAL;
    Ctx *ctx = C(c);
    if (which > 4 && !ctx->thermal) FAIL(C2R_ESTATE, "arrays 5 and 6 exist in non-isothermal runs only (c2r_set_thermal)");
    HIP_TRY(hipMemcpyAsync(ctx->grid[which], host, grid_bytes(ctx, which), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (which == 4 || which == 5) { ctx->rates_clean = false; ctx->sparse_valid = false; }     // the caller's rates: not a pass over zeroed ones
    return C2R_OK;
}

int c2r_download(c2r_ctx *c, int32_t which, void *host)
{
    if (!c || !host || which < 0 || which > 6) return C2R_EINVAL;
    Ctx *ctx = C(c);
    if (which > 4 && !ctx->thermal) FAIL(C2R_ESTATE, "arrays 5 and 6 exist in non-isothermal runs only (c2r_set_thermal)");
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

// do_grid over this rank's sources.  fz (c2r_iterate, one small batch): the batch's graph also carries what precedes and
// follows the pass (sweep_batch); sweep_prepare / sweep_finish are then fz->pre / fz->post, not called here.
// no_wait (iterate_impl): return with sweep_finish enqueued and not waited for -- the caller enqueues the global pass behind
// it, waits once and reads the totals itself (they are in h_sc after that wait).
static int pass_sources_impl(Ctx *ctx, FusedIter *fz, double *photon_loss, int64_t *sum_nbox, int64_t *visited,
                             bool no_wait = false)
{
    int rc;
    if ((rc = sync_step(ctx))) return rc;
    balance_before_pass(ctx);
    const int nloc = n_local_sources(ctx);
    long long vis = 0;
    // the sparse exchange (c2r_allreduce_rates) is only right for ONE pass over rates the library itself had zeroed: everything
    // outside this pass's sub-boxes is then zero on every rank.  fz: the fused iteration zeroes them itself (fz->pre)
    ++ctx->pass_id; ctx->sparse_valid = ctx->rates_clean || fz != nullptr; ctx->rates_clean = false;
    ctx->last_nbox.clear();
    ctx->h_sc->photon_loss = 0.0; ctx->h_sc->sum_nbox = 0;      // (the stream is idle between calls)
    if (nloc > 0) {
        rc = ensure_sweep_scratch(ctx, nloc);
        if (rc) return rc;
        if (fz && nloc > ctx->batch_cap) FAIL(C2R_ESTATE, "fused iteration needs the sources in one batch");
        if (!fz && (rc = sweep_prepare(ctx))) return rc;
        std::vector<int> nb;
        for (int first = 0; first < nloc; first += ctx->batch_cap) {
            const int count = std::min(ctx->batch_cap, nloc - first);
            rc = sweep_batch(ctx, first, count, first == 0, nullptr, &nb, nullptr, fz);
            if (rc) return rc;
            for (int v : nb) { vis += visited_for_nbox(ctx, v); ctx->last_nbox.push_back(v); }
        }
        if (!fz && (rc = sweep_finish(ctx))) return rc;
        ctx->box_hint = 0;
        for (int v : ctx->last_nbox) ctx->box_hint = std::max(ctx->box_hint, v);
    } else if (fz && (rc = fz->pre())) return rc;
    if (visited) *visited = vis;
    if (no_wait) return C2R_OK;
    if (!fz) HIP_TRY(hipStreamSynchronize(ctx->stream));        // k_batch_totals stored the totals in h_sc
    prof_collect(ctx);
    if ((rc = balance_after_pass(ctx))) return rc;
    if (photon_loss) *photon_loss = ctx->h_sc->photon_loss;
    if (sum_nbox) *sum_nbox = ctx->h_sc->sum_nbox;
    if (visited) *visited = vis;
    return C2R_OK;
}

int c2r_pass_sources(c2r_ctx *c, double *photon_loss, int64_t *sum_nbox, int64_t *visited)
{
    if (!c) return C2R_EINVAL;
    Ctx *ctx = C(c);
    int rc = check_ready(ctx);
    if (rc) return rc;
    return pass_sources_impl(ctx, nullptr, photon_loss, sum_nbox, visited);
}

int c2r_allreduce_rates(c2r_ctx *c)
{
    if (!c) return C2R_EINVAL;
    Ctx *ctx = C(c);
    if (ctx->nranks <= 1 || !ctx->ar) return C2R_OK;
    const c2r_params &p = ctx->prm;
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

static int global_pass_impl(Ctx *ctx, double dt, int64_t *conv_flag, double *sum_xh1, double *stats_dst, size_t cell_off, size_t cell_cnt);

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
    const double xav1 = std::max(xh_av[idx], p.epsilon), xav0 = std::max(1.0 - xav1, p.epsilon);
    const double nhi = xav0 * (double)ndens[idx];
    const size_t idt = (size_t)pos[1] + (size_t)p.mesh[1] * ((size_t)pos[0] + (size_t)p.mesh[0] * (size_t)pos[2]);
    {
        double *hd = reinterpret_cast<double *>(ctx->h_batch);            // >= 3 doubles + 11 ints for a batch of one
        int *hi = reinterpret_cast<int *>(hd + 3);
        hd[0] = nflux; hd[1] = nhi;
        for (int d = 0; d < 3; ++d) { hi[d] = sp[d]; hi[3 + d] = spw[d]; }
        HIP_TRY(hipMemcpyAsync(ctx->d_srcpos_b, hi, 3 * sizeof(int), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(ctx->d_srcw_b, hi + 3, 3 * sizeof(int), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(ctx->d_nflux_b, hd, sizeof(double), hipMemcpyHostToDevice, st));
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
    KParams k = make_kparams(ctx);
    double *d_out = ctx->d_sum_out;                              // 4 doubles of device scratch
#define C2R_LAUNCH_CELL(L) do { if (ctx->thermal) hipLaunchKernelGGL((k_evolve0d_cell<L, true>), dim3(1), dim3(64), 0, st, k, sa, face, a, b, is_source ? 1 : 0, on_surface ? 1 : 0, cv[0], cv[1], cv[2], cv[3], d_out); \
                                else hipLaunchKernelGGL((k_evolve0d_cell<L, false>), dim3(1), dim3(64), 0, st, k, sa, face, a, b, is_source ? 1 : 0, on_surface ? 1 : 0, cv[0], cv[1], cv[2], cv[3], d_out); } while (0)
    switch (ctx->lls_type) { case 1: C2R_LAUNCH_CELL(1); break; case 2: C2R_LAUNCH_CELL(2); break; default: C2R_LAUNCH_CELL(3); break; }
#undef C2R_LAUNCH_CELL
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

// evolve0D_global(dt,pos,conv_flag) (evolve_point.F90:305-406) for ONE cell (pos 1-based) on the caller's arrays: the
// same kernel as the mesh-wide pass on a one-cell slab.  Non-isothermal contexts: phiheat_grid / temperature_grid (3 x f32 per
// cell) of the caller as well.  conv_flag is incremented when the cell has not converged.  Slow by construction.
int c2r_global_pass_cell_host(c2r_ctx *c, double dt, const int32_t pos[3], const float *ndens, const double *xh, double *xh_av,
                              double *xh_intermed, const double *phih_grid, const double *phiheat_grid, float *temperature_grid,
                              int32_t *conv_flag)
{
    if (!c || !pos || !ndens || !xh || !xh_av || !xh_intermed || !phih_grid) return C2R_EINVAL;
    Ctx *ctx = C(c);
    int rc = check_ready(ctx);
    if (rc) return rc;
    const c2r_params &p = ctx->prm;
    for (int d = 0; d < 3; ++d) if (pos[d] < 1 || pos[d] > p.mesh[d]) FAIL(C2R_EINVAL, "evolve0D_global: mesh position out of range");
    if (ctx->thermal && (!phiheat_grid || !temperature_grid)) FAIL(C2R_EINVAL, "non-isothermal run: evolve0D_global needs phiheat_grid and temperature_grid");
    const size_t idx = (size_t)(pos[0] - 1) + (size_t)p.mesh[0] * ((size_t)(pos[1] - 1) + (size_t)p.mesh[1] * (size_t)(pos[2] - 1));
    hipStream_t st = ctx->stream;
    HIP_TRY(hipMemcpyAsync((float *)ctx->grid[0] + idx, ndens + idx, sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync((double *)ctx->grid[1] + idx, xh + idx, sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync((double *)ctx->grid[2] + idx, xh_av + idx, sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync((double *)ctx->grid[4] + idx, phih_grid + idx, sizeof(double), hipMemcpyHostToDevice, st));
    if (ctx->thermal) {
        HIP_TRY(hipMemcpyAsync((double *)ctx->grid[5] + idx, phiheat_grid + idx, sizeof(double), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync((float *)ctx->grid[6] + 3 * idx, temperature_grid + 3 * idx, 3 * sizeof(float), hipMemcpyHostToDevice, st));
    }
    int64_t nonconv = 0;
    if ((rc = global_pass_impl(ctx, dt, &nonconv, nullptr, nullptr, idx, 1))) return rc;
    HIP_TRY(hipMemcpyAsync(xh_av + idx, (double *)ctx->grid[2] + idx, sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(xh_intermed + idx, (double *)ctx->grid[3] + idx, sizeof(double), hipMemcpyDeviceToHost, st));
    if (ctx->thermal)
        HIP_TRY(hipMemcpyAsync(temperature_grid + 3 * idx, (float *)ctx->grid[6] + 3 * idx, 3 * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (conv_flag) *conv_flag += (int32_t)nonconv;
    return C2R_OK;
}

int c2r_do_source_host(c2r_ctx *c, int32_t ns, const float *ndens, const double *xh_av, double *phih_grid,
                       double *coldensh_out, double *photon_loss_src, int32_t *nbox)
{
    if (!c || !ndens || !xh_av || !phih_grid) return C2R_EINVAL;
    Ctx *ctx = C(c);
    int rc;
    if ((rc = c2r_upload(c, 0, ndens))) return rc;
    if ((rc = c2r_upload(c, 2, xh_av))) return rc;
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
    if ((rc = c2r_upload(c, 2, xh_av))) return rc;
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

int c2r_global_pass_host(c2r_ctx *c, double dt, const float *ndens, const double *xh, double *xh_av,
                         double *xh_intermed, const double *phih_grid, int64_t *conv_flag)
{
    if (!c || !ndens || !xh || !xh_av || !xh_intermed || !phih_grid) return C2R_EINVAL;
    int rc;
    if ((rc = c2r_upload(c, 0, ndens))) return rc;
    if ((rc = c2r_upload(c, 1, xh))) return rc;
    if ((rc = c2r_upload(c, 2, xh_av))) return rc;
    if ((rc = c2r_upload(c, 4, phih_grid))) return rc;
    if ((rc = c2r_global_pass(c, dt, conv_flag, nullptr))) return rc;
    if ((rc = c2r_download(c, 2, xh_av))) return rc;
    return c2r_download(c, 3, xh_intermed);
}

int c2r_sum(c2r_ctx *c, int32_t which, double *sum)
{
    if (!c || !sum || which < 1 || which > 4) return C2R_EINVAL;
    Ctx *ctx = C(c);
    HIP_TRY(hipSetDevice(ctx->prm.device));
    hipLaunchKernelGGL(k_sum_partial, dim3(kSumBlocks), dim3(256), 0, ctx->stream, ctx->ncell,
                       (const double *)ctx->grid[which], ctx->d_sum_partial);
    hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(256), 0, ctx->stream, kSumBlocks, ctx->d_sum_partial, &ctx->d_hsc->sum);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    *sum = ctx->h_sc->sum;
    return C2R_OK;
}

// the four mesh sums of photonstatistics.F90 into dst[4] (device-visible: mapped pinned memory), no host wait
static int photon_sums_launch(Ctx *ctx, int which_l, int which_r, double *dst)
{
    const c2r_params &p = ctx->prm;
    // photonstatistics.F90:166-172: same rate coefficients as doric, host libm
    hipLaunchKernelGGL(k_photon_sums, dim3(kSumBlocks), dim3(256), 0, ctx->stream, ctx->ncell,
                       (const float *)ctx->grid[0], (const double *)ctx->grid[which_l],
                       (const double *)ctx->grid[which_r], p.abu_c, (double)ctx->clumping, (const float *)ctx->d_clump,
                       p.bh00, pow(ctx->temper / 1e4, p.albpow), p.colh0, sqrt(ctx->temper),
                       exp(-p.temph0 / ctx->temper), ctx->d_sum_partial, ctx->thermal ? (const float *)ctx->grid[6] : nullptr,
                       p.albpow, p.temph0);
    hipLaunchKernelGGL(k_sum_final, dim3(4), dim3(256), 0, ctx->stream, kSumBlocks, ctx->d_sum_partial, dst);
    HIP_TRY(hipGetLastError());
    return C2R_OK;
}

int c2r_photon_sums(c2r_ctx *c, int32_t which_l, int32_t which_r, double out[4])
{
    if (!c || !out || which_l < 1 || which_l > 3 || which_r < 1 || which_r > 3) return C2R_EINVAL;
    Ctx *ctx = C(c);
    int rc = check_ready(ctx);
    if (rc) return rc;
    if ((rc = photon_sums_launch(ctx, which_l, which_r, ctx->d_hsc->four))) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int m = 0; m < 4; ++m) out[m] = ctx->h_sc->four[m];
    return C2R_OK;
}

// the launches of a global pass, no host wait; gate: see k_transpose_xy
static int global_pass_enqueue(Ctx *ctx, double dt, double *stats_dst, size_t cell_off, size_t cell_cnt, const int *gate,
                               bool count_pass = false)
{
    const c2r_params &p = ctx->prm;
    ChemParams cp{};
    // (dt and the step's rate coefficients reach the kernel through the step block -- callers have run sync_step with this
    // dt --; the by-value copies below only document what the kernel overwrites at entry)
    cp.step = reinterpret_cast<const StepBlock *>(ctx->d_step);
    cp.dt = 0.0 * dt; cp.eps = p.epsilon; cp.min_frac_change = p.minimum_fractional_change;
    cp.min_frac_atoms = p.minimum_fraction_of_atoms; cp.abu_c = p.abu_c; cp.deltht_small = C2R_DELTHT_SMALL;
    cp.max_iter = p.max_chem_iter;
    // doric.f90:73,78 -- temperature is uniform (isothermal), so both rate coefficients are
    // per-call constants; evaluated with the host libm like the reference does at run time
    cp.bh00 = p.bh00; cp.clump = ctx->d_clump ? ctx->d_clump + cell_off : nullptr;
    cp.colh0 = p.colh0;
    cp.stat_partial = ctx->d_stat_partial;
    if (ctx->thermal) {
        const c2r_thermal_params &t = ctx->tprm;
        if (t.cosmological && !ctx->have_zred) FAIL(C2R_ESTATE, "non-isothermal run: c2r_set_redshift has not been called (cosmo_cool needs zred)");
        cp.temper = (float *)ctx->grid[6] + 3 * cell_off; cp.phiheat = (const double *)ctx->grid[5] + cell_off; cp.cool = ctx->d_cool;
        cp.cool_mintemp = t.cool_mintemp; cp.cool_dtemp = t.cool_dtemp; cp.cool_points = t.cool_points;
        cp.thermal_max_steps = t.thermal_max_steps;
        cp.k_B = t.k_B; cp.gamma1 = t.gamma1; cp.minitemp = t.minitemp; cp.rel_denergy = t.relative_denergy;
        cp.rate_floor = t.thermal_rate_floor; cp.time_tol = t.thermal_time_tol;
        cp.temph0 = p.temph0; cp.albpow = p.albpow;
        cp.tconv_rel = t.temp_conv_rel; cp.tconv_abs = t.temp_conv_abs;
    }
    prof_begin(ctx, ctx->ev_chem, ctx->ev_chem_used);
#define C2R_LAUNCH_GLOBAL(S, T) hipLaunchKernelGGL((k_global_pass<S, T>), dim3(kSumBlocks), dim3(256), 0, ctx->stream, cp, cell_cnt, \
                           (const float *)ctx->grid[0] + cell_off, (const double *)ctx->grid[1] + cell_off, (double *)ctx->grid[2] + cell_off, \
                           (double *)ctx->grid[3] + cell_off, (const double *)ctx->grid[4] + cell_off, ctx->d_sum_partial, ctx->d_conv, \
                           ctx->d_chemfail, gate)
    if (stats_dst) { if (ctx->thermal) C2R_LAUNCH_GLOBAL(true, true); else C2R_LAUNCH_GLOBAL(true, false); }
    else { if (ctx->thermal) C2R_LAUNCH_GLOBAL(false, true); else C2R_LAUNCH_GLOBAL(false, false); }
#undef C2R_LAUNCH_GLOBAL
    prof_end(ctx, ctx->ev_chem, ctx->ev_chem_used);
    if (stats_dst)
        hipLaunchKernelGGL(k_sum_final, dim3(4), dim3(256), 0, ctx->stream, kSumBlocks, ctx->d_stat_partial, stats_dst, gate);
    // last: its final store is the pass counter a fused iteration's host polls (count_pass)
    hipLaunchKernelGGL(k_pass_final, dim3(1), dim3(256), 0, ctx->stream, kSumBlocks, ctx->d_sum_partial, ctx->d_conv,
                       ctx->d_chemfail, &ctx->d_hsc->sum, &ctx->d_hsc->conv, &ctx->d_hsc->chemfail, gate,
                       count_pass ? ctx->d_seq : nullptr, count_pass ? &ctx->d_hsc->seq : nullptr);
    HIP_TRY(hipGetLastError());
    return C2R_OK;
}

// global_pass (evolve.F90:499-573); stats_dst (device-visible, 4 doubles, or null): the photon-statistics sums of
// (xh_intermed, xh_av) as the pass leaves them, from the same kernel
static int global_pass_impl(Ctx *ctx, double dt, int64_t *conv_flag, double *sum_xh1, double *stats_dst, size_t cell_off, size_t cell_cnt)
{
    if (cell_cnt == (size_t)-1) cell_cnt = ctx->ncell;        // (a slab [cell_off, cell_off+cell_cnt): slab chemistry)
    ctx->step_dt = dt;
    int rc = sync_step(ctx);
    if (rc) return rc;
    rc = global_pass_enqueue(ctx, dt, stats_dst, cell_off, cell_cnt, nullptr);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    prof_collect(ctx);
    if (conv_flag) *conv_flag = (int64_t)ctx->h_sc->conv;
    if (sum_xh1) *sum_xh1 = ctx->h_sc->sum;
    return C2R_OK;
}

int c2r_global_pass(c2r_ctx *c, double dt, int64_t *conv_flag, double *sum_xh1)
{
    if (!c) return C2R_EINVAL;
    Ctx *ctx = C(c);
    int rc = check_ready(ctx);
    if (rc) return rc;
    return global_pass_impl(ctx, dt, conv_flag, sum_xh1, nullptr, 0, (size_t)-1);
}

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
    bool can_fuse = ctx->fused_iter && ctx->nranks == 1 && !ctx->balance && ctx->use_graph && ctx->sched_hint && ctx->prof == 0 && nloc > 0 &&
                    nloc <= kFewSources && ctx->box_hint >= 1 &&
                    !(ctx->thermal && ctx->tprm.cosmological && !ctx->have_zred);
    if (can_fuse) {
        if ((rc = ensure_sweep_scratch(ctx, nloc))) return rc;
        can_fuse = nloc <= ctx->batch_cap;
    }
    auto zero_rates = [ctx]() -> int {
        HIP_TRY(hipMemsetAsync(ctx->grid[4], 0, grid_bytes(ctx, 4), ctx->stream));
        if (ctx->thermal) HIP_TRY(hipMemsetAsync(ctx->grid[5], 0, grid_bytes(ctx, 5), ctx->stream));     // evolve.F90:435
        ctx->rates_clean = true; ctx->sparse_valid = false;
        return C2R_OK;
    };
    double *four = stats_host ? ctx->d_hsc->four : nullptr;
    if (!can_fuse) {
        // the three steps, with one wait behind the global pass instead of one behind each of the last two
        if ((rc = zero_rates())) return rc;
        if ((rc = pass_sources_impl(ctx, nullptr, nullptr, nullptr, vis, ctx->nranks == 1 && !ctx->balance))) return rc;
        if ((rc = global_pass_impl(ctx, dt, conv, sum1, four, 0, (size_t)-1))) return rc;
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
        const double sum0 = (double)(float)ctx->ncell - sum1;                          // :184
        const double rel1 = sum1 > 0.0 ? fabs(sum1 - prev1) / sum1 : 1.0;
        const double rel0 = sum0 > 0.0 ? fabs(sum0 - prev0) / sum0 : 1.0;
        if (niter > 0 && niter <= C2R_MAX_ITER_LOG) {
            rep->it_rel_change_xh1[niter - 1] = rel1; rep->it_rel_change_xh0[niter - 1] = rel0;
            rep->it_sum_xh1[niter - 1] = sum1;
        }
        if (conv_flag < conv_criterion || (rel1 < p.convergence_fraction && rel0 < p.convergence_fraction)) {   // :212
            HIP_TRY(hipMemcpyAsync(ctx->grid[1], ctx->grid[3], grid_bytes(ctx, 1), hipMemcpyDeviceToDevice, ctx->stream));   // :218
            if (ctx->thermal)                                                          // :220 set_final_temperature_point
                hipLaunchKernelGGL(k_final_temperature, dim3(kSumBlocks), dim3(256), 0, ctx->stream, ctx->ncell, (float *)ctx->grid[6]);
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
        pin_host_array(ctx, host, grid_bytes(ctx, which));
        HIP_TRY(hipMemcpyAsync(ctx->grid[which], host, grid_bytes(ctx, which), hipMemcpyHostToDevice, ctx->stream));
        if (which == 4 || which == 5) { ctx->rates_clean = false; ctx->sparse_valid = false; }
        return C2R_OK;
    }
    int down(int which, void *host)
    {
        if (!host) return C2R_OK;
        pin_host_array(ctx, host, grid_bytes(ctx, which));
        HIP_TRY(hipMemcpyAsync(host, ctx->grid[which], grid_bytes(ctx, which), hipMemcpyDeviceToHost, ctx->stream));
        return C2R_OK;
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
