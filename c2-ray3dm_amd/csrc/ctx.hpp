// The context behind the C ABI of include/c2ray_hip.h and what the library's translation units share:
//   api.hip         life cycle, setters, array transfer, the host-array forms of the piecewise entries
//   sweep.hip       the per-shell launch schedule of the source sweep (BatchSweep), pass over all sources, per-cell entry
//   exchange.hip    ranks: source shares (static / LPT), the (sparse) all-reduce of the rates, slab geometry
//   chemistry.hip   the global pass, the photon-statistics sums, the fixed-order reductions
//   evolve_loop.hip one outer iteration (c2r_iterate) and the evolve3D loop, device-resident and host-pointer forms
#pragma once
#include "../../include/c2ray_hip.h"
#include "../../include/c2ray_constants.h"
#include "kernels_common.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

namespace c2r {

struct BoxDesc;                       // kernels_exchange.hpp

constexpr int kSumBlocks = 1024;      // fixed grid of every deterministic reduction
constexpr int kFoldLossMax = 64;      // up to this many active sources k_box_decide also sums the last shell's loss partials
constexpr int kFusedQmax = 10;        // sub-boxes ending at q <= this run in k_sweep_box_fused (one launch per sub-box)
constexpr int kMaxSlabRanks = 64;     // slab chemistry keeps every rank's slab offsets in fixed arrays (evolve3d_worker)
constexpr int kFewSources = 32;       // a batch of up to this many sources is nothing but launch latency (see sweep_batch)

// What ONE batch of sources in flight reads and writes -- a "chain": its own stream, plane sets, staging block, active
// lists, loss partials, counts read back per sub-box.  A rank's sources of a pass are normally ONE chain on the context's
// stream.  With 64 - 512 sources a pass is split into several chains whose launches interleave on separate streams
// (sweep.hip run_chains): shell q+1 of a source depends on shell q of the SAME source only, so while one chain's launch
// drains (its last workgroups, the gap to the next launch) another chain's fills the GPU -- what one GPU's share of a
// multi-GPU run looks like (125 sources: launches of 10 - 200 us).
constexpr int kMaxChains = 4;
struct SweepScratch {
    hipStream_t stream = nullptr; bool own_stream = false;   // chain 0 runs on the context's stream
    int cap = 0;                                             // sources this chain's arrays hold
    double *d_planes = nullptr;                              // [cap][2][6][P][P]
    double *d_gbox = nullptr;   // deterministic mode: [cap][2][ncell]
    double *d_gbox_h = nullptr; // ... and the per-source heating rates of a non-isothermal run
    int *d_srcpos_b = nullptr, *d_srcw_b = nullptr; double *d_nflux_b = nullptr, *d_nflux_x = nullptr;
    // one device block + one pinned staging block hold the small per-batch arrays (one copy per batch)
    char *d_batch = nullptr, *h_batch = nullptr; size_t batch_bytes = 0;
    char *d_batch_init = nullptr; std::vector<char> batch_image;   // fused iteration: the state block a small batch starts from, on the device / as last sent
    char *d_hbatch = nullptr;                                // h_batch as the device sees it (k_box_decide_small writes results there)
    int *d_active[2] = {nullptr, nullptr};
    int *d_nactive = nullptr;                                // [2]: length of d_active[0/1]
    int *h_nactive = nullptr;                                // pinned, one slot per sub-box
    int *d_hnactive = nullptr;                               // the same slots as the device sees them
    std::vector<hipEvent_t> ev_box;                          // 'slot written' events
    hipEvent_t ev_done = nullptr;                            // 'this chain's launches of the pass have run'
    double *d_loss_partial = nullptr, *d_loss_acc = nullptr, *d_final_loss = nullptr;
    int *d_final_nbox = nullptr;
    int *d_perm = nullptr, *h_perm = nullptr;                // [3][cap]: the batch's traceable sources sorted by wrapped position along each axis (k_sweep_shell_xcd)
};

struct Ctx {
    c2r_params prm{};
    hipStream_t stream = nullptr;
    bool own_stream = false;
    size_t ncell = 0;
    // grids: 0 ndens(f32) 1 xh 2 xh_av 3 xh_intermed 4 phih_grid; non-isothermal runs: 5 phiheat_grid 6 temperature_grid (3 x f32 per cell)
    // drivers built with -DALLFRAC (c2r_params.allfrac): 7 xh0, 8 xh_av0, 9 xh_intermed0 -- the stored neutral fractions, the (:,:,:,0) halves
    void *grid[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool allfrac = false;
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
    // the second source type of photoion_rates (builds with use_xray_SED=.true.): c2r_set_xray
    bool xray = false; std::vector<double> nflux_x; double *d_xthick = nullptr, *d_xthin = nullptr;
    bool have_xheat = false; double *d_xhthick = nullptr, *d_xhthin = nullptr;          // its heating tables (c2r_set_xray_heat_tables: non-isothermal runs)
    int nsrc = 0, rank = 0, nranks = 1;
    bool explicit_share = false; std::vector<int32_t> share;   // c2r_set_source_share: this rank's sources
    std::vector<int32_t> last_nbox;                              // final sub-box count per local source, last pass
    int box_hint = 0;                                            // largest of them: how far the next pass is expected to go
    // hipGraph of a small batch's launch sequence up to box_hint (see sweep_batch); gen counts everything that
    // the captured kernel arguments depend on (tables, buffers, stream, scratch, physics switches; NOT the step's scalars: sync_step)
    struct BatchGraph { hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; unsigned long long gen = 0; int count = 0, n_active = 0, hint = 0;
                        bool fused = false; bool stats = false; const void *acc = nullptr;   // acc: the accumulator the captured launches add into
    };
    std::map<int, BatchGraph> graphs;                            // key: 2 x (first source of the batch) + (fused iteration ? 1 : 0)
    // A chain's launch sequence of a whole pass as ONE replayed hipGraph (sweep.hip run_chains): the sub-boxes the chain's previous
    // pass went through, every launch sized for the count that pass left (+ a margin); the decision kernels guard the sizes on the
    // device (k_box_decide next_bound) and the host finishes eagerly whatever the sequence did not cover.  bounds[nbox] (1..H):
    // sources the launches of sub-box nbox were sized for.  profile[k]: sources active after sub-box k of the chain's last pass
    // (profile[0]: the traceable sources it started with); empty: not known.
    struct ChainGraph { hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; unsigned long long gen = 0; int count = 0, n_active = 0, shape_count = 0;
                        const void *acc = nullptr; bool perm = false; int H = 0, launches = 0; std::vector<int> bounds, profile, profile_prev; int passes_since_capture = 0; };
    std::map<int, ChainGraph> chain_graphs;                      // key: first local source of the chain
    bool chain_tail = true;                                      // option chain_tail = 0: the iteration's tail is never enqueued behind the replayed chains' gate (A/B)
    bool chain_graph = true;                                     // option chain_graph = 0: chains are always driven launch by launch
    long long chain_replays = 0, chain_halts = 0, chain_eager = 0;   // chain passes replayed / replays the device halted (a launch too small) / driven launch by launch
    unsigned long long gen = 1;
    long long captures = 0;                                      // launch sequences captured so far (c2r_info; tests: a new time step must not add one)
    bool use_graph = true;                                       // option graph = 0: never
    bool fused_iter = true;                                      // option fused_iter = 0: c2r_iterate always runs its three steps in turn
    bool fold_source_cell = true;                                // option fold_source_cell = 0: k_source_cells is always its own launch
    bool pair_shells = true;                                     // option pair_shells = 0: never two shells per launch
    // cost-balanced distribution inside the library (c2r_set_balance): every rank learns every source's last
    // sub-box count through the all-reduce callback and computes the same LPT partition
    bool balance = false, auto_share = false;                    // auto_share: `share` was set by the balancer, not the caller
    std::vector<int32_t> nbox_all;                               // [nsrc] after a balanced pass (empty: not known yet)
    double *d_nbox_all = nullptr, *h_nbox_all = nullptr; int nbox_all_cap = 0;   // device buffer + pinned staging
    c2r_allreduce_fn ar = nullptr;
    void *ar_user = nullptr;
    // sparse exchange of the rates (c2r_allreduce_rates): while the sources' final sub-boxes cover a small part of the mesh only
    // the boxes travel (option sparse_exchange = 0: always the whole grid; sparse_fraction: the largest sum of box volumes, in
    // units of the mesh, that still goes packed)
    bool sparse_exchange = true; double sparse_fraction = 0.5;
    long long pass_id = 0, nbox_all_pass = -1;                   // passes swept so far; the pass nbox_all was gathered for
    bool sparse_valid = false;                                   // the rates in phih_grid are those of ONE c2r_pass_sources over rates the library had zeroed
    bool rates_clean = false;                                    // phih_grid (phiheat_grid) zeroed by the library and not written since
    double *d_pack = nullptr; size_t pack_cap = 0; BoxDesc *d_boxdesc = nullptr, *h_boxdesc = nullptr; int boxdesc_cap = 0;
    long long xchg_calls = 0, xchg_sparse = 0, xchg_bytes_last = 0, xchg_bytes_total = 0, xchg_overlapped = 0;
    // the exchange overlapped with the sweep (c2r_set_exchange_overlap; sweep.hip pass_sources_impl): a pass as two halves of the
    // rank's sources into two pairs of accumulators, the first half's all-reduce on a second stream while the second is swept
    bool exchange_overlap = false; int overlap_min_sources = 2 * kFewSources;    // (option exchange_overlap_min)
    double *d_phih2 = nullptr, *d_phih2_T = nullptr, *acc_phih = nullptr, *acc_phih_T = nullptr;   // acc_*: what the launches being enqueued add into (null: phih_grid / d_phih_T)
    hipStream_t xstream = nullptr; hipEvent_t ev_half = nullptr, ev_xdone = nullptr;
    long long rates_reduced_pass = -1;                           // the pass whose rates in phih_grid are already summed over the ranks
    // slab chemistry (c2r_set_slab_chemistry): reduce-scatter of the rates by z-slabs, the global pass on the own slab,
    // all-gather of its outputs -- instead of the all-reduce and a replicated global pass
    c2r_reduce_scatter_fn rs = nullptr; c2r_allgather_fn ag = nullptr; void *slab_user = nullptr;
    // sources handed out on request (c2r_set_source_queue: do_grid_master / do_grid_slave, master_slave.F90:124-330)
    c2r_next_sources_fn queue_next = nullptr; void *queue_user = nullptr; int queue_chunk = 0;
    c2r_iteration_fn iter_hook = nullptr;
    void *iter_user = nullptr;
    // sweep geometry
    int hl[3], hr[3], nbox_max = 0, Qmax = 0, R = 0, P = 1;
    size_t PP = 1;
    int tiles_cap = 0;          // ceil(P*P/256): most tiles any face plane needs
    // sweep scratch: `nchains` chains of `chain_cap` sources each are in flight at once (SweepScratch); batch_cap = their sum
    int batch_cap = 0, batch_want = 0;
    SweepScratch sc[kMaxChains];
    int nchains = 1, chain_cap = 0;
    int chains_env = 0;         // option chains = n: force n chains (0: the rule of choose_chains)
    int batch_cap_opt = 0;      // option batch_cap = n: at most n sources in flight per round (0: what the scratch budget allows)
    hipEvent_t ev_prepared = nullptr;      // 'the pass's inputs are ready' on the context's stream, for the other chains' streams
    bool stream_hint = false;   // non-temporal cache policy of k_sweep_shell: meshes whose n_HI array outgrows the L2s
    bool fuse_small = true;     // option fuse_small = 0 disables the fused first sub-boxes (A/B tests)
    bool sched_hint = true;     // option sched_hint = 0: always one sub-box ahead (see sweep_batch)
    // the time step's scalars as the kernels read them (kernels.hpp StepBlock + ShellStep[Qmax + 1]): device copy, the image last sent
    char *d_step = nullptr, *h_step = nullptr; std::vector<char> step_image; double step_dt = 0.0;      // h_step: pinned staging of the copy
    hipEvent_t ev_step = nullptr; bool ev_step_recorded = false;                                          // ... and 'the copy has read it'

    double *d_photon_loss = nullptr; long long *d_sum_nbox = nullptr;
    // reductions
    double *d_sum_partial = nullptr, *d_sum_out = nullptr, *d_stat_partial = nullptr;
    unsigned long long *d_conv = nullptr; unsigned int *d_chemfail = nullptr;
    struct HostScalars { double sum; double photon_loss; long long sum_nbox; unsigned long long conv; unsigned int chemfail; double pair[2]; double four[4];
                         unsigned long long seq; double before[4], after[4]; } *h_sc = nullptr,  // pinned
      *d_hsc = nullptr;                       // ... and its device alias: kernels store results there directly
    double *d_dbg = nullptr, *d_pair = nullptr;
    int *d_gate = nullptr;                                       // run_chains: 0 iff the replayed chains of a pass have all run to their end (k_chain_gate)
    long long chain_tails = 0;                                   // iterations whose tail (totals, fold, global pass) ran behind that gate: one host wait
    unsigned long long *d_seq = nullptr, seq_seen = 0;           // passes completed by fused iterations (k_pass_final counts, the host polls h_sc->seq)
    bool spin_wait = true;                                       // option spin_wait = 0: always hipStreamSynchronize
    // XCD-aware, plane-ordered block mapping of the far shells (k_sweep_shell_xcd): option xcd_order = 0 never, 1 always where it
    // can run, -1 (default): where at least xcd_min_per_plane sources share a mesh plane and face sign (sources / mesh planes)
    int xcd_order = -1; double xcd_min_per_plane = 1.5; double xcd_min_alive = 0.9; int xcd_qmin = 16; int xcd_min_sources = 64;
    long long xcd_launches = 0;
    bool poll_wait = true;                                       // option poll_wait = 0: the sub-box counts are waited for with hipEventSynchronize alone
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

inline size_t grid_bytes(const Ctx *ctx, int which) { return ctx->ncell * (which == 0 ? sizeof(float) : (which == 6 ? 3 * sizeof(float) : sizeof(double))); }

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

// ---- sweep.hip ---------------------------------------------------------------------------------------------------
void free_sweep_scratch(Ctx *ctx);
int n_local_sources(const Ctx *ctx);
int ensure_sweep_scratch(Ctx *ctx, int want);
int sync_step(Ctx *ctx);
void prof_begin(Ctx *ctx, std::vector<std::pair<hipEvent_t, hipEvent_t>> &pool, size_t &used);
void prof_end(Ctx *ctx, std::vector<std::pair<hipEvent_t, hipEvent_t>> &pool, size_t &used);
void prof_collect(Ctx *ctx);
int sweep_prepare(Ctx *ctx, bool zero_rates = false, bool copy_batch = false);
int sweep_finish(Ctx *ctx, const int *gate = nullptr);
long long visited_for_nbox(const Ctx *ctx, int nbox);
int upload_lls_grid(Ctx *ctx, const float *lls_grid);          // LLS_grid and its (x,y)-transposed replica
// do_grid over this rank's sources (c2r_pass_sources; iterate_impl: fz / no_wait)
int pass_sources_impl(Ctx *ctx, FusedIter *fz, double *photon_loss, int64_t *sum_nbox, int64_t *visited, bool no_wait = false, FusedIter *chain_tail = nullptr);
// ---- exchange.hip ------------------------------------------------------------------------------------------------
void slab_of(const Ctx *ctx, int r, int P, size_t *off, size_t *cnt);
void balance_before_pass(Ctx *ctx);
int balance_after_pass(Ctx *ctx);
// ---- chemistry.hip -----------------------------------------------------------------------------------------------
int photon_sums_launch(Ctx *ctx, int which_l, int which_r, double *dst);
int global_pass_enqueue(Ctx *ctx, double dt, double *stats_dst, size_t cell_off, size_t cell_cnt, const int *gate, bool count_pass = false);
int global_pass_impl(Ctx *ctx, double dt, int64_t *conv_flag, double *sum_xh1, double *stats_dst, size_t cell_off, size_t cell_cnt);
int final_temperature_enqueue(Ctx *ctx);                       // set_final_temperature_point (temperature_module.F90:172-183)
// ---- api.hip -----------------------------------------------------------------------------------------------------
int check_ready(Ctx *ctx);
// host array `host` of the driver <-> device array `which` (1 xh, 2 xh_av, 3 xh_intermed: with -DALLFRAC drivers the host array is
// (mesh,0:1) -- its first half goes to / comes from array which + 6; everything else: one plain copy), on the context's stream, no wait
int copy_in(Ctx *ctx, int which, const void *host);
int copy_out(Ctx *ctx, int which, void *host);

}  // namespace c2r
