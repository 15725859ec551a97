// TEST INFRASTRUCTURE: a host WITHOUT Python, torch or MPI that drives the C ABI (include/c2ray_hip.h) with several
// ranks in one process -- one thread per rank, one context per rank -- the way an MPI build of the Fortran driver
// does with one process per rank (mpi.F90:83-160, master_slave.F90:74-96, evolve.F90:577-616):
//     c2r_create(device = rank mod visible devices) / c2r_set_rank + all-reduce callback / c2r_set_balance /
//     c2r_evolve3d (host-pointer entry, what the Fortran shim calls).
// Collective:  "rccl"  libc2ray_rccl.so (ncclAllReduce over xGMI; needs one device per rank),
//              "host"  a rank-ordered sum staged through host memory (any device count, also 1 GPU).
// Slab chemistry (c2r_set_slab_chemistry: reduce-scatter of the rates, global pass on the own z-slab, all-gather): optional 6th
// argument 1; host-staged reduce-scatter / all-gather in rank order, or the RCCL binding's.
// usage: mgpu_harness <in.bin> <out.bin> <nranks> <host|rccl> <balance 0|1> [slab 0|1]
//   in.bin : int32 mesh, nsrc | f64 dr, vol, lls, dt | f32 ndens[N^3] | f64 xh[N^3] | int32 srcpos[3 nsrc] | f64 normflux[nsrc]
//   out.bin: int32 niter, converged, nranks_used_rccl | int64 sum_nbox, conv_flag[niter] | f64 loss | f64 xh[N^3] | f64 phih[N^3]
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <atomic>
#include <vector>
#include "../../include/c2ray_hip.h"
#include "../../include/c2ray_rccl.h"

struct Problem {
    int32_t mesh, nsrc; double dr, vol, lls, dt;
    std::vector<float> ndens; std::vector<double> xh, xh0 /* allfrac: the stored neutral fraction, or empty */, normflux; std::vector<int32_t> srcpos;
    std::vector<double> thick, thin;
};
struct Shared {
    int nranks; bool rccl, balance, slab = false; const Problem *pb;
    pthread_barrier_t bar;
    std::vector<double *> stage; std::vector<size_t> stage_cap;   // host all-reduce: one PINNED staging buffer per rank
    unsigned char uid[C2R_RCCL_ID_BYTES];
    std::vector<int> rc; std::vector<std::string> err;
    c2r_report rep0; std::vector<double> xh0, xh0n, phih0;
    // C2R_HARNESS_QUEUE=chunk: sources handed out on request (c2r_set_source_queue) from these two counters, one per parity of the pass
    int queue_chunk = 0; std::atomic<int> queue[2]; std::vector<long long> swept;      // swept[rank]: sources the rank took, all passes
    int64_t xchg[4] = {0, 0, 0, 0};                               // rank 0's c2r_exchange_stats: calls, packed calls, bytes of the last call, bytes in all
    bool rotate = false;                                          // C2R_HARNESS_ROTATE_SUM=1: the host all-reduce sums element i starting at rank i mod nranks
    std::vector<unsigned long long> hash;                         // per rank: a hash of the bits of xh and phih_grid the step left (replicas must agree)
    std::string info0;
};
struct RankArg { Shared *sh; int rank; };

// c2r_allreduce_fn: sum over the ranks (threads) in rank order, staged through host memory
static int host_allreduce(void *user, void *dev_buf, size_t count, void *stream)
{
    RankArg *a = static_cast<RankArg *>(user);
    Shared *sh = a->sh;
    if (sh->stage_cap[a->rank] < count) {                      // (re)allocate between collectives only: no one reads it now
        if (sh->stage[a->rank]) (void)hipHostFree(sh->stage[a->rank]);
        if (hipHostMalloc((void **)&sh->stage[a->rank], 2 * count * sizeof(double)) != hipSuccess) return 1;   // [0,count) mine, [count,2count) sum
        sh->stage_cap[a->rank] = count;
    }
    double *mine = sh->stage[a->rank], *sum = mine + sh->stage_cap[a->rank];
    if (hipMemcpyAsync(mine, dev_buf, count * sizeof(double), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return 1;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return 1;
    pthread_barrier_wait(&sh->bar);
    for (size_t i = 0; i < count; ++i) sum[i] = 0.0;
    if (!sh->rotate) {
        for (int r = 0; r < sh->nranks; ++r)
            for (size_t i = 0; i < count; ++i) sum[i] = sum[i] + sh->stage[r][i];
    } else {
        // an OFFSET-dependent order, as a ring / tree collective has (RCCL's order depends on where in the buffer an element
        // lies): element i is summed starting at rank i mod nranks -- the same on every rank, different from element to element
        for (size_t i = 0; i < count; ++i)
            for (int k = 0; k < sh->nranks; ++k) sum[i] = sum[i] + sh->stage[(k + (int)(i % (size_t)sh->nranks)) % sh->nranks][i];
    }
    pthread_barrier_wait(&sh->bar);                  // everyone has read every staging buffer
    if (hipMemcpyAsync(dev_buf, sum, count * sizeof(double), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) return 1;
    return hipStreamSynchronize((hipStream_t)stream) == hipSuccess ? 0 : 1;
}

static int stage_for(Shared *sh, int rank, size_t doubles)
{
    if (sh->stage_cap[rank] < doubles) {
        if (sh->stage[rank]) (void)hipHostFree(sh->stage[rank]);
        if (hipHostMalloc((void **)&sh->stage[rank], 2 * doubles * sizeof(double)) != hipSuccess) return 1;
        sh->stage_cap[rank] = doubles;
    }
    return 0;
}
// c2r_reduce_scatter_fn: every rank stages its whole array; rank r sums ITS slab over the ranks in rank order
static int host_reduce_scatter(void *user, void *dev_buf, const size_t *off, const size_t *cnt, int32_t nranks, void *stream)
{
    RankArg *a = static_cast<RankArg *>(user);
    Shared *sh = a->sh;
    const size_t total = off[nranks - 1] + cnt[nranks - 1];
    if (stage_for(sh, a->rank, total)) return 1;
    double *mine = sh->stage[a->rank], *sum = mine + sh->stage_cap[a->rank];
    if (hipMemcpyAsync(mine, dev_buf, total * sizeof(double), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return 1;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return 1;
    pthread_barrier_wait(&sh->bar);
    const size_t o = off[a->rank], n = cnt[a->rank];
    for (size_t i = 0; i < n; ++i) sum[i] = 0.0;
    for (int r = 0; r < sh->nranks; ++r)
        for (size_t i = 0; i < n; ++i) sum[i] = sum[i] + sh->stage[r][o + i];
    pthread_barrier_wait(&sh->bar);
    if (hipMemcpyAsync(static_cast<double *>(dev_buf) + o, sum, n * sizeof(double), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) return 1;
    return hipStreamSynchronize((hipStream_t)stream) == hipSuccess ? 0 : 1;
}
// c2r_allgather_fn: every rank stages its own slab (bytes); every rank then copies the others' slabs up
static int host_allgather(void *user, void *dev_buf, const size_t *off, const size_t *cnt, int32_t nranks, void *stream)
{
    RankArg *a = static_cast<RankArg *>(user);
    Shared *sh = a->sh;
    const size_t total = off[nranks - 1] + cnt[nranks - 1];
    if (stage_for(sh, a->rank, (total + 7) / 8)) return 1;
    char *mine = reinterpret_cast<char *>(sh->stage[a->rank]);
    char *dev = static_cast<char *>(dev_buf);
    if (hipMemcpyAsync(mine + off[a->rank], dev + off[a->rank], cnt[a->rank], hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return 1;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return 1;
    pthread_barrier_wait(&sh->bar);
    for (int r = 0; r < sh->nranks; ++r)
        if (r != a->rank && cnt[r] &&
            hipMemcpyAsync(dev + off[r], reinterpret_cast<char *>(sh->stage[r]) + off[r], cnt[r], hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) return 1;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return 1;
    pthread_barrier_wait(&sh->bar);                  // nobody restages before everyone has copied
    return 0;
}

// the harness's source queue: first come, first served (what do_grid_master does with MPI_Send / MPI_Recv, master_slave.F90:124-230)
struct QueueArg { Shared *sh; int rank; long long last_pass = -1; };
static int next_sources(void *user, int64_t pass, int32_t want, int32_t *first, int32_t *count)
{
    QueueArg *q = static_cast<QueueArg *>(user);
    Shared *sh = q->sh;
    const int p = (int)(pass & 1);
    if (q->last_pass != pass) { q->last_pass = pass; sh->queue[p ^ 1].store(0); }     // (nobody uses the other counter during this pass)
    const int v = sh->queue[p].fetch_add(want);
    const int n = sh->pb->nsrc;
    if (v >= n) { *first = 0; *count = 0; return 0; }
    *first = v; *count = std::min<int>(want, n - v);
    sh->swept[q->rank] += *count;
    return 0;
}

#define TRY(expr) do { int rc_ = (expr); if (rc_ != 0) { sh->rc[rank] = rc_; sh->err[rank] = std::string(#expr) + ": " + (ctx ? c2r_last_error(ctx) : ""); goto done; } } while (0)

static void *rank_main(void *p)
{
    RankArg *a = static_cast<RankArg *>(p);
    Shared *sh = a->sh;
    const Problem &pb = *sh->pb;
    const int rank = a->rank;
    c2r_ctx *ctx = nullptr;
    int ndev = 0;
    (void)hipGetDeviceCount(&ndev);
    const size_t ncell = (size_t)pb.mesh * pb.mesh * pb.mesh;
    // every rank owns its arrays, as every MPI process does (the library page-locks the arrays it is handed)
    std::vector<float> ndens = pb.ndens;
    // C2R_HARNESS_ALLFRAC=1: a driver built with -DALLFRAC -- xh / xh_av / xh_intermed are (mesh,0:1) arrays, the stored neutral
    // half first; it comes from <in.bin>.x0 (ncell doubles) or, without that file, is 1 - x as xfrac_restart_init sets it
    const bool allfrac = getenv("C2R_HARNESS_ALLFRAC") && atoi(getenv("C2R_HARNESS_ALLFRAC")) != 0;
    std::vector<double> xh = pb.xh, xh_av(allfrac ? 2 * ncell : ncell), xh_int(allfrac ? 2 * ncell : ncell), phih(ncell);
    if (allfrac) {
        xh.resize(2 * ncell);
        for (size_t i = 0; i < ncell; ++i) { xh[ncell + i] = pb.xh[i]; xh[i] = pb.xh0.empty() ? 1.0 - pb.xh[i] : pb.xh0[i]; }
    }
    c2r_report rep;
    QueueArg qarg{sh, rank};
    {
        c2r_params prm;
        c2r_default_params(&prm);
        prm.mesh[0] = prm.mesh[1] = prm.mesh[2] = pb.mesh;
        prm.device = rank % ndev;
        if (const char *e = getenv("C2R_SWEEP_MODE")) prm.sweep_mode = atoi(e) ? C2R_SWEEP_FAST : C2R_SWEEP_EXACT;
        if (const char *e = getenv("C2R_HARNESS_DETERMINISTIC")) prm.deterministic_rates = atoi(e) ? 1 : 0;   // ordered per-source sums: runs comparable bit for bit
        prm.allfrac = allfrac ? 1 : 0;
        TRY(c2r_create(&ctx, &prm));
        // the library reads no environment variable for its schedule switches: this harness (test code) hands its own over
        static const char *const opts[][2] = {{"C2R_EXCHANGE_OVERLAP", "exchange_overlap"}, {"C2R_EXCHANGE_OVERLAP_MIN", "exchange_overlap_min"},
                                              {"C2R_SPARSE_EXCHANGE", "sparse_exchange"}, {"C2R_SPARSE_FRACTION", "sparse_fraction"},
                                              {"C2R_CHAINS", "chains"}, {"C2R_CHAIN_GRAPH", "chain_graph"}};
        for (const auto &o : opts)
            if (const char *e = getenv(o[0])) TRY(c2r_set_option(ctx, o[1], atof(e)));
        TRY(c2r_set_tables(ctx, pb.thick.data(), pb.thin.data(), (int32_t)pb.thick.size()));
        const double dr[3] = {pb.dr, pb.dr, pb.dr};
        TRY(c2r_set_step(ctx, dr, pb.vol, pb.lls, 1.0f, 1.0e4));
        TRY(c2r_set_sources(ctx, pb.srcpos.data(), pb.normflux.data(), pb.nsrc));
        if (sh->nranks > 1) {
            if (sh->rccl) {
                if (rank == 0) TRY(c2r_rccl_unique_id(sh->uid));
                pthread_barrier_wait(&sh->bar);
                TRY(c2r_rccl_attach(ctx, sh->uid, rank, sh->nranks));
            } else {
                TRY(c2r_set_rank(ctx, rank, sh->nranks, host_allreduce, a));
            }
            TRY(c2r_set_balance(ctx, sh->balance ? 1 : 0));
            if (sh->slab) {
                if (sh->rccl) TRY(c2r_rccl_slab_chemistry(ctx, 1));
                else TRY(c2r_set_slab_chemistry(ctx, host_reduce_scatter, host_allgather, a));
            }
        }
        if (sh->queue_chunk > 0) TRY(c2r_set_source_queue(ctx, next_sources, &qarg, sh->queue_chunk));
        TRY(c2r_evolve3d(ctx, pb.dt, ndens.data(), xh.data(), xh_av.data(), xh_int.data(), phih.data(), &rep));
        {   // FNV-1a over the bits of the replicated results
            unsigned long long h = 1469598103934665603ULL;
            auto mix = [&h](const std::vector<double> &v) { const unsigned char *b = reinterpret_cast<const unsigned char *>(v.data());
                                                           for (size_t i = 0; i < v.size() * sizeof(double); ++i) { h ^= b[i]; h *= 1099511628211ULL; } };
            mix(xh); mix(phih);
            sh->hash[rank] = h;
        }
        if (rank == 0) {
            sh->info0 = c2r_info(ctx);
            sh->rep0 = rep; sh->phih0 = phih;
            sh->xh0.assign(xh.begin() + (allfrac ? (long)ncell : 0), xh.end());      // the ionized half
            if (allfrac) sh->xh0n.assign(xh.begin(), xh.begin() + (long)ncell);     // ... and the stored neutral half
            (void)c2r_exchange_stats(ctx, &sh->xchg[0], &sh->xchg[1], &sh->xchg[2], &sh->xchg[3]);
        }
        if (sh->rccl && sh->nranks > 1) c2r_rccl_detach(ctx);
    }
done:
    if (ctx) c2r_destroy(ctx);
    return nullptr;
}

template <typename T> static void rd(FILE *f, T *p, size_t n) { if (fread(p, sizeof(T), n, f) != n) { fprintf(stderr, "short input\n"); exit(2); } }

int main(int argc, char **argv)
{
    if (argc < 6) { fprintf(stderr, "usage: %s in.bin out.bin nranks host|rccl balance\n", argv[0]); return 2; }
    Problem pb;
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    rd(f, &pb.mesh, 1); rd(f, &pb.nsrc, 1); rd(f, &pb.dr, 1); rd(f, &pb.vol, 1); rd(f, &pb.lls, 1); rd(f, &pb.dt, 1);
    const size_t ncell = (size_t)pb.mesh * pb.mesh * pb.mesh;
    pb.ndens.resize(ncell); pb.xh.resize(ncell); pb.srcpos.resize(3 * (size_t)pb.nsrc); pb.normflux.resize(pb.nsrc);
    rd(f, pb.ndens.data(), ncell); rd(f, pb.xh.data(), ncell); rd(f, pb.srcpos.data(), pb.srcpos.size()); rd(f, pb.normflux.data(), pb.normflux.size());
    fclose(f);
    if (FILE *f0 = fopen((std::string(argv[1]) + ".x0").c_str(), "rb")) { pb.xh0.resize(ncell); rd(f0, pb.xh0.data(), ncell); fclose(f0); }
    c2r_sed_params sed;
    c2r_default_sed(&sed);
    pb.thick.resize(sed.numtau + 1); pb.thin.resize(sed.numtau + 1);
    if (c2r_build_tables(&sed, pb.thick.data(), pb.thin.data(), sed.numtau + 1, nullptr) != 0) { fprintf(stderr, "c2r_build_tables failed\n"); return 1; }
    Shared sh;
    sh.nranks = atoi(argv[3]); sh.rccl = strcmp(argv[4], "rccl") == 0; sh.balance = atoi(argv[5]) != 0; sh.pb = &pb;
    sh.slab = argc > 6 && atoi(argv[6]) != 0;
    if (const char *e = getenv("C2R_HARNESS_ROTATE_SUM")) sh.rotate = atoi(e) != 0;
    if (const char *e = getenv("C2R_HARNESS_QUEUE")) sh.queue_chunk = atoi(e);
    sh.queue[0].store(0); sh.queue[1].store(0); sh.swept.assign(sh.nranks, 0);
    sh.hash.assign(sh.nranks, 0ULL);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { fprintf(stderr, "no HIP device\n"); return 1; }
    if (sh.rccl && ndev < sh.nranks) { printf("SKIP: %d ranks over RCCL need %d devices, %d visible\n", sh.nranks, sh.nranks, ndev); return 77; }
    pthread_barrier_init(&sh.bar, nullptr, sh.nranks);
    sh.stage.assign(sh.nranks, nullptr); sh.stage_cap.assign(sh.nranks, 0); sh.rc.assign(sh.nranks, 0); sh.err.resize(sh.nranks);
    std::vector<pthread_t> th(sh.nranks);
    std::vector<RankArg> args(sh.nranks);
    for (int r = 0; r < sh.nranks; ++r) { args[r] = {&sh, r}; pthread_create(&th[r], nullptr, rank_main, &args[r]); }
    for (int r = 0; r < sh.nranks; ++r) pthread_join(th[r], nullptr);
    for (int r = 0; r < sh.nranks; ++r)
        if (sh.rc[r]) { fprintf(stderr, "rank %d failed (%d): %s\n", r, sh.rc[r], sh.err[r].c_str()); return 1; }
    f = fopen(argv[2], "wb");
    const int32_t head[3] = {sh.rep0.niter, sh.rep0.converged, sh.rccl ? sh.nranks : 0};
    fwrite(head, sizeof(int32_t), 3, f);
    fwrite(&sh.rep0.sum_nbox_all, sizeof(int64_t), 1, f);
    fwrite(sh.rep0.it_conv_flag, sizeof(int64_t), sh.rep0.niter, f);
    fwrite(&sh.rep0.photon_loss_all, sizeof(double), 1, f);
    fwrite(sh.xh0.data(), sizeof(double), ncell, f);
    fwrite(sh.phih0.data(), sizeof(double), ncell, f);
    if (!sh.xh0n.empty()) fwrite(sh.xh0n.data(), sizeof(double), ncell, f);      // (C2R_HARNESS_ALLFRAC: the neutral half of xh behind everything else)
    fclose(f);
    printf("ok: %d rank(s) on %d device(s), %s %s, balance %d: niter %d sum_nbox %lld\n", sh.nranks, ndev < sh.nranks ? ndev : sh.nranks,
           sh.rccl ? "rccl" : "host", sh.slab ? "reduce-scatter + slab chemistry + all-gather" : "all-reduce", (int)sh.balance, sh.rep0.niter,
           (long long)sh.rep0.sum_nbox_all);
    if (sh.queue_chunk > 0) { printf("queue: chunk %d, sources taken per rank over all passes:", sh.queue_chunk); for (long long v : sh.swept) printf(" %lld", v); printf("\n"); }
    bool same = true;
    for (int r = 1; r < sh.nranks; ++r) same = same && sh.hash[r] == sh.hash[0];
    printf("replicas identical: %s\n", same ? "yes" : "NO");
    printf("info: %s\n", sh.info0.c_str());
    if (!same) return 1;
    printf("exchange: calls %lld packed %lld bytes_last %lld bytes_total %lld\n", (long long)sh.xchg[0], (long long)sh.xchg[1],
           (long long)sh.xchg[2], (long long)sh.xchg[3]);
    return 0;
}
