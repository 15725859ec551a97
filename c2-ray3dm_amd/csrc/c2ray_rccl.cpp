// libc2ray_rccl.so: the all-reduce callback of the evolve hot path on RCCL (include/c2ray_rccl.h).
// Replaces MPI_ALLREDUCE of phih_grid / photon_loss / sum_nbox (evolve.F90:587-613) for
// one-process-per-GPU hosts.  Kept out of libc2ray_hip.so so that the core has no RCCL dependency.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <map>
#include <mutex>

#include "../../include/c2ray_rccl.h"

static_assert(sizeof(ncclUniqueId) == C2R_RCCL_ID_BYTES, "ncclUniqueId size");

namespace {
struct Link { ncclComm_t comm = nullptr; };
std::map<c2r_ctx *, Link *> g_links;
std::mutex g_mu;

Link *find(c2r_ctx *ctx)
{
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_links.find(ctx);
    return it == g_links.end() ? nullptr : it->second;
}

// c2r_allreduce_fn: sum `count` f64 in place at the device pointer, ordered on the given stream
int rccl_cb(void *user, void *dev_buf, size_t count, void *stream)
{
    Link *l = static_cast<Link *>(user);
    return (int)ncclAllReduce(dev_buf, dev_buf, count, ncclDouble, ncclSum, l->comm, (hipStream_t)stream);
}
// c2r_reduce_scatter_fn: rank r receives the sum over ranks of ITS slab, in place.  The slabs are whole z-planes and need
// not be equal, so this is one ncclReduce per slab (root = the slab's rank) inside a group -- the traffic of a reduce-scatter.
int rccl_rs(void *user, void *dev_buf, const size_t *off, const size_t *cnt, int32_t nranks, void *stream)
{
    Link *l = static_cast<Link *>(user);
    double *b = static_cast<double *>(dev_buf);
    ncclResult_t r = ncclGroupStart();
    for (int k = 0; k < nranks && r == ncclSuccess; ++k)
        if (cnt[k]) r = ncclReduce(b + off[k], b + off[k], cnt[k], ncclDouble, ncclSum, k, l->comm, (hipStream_t)stream);
    const ncclResult_t e = ncclGroupEnd();
    return (int)(r != ncclSuccess ? r : e);
}
// c2r_allgather_fn: every rank's slab (bytes) to every rank, in place: one ncclBroadcast per slab inside a group
int rccl_ag(void *user, void *dev_buf, const size_t *off, const size_t *cnt, int32_t nranks, void *stream)
{
    Link *l = static_cast<Link *>(user);
    char *b = static_cast<char *>(dev_buf);
    ncclResult_t r = ncclGroupStart();
    for (int k = 0; k < nranks && r == ncclSuccess; ++k)
        if (cnt[k]) r = ncclBroadcast(b + off[k], b + off[k], cnt[k], ncclChar, k, l->comm, (hipStream_t)stream);
    const ncclResult_t e = ncclGroupEnd();
    return (int)(r != ncclSuccess ? r : e);
}
}  // namespace

extern "C" {

int c2r_rccl_slab_chemistry(c2r_ctx *ctx, int32_t on)
{
    Link *l = find(ctx);
    if (!l) return C2R_ESTATE;
    return on ? c2r_set_slab_chemistry(ctx, rccl_rs, rccl_ag, l) : c2r_set_slab_chemistry(ctx, nullptr, nullptr, nullptr);
}

int c2r_rccl_unique_id(void *id)
{
    if (!id) return C2R_EINVAL;
    return (int)ncclGetUniqueId(static_cast<ncclUniqueId *>(id));
}

int c2r_rccl_attach(c2r_ctx *ctx, const void *id, int32_t rank, int32_t nranks)
{
    if (!ctx || !id || nranks < 1 || rank < 0 || rank >= nranks) return C2R_EINVAL;
    if (find(ctx)) return C2R_ESTATE;                       // already attached
    int32_t dev = 0;
    if (c2r_get_device(ctx, &dev) != C2R_OK || hipSetDevice(dev) != hipSuccess) return C2R_ESTATE;   // the communicator lives on the context's device
    Link *l = new Link();
    ncclUniqueId uid = *static_cast<const ncclUniqueId *>(id);
    ncclResult_t r = ncclCommInitRank(&l->comm, nranks, uid, rank);
    if (r != ncclSuccess) { delete l; return (int)r; }
    // a single rank keeps the callback too: c2r_rccl_allreduce stays usable, the library skips it
    int rc = c2r_set_rank(ctx, rank, nranks, rccl_cb, l);
    if (rc != C2R_OK) { ncclCommDestroy(l->comm); delete l; return rc; }
    std::lock_guard<std::mutex> lk(g_mu);
    g_links[ctx] = l;
    return C2R_OK;
}

int c2r_rccl_allreduce(c2r_ctx *ctx, void *dev_buf, size_t count, void *hip_stream)
{
    Link *l = find(ctx);
    if (!l || !dev_buf) return l ? C2R_EINVAL : C2R_ESTATE;
    return rccl_cb(l, dev_buf, count, hip_stream);
}

int c2r_rccl_detach(c2r_ctx *ctx)
{
    Link *l = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_links.find(ctx);
        if (it == g_links.end()) return C2R_ESTATE;
        l = it->second;
        g_links.erase(it);
    }
    c2r_set_rank(ctx, 0, 1, nullptr, nullptr);
    c2r_set_slab_chemistry(ctx, nullptr, nullptr, nullptr);
    ncclResult_t r = ncclCommDestroy(l->comm);
    delete l;
    return (int)r;
}

}  // extern "C"
