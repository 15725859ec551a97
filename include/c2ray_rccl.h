/* c2ray_rccl.h -- optional RCCL binding of the Gamma all-reduce for hosts without torch.distributed.
 *
 * The core library (c2ray_hip.h) takes the all-reduce of evolve.F90:577-616
 * (mpi_accumulate_grid_quantities: MPI_ALLREDUCE of phih_grid, photon_loss, sum_nbox) as a callback so
 * that it does not depend on a communication library.  libc2ray_rccl.so supplies that callback with
 * RCCL (ncclAllReduce, ncclDouble, ncclSum on the context's stream) for one-process-per-GPU hosts that
 * already have a way to hand 128 bytes from rank 0 to every rank -- in the reference's -DMPI builds that
 * is one MPI_BCAST next to the ones mpi.F90 already issues.
 *
 * Every function returns 0 on success, a negative C2R_E* code, or a positive ncclResult_t.
 */
#ifndef C2RAY_RCCL_H
#define C2RAY_RCCL_H

#include "c2ray_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define C2R_RCCL_ID_BYTES 128            /* sizeof(ncclUniqueId) */

/* rank 0: make the rendezvous token; the host distributes it (MPI_BCAST of 128 bytes). */
int c2r_rccl_unique_id(void *id /* C2R_RCCL_ID_BYTES */);

/* every rank: join the communicator on the context's device and install the RCCL all-reduce as the
 * context's callback (c2r_set_rank(ctx, rank, nranks, ...)): sources are then distributed
 * 1+rank, 1+rank+nranks, ... (master_slave.F90:85) and summed once per outer iteration. */
int c2r_rccl_attach(c2r_ctx *ctx, const void *id, int32_t rank, int32_t nranks);

/* all-reduce `count` f64 at a device pointer through the attached communicator (what the callback does);
 * exported for tests and for hosts that reduce their own quantities the same way. */
int c2r_rccl_allreduce(c2r_ctx *ctx, void *dev_buf, size_t count, void *hip_stream);

/* after c2r_rccl_attach: slab chemistry over the same communicator (c2r_set_slab_chemistry: the rates reduce-scattered by
 * z-slabs -- grouped ncclReduce, one per slab, since whole planes rarely divide evenly --, the chemistry on the own slab, its
 * outputs all-gathered with grouped ncclBroadcast); on = 0 returns to the all-reduce + replicated global pass. */
int c2r_rccl_slab_chemistry(c2r_ctx *ctx, int32_t on);

/* leave the communicator; the context goes back to a single rank. */
int c2r_rccl_detach(c2r_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
