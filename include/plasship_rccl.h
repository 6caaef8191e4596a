/* plasship_rccl — the native communicator of a sharded run: RCCL point-to-point over xGMI behind plasship_ctx_set_comm.
 *
 * replaces: the reference's MPI split of kmermatcher (`$RUNNER` = mpirun in data/assemble.sh:92,103; MMseqsMPI::init,
 * mm/commons/MMseqsMPI.cpp; hash-range split + file merge mm/linclust/kmermatcher.cpp:631-660,736-778).  One process and one
 * plasship_ctx per GPU; a C++ host binds exactly this:
 *
 *     unsigned char id[PLASSHIP_RCCL_ID_BYTES];
 *     if (rank == 0) plasship_rccl_get_unique_id(id);
 *     <broadcast id to all ranks: MPI_Bcast, a file, the launcher's key-value store …>
 *     plasship_rccl_comm *c; plasship_rccl_comm_create(ctx, rank, world, id, &c);      // collective; installs itself on ctx
 *     … plasship_kmermatch / plasship_rescore / plasship_assemble as on one GPU …
 *     plasship_rccl_comm_destroy(ctx, c);
 *
 * The all-to-all(v) of k-mer and grouped records and the all-gather(v) of extended sequences are groups of ncclSend / ncclRecv
 * pairs, enqueued on the CONTEXT'S stream in pieces of at most 256 MiB (RCCL 2.26 drops part of a message beyond 2^30 bytes,
 * tools/rccl_a2a_probe.py): they are ordered with the kernels around them and need no host synchronisation; the piece a rank
 * keeps for itself is a device copy.  Small host arrays (bucket counts, run heads) travel through a pinned staging buffer and
 * ncclAllGather.  RCCL is loaded at run time (dlopen of librccl.so.1; PLASSHIP_RCCL_LIB overrides the path), so a single-GPU
 * user of libplasship.so has no dependency on it.
 */
#ifndef PLASSHIP_RCCL_H
#define PLASSHIP_RCCL_H
#include "plasship.h"
#ifdef __cplusplus
extern "C" {
#endif

#define PLASSHIP_RCCL_ID_BYTES 128
typedef struct plasship_rccl_comm plasship_rccl_comm;

int plasship_rccl_get_unique_id(void *id_out);
int plasship_rccl_comm_create(plasship_ctx *ctx, int rank, int world, const void *id, plasship_rccl_comm **out);
/* device bytes this rank sent to other ranks, seconds spent inside the collectives (host clock), calls — since creation / the last reset */
int plasship_rccl_comm_stats(plasship_rccl_comm *c, uint64_t *bytes_sent, double *seconds, uint64_t *calls, int reset);
void plasship_rccl_comm_destroy(plasship_ctx *ctx, plasship_rccl_comm *c);

#ifdef __cplusplus
}
#endif
#endif
