/* farnn_rccl.h -- the tag gather of the multi-GPU tagging path as a C-ABI of its own (libfarnn_rccl.so), for hosts that do not
 * go through torch.distributed.
 *
 * One process per GPU tags its share of a batch with its own handle (farnn.h); ONE collective returns every rank's tag ids to
 * every rank: an all-gather of int32 [rows_per_rank][L] blocks over RCCL / xGMI.  The reference has no multi-GPU facility
 * (SURVEY.md 2a, 8e); this entry replaces the build's own `re2nn_seq_amd.dist.gather_tags_balanced`
 * (re2nn-seq_amd/dist.py: torch.distributed.all_gather_into_tensor over the "nccl" = RCCL backend) -- the same collective,
 * the same buffers, without torch.  The library links librccl.so; libfarnn_hip.so does not depend on it.
 *
 * Protocol: rank 0 calls farnn_rccl_unique_id and hands the 128 bytes to the other ranks by whatever channel the host has
 * (a file, a socket, MPI, torch's store); every rank then calls farnn_rccl_comm_create (collective: it returns when all
 * ranks have joined).  Calls return 0 or a negative errno-style code; farnn_rccl_last_error() has the message of the calling
 * thread's last failure.  Nothing throws across the ABI. */
#ifndef FARNN_RCCL_H
#define FARNN_RCCL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FARNN_RCCL_ID_BYTES 128

/* rank 0: a fresh communicator id (ncclGetUniqueId) into id_out[FARNN_RCCL_ID_BYTES] */
int farnn_rccl_unique_id(void *id_out);

/* every rank (collective): joins the communicator `id` of `nranks` ranks as `rank`, on HIP device `device` */
int farnn_rccl_comm_create(const void *id, int nranks, int rank, int device, void **comm_out);

/* every rank (collective, asynchronous on `stream`): gathered[r * rows_per_rank + i][:] = rank r's local[i][:] for every rank r.
 * local: device int32 [rows_per_rank][L] (ranks with fewer rows pad with -1 rows, as dist.gather_tags_balanced does);
 * gathered: device int32 [nranks * rows_per_rank][L]; stream: a hipStream_t (0 = the null stream). */
int farnn_rccl_gather_tags(void *comm, const int32_t *local, int64_t rows_per_rank, int L, int32_t *gathered, void *stream);

/* the number of ranks RCCL itself reports for the communicator (ncclCommCount), or a negative error: what the N > 1 bench line
 * prints and asserts equal to the launcher's world size */
int farnn_rccl_comm_count(void *comm);

int farnn_rccl_comm_destroy(void *comm);

/* RCCL's version code (ncclGetVersion), or a negative error */
int farnn_rccl_version(void);

const char *farnn_rccl_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
