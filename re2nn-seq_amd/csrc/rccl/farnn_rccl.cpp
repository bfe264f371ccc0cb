// libfarnn_rccl.so -- include/farnn_rccl.h: the tag gather of the multi-GPU tagging path over RCCL, without torch.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cerrno>
#include <cstdio>
#include <cstring>

#include "../../../include/farnn_rccl.h"

namespace {

thread_local char g_err[256] = "";

int fail(int code, const char *what, const char *detail) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, detail ? detail : "");
    return code;
}

struct Comm {
    ncclComm_t comm;
    int nranks, rank, device;
};

static_assert(sizeof(ncclUniqueId) <= FARNN_RCCL_ID_BYTES, "the id buffer of the ABI holds an ncclUniqueId");

}  // namespace

extern "C" {

int farnn_rccl_unique_id(void *id_out) {
    if (!id_out) return fail(-EINVAL, "farnn_rccl_unique_id", "null buffer");
    ncclUniqueId id;
    const ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return fail(-EIO, "ncclGetUniqueId", ncclGetErrorString(r));
    memset(id_out, 0, FARNN_RCCL_ID_BYTES);
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

int farnn_rccl_comm_create(const void *id, int nranks, int rank, int device, void **comm_out) {
    if (!id || !comm_out || nranks < 1 || rank < 0 || rank >= nranks) return fail(-EINVAL, "farnn_rccl_comm_create", "arguments");
    hipError_t he = hipSetDevice(device);
    if (he != hipSuccess) return fail(-ENODEV, "hipSetDevice", hipGetErrorString(he));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    Comm *c = new Comm{nullptr, nranks, rank, device};
    const ncclResult_t r = ncclCommInitRank(&c->comm, nranks, uid, rank);
    if (r != ncclSuccess) {
        delete c;
        return fail(-EIO, "ncclCommInitRank", ncclGetErrorString(r));
    }
    *comm_out = c;
    return 0;
}

int farnn_rccl_gather_tags(void *comm, const int32_t *local, int64_t rows_per_rank, int L, int32_t *gathered, void *stream) {
    Comm *c = static_cast<Comm *>(comm);
    if (!c || !local || !gathered || rows_per_rank < 0 || L < 0) return fail(-EINVAL, "farnn_rccl_gather_tags", "arguments");
    if (rows_per_rank == 0 || L == 0) return 0;
    hipError_t he = hipSetDevice(c->device);
    if (he != hipSuccess) return fail(-ENODEV, "hipSetDevice", hipGetErrorString(he));
    const ncclResult_t r = ncclAllGather(local, gathered, (size_t)rows_per_rank * (size_t)L, ncclInt32, c->comm,
                                         static_cast<hipStream_t>(stream));
    if (r != ncclSuccess) return fail(-EIO, "ncclAllGather", ncclGetErrorString(r));
    return 0;
}

int farnn_rccl_comm_count(void *comm) {
    Comm *c = static_cast<Comm *>(comm);
    if (!c) return fail(-EINVAL, "farnn_rccl_comm_count", "null communicator");
    int n = 0;
    const ncclResult_t r = ncclCommCount(c->comm, &n);
    if (r != ncclSuccess) return fail(-EIO, "ncclCommCount", ncclGetErrorString(r));
    return n;
}

int farnn_rccl_comm_destroy(void *comm) {
    Comm *c = static_cast<Comm *>(comm);
    if (!c) return 0;
    const ncclResult_t r = ncclCommDestroy(c->comm);
    delete c;
    if (r != ncclSuccess) return fail(-EIO, "ncclCommDestroy", ncclGetErrorString(r));
    return 0;
}

int farnn_rccl_version(void) {
    int v = 0;
    const ncclResult_t r = ncclGetVersion(&v);
    if (r != ncclSuccess) return fail(-EIO, "ncclGetVersion", ncclGetErrorString(r));
    return v;
}

const char *farnn_rccl_last_error(void) { return g_err; }

}  // extern "C"
