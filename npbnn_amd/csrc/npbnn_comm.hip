// RCCL side of the C ABI: the MC3 temperature-swap exchange (reference: np_bnn/BNN_mc3.py:98-112).
// One communicator per process / GPU; payloads are a few dozen bytes, so every call is a staged
// host -> device copy, one collective on the communicator's stream, a copy back and a stream sync.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>

#include "npbnn_hip.h"

extern "C" void npbnn_set_global_error_(const char* msg);

struct npbnn_comm {
    int device = 0, rank = 0, nranks = 1;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    void* d_send = nullptr;
    void* d_recv = nullptr;
    size_t cap_send = 0, cap_recv = 0;
    bool dead = false;          // aborted after a failure in the middle of an exchange run: every later call fails at once
};

namespace {

int cfail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    npbnn_set_global_error_(buf);
    return code;
}

#define C_HIP(call)                                                                                     \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess) return cfail(NPBNN_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)
#define C_NCCL(call)                                                                                       \
    do {                                                                                                   \
        ncclResult_t r_ = (call);                                                                          \
        if (r_ != ncclSuccess) return cfail(NPBNN_E_COMM, "%s failed: %s", #call, ncclGetErrorString(r_)); \
    } while (0)

int ensure(npbnn_comm* c, size_t send_bytes, size_t recv_bytes) {
    if (send_bytes > c->cap_send) {
        if (c->d_send) (void)hipFree(c->d_send);
        c->d_send = nullptr;
        C_HIP(hipMalloc(&c->d_send, send_bytes));
        c->cap_send = send_bytes;
    }
    if (recv_bytes > c->cap_recv) {
        if (c->d_recv) (void)hipFree(c->d_recv);
        c->d_recv = nullptr;
        C_HIP(hipMalloc(&c->d_recv, recv_bytes));
        c->cap_recv = recv_bytes;
    }
    return NPBNN_OK;
}

}  // namespace

extern "C" {

int npbnn_comm_unique_id(char out[128]) {
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
    if (!out) return cfail(NPBNN_E_ARG, "null out");
    ncclUniqueId id;
    C_NCCL(ncclGetUniqueId(&id));
    memcpy(out, &id, 128);
    return NPBNN_OK;
}

int npbnn_comm_init(int device_id, int rank, int nranks, const char id[128], npbnn_comm** out) {
    if (!out || !id || nranks < 1 || rank < 0 || rank >= nranks) return cfail(NPBNN_E_ARG, "comm_init: bad arguments");
    *out = nullptr;
    C_HIP(hipSetDevice(device_id));
    npbnn_comm* c = new npbnn_comm();
    c->device = device_id;
    c->rank = rank;
    c->nranks = nranks;
    ncclUniqueId uid;
    memcpy(&uid, id, 128);
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return cfail(NPBNN_E_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
    }
    ncclResult_t r = ncclCommInitRank(&c->comm, nranks, uid, rank);
    if (r != ncclSuccess) {
        (void)hipStreamDestroy(c->stream);
        delete c;
        return cfail(NPBNN_E_COMM, "ncclCommInitRank failed: %s", ncclGetErrorString(r));
    }
    *out = c;
    return NPBNN_OK;
}

int npbnn_comm_allgather_f64(npbnn_comm* c, const double* send, int count, double* recv) {
    if (!c || !send || !recv || count < 1) return cfail(NPBNN_E_ARG, "comm_allgather: bad arguments");
    if (c->dead) return cfail(NPBNN_E_COMM, "comm_allgather: this communicator was aborted after an earlier failure");
    C_HIP(hipSetDevice(c->device));
    const size_t sb = (size_t)count * sizeof(double), rb = sb * c->nranks;
    int rc = ensure(c, sb, rb);
    if (rc) return rc;
    C_HIP(hipMemcpyAsync(c->d_send, send, sb, hipMemcpyHostToDevice, c->stream));
    C_NCCL(ncclAllGather(c->d_send, c->d_recv, (size_t)count, ncclDouble, c->comm, c->stream));
    C_HIP(hipMemcpyAsync(recv, c->d_recv, rb, hipMemcpyDeviceToHost, c->stream));
    C_HIP(hipStreamSynchronize(c->stream));
    return NPBNN_OK;
}

int npbnn_comm_bcast_i64(npbnn_comm* c, int64_t* buf, int count, int root) {
    if (!c || !buf || count < 1 || root < 0 || root >= c->nranks) return cfail(NPBNN_E_ARG, "comm_bcast: bad arguments");
    if (c->dead) return cfail(NPBNN_E_COMM, "comm_bcast: this communicator was aborted after an earlier failure");
    C_HIP(hipSetDevice(c->device));
    const size_t b = (size_t)count * sizeof(int64_t);
    int rc = ensure(c, b, b);
    if (rc) return rc;
    C_HIP(hipMemcpyAsync(c->d_send, buf, b, hipMemcpyHostToDevice, c->stream));
    C_NCCL(ncclBroadcast(c->d_send, c->d_recv, (size_t)count, ncclInt64, root, c->comm, c->stream));
    C_HIP(hipMemcpyAsync(buf, c->d_recv, b, hipMemcpyDeviceToHost, c->stream));
    C_HIP(hipStreamSynchronize(c->stream));
    return NPBNN_OK;
}

// internal (npbnn_capi.hip, exchange run): the records of one exchange all-gathered IN PLACE on a chain's own stream -
// d_buf holds nranks * count doubles, this rank's share already at d_buf + rank * count - with no host synchronisation
int npbnn_comm_allgather_inplace_stream_(npbnn_comm* c, double* d_buf, int count, void* stream) {
    if (!c || !d_buf || count < 1) return cfail(NPBNN_E_ARG, "comm_allgather_inplace: bad arguments");
    if (c->dead) return cfail(NPBNN_E_COMM, "comm_allgather_inplace: this communicator was aborted after an earlier failure");
    C_NCCL(ncclAllGather(d_buf + (size_t)c->rank * count, d_buf, (size_t)count, ncclDouble, c->comm, (hipStream_t)stream));
    return NPBNN_OK;
}

// internal: a rank that fails after its peers have collectives of this communicator in flight cannot leave them paired with
// whatever it would issue next - it tears the communicator down (the peers' pending collectives end with an error instead of
// waiting for ever) and every later call on the handle fails at once.  The caller starts over with a new communicator.
void npbnn_comm_abort_(npbnn_comm* c) {
    if (!c || c->dead) return;
    c->dead = true;
    if (c->comm) {
        (void)ncclCommAbort(c->comm);
        c->comm = nullptr;
    }
}

int npbnn_comm_info_(const npbnn_comm* c, int* device, int* rank, int* nranks) {
    if (!c) return cfail(NPBNN_E_ARG, "null communicator");
    if (device) *device = c->device;
    if (rank) *rank = c->rank;
    if (nranks) *nranks = c->nranks;
    return NPBNN_OK;
}

void npbnn_comm_destroy(npbnn_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (c->d_send) (void)hipFree(c->d_send);
    if (c->d_recv) (void)hipFree(c->d_recv);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

}  // extern "C"
