// RCCL side of the C ABI: the MC3 temperature-swap exchange (reference: np_bnn/BNN_mc3.py:98-112).
// One communicator per process / GPU; payloads are a few dozen bytes, so every call is a staged
// host -> device copy, one collective on the communicator's stream, a copy back and a stream sync.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

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

// A collective whose peer has died never completes, and hipStreamSynchronize on it never returns.  Every host-side wait on a stream
// that carries collectives of this communicator polls instead: the stream, the communicator's asynchronous error state, and a
// deadline (NPBNN_COMM_TIMEOUT_S, default 300 s - a swap exchange takes microseconds).  On an error or at the deadline the
// communicator is aborted (its kernels leave the GPU) and the call fails with NPBNN_E_COMM: a rank that lost a peer ends, it does
// not hang.
double comm_timeout_s() {
    static const double t = [] {
        const char* e = getenv("NPBNN_COMM_TIMEOUT_S");
        const double v = e ? atof(e) : 0.0;
        return v > 0.0 ? v : 300.0;
    }();
    return t;
}

void abort_comm(npbnn_comm* c) {
    if (!c || c->dead) return;
    c->dead = true;
    if (c->comm) {
        (void)ncclCommAbort(c->comm);
        c->comm = nullptr;
    }
}

int wait_stream(npbnn_comm* c, hipStream_t stream, const char* what) {
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    for (;;) {
        const hipError_t q = hipStreamQuery(stream);
        if (q == hipSuccess) return NPBNN_OK;
        if (q != hipErrorNotReady) {
            abort_comm(c);
            return cfail(NPBNN_E_HIP, "%s: the stream failed: %s", what, hipGetErrorString(q));
        }
        if ((++spins & 63) == 0) {
            ncclResult_t async = ncclSuccess;
            if (c->comm && ncclCommGetAsyncError(c->comm, &async) == ncclSuccess && async != ncclSuccess && async != ncclInProgress) {
                abort_comm(c);
                return cfail(NPBNN_E_COMM, "%s: the communicator reports %s (a peer rank has gone?)", what, ncclGetErrorString(async));
            }
            const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (el > comm_timeout_s()) {
                abort_comm(c);
                return cfail(NPBNN_E_COMM, "%s: no completion after %.0f s (NPBNN_COMM_TIMEOUT_S) - a peer rank has gone or never "
                                           "entered the collective; the communicator was aborted", what, el);
            }
            if (el > 2e-3) std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    }
}

}  // namespace

// The RCCL this library was compiled against (rccl.h of /opt/rocm) and the one the process has mapped must be the same release line:
// libnpbnn_hip.so names librccl.so.1 with a run path of /opt/rocm/lib, but a process that loaded another librccl.so.1 first (the
// copy bundled with a PyTorch wheel, say) hands that one to the dynamic linker for the same name.
static int rccl_runtime(int* runtime_version, const char** path) {
    int v = 0;
    ncclResult_t r = ncclGetVersion(&v);
    if (r != ncclSuccess) return cfail(NPBNN_E_COMM, "ncclGetVersion failed: %s", ncclGetErrorString(r));
    *runtime_version = v;
    static thread_local char where[512];
    where[0] = 0;
    Dl_info info;
    if (dladdr(reinterpret_cast<void*>(&ncclGetVersion), &info) && info.dli_fname) snprintf(where, sizeof where, "%s", info.dli_fname);
    *path = where;
    return NPBNN_OK;
}

extern "C" {

int npbnn_comm_runtime(int* runtime_version, int* header_version, char* path, int path_cap) {
    int v = 0;
    const char* where = "";
    int rc = rccl_runtime(&v, &where);
    if (rc) return rc;
    if (runtime_version) *runtime_version = v;
    if (header_version) *header_version = NCCL_VERSION_CODE;
    if (path && path_cap > 0) snprintf(path, (size_t)path_cap, "%s", where);
    return NPBNN_OK;
}

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
    {
        int v = 0;
        const char* where = "";
        int rc = rccl_runtime(&v, &where);
        if (rc) return rc;
        if (v / 100 != NCCL_VERSION_CODE / 100)
            return cfail(NPBNN_E_COMM, "comm_init: this process has RCCL %d.%d.%d mapped (%s) but libnpbnn_hip.so was built against %d.%d.%d "
                                       "(/opt/rocm): another librccl.so.1 was loaded first - do not import torch (or anything that bundles "
                                       "its own RCCL) before the communicator is created", v / 10000, (v / 100) % 100, v % 100, where,
                         NCCL_MAJOR, NCCL_MINOR, NCCL_PATCH);
    }
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
    return wait_stream(c, c->stream, "comm_allgather");
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
    return wait_stream(c, c->stream, "comm_bcast");
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
void npbnn_comm_abort_(npbnn_comm* c) { abort_comm(c); }

// internal (npbnn_capi.hip, exchange run): wait for a chain's stream that carries in-place all-gathers of `c` (see wait_stream)
int npbnn_comm_wait_stream_(npbnn_comm* c, void* stream, const char* what) {
    if (!c) return cfail(NPBNN_E_ARG, "null communicator");
    if (c->dead) return cfail(NPBNN_E_COMM, "%s: this communicator was aborted after an earlier failure", what);
    return wait_stream(c, (hipStream_t)stream, what);
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
