// Stand-alone device operators on host arrays: the reference's small call-surface helpers (activation functions,
// output functions, likelihoods and accuracy statistics applied to an explicit prediction matrix) when user code
// calls them directly instead of through the sampler's fused path.  float64 in, float64 compute, float64 out;
// every reduction is two-stage (per-block partials, fixed-order sum on the host) and therefore deterministic.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <vector>

#include "npbnn_hip.h"

extern "C" void npbnn_set_global_error_(const char* msg);

namespace {

int ofail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    npbnn_set_global_error_(buf);
    return code;
}

#define O_HIP(call)                                                                                     \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess) return ofail(NPBNN_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

struct DevBuf {          // RAII device buffer
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
    template <typename T> T* as() { return static_cast<T*>(p); }
};

constexpr int kBlock = 256;

__global__ void act_kernel(double* z, long long n, int kind, double prm) {
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
        const double v = z[i];
        double r;
        switch (kind) {
            case NPBNN_ACT_RELU: r = v < 0 ? 0.0 : v; break;                      // BNN_lib.py:50-52
            case NPBNN_ACT_LEAKY: r = v < 0 ? prm * v : v; break;                 // BNN_lib.py:54-56
            case NPBNN_ACT_SWISH: r = v / (1.0 + exp(-v)); break;                 // BNN_lib.py:58-61
            case NPBNN_ACT_TANH: r = 1.0 - 2.0 / (exp(2.0 * v) + 1.0); break;     // BNN_lib.py:63-65
            default: r = fmax(v, 0.0) + log1p(exp(-fabs(v))); break;              // 4: softplus, BNN_lib.py:170-172
        }
        z[i] = r;
    }
}

// rows of a matrix: softmax (out_kind 0) or softplus on the columns >= ind (out_kind 2)
__global__ void output_kernel(double* z, long long rows, int cols, int out_kind, int ind) {
    for (long long r = (long long)blockIdx.x * kBlock + threadIdx.x; r < rows; r += (long long)gridDim.x * kBlock) {
        double* row = z + r * cols;
        if (out_kind == NPBNN_OUT_SOFTMAX) {                                      // scipy.special.softmax, BNN_lib.py:166-168
            double m = -INFINITY;
            for (int c = 0; c < cols; ++c) m = fmax(m, row[c]);
            double s = 0.0;
            for (int c = 0; c < cols; ++c) s += exp(row[c] - m);
            for (int c = 0; c < cols; ++c) row[c] = exp(row[c] - m) / s;
        } else if (out_kind == NPBNN_OUT_SOFTPLUS_HALF) {                         // RegressTransformError, BNN_lib.py:177-182
            for (int c = ind; c < cols; ++c) row[c] = fmax(row[c], 0.0) + log1p(exp(-fabs(row[c])));
        }
    }
}

__device__ double block_reduce(double v, double* sh) {
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int h = kBlock / 2; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) sh[threadIdx.x] += sh[threadIdx.x + h];
        __syncthreads();
    }
    return sh[0];
}

// per-row log-likelihood terms of a given prediction matrix, summed per block
__global__ void lik_kernel(int lik_kind, const double* pred, long long rows, int cols, const long long* labels, const double* targets,
                           int k, const double* inst_w, const double* class_w, const double* sigma, double* partial) {
    __shared__ double sh[kBlock];
    double acc = 0.0;
    for (long long r = (long long)blockIdx.x * kBlock + threadIdx.x; r < rows; r += (long long)gridDim.x * kBlock) {
        const double* p = pred + r * cols;
        if (lik_kind == NPBNN_LIK_CATEGORICAL) {                                  // BNN_lib.py:100-121
            const long long lab = labels[r];
            double t = log(p[lab]);
            if (inst_w) t *= inst_w[r];
            if (class_w) t *= class_w[lab];
            acc += t;
        } else if (lik_kind == NPBNN_LIK_GAUSS || lik_kind == NPBNN_LIK_GAUSS_PRED_SIGMA) {   // BNN_lib.py:123-143
            for (int j = 0; j < k; ++j) {
                const double sg = lik_kind == NPBNN_LIK_GAUSS ? sigma[j] : p[k + j];
                const double d = (targets[r * k + j] - p[j]) / sg;
                acc += -0.9189385332046727418 - log(sg) - 0.5 * d * d;
            }
        } else if (lik_kind == NPBNN_LIK_POISSON) {                               // BNN_lik.py:5-14
            const double y = targets[r * k], eta = p[0];
            acc += y * eta - exp(eta) - lgamma(y + 1.0);
        } else {                                                                  // BNN_lik.py:16-66
            const int kk = lik_kind == NPBNN_LIK_NEGBIN2D ? k : 1;
            for (int j = 0; j < kk; ++j) {
                const double y = targets[r * k + j];
                const double e0 = p[j], e1 = p[(lik_kind == NPBNN_LIK_NEGBIN2D ? k : 1) + j];
                double mean, pr;
                if (lik_kind == NPBNN_LIK_NEGBIN_BASE10) { mean = exp(2.302585092994046 * e0); pr = 1.0 / (1.0 + exp(-2.302585092994046 * e1)); }
                else { mean = exp(e0); pr = 1.0 / (1.0 + exp(-e1)); }
                const double nn = pr * mean / (1.0 - pr);
                acc += lgamma(y + nn) - lgamma(y + 1.0) - lgamma(nn) + nn * log(pr) + y * log1p(-pr);
            }
        }
    }
    const double s = block_reduce(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// argmax per row (first maximum wins, np.argmax) -> confusion counts [true][predicted] or predicted-class counts
__global__ void confusion_kernel(const double* pred, long long rows, int cols, const long long* labels, unsigned long long* conf,
                                 unsigned long long* pred_counts) {
    for (long long r = (long long)blockIdx.x * kBlock + threadIdx.x; r < rows; r += (long long)gridDim.x * kBlock) {
        const double* p = pred + r * cols;
        int best = 0;
        for (int c = 1; c < cols; ++c)
            if (p[c] > p[best]) best = c;
        atomicAdd(pred_counts + best, 1ull);
        if (labels) atomicAdd(conf + labels[r] * cols + best, 1ull);
    }
}

// per-column sum of squared errors between link(pred[:, j]) and targets[:, j]; link 0 identity, 1 exp, 2 10^x
__global__ void sse_kernel(const double* pred, const double* targets, long long rows, int cols_pred, int k, int link, double* partial) {
    __shared__ double sh[kBlock];
    for (int j = 0; j < k; ++j) {
        double acc = 0.0;
        for (long long r = (long long)blockIdx.x * kBlock + threadIdx.x; r < rows; r += (long long)gridDim.x * kBlock) {
            double v = pred[r * cols_pred + j];
            if (link == 1) v = exp(v);
            else if (link == 2) v = exp(2.302585092994046 * v);
            const double d = v - targets[r * k + j];
            acc += d * d;
        }
        const double s = block_reduce(acc, sh);
        if (threadIdx.x == 0) partial[(size_t)j * gridDim.x + blockIdx.x] = s;
        __syncthreads();
    }
}

int grid_for(long long n) {
    long long g = (n + kBlock - 1) / kBlock;
    if (g > 1024) g = 1024;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" {

int npbnn_op_activation(int device, int kind, double prm, double* inout, int64_t n) {
    if (!inout || n < 0 || kind < 0 || kind > 4) return ofail(NPBNN_E_ARG, "op_activation: bad arguments");
    if (n == 0) return NPBNN_OK;
    O_HIP(hipSetDevice(device));
    DevBuf d;
    O_HIP(d.alloc((size_t)n * 8));
    O_HIP(hipMemcpy(d.p, inout, (size_t)n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(act_kernel, dim3(grid_for(n)), dim3(kBlock), 0, 0, d.as<double>(), (long long)n, kind, prm);
    O_HIP(hipGetLastError());
    O_HIP(hipMemcpy(inout, d.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    return NPBNN_OK;
}

int npbnn_op_output(int device, int out_kind, double* inout, int64_t rows, int32_t cols, int32_t ind) {
    if (!inout || rows < 0 || cols < 1 || out_kind < 0 || out_kind > NPBNN_OUT_SOFTPLUS_HALF) return ofail(NPBNN_E_ARG, "op_output: bad arguments");
    if (rows == 0 || out_kind == NPBNN_OUT_IDENTITY) return NPBNN_OK;
    if (ind < 0) ind = cols / 2;
    O_HIP(hipSetDevice(device));
    DevBuf d;
    const size_t bytes = (size_t)rows * cols * 8;
    O_HIP(d.alloc(bytes));
    O_HIP(hipMemcpy(d.p, inout, bytes, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(output_kernel, dim3(grid_for(rows)), dim3(kBlock), 0, 0, d.as<double>(), (long long)rows, cols, out_kind, ind);
    O_HIP(hipGetLastError());
    O_HIP(hipMemcpy(inout, d.p, bytes, hipMemcpyDeviceToHost));
    return NPBNN_OK;
}

int npbnn_op_likelihood(int device, int lik_kind, const double* pred, int64_t rows, int32_t cols, const int64_t* labels,
                        const double* targets, int32_t k, const double* inst_w, const double* class_w, int32_t n_class_w,
                        double lik_temp, const double* sigma, double* out) {
    if (!pred || !out || rows < 1 || cols < 1 || lik_kind < 0 || lik_kind >= NPBNN_LIK_NONE) return ofail(NPBNN_E_ARG, "op_likelihood: bad arguments");
    if (lik_kind == NPBNN_LIK_CATEGORICAL) {
        if (!labels) return ofail(NPBNN_E_ARG, "op_likelihood: labels missing");
        for (int64_t r = 0; r < rows; ++r)
            if (labels[r] < 0 || labels[r] >= cols) return ofail(NPBNN_E_ARG, "op_likelihood: label %lld outside 0..%d", (long long)labels[r], cols - 1);
    } else {
        if (!targets || k < 1) return ofail(NPBNN_E_ARG, "op_likelihood: targets missing");
        int need = 1;
        if (lik_kind == NPBNN_LIK_GAUSS) need = k;
        else if (lik_kind == NPBNN_LIK_GAUSS_PRED_SIGMA || lik_kind == NPBNN_LIK_NEGBIN2D) need = 2 * k;
        else if (lik_kind != NPBNN_LIK_POISSON) need = 2;
        if (cols < need) return ofail(NPBNN_E_ARG, "op_likelihood: prediction has %d columns, likelihood needs %d", cols, need);
        if (lik_kind == NPBNN_LIK_GAUSS && !sigma) return ofail(NPBNN_E_ARG, "op_likelihood: sigma missing");
    }
    O_HIP(hipSetDevice(device));
    DevBuf dp, dl, dt, dw, dc, ds, dpart;
    O_HIP(dp.alloc((size_t)rows * cols * 8));
    O_HIP(hipMemcpy(dp.p, pred, (size_t)rows * cols * 8, hipMemcpyHostToDevice));
    if (labels) { O_HIP(dl.alloc((size_t)rows * 8)); O_HIP(hipMemcpy(dl.p, labels, (size_t)rows * 8, hipMemcpyHostToDevice)); }
    if (targets) { O_HIP(dt.alloc((size_t)rows * k * 8)); O_HIP(hipMemcpy(dt.p, targets, (size_t)rows * k * 8, hipMemcpyHostToDevice)); }
    if (inst_w) { O_HIP(dw.alloc((size_t)rows * 8)); O_HIP(hipMemcpy(dw.p, inst_w, (size_t)rows * 8, hipMemcpyHostToDevice)); }
    if (class_w) {
        if (n_class_w < cols) return ofail(NPBNN_E_ARG, "op_likelihood: %d class weights for %d classes", n_class_w, cols);
        O_HIP(dc.alloc((size_t)n_class_w * 8));
        O_HIP(hipMemcpy(dc.p, class_w, (size_t)n_class_w * 8, hipMemcpyHostToDevice));
    }
    if (sigma) { O_HIP(ds.alloc((size_t)k * 8)); O_HIP(hipMemcpy(ds.p, sigma, (size_t)k * 8, hipMemcpyHostToDevice)); }
    const int g = grid_for(rows);
    O_HIP(dpart.alloc((size_t)g * 8));
    hipLaunchKernelGGL(lik_kernel, dim3(g), dim3(kBlock), 0, 0, lik_kind, dp.as<double>(), (long long)rows, cols, dl.as<long long>(),
                       dt.as<double>(), k, inst_w ? dw.as<double>() : nullptr, class_w ? dc.as<double>() : nullptr,
                       sigma ? ds.as<double>() : nullptr, dpart.as<double>());
    O_HIP(hipGetLastError());
    std::vector<double> part((size_t)g);
    O_HIP(hipMemcpy(part.data(), dpart.p, (size_t)g * 8, hipMemcpyDeviceToHost));
    double s = 0.0;
    for (double v : part) s += v;
    const bool tempered = lik_kind <= NPBNN_LIK_GAUSS_PRED_SIGMA;     // the count likelihoods ignore lik_temp (BNN_lik.py)
    *out = tempered ? lik_temp * s : s;
    return NPBNN_OK;
}

int npbnn_op_confusion(int device, const double* pred, int64_t rows, int32_t cols, const int64_t* labels, int64_t* conf,
                       int64_t* pred_counts) {
    if (!pred || !pred_counts || rows < 1 || cols < 1) return ofail(NPBNN_E_ARG, "op_confusion: bad arguments");
    if (labels) {
        if (!conf) return ofail(NPBNN_E_ARG, "op_confusion: conf missing");
        for (int64_t r = 0; r < rows; ++r)
            if (labels[r] < 0 || labels[r] >= cols) return ofail(NPBNN_E_ARG, "op_confusion: label %lld outside 0..%d", (long long)labels[r], cols - 1);
    }
    O_HIP(hipSetDevice(device));
    DevBuf dp, dl, dc, dn;
    O_HIP(dp.alloc((size_t)rows * cols * 8));
    O_HIP(hipMemcpy(dp.p, pred, (size_t)rows * cols * 8, hipMemcpyHostToDevice));
    if (labels) { O_HIP(dl.alloc((size_t)rows * 8)); O_HIP(hipMemcpy(dl.p, labels, (size_t)rows * 8, hipMemcpyHostToDevice)); }
    O_HIP(dc.alloc((size_t)cols * cols * 8));
    O_HIP(hipMemset(dc.p, 0, (size_t)cols * cols * 8));
    O_HIP(dn.alloc((size_t)cols * 8));
    O_HIP(hipMemset(dn.p, 0, (size_t)cols * 8));
    hipLaunchKernelGGL(confusion_kernel, dim3(grid_for(rows)), dim3(kBlock), 0, 0, dp.as<double>(), (long long)rows, cols,
                       labels ? dl.as<long long>() : nullptr, dc.as<unsigned long long>(), dn.as<unsigned long long>());
    O_HIP(hipGetLastError());
    if (labels) O_HIP(hipMemcpy(conf, dc.p, (size_t)cols * cols * 8, hipMemcpyDeviceToHost));
    O_HIP(hipMemcpy(pred_counts, dn.p, (size_t)cols * 8, hipMemcpyDeviceToHost));
    return NPBNN_OK;
}

int npbnn_op_sse(int device, const double* pred, const double* targets, int64_t rows, int32_t cols_pred, int32_t k, int link,
                 double* out_per_col) {
    if (!pred || !targets || !out_per_col || rows < 1 || k < 1 || cols_pred < k || link < 0 || link > 2)
        return ofail(NPBNN_E_ARG, "op_sse: bad arguments");
    O_HIP(hipSetDevice(device));
    DevBuf dp, dt, dpart;
    O_HIP(dp.alloc((size_t)rows * cols_pred * 8));
    O_HIP(hipMemcpy(dp.p, pred, (size_t)rows * cols_pred * 8, hipMemcpyHostToDevice));
    O_HIP(dt.alloc((size_t)rows * k * 8));
    O_HIP(hipMemcpy(dt.p, targets, (size_t)rows * k * 8, hipMemcpyHostToDevice));
    const int g = grid_for(rows);
    O_HIP(dpart.alloc((size_t)g * k * 8));
    hipLaunchKernelGGL(sse_kernel, dim3(g), dim3(kBlock), 0, 0, dp.as<double>(), dt.as<double>(), (long long)rows, cols_pred, k, link,
                       dpart.as<double>());
    O_HIP(hipGetLastError());
    std::vector<double> part((size_t)g * k);
    O_HIP(hipMemcpy(part.data(), dpart.p, part.size() * 8, hipMemcpyDeviceToHost));
    for (int j = 0; j < k; ++j) {
        double s = 0.0;
        for (int b = 0; b < g; ++b) s += part[(size_t)j * g + b];
        out_per_col[j] = s;
    }
    return NPBNN_OK;
}

}  // extern "C"
