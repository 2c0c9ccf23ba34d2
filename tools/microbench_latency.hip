// Micro-benchmarks behind DESIGN.md's step-kernel numbers: how long do tiny single-workgroup kernels take on MI355X?
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_latency.hip -o /tmp/mb && rocprofv3 --kernel-trace --stats -- /tmp/mb
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty(int* p) { if (threadIdx.x == 9999) p[0] = 1; }
__global__ void k_load1(const double* a, double* out) { out[threadIdx.x] = a[threadIdx.x * 33]; }
__global__ void k_chain2(const int* idx, const double* a, double* out) { out[threadIdx.x] = a[idx[threadIdx.x]]; }
__global__ void k_chain3(const int* idx, const int* idx2, const double* a, double* out) { out[threadIdx.x] = a[idx2[idx[threadIdx.x]]]; }
__global__ void k_reduce(const double* a, double* out) {
    __shared__ double red[16];
    double s = a[threadIdx.x * 33] + a[(threadIdx.x + 1024) * 33];
    for (int sh = 32; sh > 0; sh >>= 1) {
        int lo = __double2loint(s), hi = __double2hiint(s);
        lo = __shfl_xor(lo, sh); hi = __shfl_xor(hi, sh);
        s += __hiloint2double(hi, lo);
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0; for (int i = 0; i < 16; ++i) t += red[i]; out[0] = t; }
}
__global__ void k_writer(double* a, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) a[i * 33] = i; }
int main() {
    double *a, *out; int *idx, *idx2;
    hipMalloc(&a, 4096 * 33 * 8); hipMalloc(&out, 4096 * 8); hipMalloc(&idx, 4096 * 4); hipMalloc(&idx2, 4096 * 4);
    hipMemset(a, 0, 4096 * 33 * 8); hipMemset(idx, 0, 4096 * 4); hipMemset(idx2, 0, 4096 * 4);
    for (int rep = 0; rep < 50; ++rep) {
        hipLaunchKernelGGL(k_writer, dim3(256), dim3(16), 0, 0, a, 4096);     // partials written by many CUs, as the eval kernel does
        hipLaunchKernelGGL(k_empty, dim3(1), dim3(1024), 0, 0, idx);
        hipLaunchKernelGGL(k_load1, dim3(1), dim3(1024), 0, 0, a, out);
        hipLaunchKernelGGL(k_writer, dim3(256), dim3(16), 0, 0, a, 4096);
        hipLaunchKernelGGL(k_reduce, dim3(1), dim3(1024), 0, 0, a, out);
        hipLaunchKernelGGL(k_chain2, dim3(1), dim3(1024), 0, 0, idx, a, out);
        hipLaunchKernelGGL(k_chain3, dim3(1), dim3(1024), 0, 0, idx, idx2, a, out);
    }
    hipDeviceSynchronize();
    printf("done\n");
    return 0;
}
