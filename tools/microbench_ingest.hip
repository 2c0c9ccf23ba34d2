// What can a compute unit take in per clock?  The weight-streamed path (npbnn_amd/csrc/npbnn_wide.hip.h) moves BOTH operands of a
// tiled matrix product into LDS by LDS-DMA; its first version ran at ~10 B/clk per compute unit whatever the tiling.  This measures
// the ceiling directly: every workgroup streams 1-KiB pieces (16 rows x 64 B, the path's piece shape, or 1 KiB contiguous) from a
// region of a given size (small: L2 hits; 128 MB: Infinity Cache; 2 GB: HBM) with a given number of pieces in flight per wave,
//   mode 0: global_load_lds_dwordx4 into LDS      mode 1: global_load_dwordx4 into registers
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_ingest.hip -o /tmp/mbi && /tmp/mbi
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(1))) const void gvoid;
typedef __attribute__((address_space(3))) void lvoid;
typedef float f32x4 __attribute__((ext_vector_type(4)));

// region: floats; every workgroup walks its own window of `win` floats, `iters` pieces per wave
template <int MODE, int DEPTH>
__global__ void __launch_bounds__(1024) ingest(const float* __restrict__ src, long long region, long long win, int iters, int strided, long long row_stride, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int n = lane & 15, kq = lane >> 4;
    const long long base = ((long long)blockIdx.x * win) % region;
    char* ring = smem + wave * DEPTH * 1024;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x4 r[DEPTH];
    // piece p of this wave: strided = 16 rows x 64 B (rows row_stride floats apart), else 1 KiB contiguous
    auto addr = [&](int p) -> const float* {
        long long off;
        if (strided) {
            const long long tile = (long long)(p * nw + wave);
            const long long cols = row_stride / 16;                       // pieces along a row
            off = (tile / cols) * 16 * row_stride + (tile % cols) * 16 + (long long)n * row_stride + 4 * kq;
        } else {
            off = ((long long)(p * nw + wave)) * 256 + lane * 4;
        }
        return src + (base + off % win) % region;
    };
    for (int p = 0; p < DEPTH && p < iters; ++p) {
        if (MODE == 0) __builtin_amdgcn_global_load_lds((gvoid*)addr(p), (lvoid*)(ring + p * 1024), 16, 0, 0);
        else r[p] = *reinterpret_cast<const f32x4*>(addr(p));
    }
    for (int p = 0; p < iters; ++p) {
        if (MODE == 0) {
            if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            acc += *reinterpret_cast<const f32x4*>(ring + (p % DEPTH) * 1024 + lane * 16);
            if (p + DEPTH < iters) __builtin_amdgcn_global_load_lds((gvoid*)addr(p + DEPTH), (lvoid*)(ring + (p % DEPTH) * 1024), 16, 0, 0);
        } else {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d)
                if (p % DEPTH == d) {
                    acc += r[d];
                    if (p + DEPTH < iters) r[d] = *reinterpret_cast<const f32x4*>(addr(p + DEPTH));
                }
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345e30f) sink[0] = acc[0];
}

template <int MODE, int DEPTH>
void run(const float* d, long long region, const char* what, int waves, int strided, long long row_stride, float* sink) {
    const int grid = 256, iters = 512;
    const long long win = (long long)iters * waves * 256;          // floats a workgroup touches
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(ingest<MODE, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const size_t lds = (size_t)waves * DEPTH * 1024;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((ingest<MODE, DEPTH>), dim3(grid), dim3(waves * 64), lds, 0, d, region, win, iters, strided, row_stride, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 10.0 * grid * waves * iters * 1024.0;
    printf("%-14s %s %s depth %d, %2d waves/CU: %6.2f TB/s = %5.1f GB/s per CU = %4.1f B/clk/CU at 2.4 GHz\n", what, MODE ? "load->regs" : "LDS-DMA   ",
           strided ? "16x64B rows" : "1KiB contig", DEPTH, waves, bytes / ms / 1e9, bytes / ms / 1e6 / grid, bytes / ms / 1e6 / grid / 2.4);
}

int main() {
    const long long big = 2ll << 30;       // bytes
    float* d = nullptr;
    float* sink = nullptr;
    if (hipMalloc(&d, big) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&sink, 64);
    hipMemset(d, 0, big);
    struct { long long bytes; const char* what; } regions[] = {{1ll << 20, "L2 (1 MB)"}, {96ll << 20, "MALL (96 MB)"}, {2ll << 30, "HBM (2 GB)"}};
    for (auto& rg : regions) {
        const long long region = rg.bytes / 4;
        for (int waves : {4, 8, 16}) {
            run<0, 4>(d, region, rg.what, waves, 0, 0, sink);
            run<0, 8>(d, region, rg.what, waves, 0, 0, sink);
            run<0, 8>(d, region, rg.what, waves, 1, 1024, sink);
            run<1, 4>(d, region, rg.what, waves, 0, 0, sink);
            run<1, 8>(d, region, rg.what, waves, 0, 0, sink);
        }
    }
    return 0;
}
