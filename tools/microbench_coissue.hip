// Can a SIMD of gfx950 run matrix-core work and ordinary vector work at the same time - inside one wave's instruction stream, and
// across two waves of the same SIMD?  (in the pass kernel matrix-pipe busy + VALU active add up to the tile time.)
//   one workgroup per CU; W waves per workgroup; each wave runs R rounds of one of these bodies and reports s_memtime ticks:
//     mode 0  MFMA only:   18 x v_mfma_f32_16x16x32_f16 on 6 independent accumulators (a layer-0 K-step of the pass kernel)
//     mode 1  VALU only:   V x v_fma_f32 on 8 independent chains
//     mode 2  both in one wave: the 18 MFMAs, then the V FMAs (the FMAs do not depend on the MFMAs)
//     mode 3  waves alternate: even waves MFMA only, odd waves VALU only (2 waves per SIMD when W = 8)
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_coissue.hip -o /tmp/mbc && /tmp/mbc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int V>
__device__ __forceinline__ void valu_block(float (&c)[8], float m) {
#pragma unroll
    for (int i = 0; i < V; ++i) c[i & 7] = __builtin_fmaf(c[i & 7], m, 0.25f);
}
__device__ __forceinline__ void mfma_block(f32x4 (&acc)[6], const f16x8& a, const f16x8& b) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int i = 0; i < 6; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
}

template <int V>
__global__ void __launch_bounds__(1024) bench(int mode, int rounds, unsigned long long* ticks, float* sink) {
    const int wave = threadIdx.x >> 6;
    f32x4 acc[6];
    for (int i = 0; i < 6; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float c[8];
    for (int i = 0; i < 8; ++i) c[i] = (float)threadIdx.x * 1e-3f + i;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x & 15) + i); b[i] = (_Float16)(0.002f * i); }
    const float m = 0.999f;
    const bool do_m = mode == 0 || mode == 2 || (mode == 3 && (wave & 4) == 0);     // (wave w sits on SIMD w % 4: waves 0-3 / 4-7 pair up)
    const bool do_v = mode == 1 || mode == 2 || (mode == 3 && (wave & 4) != 0);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < rounds; ++r) {
        if (do_m) mfma_block(acc, a, b);
        if (do_v) valu_block<V>(c, m);
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 6; ++i) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += c[i];
    if (s == 1.2345e30f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) ticks[(size_t)blockIdx.x * 16 + wave] = t1 - t0;
}

int main() {
    unsigned long long* d_ticks; float* d_sink;
    hipMalloc(&d_ticks, 256 * 16 * sizeof(unsigned long long));
    hipMalloc(&d_sink, 4);
    const int rounds = 2000;
    const char* names[4] = {"MFMA only", "VALU only", "both, one wave", "MFMA waves + VALU waves"};
    for (int W : {4, 8, 12}) {
        for (int mode = 0; mode < 4; ++mode) {
            if (mode == 3 && W != 8) continue;
            hipMemset(d_ticks, 0, 256 * 16 * sizeof(unsigned long long));
            hipLaunchKernelGGL(bench<72>, dim3(256), dim3(64 * W), 0, 0, mode, rounds, d_ticks, d_sink);
            hipDeviceSynchronize();
            std::vector<unsigned long long> h(256 * 16);
            hipMemcpy(h.data(), d_ticks, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            double mx = 0, sm = 0; int n = 0;
            for (int b = 0; b < 256; ++b) for (int w = 0; w < W; ++w) { double t = (double)h[b * 16 + w]; sm += t; ++n; if (t > mx) mx = t; }
            printf("%2d waves per CU, %-24s: %.0f cycles per round and wave (mean), %.0f (slowest)   [18 MFMA = 288 pipe cycles, 72 FMA = 288 issue cycles]\n",
                   W, names[mode], sm / n / rounds, mx / rounds);
        }
    }
    return 0;
}
