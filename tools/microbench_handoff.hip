// Groundwork for ordering OVERLAPPING launches by device-side flags instead of kernel boundaries ("next levers"):
// can one workgroup hand a few KB to workgroups of ANOTHER, concurrently running launch on other XCDs - and what does it cost?
//   producer kernel (1 workgroup, stream A): waits a little, writes N doubles, then raises a flag
//   consumer kernel (255 workgroups x 64 threads, stream B, running at the same time): thread 0 of every workgroup polls the
//   flag (bounded spin), then the workgroup reads the N doubles and counts values that are not this round's
// (every consumer first reads the payload of the round before, so its caches hold stale lines when the flag arrives)
// three protocols:  0  payload by plain stores, flag by release store / acquire load at agent scope, payload read by plain loads
//                   1  payload and flag by relaxed agent-scope atomics (write-through), payload read by agent-scope atomic loads
//                   2  as 0, but the consumer reads the payload with agent-scope atomic loads
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_handoff.hip -o /tmp/mbh && /tmp/mbh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int kN = 1536;          // doubles handed over (the patch values of three candidates: 3 x 428, rounded up)
constexpr long long kSpinMax = 4000000;

__global__ void producer(double* payload, int* flag, int round, int mode, unsigned long long* t_raise) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 300) {}                       // 3 us: let the consumers get resident and start polling
    const double v = (double)round;
    if (mode == 1) {
        for (int i = threadIdx.x; i < kN; i += blockDim.x) __hip_atomic_store(payload + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        for (int i = threadIdx.x; i < kN; i += blockDim.x) payload[i] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (mode == 1) {
            __builtin_amdgcn_s_waitcnt(0);
            __hip_atomic_store(flag, round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __hip_atomic_store(flag, round, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        *t_raise = wall_clock64();
    }
}

__global__ void consumer(const double* payload, int* flag, int round, int mode, unsigned* bad, unsigned* timeouts,
                         unsigned long long* t_seen, unsigned long long* fence_ticks) {
    __shared__ int ok;
    // warm this XCD's L2 (and the CU's L1) with the OLD payload first: the question is whether the acquire below drops it
    double warm = 0.0;
    for (int i = threadIdx.x; i < kN; i += blockDim.x) warm += payload[i];
    if (warm == 1.2345e300) atomicAdd(bad, 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        long long spins = 0;
        int f;
        if (mode == 1) {
            while ((f = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != round && ++spins < kSpinMax) {}
        } else {
            while ((f = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != round && ++spins < kSpinMax) {}
            const unsigned long long a = wall_clock64();
            __atomic_thread_fence(__ATOMIC_ACQUIRE);           // (agent scope by default for HIP device code)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            fence_ticks[blockIdx.x] = wall_clock64() - a;
        }
        t_seen[blockIdx.x] = wall_clock64();
        ok = f == round;
        if (!ok) atomicAdd(timeouts, 1u);
    }
    __syncthreads();
    if (!ok) return;
    unsigned n_bad = 0;
    const double want = (double)round;
    for (int i = threadIdx.x; i < kN; i += blockDim.x) {
        const double v = (mode == 0) ? payload[i] : __hip_atomic_load(payload + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        n_bad += v != want;
    }
    if (n_bad) atomicAdd(bad, n_bad);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    double* payload;
    int* flag;
    unsigned *bad, *timeouts;
    unsigned long long *t_raise, *t_seen, *fence_ticks;
    const int G = 255;
    CK(hipMalloc(&payload, kN * sizeof(double)));
    CK(hipMalloc(&flag, 256));
    CK(hipMalloc(&bad, 4));
    CK(hipMalloc(&timeouts, 4));
    CK(hipMalloc(&t_raise, 8));
    CK(hipMalloc(&t_seen, G * 8));
    CK(hipMalloc(&fence_ticks, G * 8));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    for (int mode = 0; mode < 3; ++mode) {
        CK(hipMemset(payload, 0, kN * sizeof(double)));
        CK(hipMemset(flag, 0, 256));
        CK(hipMemset(bad, 0, 4));
        CK(hipMemset(timeouts, 0, 4));
        CK(hipMemset(fence_ticks, 0, G * 8));
        CK(hipDeviceSynchronize());
        double lat_sum = 0, lat_max = 0, fence_sum = 0;
        const int rounds = 300;
        for (int r = 1; r <= rounds; ++r) {
            hipLaunchKernelGGL(consumer, dim3(G), dim3(64), 0, sb, (const double*)payload, flag, r, mode, bad, timeouts, t_seen, fence_ticks);
            hipLaunchKernelGGL(producer, dim3(1), dim3(256), 0, sa, payload, flag, r, mode, t_raise);
            CK(hipStreamSynchronize(sa));
            CK(hipStreamSynchronize(sb));
            unsigned long long tr, ts[G], ft[G];
            CK(hipMemcpy(&tr, t_raise, 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(ts, t_seen, G * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(ft, fence_ticks, G * 8, hipMemcpyDeviceToHost));
            for (int b = 0; b < G; ++b) {
                const double d = ts[b] > tr ? (double)(ts[b] - tr) * 0.01 : 0.0;
                lat_sum += d;
                if (d > lat_max) lat_max = d;
                fence_sum += (double)ft[b] * 0.01;
            }
        }
        unsigned hb = 0, ht = 0;
        CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&ht, timeouts, 4, hipMemcpyDeviceToHost));
        printf("mode %d: %d rounds x %d consumer workgroups: stale values %u of %lld, poll time-outs %u, flag seen %.2f us after it was raised "
               "(max %.2f), acquire fence %.2f us\n", mode, rounds, G, hb, (long long)rounds * G * kN, ht, lat_sum / (rounds * G), lat_max,
               fence_sum / (rounds * G));
    }
    return 0;
}
