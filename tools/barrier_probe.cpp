// Measures the cost of a grid-wide barrier inside ONE persistent kernel on MI355X (all workgroups co-resident),
// to compare against the 1.6-1.7 us floor of a dependent kernel launch in a hipGraph (tools/latency_probe.cpp):
// the question is whether a persistent multi-stage decode kernel could beat per-stage launches.
// Variants: barrier only (relaxed / release-acquire), and barrier + a 2 KB-per-workgroup data hand-off that
// is verified (plain stores + agent-scope fences, or agent-scope atomic stores/loads that bypass L2).
// Every spin has a bounded poll count and a sticky abort flag, so the grid always drains.
// (Measurement tool; not part of the product path.)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)

struct Sync { unsigned count; unsigned abort; unsigned bad; unsigned pad[29]; };

template <bool RELACQ>
__device__ __forceinline__ void grid_barrier(Sync* s, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        if (RELACQ) __hip_atomic_fetch_add(&s->count, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(&s->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned polls = 0;
        while (true) {
            const unsigned c = RELACQ ? __hip_atomic_load(&s->count, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)
                                      : __hip_atomic_load(&s->count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (c >= target) break;
            if (++polls > (1u << 18) || __hip_atomic_load(&s->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(&s->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
}

// MODE 0: relaxed barrier only; 1: release/acquire barrier only; 2: plain-store hand-off + release/acquire barrier;
// 3: agent-scope atomic store/load hand-off + relaxed barrier
template <int MODE>
__global__ void k_persist(Sync* s, float* buf, int n_iter) {
    const unsigned G = gridDim.x;
    const int T = blockDim.x;
    for (int it = 0; it < n_iter; ++it) {
        float* cur = buf + (size_t)(it & 1) * G * T;
        const float val = (float)(it * 7 + (int)blockIdx.x);
        if (MODE == 2) cur[(size_t)blockIdx.x * T + threadIdx.x] = val;
        if (MODE == 3) __hip_atomic_store(cur + (size_t)blockIdx.x * T + threadIdx.x, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 3) __builtin_amdgcn_s_waitcnt(0);          // stores issued before the arrive
        grid_barrier<(MODE == 1 || MODE == 2)>(s, (unsigned)(it + 1) * G);
        if (MODE >= 2) {
            const unsigned src = (blockIdx.x + 37u) % G;
            float got;
            if (MODE == 2) got = cur[(size_t)src * T + threadIdx.x];
            else got = __hip_atomic_load(cur + (size_t)src * T + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (got != (float)(it * 7 + (int)src)) atomicAdd(&s->bad, 1u);
        }
    }
}

template <int MODE>
static int run(const char* name, int grid, int block, int n_iter, hipStream_t st) {
    Sync* s; float* buf;
    CK(hipMalloc(&s, sizeof(Sync)));
    CK(hipMalloc(&buf, sizeof(float) * 2 * grid * block));
    double best = 1e30;
    Sync h{};
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemsetAsync(s, 0, sizeof(Sync), st));
        CK(hipMemsetAsync(buf, 0, sizeof(float) * 2 * grid * block, st));
        CK(hipStreamSynchronize(st));
        auto t0 = std::chrono::steady_clock::now();
        k_persist<MODE><<<grid, block, 0, st>>>(s, buf, n_iter);
        CK(hipStreamSynchronize(st));
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (us < best) best = us;
        CK(hipMemcpy(&h, s, sizeof(Sync), hipMemcpyDeviceToHost));
        if (h.abort) break;
    }
    printf("%-46s grid %4d x %4d : %.2f us per barrier%s  (mismatches %u)\n", name, grid, block, best / n_iter,
           h.abort ? "  ABORTED (spin bound hit)" : "", h.bad);
    CK(hipFree(s)); CK(hipFree(buf));
    return h.abort ? 2 : 0;
}

int main() {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    const int N = 2000;
    const int shapes[][2] = {{256, 512}, {256, 1024}, {512, 256}, {128, 512}, {64, 512}, {8, 512}};
    for (auto& sh : shapes) {
        int rc = 0;
        rc |= run<0>("barrier, relaxed atomics", sh[0], sh[1], N, st);
        if (rc) return rc;                                     // after an abort: stop, do not keep launching
        rc |= run<1>("barrier, release/acquire (L2 wb + inv)", sh[0], sh[1], N, st);
        if (rc) return rc;
        rc |= run<2>("plain-store hand-off + rel/acq barrier", sh[0], sh[1], N, st);
        if (rc) return rc;
        rc |= run<3>("agent-scope atomic hand-off + relaxed barrier", sh[0], sh[1], N, st);
        if (rc) return rc;
    }
    return 0;
}
