// Calibrates the per-kernel floor for chains of small dependent kernels on MI355X: N launches of each
// variant captured in one hipGraph, wall time / N.  Variants differ in the number of dependent memory
// round trips inside the kernel.  (Measurement tool; not part of the product path.)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)

struct Args { const float* a; const int* idx; float* out; int n; int pad[16]; };

__global__ void k_empty(Args) {}
__global__ void k_store(Args p) { p.out[blockIdx.x * blockDim.x + threadIdx.x] = 1.0f; }
__global__ void k_load_store(Args p) { const int i = blockIdx.x * blockDim.x + threadIdx.x; p.out[i] = p.a[i] + 1.0f; }
__global__ void k_load2_store(Args p) {   // two dependent loads
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = p.idx[i];
    p.out[i] = p.a[j] + 1.0f;
}
__global__ void k_load_sync_store(Args p) {   // load -> LDS reduce -> store
    __shared__ float s[512];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    s[threadIdx.x] = p.a[i];
    __syncthreads();
    p.out[i] = s[threadIdx.x ^ 1] + 1.0f;
}

// each workgroup pulls `per_wg` bytes (16 B per lane per load, all loads independent) from the buffer the
// PREVIOUS kernel wrote (ping-pong), reduces, and writes 2 KB: the shape of a skinny decode GEMM
template <int NLOAD>
__global__ __launch_bounds__(512) void k_pull(const float4* src, float4* dst, int wg_stride4) {
    const float4* p = src + (size_t)blockIdx.x * wg_stride4 + threadIdx.x;
    float4 v[NLOAD];
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) v[i] = p[i * 512];
    float4 s = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) { s.x += v[i].x; s.y += v[i].y; s.z += v[i].z; s.w += v[i].w; }
    dst[(size_t)(blockIdx.x % 64) * 512 + threadIdx.x] = s;
}
template <int NLOAD>
static int run_pull(int grid, hipStream_t st, float4* a, float4* b, int n) {
    hipGraph_t g; hipGraphExec_t e;
    const int stride4 = NLOAD * 512;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < n; ++i) k_pull<NLOAD><<<grid, 512, 0, st>>>((i & 1) ? b : a, (i & 1) ? a : b, stride4);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&e, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(e, st)); CK(hipStreamSynchronize(st));
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(e, st));
    CK(hipStreamSynchronize(st));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (5.0 * n);
    printf("pull %3d KB per WG    grid %4d x 512 : %.2f us per kernel\n", NLOAD * 8, grid, us);
    hipGraphExecDestroy(e); hipGraphDestroy(g);
    return 0;
}

template <typename K>
static int run(const char* name, K kern, Args a, int grid, int block, hipStream_t st, int n) {
    hipGraph_t g; hipGraphExec_t e;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < n; ++i) kern<<<grid, block, 0, st>>>(a);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&e, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(e, st)); CK(hipStreamSynchronize(st));
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(e, st));
    CK(hipStreamSynchronize(st));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (5.0 * n);
    printf("%-20s grid %4d x %3d : %.2f us per kernel\n", name, grid, block, us);
    hipGraphExecDestroy(e); hipGraphDestroy(g);
    return 0;
}

int main() {
    const int N = 1 << 20;
    float *a, *out; int* idx;
    CK(hipMalloc((void**)&a, N * 4)); CK(hipMalloc((void**)&out, N * 4)); CK(hipMalloc((void**)&idx, N * 4));
    std::vector<int> h(N); for (int i = 0; i < N; ++i) h[i] = (i * 7919) % N;
    CK(hipMemcpy(idx, h.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemset(a, 0, N * 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    Args p{a, idx, out, N, {}};
    for (int grid : {32, 128, 512}) {
        run("empty", k_empty, p, grid, 512, st, 1000);
        run("store", k_store, p, grid, 512, st, 1000);
        run("load->store", k_load_store, p, grid, 512, st, 1000);
        run("load->load->store", k_load2_store, p, grid, 512, st, 1000);
        run("load->lds->store", k_load_sync_store, p, grid, 512, st, 1000);
    }
    float4 *pa, *pb;
    CK(hipMalloc((void**)&pa, (size_t)384 * 32 * 512 * 16)); CK(hipMalloc((void**)&pb, (size_t)384 * 32 * 512 * 16));
    CK(hipMemset(pa, 0, (size_t)384 * 32 * 512 * 16)); CK(hipMemset(pb, 0, (size_t)384 * 32 * 512 * 16));
    for (int grid : {32, 128, 384}) {
        run_pull<1>(grid, st, pa, pb, 1000);
        run_pull<4>(grid, st, pa, pb, 1000);
        run_pull<8>(grid, st, pa, pb, 1000);
        run_pull<16>(grid, st, pa, pb, 1000);
        run_pull<32>(grid, st, pa, pb, 1000);
    }
    // where do re-read weights come from?  read-only pulls of 32 KB per WG, 128 WGs (4 MB per launch), cycling over
    // `nbuf` distinct 4 MB buffers: 1 -> whatever survives a kernel boundary closest to the CU; 8 (32 MB) -> beyond one
    // XCD's 4 MB L2; 48 (192 MB) -> Infinity Cache; 160 (640 MB) -> HBM
    {
        const size_t per = 128 * 4 * 512 * 16;      // bytes per buffer
        float4* big; CK(hipMalloc((void**)&big, per * 160)); CK(hipMemset(big, 0, per * 160));
        for (int nbuf : {1, 8, 48, 160}) {
            hipGraph_t g; hipGraphExec_t e;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < 960; ++i)
                k_pull<4><<<128, 512, 0, st>>>(big + (size_t)(i % nbuf) * (per / 16), pb, 4 * 512);
            CK(hipStreamEndCapture(st, &g));
            CK(hipGraphInstantiate(&e, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(e, st)); CK(hipStreamSynchronize(st));
            auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(e, st));
            CK(hipStreamSynchronize(st));
            printf("read-only 32 KB/WG x 128 WGs cycling %3d x 4 MB buffers: %.2f us per kernel\n", nbuf,
                   std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (3.0 * 960));
            hipGraphExecDestroy(e); hipGraphDestroy(g);
        }
    }
    // does it matter that consecutive kernels are DIFFERENT code?  same tiny kernel x1000 vs 5 distinct kernels in rotation
    {
        auto run_rot = [&](int nk, const char* name) -> int {
            hipGraph_t g; hipGraphExec_t e;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < 1000; ++i) {
                switch (i % nk) {
                    case 0: k_store<<<128, 512, 0, st>>>(p); break;
                    case 1: k_load_store<<<128, 512, 0, st>>>(p); break;
                    case 2: k_load2_store<<<128, 512, 0, st>>>(p); break;
                    case 3: k_load_sync_store<<<128, 512, 0, st>>>(p); break;
                    default: k_pull<4><<<128, 512, 0, st>>>(pa, pb, 4 * 512); break;
                }
            }
            CK(hipStreamEndCapture(st, &g));
            CK(hipGraphInstantiate(&e, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(e, st)); CK(hipStreamSynchronize(st));
            auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(e, st));
            CK(hipStreamSynchronize(st));
            printf("%-44s %.2f us per kernel\n", name, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 5000.0);
            hipGraphExecDestroy(e); hipGraphDestroy(g);
            return 0;
        };
        run_rot(1, "rotation of 1 kernel (k_store)");
        run_rot(2, "rotation of 2 distinct kernels");
        run_rot(5, "rotation of 5 distinct kernels");
    }
    return 0;
}
