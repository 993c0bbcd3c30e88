// What sets the cost of a dependent launch on MI355X?  Chains of 1000 trivial kernels in one hipGraph, wall time / 1000, varying
// one thing at a time: workgroup size, grid, dynamic LDS, leading scalar arguments (kernarg preload), a by-value argument struct.
// Build twice: with and without -mllvm -amdgpu-kernarg-preload-count=16.  (Measurement tool; not part of the product path.)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Big { const float* p[8]; int v[24]; };          // 160 bytes, like DecGemmArgs

__global__ void k_plain(float* out) { if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = 1.f; }
__global__ void k_lds(float* out) {
    extern __shared__ float s[];
    if (threadIdx.x == 0) s[0] = 1.f;
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = s[0];
}
__global__ void k_scalars(const float* a, const float* b, const float* c, const float* d, float* out, int i0, int i1, int i2, int i3) {
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = a[i0] + b[i1] + c[i2] + d[i3];
}
__global__ void k_struct(Big g, float* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = g.p[0][g.v[0]] + g.p[1][g.v[1]] + g.p[2][g.v[2]] + g.p[3][g.v[3]];
}
__global__ void k_scalars_struct(const float* a, const float* b, const float* c, const float* d, float* out, int i0, int i1, int i2, int i3, Big g) {
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = a[i0] + b[i1] + c[i2] + d[i3] + (float)g.v[5];
}

template <typename F>
static int chain(const char* name, F launch, hipStream_t st) {
    hipGraph_t g; hipGraphExec_t e;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 1000; ++i) launch();
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&e, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(e, st)); CK(hipStreamSynchronize(st));
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(e, st));
    CK(hipStreamSynchronize(st));
    printf("%-64s %.2f us per kernel\n", name, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 5000.0);
    hipGraphExecDestroy(e); hipGraphDestroy(g);
    return 0;
}

int main() {
    float *a, *out;
    CK(hipMalloc((void**)&a, 1 << 20)); CK(hipMalloc((void**)&out, 4096));
    CK(hipMemset(a, 0, 1 << 20));
    hipStream_t st; CK(hipStreamCreate(&st));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    Big g{}; for (int i = 0; i < 8; ++i) g.p[i] = a;
    chain("plain, 1 x 64", [&] { k_plain<<<1, 64, 0, st>>>(out); }, st);
    chain("plain, 256 x 512", [&] { k_plain<<<256, 512, 0, st>>>(out); }, st);
    chain("plain, 512 x 512", [&] { k_plain<<<512, 512, 0, st>>>(out); }, st);
    chain("dynamic LDS 16 KB, 256 x 512", [&] { k_lds<<<256, 512, 16 * 1024, st>>>(out); }, st);
    chain("dynamic LDS 64 KB, 256 x 512", [&] { k_lds<<<256, 512, 64 * 1024, st>>>(out); }, st);
    chain("dynamic LDS 143 KB, 256 x 512", [&] { k_lds<<<256, 512, 143 * 1024, st>>>(out); }, st);
    chain("dynamic LDS 143 KB, 128 x 512", [&] { k_lds<<<128, 512, 143 * 1024, st>>>(out); }, st);
    chain("9 leading scalar args (uses them), 256 x 512", [&] { k_scalars<<<256, 512, 0, st>>>(a, a, a, a, out, 1, 2, 3, 4); }, st);
    chain("160-byte struct by value (uses it), 256 x 512", [&] { k_struct<<<256, 512, 0, st>>>(g, out); }, st);
    chain("9 scalars + 160-byte struct, 256 x 512", [&] { k_scalars_struct<<<256, 512, 0, st>>>(a, a, a, a, out, 1, 2, 3, 4, g); }, st);
    return 0;
}
