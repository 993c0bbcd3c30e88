// Does the 256 MB Infinity Cache (memory-side) make a re-read of a streamed buffer faster than HBM?  Times a 16-byte-per-lane
// streaming read of 32..192 MB cold (right after a 1 GB write pushed everything out) and again right after (L2 holds 32 MB at
// most, so whatever is faster beyond that comes from the Infinity Cache), with plain and non-temporal loads, and after a
// low-occupancy "prefetch" pass (one wave per CU) -- the question behind prefetching the next layer's KV cache on a second
// stream while the latency-bound GEMMs leave HBM idle.  (Measurement tool; not part of the product path.)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void k_read(const u32x4* src, size_t n16, unsigned* sink) {
    u32x4 acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride * 4) {
        u32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const size_t k = i + j * stride;
            const u32x4* p = src + (k < n16 ? k : i);
            v[j] = NT ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc ^= v[j];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[blockIdx.x] = 1;
}
__global__ void k_fill(u32x4* dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = (u32x4){(unsigned)i, 1u, 2u, 3u};
}

static float timed(hipStream_t st, hipEvent_t e0, hipEvent_t e1, void (*launch)(hipStream_t, const u32x4*, size_t, unsigned*), const u32x4* b, size_t n16, unsigned* sink) {
    hipEventRecord(e0, st); launch(st, b, n16, sink); hipEventRecord(e1, st); hipStreamSynchronize(st);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f;
}
static void l_plain(hipStream_t st, const u32x4* b, size_t n, unsigned* s) { k_read<false><<<2048, 256, 0, st>>>(b, n, s); }
static void l_nt(hipStream_t st, const u32x4* b, size_t n, unsigned* s) { k_read<true><<<2048, 256, 0, st>>>(b, n, s); }
static void l_prefetch(hipStream_t st, const u32x4* b, size_t n, unsigned* s) { k_read<false><<<256, 64, 0, st>>>(b, n, s); }
static void l_prefetch_nt(hipStream_t st, const u32x4* b, size_t n, unsigned* s) { k_read<true><<<256, 64, 0, st>>>(b, n, s); }

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    u32x4 *buf, *flush; unsigned* sink;
    const size_t flush_bytes = (size_t)1 << 30;
    CK(hipMalloc(&buf, (size_t)192 << 20)); CK(hipMalloc(&flush, flush_bytes)); CK(hipMalloc(&sink, 4096 * 4));
    k_fill<<<2048, 256, 0, st>>>(buf, ((size_t)192 << 20) / 16);
    CK(hipStreamSynchronize(st));
    for (int mb : {32, 64, 128, 192}) {
        const size_t n16 = ((size_t)mb << 20) / 16;
        auto flush_all = [&]() { k_fill<<<2048, 256, 0, st>>>(flush, flush_bytes / 16); hipStreamSynchronize(st); };
        auto gbps = [&](float us) { return mb * 1.048576 / us * 1e3; };   // GB/s
        flush_all();
        const float c1 = timed(st, e0, e1, l_plain, buf, n16, sink), w1 = timed(st, e0, e1, l_plain, buf, n16, sink), w2 = timed(st, e0, e1, l_plain, buf, n16, sink);
        flush_all();
        const float c2 = timed(st, e0, e1, l_nt, buf, n16, sink), w3 = timed(st, e0, e1, l_nt, buf, n16, sink);
        flush_all();
        const float p1 = timed(st, e0, e1, l_prefetch, buf, n16, sink), w4 = timed(st, e0, e1, l_nt, buf, n16, sink);
        flush_all();
        const float p2 = timed(st, e0, e1, l_prefetch_nt, buf, n16, sink), w5 = timed(st, e0, e1, l_nt, buf, n16, sink);
        printf("%3d MB  plain: cold %6.1f us (%5.0f GB/s)  again %6.1f (%5.0f)  again %6.1f | nt: cold %6.1f (%5.0f)  again %6.1f (%5.0f) | "
               "after 1-wave/CU prefetch (%6.1f us): nt read %6.1f (%5.0f) | after nt prefetch (%6.1f us): nt read %6.1f (%5.0f)\n",
               mb, c1, gbps(c1), w1, gbps(w1), w2, c2, gbps(c2), w3, gbps(w3), p1, w4, gbps(w4), p2, w5, gbps(w5));
    }
    return 0;
}
