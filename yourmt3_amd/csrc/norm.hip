// RMS norm (TP: transformers/models/t5/modeling_t5.py:50-72 -- no mean subtraction, fp32 sum) of
// the fp32 residual stream into the bf16 GEMM operand, one wave per row; and the fp32 -> bf16
// cast used for the log-mel input.  HBM-bound element-wise work: 16-byte loads, 8/16-byte stores.
// Oracle: oracle/ymt3_oracle.py::rmsnorm.
#include "common.h"
#include "kernels.h"

namespace {

__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* __restrict__ x, const float* __restrict__ gain,
                                                      bf16_t* __restrict__ out, int M, int d, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * d);
    const int nv = d >> 2;                          // float4 per row (d % 4 == 0)
    float ss = 0.f;
    for (int i = lane; i < nv; i += 64) {
        const float4 v = xr[i];
        ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    ss = wave_sum(ss);
    const float sc = rsqrtf(ss / (float)d + eps);
    const float4* gr = reinterpret_cast<const float4*>(gain);
    uint2* orow = reinterpret_cast<uint2*>(out + (size_t)row * d);
    for (int i = lane; i < nv; i += 64) {
        const float4 v = xr[i];
        const float4 g = gr[i];
        orow[i] = make_uint2(pack_bf16x2(v.x * sc * g.x, v.y * sc * g.y), pack_bf16x2(v.z * sc * g.z, v.w * sc * g.w));
    }
}

__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        reinterpret_cast<uint2*>(out)[i] = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
    }
}

__global__ __launch_bounds__(256) void broadcast_kernel(const bf16_t* __restrict__ src, float* __restrict__ out, size_t n4, size_t total4) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const uint2 v = reinterpret_cast<const uint2*>(src)[i % n4];
        reinterpret_cast<float4*>(out)[i] = make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u),
                                                        __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
    }
}

}  // namespace

int launch_rmsnorm(const float* x, const float* gain, bf16_t* out, int M, int d, float eps, hipStream_t stream) {
    if (M <= 0) return 0;
    if (d % 4) return -1;
    rmsnorm_kernel<<<(M + 3) / 4, 256, 0, stream>>>(x, gain, out, M, d, eps);
    return 0;
}

int launch_f32_to_bf16(const float* x, bf16_t* out, size_t n, hipStream_t stream) {
    if (n == 0) return 0;
    if (n % 4) return -1;
    const size_t n4 = n / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    cast_kernel<<<(int)blocks, 256, 0, stream>>>(x, out, n4);
    return 0;
}

int launch_broadcast_bf16(const bf16_t* src, float* out, int B, size_t n, hipStream_t stream) {
    if (B <= 0 || n == 0) return 0;
    if (n % 4) return -1;
    const size_t total4 = (size_t)B * n / 4;
    size_t blocks = (total4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    broadcast_kernel<<<(int)blocks, 256, 0, stream>>>(src, out, n / 4, total4);
    return 0;
}
