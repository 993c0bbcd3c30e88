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

// d = 128 (the Perceiver-TF latents: 2.1 M rows of 512 bytes at configs[2], nineteen launches per batch): the kernel above spends a 64-lane wave
// and a quarter of a 256-thread workgroup per row -- half the lanes idle, 524 288 workgroups -- and ran at 3.1 TB/s.  Here a row is 32 lanes, a workgroup
// walks 32 rows per iteration (four in flight per half-wave) over a persistent grid.  Same arithmetic: the 64-lane butterfly's first step (lane ^ 32) added
// the idle half's zeros, the remaining five steps are these, in the same order.
__global__ __launch_bounds__(256) void rmsnorm_d128_kernel(const float* __restrict__ x, const float* __restrict__ gain, bf16_t* __restrict__ out,
                                                           long long M, float eps) {
    const int hw = threadIdx.x >> 5, l = threadIdx.x & 31;
    const float4 g = reinterpret_cast<const float4*>(gain)[l];
    const float4* xv = reinterpret_cast<const float4*>(x);
    uint2* ov = reinterpret_cast<uint2*>(out);
    for (long long row0 = (long long)blockIdx.x * 32; row0 < M; row0 += (long long)gridDim.x * 32) {
        float4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long row = row0 + j * 8 + hw;
            v[j] = row < M ? xv[row * 32 + l] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long row = row0 + j * 8 + hw;
            float ss = 0.f;
            ss += v[j].x * v[j].x + v[j].y * v[j].y + v[j].z * v[j].z + v[j].w * v[j].w;
            ss += lane_xor16(ss);
            ss = add_xor8(ss);
            {
                int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ss), 0x104, 0xF, 0x5, false);      // row_shl:4 -> banks 0, 2   (wave_sum's lane ^ 4 step)
                t = __builtin_amdgcn_update_dpp(t, __builtin_bit_cast(int, ss), 0x114, 0xF, 0xA, false);          // row_shr:4 -> banks 1, 3
                ss += __builtin_bit_cast(float, t);
            }
            ss = add_xor2(ss);
            ss = add_xor1(ss);
            const float sc = rsqrtf(ss / 128.f + eps);
            if (row < M)
                ov[row * 32 + l] = make_uint2(pack_bf16x2(v[j].x * sc * g.x, v[j].y * sc * g.y), pack_bf16x2(v[j].z * sc * g.z, v[j].w * sc * g.w));
        }
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
    if (d == 128 && M >= 4096) {
        const long long iters = ((long long)M + 31) / 32;
        rmsnorm_d128_kernel<<<(int)(iters < 4096 ? iters : 4096), 256, 0, stream>>>(x, gain, out, (long long)M, eps);
        return 0;
    }
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
