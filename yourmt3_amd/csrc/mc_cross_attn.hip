// Cross-attention of the multi-channel decoder (a10 of SURVEY.md section 8): the K channels of one segment decode in
// lock-step and attend to the SAME encoder K/V, so one workgroup per (segment, head) serves all of them -- the
// 64 KB K/V slab and the head's 64 x 512 query-projection weights are read once instead of once per channel
// (13x less traffic at K = 13).  With <= 16 query rows the work is MFMA-shaped:
//
//   q   = R( R(rmsnorm(h_c) * gain) . Wq[head]^T )          16 x 512 by 512 x 64, v_mfma_f32_16x16x32_bf16
//   S^T = K q^T                                              the 4 waves split the T keys; lane = one query column
//   p   = exp(S - max),  fp32; max and sum merged across the waves through LDS
//   O^T = V^T (p_hi + p_lo)^T                                p split into two bf16 terms so that P.V keeps the
//                                                            fp32-softmax accuracy of the numerics contract;
//                                                            V^T fragments by ds_read_b64_tr_b16 from a wave-private strip
//
// Same arithmetic contract as dec_attn_kernel<false, true>; oracle: oracle/ymt3_oracle.py::decoder_step.
#include "common.h"
#include "kernels.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int DKV = 64;
constexpr int XP = 520;     // xs row pitch (bf16 elements): 512 + 8
constexpr int QP = 72;      // qs / sV row pitch (bf16 elements): 64 + 8 -> 144 bytes

template <int T>
__global__ __launch_bounds__(256) void mc_cross_attn_kernel(const float* __restrict__ pX, const float* __restrict__ pGain, const float* __restrict__ pSsq,
                                                            const bf16_t* __restrict__ pWq, const bf16_t* __restrict__ pK, const bf16_t* __restrict__ pV,
                                                            int row0, int n_channels, McCrossArgs a) {
    // leading scalar arguments: kernarg preload (see dec_gemm_kernel in decode.hip)
    constexpr int NKT = T / 64;              // 16-key tiles per wave
    constexpr int KPW = T / 4;               // keys per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem);                         // [16][XP]
    bf16_t* qs = xs + 16 * XP;                                            // [16][QP]
    bf16_t* sV = qs + 16 * QP;                                            // [4][KPW][QP]
    float* so = reinterpret_cast<float*>(sV + 4 * KPW * QP);              // [4][64][16]
    float* smx = so + 4 * 64 * 16;                                        // [4][16]
    float* ssum = smx + 64;                                               // [4][16]
    float* sscale = ssum + 64;                                            // [16]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
    const int h = blockIdx.x, seg = blockIdx.y;
    const int nc = n_channels, r0 = row0 + seg * nc;
    const size_t slab = ((size_t)seg * a.H + h) * T * DKV;

    // --- every operand is requested now, in the order of first use (vector loads return in order): the norm's inputs (sum(h^2) partials, the
    // channel rows, the gain), the head's query-projection fragments, then the K fragments and the V rows, which stay in flight under the projection
    float ss = 0.f;
    {
        const int row = tid >> 4, part = tid & 15;
        if (row < nc) ss = pSsq[(size_t)(2 * part) * a.ssq_stride + r0 + row] + pSsq[(size_t)(2 * part + 1) * a.ssq_stride + r0 + row];
    }
    float4 xr[4][2], gr[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) gr[i] = reinterpret_cast<const float4*>(pGain)[lane + 64 * i];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wave + 4 * j;
            xr[j][i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < nc) xr[j][i] = reinterpret_cast<const float4*>(pX + (size_t)(r0 + row) * 512)[lane + 64 * i];
        }
    u32x4 kf[NKT][2], vv[KPW / 8], wf[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
        wf[ks] = *reinterpret_cast<const u32x4*>(pWq + ((size_t)h * DKV + wave * 16 + li) * 512 + ks * 32 + g * 8);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            kf[kt][ks] = *reinterpret_cast<const u32x4*>(pK + slab + (size_t)(wave * KPW + kt * 16 + li) * DKV + ks * 32 + g * 8);
#pragma unroll
    for (int i = 0; i < KPW / 8; ++i)
        vv[i] = *reinterpret_cast<const u32x4*>(pV + slab + (size_t)(wave * KPW + i * 8 + (lane >> 3)) * DKV + (lane & 7) * 8);
    __builtin_amdgcn_sched_barrier(0);

    // --- RMS norm of the channel rows into LDS (rows >= n_channels are zero)
    {
        const int row = tid >> 4, part = tid & 15;
        ss = add_xor8(sum8(ss));                 // lanes ^1, ^2, ^4, ^8 by DPP moves (common.h): same pairs, same bits
        if (part == 0) sscale[row] = rsqrtf(ss / 512.f + a.eps);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = wave + 4 * j;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            uint2 pk = make_uint2(0u, 0u);
            if (row < nc) {
                const float4 v = xr[j][i];
                const float4 gg = gr[i];
                const float sc = sscale[row];
                pk = make_uint2(pack_bf16x2(v.x * sc * gg.x, v.y * sc * gg.y), pack_bf16x2(v.z * sc * gg.z, v.w * sc * gg.w));
            }
            *reinterpret_cast<uint2*>(xs + row * XP + (lane + 64 * i) * 4) = pk;
        }
    }
    __syncthreads();

    // --- query projection: wave w owns head dims 16w .. 16w+15
    {
        f32x4 qa = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xs + li * XP + ks * 32 + g * 8);
            qa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[ks]), xf, qa, 0, 0, 0);
        }
        // lane: query li, head dims 16w + 4g + r
        *reinterpret_cast<uint2*>(qs + li * QP + wave * 16 + 4 * g) = make_uint2(pack_bf16x2(qa[0], qa[1]), pack_bf16x2(qa[2], qa[3]));
    }
    __syncthreads();

    // --- scores of this wave's keys: S^T tiles, lane = query li, keys kt*16 + 4g + r
    bf16x8 qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qs + li * QP + ks * 32 + g * 8);
    f32x4 s[NKT];
    float mx = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kf[kt][ks]), qf[ks], s[kt], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
    }
    mx = fmaxf(mx, lane_xor16(mx));
    mx = fmaxf(mx, lane_xor32(mx));
    if (g == 0) smx[wave * 16 + li] = mx;
    __syncthreads();
    const float M = fmaxf(fmaxf(smx[li], smx[16 + li]), fmaxf(smx[32 + li], smx[48 + li]));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = __expf(s[kt][r] - M);
            s[kt][r] = e;
            sum += e;
        }
    sum += lane_xor16(sum);
    sum += lane_xor32(sum);
    if (g == 0) ssum[wave * 16 + li] = sum;

    // the wave's V rows -> its private strip, consumed by this wave's transposed reads below (a wave's LDS accesses execute in order: no barrier)
    bf16_t* myV = sV + wave * KPW * QP;
#pragma unroll
    for (int i = 0; i < KPW / 8; ++i) *reinterpret_cast<u32x4*>(myV + (i * 8 + (lane >> 3)) * QP + (lane & 7) * 8) = vv[i];

    // --- partial O^T over this wave's keys: k-step = key tiles (2kp, 2kp+1); P as hi + lo bf16 terms
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
#pragma unroll
    for (int kp = 0; kp < NKT / 2; ++kp) {
        bf16x8 ph, pl;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e0 = s[2 * kp][r], e1 = s[2 * kp + 1][r];
            ph[r] = (__bf16)e0;     pl[r] = (__bf16)(e0 - (float)ph[r]);
            ph[4 + r] = (__bf16)e1; pl[4 + r] = (__bf16)(e1 - (float)ph[4 + r]);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const int rr = (2 * kp) * 16 + 4 * g + (li >> 2);
            const int col = dt * 16 + 4 * (li & 3);
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(myV + rr * QP + col));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(myV + (rr + 16) * QP + col));
            const bf16x8 vf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, ph, o[dt], 0, 0, 0);
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pl, o[dt], 0, 0, 0);
        }
    }
    // lane: query li, head dims dt*16 + 4g + r
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) so[(wave * 64 + dt * 16 + 4 * g + r) * 16 + li] = o[dt][r];
    __syncthreads();

    // --- merge the four waves in a fixed order, normalise, store
    {
        const int q = tid & 15, d0 = tid >> 4;
        if (q < nc) {
            const float L = (ssum[q] + ssum[16 + q]) + (ssum[32 + q] + ssum[48 + q]);
            bf16_t* orow = a.out + (size_t)(r0 + q) * a.H * DKV + h * DKV;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int d = d0 + 16 * j;
                const float v = (so[(0 * 64 + d) * 16 + q] + so[(1 * 64 + d) * 16 + q]) + (so[(2 * 64 + d) * 16 + q] + so[(3 * 64 + d) * 16 + q]);
                orow[d] = f2bf(v / L);
            }
        }
    }
}

template <int T>
constexpr size_t mc_lds() {
    return (size_t)(16 * XP + 16 * QP + 4 * (T / 4) * QP) * 2 + (size_t)(4 * 64 * 16 + 64 + 64 + 16) * 4;
}

template <int T>
int launch_mc(const McCrossArgs& a, hipStream_t stream) {
    if (a.k == nullptr)
        return hipFuncSetAttribute(reinterpret_cast<const void*>(mc_cross_attn_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)mc_lds<T>()) == hipSuccess ? 0 : -2;
    mc_cross_attn_kernel<T><<<dim3(a.H, a.n_seg), 256, mc_lds<T>(), stream>>>(a.x_f32, a.gain, a.ssq, a.wq, a.k, a.v, a.row0, a.n_channels, a);
    return 0;
}

}  // namespace

int init_mc_cross_kernels() {
    McCrossArgs z{};
    return launch_mc<128>(z, nullptr) | launch_mc<256>(z, nullptr) | launch_mc<512>(z, nullptr);
}

// n_channels in [2, 16], T in {128, 256, 512}; anything else -> -1 and the caller falls back to dec_attn_kernel
int launch_mc_cross_attention(const McCrossArgs& a, hipStream_t stream) {
    if (a.n_seg <= 0) return 0;
    if (a.n_channels < 2 || a.n_channels > 16) return -1;
    switch (a.T) {
        case 128: return launch_mc<128>(a, stream);
        case 256: return launch_mc<256>(a, stream);
        case 512: return launch_mc<512>(a, stream);
        default: return -1;
    }
}
