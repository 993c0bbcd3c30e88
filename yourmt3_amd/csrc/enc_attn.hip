// Encoder self-attention (a4/a5 of SURVEY.md section 8): softmax(Q K^T + rel_bias) V per (segment,
// head), no 1/sqrt(d) scale (TP: transformers/models/t5/modeling_t5.py:196-197), bias added
// before the softmax (:159-167), one bias table shared by all layers (:739-742).
//
// One workgroup = one (segment, head): K and V of the whole segment (T <= 512 keys x 64) are staged in LDS ONCE and two
// groups of four waves walk the segment's even / odd 64-query tiles (16 queries per wave), the next tile's Q fragments in flight under the
// current tile's math.  (One workgroup per query tile re-staged the same 64 KB four times: 134 MB of L2 -> LDS traffic per
// layer for 8.6 GFLOP; the kernel was bound by that, 38 us.)  Scores are computed TRANSPOSED, S^T = K Q^T, with
// v_mfma_f32_16x16x32_bf16 so that each lane owns ONE query column and 4 consecutive keys per
// accumulator: the softmax reductions are lane-local plus two shuffles, and the exponentials,
// rounded to bf16, are already the B operand of the second product O^T = V^T P^T.  V^T fragments come
// from the row-major LDS image through ds_read_b64_tr_b16 (hardware transpose), so nothing is
// transposed in memory.  All T keys are resident, so there is no online rescale: max and sum are exact.
//
// Numerics contract: e = exp(s - max) rounded to bf16 for P.V, the normaliser sums the
// unrounded fp32 e, output rounded to bf16.  Oracle: oracle/ymt3_oracle.py::attention(round_p=True).
#include "common.h"
#include "kernels.h"

namespace {

constexpr int DKV = 64;
constexpr int ROWB = 144;   // LDS row pitch in bytes (128 + 16 pad: spreads rows over the banks)

// q rows: q + (b*T + t) * ldq + h*64;  k / v rows: k|v + (b*T + t) * ldkv + h*64.  The T5 encoder passes one fused
// qkv buffer (k = qkv + inner, v = qkv + 2*inner, ldq = ldkv = 3*inner); the latent cross-attention (a9) passes
// the latent queries and the frame K/V as separate buffers.
template <int T> constexpr int enc_attn_threads() { return T >= 512 ? 256 : 512; }   // 512 keys: 128 score registers per lane, one wave per SIMD

template <int T>
__global__ __launch_bounds__(enc_attn_threads<T>()) void enc_attn_kernel(const bf16_t* __restrict__ qp, int ldq, const bf16_t* __restrict__ kp,
                                                       const bf16_t* __restrict__ vp, int ldkv,
                                                       const float* __restrict__ bias_off, bf16_t* __restrict__ out, int H) {
    constexpr int NT = T / 16, NTH = enc_attn_threads<T>(), NG = NTH / 256;   // NG groups of four waves share the query tiles
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sK = smem;
    char* sV = smem + T * ROWB;
    float* sB = reinterpret_cast<float*>(smem + 2 * T * ROWB);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = (tid >> 6) & 3, half = tid >> 8;       // waves 0-3 take query tiles 0, NG, ..; waves 4-7 (if any) tiles 1, 1 + NG, ..
    const int h = blockIdx.y, b = blockIdx.z;
    const int inner = H * DKV;
    const size_t kvoff = (size_t)b * T * ldkv + h * DKV;

    {
        // all of this thread's K/V chunks are requested before the first is written to LDS (one memory round trip)
        constexpr int NCH = (T * 8 + NTH - 1) / NTH;
        uint4 kc[NCH], vc[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int idx = tid + i * NTH, row = (idx < T * 8 ? idx : 0) >> 3, ch = idx & 7;
            const size_t o = kvoff + (size_t)row * ldkv + ch * 8;
            kc[i] = *reinterpret_cast<const uint4*>(kp + o);
            vc[i] = *reinterpret_cast<const uint4*>(vp + o);
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int idx = tid + i * NTH, row = idx >> 3, ch = idx & 7;
            if (idx < T * 8) {
                *reinterpret_cast<uint4*>(sK + row * ROWB + ch * 16) = kc[i];
                *reinterpret_cast<uint4*>(sV + row * ROWB + ch * 16) = vc[i];
            }
        }
    }
    for (int i = tid; i < 2 * T - 1; i += NTH) sB[i] = bias_off[(size_t)h * (2 * T - 1) + i];

    const int g = lane >> 4, li = lane & 15;
    auto load_q = [&](int qt, bf16x8 (&qf)[2]) {
        const int q = qt * 64 + wave * 16 + li;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            qf[ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(qp + ((size_t)b * T + q) * ldq + h * DKV + ks * 32 + g * 8));
    };
    bf16x8 qf[2], qn[2];
    if (half < T / 64) load_q(half, qf);
    __syncthreads();
    for (int qt = half; qt < T / 64; qt += NG) {
    const int q = qt * 64 + wave * 16 + li;             // this lane's query (position in segment)
    if (qt + NG < T / 64) load_q(qt + NG, qn);

    // S^T tiles: lane -> query q, keys kt*16 + 4g + r
    f32x4 s[NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 kf = __builtin_bit_cast(
                bf16x8, *reinterpret_cast<const uint4*>(sK + (kt * 16 + li) * ROWB + (ks * 4 + g) * 16));
            s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s[kt], 0, 0, 0);
        }
    }
    float mx = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = kt * 16 + 4 * g + r;
            s[kt][r] += sB[key - q + T - 1];
            mx = fmaxf(mx, s[kt][r]);
        }
    mx = fmaxf(mx, lane_xor16(mx));
    mx = fmaxf(mx, lane_xor32(mx));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = __expf(s[kt][r] - mx);
            s[kt][r] = e;
            sum += e;
        }
    sum += lane_xor16(sum);
    sum += lane_xor32(sum);

    // O^T[d][q] = sum_key V^T[d][key] P^T[key][q]; k-step = key tiles (2kp, 2kp+1)
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
#pragma unroll
    for (int kp = 0; kp < NT / 2; ++kp) {
        bf16x8 pf;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pf[r] = (__bf16)s[2 * kp][r];
            pf[4 + r] = (__bf16)s[2 * kp + 1][r];
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            // lane 4q'+p of each 16-lane group addresses row q', columns 4p..4p+3 of its 4-key block
            const int r0 = (2 * kp) * 16 + 4 * g + (li >> 2);
            const int col = dt * 16 + 4 * (li & 3);
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sV + r0 * ROWB + col * 2));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sV + (r0 + 16) * ROWB + col * 2));
            const bf16x8 vf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[dt], 0, 0, 0);
        }
    }
    const float inv = 1.0f / sum;
    bf16_t* orow = out + ((size_t)b * T + q) * inner + h * DKV;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        const uint2 pk = make_uint2(pack_bf16x2(o[dt][0] * inv, o[dt][1] * inv), pack_bf16x2(o[dt][2] * inv, o[dt][3] * inv));
        *reinterpret_cast<uint2*>(orow + dt * 16 + 4 * g) = pk;
    }
    qf[0] = qn[0]; qf[1] = qn[1];
    }
}

template <int T>
int launch_t(const bf16_t* q, int ldq, const bf16_t* k, const bf16_t* v, int ldkv, const float* bias_off, bf16_t* out, int B,
             int H, hipStream_t stream) {
    const size_t lds = (size_t)2 * T * ROWB + (2 * T - 1) * sizeof(float) + 16;
    if (q == nullptr)       // attribute-only call from init_enc_attn_kernels()
        return hipFuncSetAttribute(reinterpret_cast<const void*>(enc_attn_kernel<T>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess ? 0 : -2;
    enc_attn_kernel<T><<<dim3(1, H, B), enc_attn_threads<T>(), lds, stream>>>(q, ldq, k, v, ldkv, bias_off, out, H);
    return 0;
}

// The same product structure for n_seq small sequences addressed through (outer, inner, step) strides (SeqAttnArgs): K/V of
// one (sequence, head) staged in LDS, every wave walks 16-query tiles.  Oracle: oracle/perceiver_oracle.py::_mha.
template <int TK> constexpr int seq_attn_threads() { return TK <= 64 ? 128 : 256; }

template <int TK>
__global__ __launch_bounds__(seq_attn_threads<TK>()) void seq_attn_kernel(SeqAttnArgs a) {
    constexpr int NT = TK / 16, NTH = seq_attn_threads<TK>(), NWV = NTH / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sK = smem;
    char* sV = smem + TK * ROWB;
    float* sB = reinterpret_cast<float*>(smem + 2 * TK * ROWB);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const long long so = s / a.inner_n, si = s % a.inner_n;
    const bf16_t* qp = a.q + so * a.q_outer + si * a.q_inner + h * DKV;
    const bf16_t* kp = a.k + so * a.kv_outer + si * a.kv_inner + h * DKV;
    const bf16_t* vp = a.v + so * a.kv_outer + si * a.kv_inner + h * DKV;
    bf16_t* op = a.out + so * a.o_outer + si * a.o_inner + h * DKV;
    {
        constexpr int NCH = (TK * 8 + NTH - 1) / NTH;
        uint4 kc[NCH], vc[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int idx = tid + i * NTH, row = (idx < TK * 8 ? idx : 0) >> 3, ch = idx & 7;
            kc[i] = *reinterpret_cast<const uint4*>(kp + (long long)row * a.kv_step + ch * 8);
            vc[i] = *reinterpret_cast<const uint4*>(vp + (long long)row * a.kv_step + ch * 8);
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int idx = tid + i * NTH, row = idx >> 3, ch = idx & 7;
            if (idx < TK * 8) {
                *reinterpret_cast<uint4*>(sK + row * ROWB + ch * 16) = kc[i];
                *reinterpret_cast<uint4*>(sV + row * ROWB + ch * 16) = vc[i];
            }
        }
    }
    for (int i = tid; i < 2 * TK - 1; i += NTH) sB[i] = a.bias_off ? a.bias_off[(size_t)h * (2 * TK - 1) + i] : 0.f;
    __syncthreads();
    const int g = lane >> 4, li = lane & 15;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    for (int qt = wave; qt < a.Tq / 16; qt += NWV) {          // wave-uniform
        const int q = qt * 16 + li;                           // this lane's query position
        bf16x8 qf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            qf[ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(qp + (long long)q * a.q_step + ks * 32 + g * 8));
        f32x4 sc[NT];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            sc[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 kf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sK + (kt * 16 + li) * ROWB + (ks * 4 + g) * 16));
                sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], sc[kt], 0, 0, 0);
            }
        }
        const int boff = a.bias_off ? TK - 1 - q : 0;         // without a bias every entry of sB is zero: index by key alone
        float mx = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sc[kt][r] += sB[kt * 16 + 4 * g + r + boff];
                mx = fmaxf(mx, sc[kt][r]);
            }
        mx = fmaxf(mx, lane_xor16(mx));
        mx = fmaxf(mx, lane_xor32(mx));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __expf(sc[kt][r] - mx);
                sc[kt][r] = e;
                sum += e;
            }
        sum += lane_xor16(sum);
        sum += lane_xor32(sum);
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kp2 = 0; kp2 < NT / 2; ++kp2) {
            bf16x8 pf;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pf[r] = (__bf16)sc[2 * kp2][r];
                pf[4 + r] = (__bf16)sc[2 * kp2 + 1][r];
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int r0 = (2 * kp2) * 16 + 4 * g + (li >> 2);
                const int col = dt * 16 + 4 * (li & 3);
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sV + r0 * ROWB + col * 2));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sV + (r0 + 16) * ROWB + col * 2));
                const bf16x8 vf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[dt], 0, 0, 0);
            }
        }
        const float inv = 1.0f / sum;
        bf16_t* orow = op + (long long)q * a.o_step;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const uint2 pk = make_uint2(pack_bf16x2(o[dt][0] * inv, o[dt][1] * inv), pack_bf16x2(o[dt][2] * inv, o[dt][3] * inv));
            *reinterpret_cast<uint2*>(orow + dt * 16 + 4 * g) = pk;
        }
    }
}

template <int TK>
int launch_seq_t(const SeqAttnArgs& a, hipStream_t stream) {
    const size_t lds = (size_t)2 * TK * ROWB + (2 * TK - 1) * sizeof(float) + 16;
    if (a.q == nullptr)       // attribute-only call from init_enc_attn_kernels()
        return hipFuncSetAttribute(reinterpret_cast<const void*>(seq_attn_kernel<TK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess ? 0 : -2;
    seq_attn_kernel<TK><<<a.n_seq * a.H, seq_attn_threads<TK>(), lds, stream>>>(a);
    return 0;
}

__global__ __launch_bounds__(256) void spec_embed_kernel(const float* __restrict__ mel, const float* __restrict__ w, const bf16_t* __restrict__ pos,
                                                         const float* __restrict__ gain, bf16_t* __restrict__ out, long long n_rows, int F, int d, float eps) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);       // one wave per spectral token
    const int lane = threadIdx.x & 63;
    if (row >= n_rows) return;
    const float m = mel[row];
    const bf16_t* pr = pos + (size_t)(row % F) * d;
    float xv[4];                                                                // d <= 256: up to 4 elements per lane
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = lane + 64 * j;
        xv[j] = c < d ? add_sep(mul_sep(m, w[c]), bf2f(pr[c])) : 0.f;   // mul, then add: the oracle's two roundings (no fma)
        ss += xv[j] * xv[j];
    }
    ss = wave_sum(ss);
    const float scl = rsqrtf(ss / (float)d + eps);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = lane + 64 * j;
        if (c < d) out[(size_t)row * d + c] = f2bf(xv[j] * scl * gain[c]);
    }
}

}  // namespace

int launch_seq_attention(const SeqAttnArgs& a, hipStream_t stream) {
    if (a.n_seq <= 0) return 0;
    if (a.Tq % 16 || a.Tq <= 0 || a.inner_n <= 0 || (a.bias_off && a.Tq != a.Tk)) return -1;
    if ((a.q_step | a.kv_step | a.o_step | a.q_outer | a.q_inner | a.kv_outer | a.kv_inner | a.o_outer | a.o_inner) % 8) return -1;   // 16-byte rows
    switch (a.Tk) {
        case 16: return -1;
        case 32: return launch_seq_t<32>(a, stream);
        case 64: return launch_seq_t<64>(a, stream);
        case 128: return launch_seq_t<128>(a, stream);
        case 256: return launch_seq_t<256>(a, stream);
        default: return -1;
    }
}

int launch_spec_embed(const float* mel, const float* w, const bf16_t* pos, const float* gain, bf16_t* out, long long n_rows, int F, int d,
                      float eps, hipStream_t stream) {
    if (n_rows <= 0) return 0;
    if (d > 256 || d <= 0 || F <= 0) return -1;
    spec_embed_kernel<<<(unsigned)((n_rows + 3) / 4), 256, 0, stream>>>(mel, w, pos, gain, out, n_rows, F, d, eps);
    return 0;
}

int init_enc_attn_kernels() {
    SeqAttnArgs z{};
    if (launch_seq_t<32>(z, nullptr) | launch_seq_t<64>(z, nullptr) | launch_seq_t<128>(z, nullptr) | launch_seq_t<256>(z, nullptr)) return -2;
    return launch_t<64>(nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr, 1, 1, nullptr) |
           launch_t<128>(nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr, 1, 1, nullptr) |
           launch_t<256>(nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr, 1, 1, nullptr) |
           launch_t<512>(nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr, 1, 1, nullptr);
}

int launch_enc_attention_qkv(const bf16_t* q, int ldq, const bf16_t* k, const bf16_t* v, int ldkv, const float* bias_off,
                             bf16_t* out, int B, int T, int H, hipStream_t stream) {
    if (B <= 0) return 0;
    if ((ldq % 8) || (ldkv % 8)) return -1;
    switch (T) {
        case 64: return launch_t<64>(q, ldq, k, v, ldkv, bias_off, out, B, H, stream);
        case 128: return launch_t<128>(q, ldq, k, v, ldkv, bias_off, out, B, H, stream);
        case 256: return launch_t<256>(q, ldq, k, v, ldkv, bias_off, out, B, H, stream);
        case 512: return launch_t<512>(q, ldq, k, v, ldkv, bias_off, out, B, H, stream);
        default: return -1;
    }
}

int launch_enc_attention(const bf16_t* qkv, const float* bias_off, bf16_t* out, int B, int T, int H, hipStream_t stream) {
    const int inner = H * DKV;
    return launch_enc_attention_qkv(qkv, 3 * inner, qkv + inner, qkv + 2 * inner, 3 * inner, bias_off, out, B, T, H, stream);
}
