// Autoregressive decoder step kernels (a7, a8, a10 of SURVEY.md section 8).
//
// Every kernel reads the decode position from DecodeShared in device memory, so ONE captured
// hipGraph of a step can be replayed for every position with no host round trip (SURVEY section 3.4:
// the per-step `.max()==0` host sync of TP: generation/utils.py:2936-2937 becomes device state).
//
//  dec_gemm_kernel      skinny GEMM, R = segments x channels rows (64..832) against a [N][K] bf16
//                       weight that is streamed exactly once per 64-row tile: one workgroup per 16
//                       output columns, the four waves split K, weight fragments go straight from
//                       global memory to MFMA operand registers (no LDS: each element is used by one
//                       wave), partial tiles are summed through LDS in a fixed order (bitwise
//                       reproducible; no float atomics).  NORM modes fuse the T5 RMS norm
//                       (TP: modeling_t5.py:50-72) of the fp32 residual rows as the prologue; the
//                       epilogues fuse KV-cache append (TP: cache_utils.py:144-145), ReLU, residual add.
//  dec_attn_kernel      one (row, head) per workgroup; K/V slabs streamed HBM -> registers with 16-byte
//                       coalesced loads, 8 in flight per lane; online softmax per lane group, merged by
//                       shuffles and one LDS pass.  Self-attention adds the unidirectional relative
//                       position bias by distance (TP: modeling_t5.py:264-279); cross-attention has none.
//  argmax_embed_kernel  fp32 argmax (first index wins ties, TP: utils.py:2925), EOS -> PAD fill
//                       (:2928-2929), token store, next-token embedding gather into the residual
//                       stream, and the step counter advance by the last workgroup to finish.
//
// Oracle: oracle/ymt3_oracle.py::decoder_step / greedy_decode.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int DKV = 64;

// ------------------------------------------------------------------------------------------------
template <int MODE, int K>
__global__ __launch_bounds__(256) void dec_gemm_kernel(DecGemmArgs a) {
    constexpr bool NORM = (MODE != DG_RESID);
    constexpr int KW = K / 4;            // K slice per wave
    constexpr int KS = KW / 32;          // MFMA k-steps per wave
    constexpr int PITCH = K + 8;         // bf16 elements per LDS row (16-byte pad)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* sA = reinterpret_cast<bf16_t*>(smem);
    float* red = reinterpret_cast<float*>(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 64;
    const int kb = wave * KW;

    f32x4 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const bf16_t* wrow = a.W + (size_t)(n0 + li) * K + kb + g * 8;

    if constexpr (NORM) {
        // weight fragments first: their latency hides under the norm prologue
        bf16x8 wf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wf[ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(wrow + ks * 32));

        // RMS norm of rows m0 + 16*wave .. +15 into the bf16 LDS tile
        constexpr int NV = K / 256;      // float4 per lane per row
#pragma unroll 4
        for (int rr = 0; rr < 16; ++rr) {
            const int row = wave * 16 + rr, m = m0 + row;
            float4 v[NV];
            float ss = 0.f;
            if (m < a.R) {
                const float4* xr = reinterpret_cast<const float4*>(a.x_f32 + (size_t)m * K);
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    v[i] = xr[lane + 64 * i];
                    ss += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
                }
            } else {
#pragma unroll
                for (int i = 0; i < NV; ++i) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            ss = wave_sum(ss);
            const float sc = rsqrtf(ss / (float)K + a.eps);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const float4 gn = reinterpret_cast<const float4*>(a.gain)[lane + 64 * i];
                *reinterpret_cast<uint2*>(sA + row * PITCH + (lane + 64 * i) * 4) =
                    make_uint2(pack_bf16x2(v[i].x * sc * gn.x, v[i].y * sc * gn.y),
                               pack_bf16x2(v[i].z * sc * gn.z, v[i].w * sc * gn.w));
            }
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const bf16x8 af = __builtin_bit_cast(
                    bf16x8, *reinterpret_cast<const uint4*>(sA + (mt * 16 + li) * PITCH + kb + ks * 32 + g * 8));
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks], af, acc[mt], 0, 0, 0);
            }
        __syncthreads();                 // sA is dead; `red` aliases it
    } else {
        // A is bf16 in global memory and each element feeds exactly one wave: no LDS staging
        const bf16_t* arow[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            int m = m0 + mt * 16 + li;
            m = m < a.R ? m : a.R - 1;
            arow[mt] = a.a_bf16 + (size_t)m * K + kb + g * 8;
        }
        constexpr int U = KS < 4 ? KS : 4;
#pragma unroll 1
        for (int k0 = 0; k0 < KS; k0 += U) {
            bf16x8 wf[U], af[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                wf[u] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(wrow + (k0 + u) * 32));
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    af[u][mt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(arow[mt] + (k0 + u) * 32));
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u], af[u][mt], acc[mt], 0, 0, 0);
        }
    }

    // fixed-order cross-wave reduction: red[wave][m (64)][n (16)]
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
        *reinterpret_cast<float4*>(red + ((wave * 64 + mt * 16 + li) * 16 + g * 4)) =
            make_float4(acc[mt][0], acc[mt][1], acc[mt][2], acc[mt][3]);
    __syncthreads();
    const int mr = tid >> 2, nq = (tid & 3) * 4;
    float4 s = *reinterpret_cast<const float4*>(red + (mr * 16 + nq));
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        const float4 t = *reinterpret_cast<const float4*>(red + ((w * 64 + mr) * 16 + nq));
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    const int m = m0 + mr, n = n0 + nq;
    if (m >= a.R) return;

    if constexpr (MODE == DG_RESID) {
        float4* p = reinterpret_cast<float4*>(a.out_f32 + (size_t)m * a.N + n);
        float4 o = *p;
        o.x += s.x; o.y += s.y; o.z += s.z; o.w += s.w;
        *p = o;
    } else if constexpr (MODE == DG_NORM_LOGITS) {
        *reinterpret_cast<float4*>(a.out_f32 + (size_t)m * a.N + n) = s;
    } else {
        if constexpr (MODE == DG_NORM_BF16_RELU) {
            s.x = fmaxf(s.x, 0.f); s.y = fmaxf(s.y, 0.f); s.z = fmaxf(s.z, 0.f); s.w = fmaxf(s.w, 0.f);
        }
        const uint2 pk = make_uint2(pack_bf16x2(s.x, s.y), pack_bf16x2(s.z, s.w));
        if constexpr (MODE == DG_NORM_QKV_CACHE) {
            const int inner = a.H * DKV;
            if (n < inner) {
                *reinterpret_cast<uint2*>(a.out_bf16 + (size_t)m * inner + n) = pk;
            } else {
                const int step = a.shared->step;
                const int nn = n - inner, kv = nn / inner, hh = (nn % inner) >> 6, dd = nn & 63;
                bf16_t* cache = kv ? a.vcache : a.kcache;
                *reinterpret_cast<uint2*>(cache + (((size_t)m * a.H + hh) * a.L + step) * DKV + dd) = pk;
            }
        } else {
            *reinterpret_cast<uint2*>(a.out_bf16 + (size_t)m * a.N + n) = pk;
        }
    }
}

// ------------------------------------------------------------------------------------------------
template <bool SELF>
__global__ __launch_bounds__(256) void dec_attn_kernel(DecAttnArgs a) {
    __shared__ float sm[4], sl[4], sacc[4][DKV];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = lane & 7, kg = lane >> 3;
    const int r = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const int n_keys = SELF ? a.shared->step + 1 : a.n_keys_const;
    const int kv_row = r / a.rows_per_kv;
    const size_t slab = ((size_t)kv_row * a.H + h) * a.slab_keys * DKV;
    const bf16_t* kb = a.k + slab + sub * 8;
    const bf16_t* vb = a.v + slab + sub * 8;
    const float* bias = SELF ? a.bias + (size_t)h * a.bias_stride : nullptr;

    float qf[8];
    unpack8(*reinterpret_cast<const uint4*>(a.q + ((size_t)r * a.H + h) * DKV + sub * 8), qf);

    float m = -1.0e30f, l = 0.f, acc[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) acc[d] = 0.f;

    constexpr int U = 4;
    for (int kw = wave * 8; kw < n_keys; kw += 32 * U) {     // wave-uniform trip count
        const int k0 = kw + kg;
        uint4 ku[U], vu[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int key = k0 + 32 * u;
            ok[u] = key < n_keys;
            const int kc = ok[u] ? key : 0;
            ku[u] = *reinterpret_cast<const uint4*>(kb + (size_t)kc * DKV);
            vu[u] = *reinterpret_cast<const uint4*>(vb + (size_t)kc * DKV);
        }
        float sc[U];
        float mn = m;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float kf[8];
            unpack8(ku[u], kf);
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < 8; ++d) s = fmaf(qf[d], kf[d], s);
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            s += __shfl_xor(s, 4, 64);
            if (SELF && ok[u]) s += bias[n_keys - 1 - (k0 + 32 * u)];
            sc[u] = s;
            if (ok[u]) mn = fmaxf(mn, s);
        }
        const float rescale = __expf(m - mn);
        l *= rescale;
#pragma unroll
        for (int d = 0; d < 8; ++d) acc[d] *= rescale;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float p = ok[u] ? __expf(sc[u] - mn) : 0.f;
            float vf[8];
            unpack8(vu[u], vf);
            l += p;
#pragma unroll
            for (int d = 0; d < 8; ++d) acc[d] = fmaf(p, vf[d], acc[d]);
        }
        m = mn;
    }
    // merge the 8 key groups of the wave (lanes with equal `sub`)
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) {
        const float mo = __shfl_xor(m, off, 64), lo = __shfl_xor(l, off, 64);
        const float mn = fmaxf(m, mo);
        const float sa = __expf(m - mn), sb = __expf(mo - mn);
        l = l * sa + lo * sb;
#pragma unroll
        for (int d = 0; d < 8; ++d) acc[d] = acc[d] * sa + __shfl_xor(acc[d], off, 64) * sb;
        m = mn;
    }
    if (kg == 0) {
#pragma unroll
        for (int d = 0; d < 8; ++d) sacc[wave][sub * 8 + d] = acc[d];
        if (sub == 0) { sm[wave] = m; sl[wave] = l; }
    }
    __syncthreads();
    if (tid < DKV) {
        const float M = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
        float L = 0.f, o = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float e = __expf(sm[w] - M);
            L += sl[w] * e;
            o += sacc[w][tid] * e;
        }
        a.out[((size_t)r * a.H + h) * DKV + tid] = f2bf(o / L);
    }
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void argmax_embed_kernel(ArgmaxArgs a) {
    __shared__ float sv[4];
    __shared__ int si[4];
    __shared__ int s_feed;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = blockIdx.x;
    DecodeShared* sh = a.shared;
    const int t = sh->step, n_steps = sh->n_steps;
    const float* row = a.logits + (size_t)r * a.V;

    float bv = -3.4e38f;
    int bi = 0x7fffffff;
    for (int i = tid; i < a.V; i += 256) {
        const float v = row[i];
        if (v > bv) { bv = v; bi = i; }       // ascending i: the first maximum is kept
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(bv, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
        int tok = bi;
        if (a.eos_id >= 0) {
            if (a.finished[r]) tok = a.pad_id;
            else if (tok == a.eos_id) a.finished[r] = 1;
        }
        sh->tokens_out[(size_t)r * n_steps + t] = tok;
        s_feed = sh->forced ? sh->forced[(size_t)r * n_steps + t] : tok;
    }
    __syncthreads();
    const int feed = s_feed;
    const bf16_t* e = a.embed + (size_t)feed * a.d;
    const bf16_t* c = a.chan_embed ? a.chan_embed + (size_t)(r % a.n_channels) * a.d : nullptr;
    for (int i = tid; i < a.d; i += 256) a.h[(size_t)r * a.d + i] = bf2f(e[i]) + (c ? bf2f(c[i]) : 0.f);
    if (sh->logits_out) {
        float* dst = sh->logits_out + ((size_t)r * n_steps + t) * a.V;
        for (int i = tid; i < a.V; i += 256) dst[i] = row[i];
    }
    // the last workgroup to finish advances the position; every workgroup has read `t` by then
    __syncthreads();
    if (tid == 0) {
        __threadfence();
        const int old = atomicAdd(&sh->done_count, 1);
        if (old == (int)gridDim.x - 1) {
            sh->done_count = 0;
            sh->step = t + 1;
        }
    }
}

__global__ __launch_bounds__(256) void decode_init_kernel(ArgmaxArgs a, int n_steps, int32_t* tokens_out,
                                                          const int32_t* forced, float* logits_out) {
    const int r = blockIdx.x, tid = threadIdx.x;
    const bf16_t* e = a.embed + (size_t)a.pad_id * a.d;
    const bf16_t* c = a.chan_embed ? a.chan_embed + (size_t)(r % a.n_channels) * a.d : nullptr;
    for (int i = tid; i < a.d; i += 256) a.h[(size_t)r * a.d + i] = bf2f(e[i]) + (c ? bf2f(c[i]) : 0.f);
    if (tid == 0) a.finished[r] = 0;
    if (r == 0 && tid == 0) {
        a.shared->step = 0;
        a.shared->done_count = 0;
        a.shared->n_steps = n_steps;
        a.shared->tokens_out = tokens_out;
        a.shared->forced = forced;
        a.shared->logits_out = logits_out;
    }
}

template <int MODE, int K>
int launch_dg(const DecGemmArgs& a, hipStream_t stream) {
    constexpr size_t lds_norm = (size_t)64 * (K + 8) * 2;
    constexpr size_t lds_red = 4 * 64 * 16 * 4;
    constexpr size_t lds = (MODE != DG_RESID) ? (lds_norm > lds_red ? lds_norm : lds_red) : lds_red;
    if (a.W == nullptr) {   // attribute-only call from init_decode_kernels()
        return hipFuncSetAttribute(reinterpret_cast<const void*>(dec_gemm_kernel<MODE, K>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess ? 0 : -2;
    }
    dec_gemm_kernel<MODE, K><<<dim3(a.N / 16, (a.R + 63) / 64), 256, lds, stream>>>(a);
    return 0;
}

}  // namespace

int init_decode_kernels() {
    DecGemmArgs z{};
    int rc = 0;
    rc |= launch_dg<DG_RESID, 512>(z, nullptr);
    rc |= launch_dg<DG_RESID, 1024>(z, nullptr);
    rc |= launch_dg<DG_RESID, 2048>(z, nullptr);
    rc |= launch_dg<DG_NORM_QKV_CACHE, 512>(z, nullptr);
    rc |= launch_dg<DG_NORM_BF16, 512>(z, nullptr);
    rc |= launch_dg<DG_NORM_BF16_RELU, 512>(z, nullptr);
    rc |= launch_dg<DG_NORM_LOGITS, 512>(z, nullptr);
    return rc;
}

int launch_dec_gemm(int mode, const DecGemmArgs& a, hipStream_t stream) {
    if (a.R <= 0) return 0;
    if (a.N % 16) return -1;
    if (mode == DG_RESID) {
        if (a.K == 512) return launch_dg<DG_RESID, 512>(a, stream);
        if (a.K == 2048) return launch_dg<DG_RESID, 2048>(a, stream);
        if (a.K == 1024) return launch_dg<DG_RESID, 1024>(a, stream);
        return -1;
    }
    if (a.K != 512) return -1;
    switch (mode) {
        case DG_NORM_QKV_CACHE: return launch_dg<DG_NORM_QKV_CACHE, 512>(a, stream);
        case DG_NORM_BF16: return launch_dg<DG_NORM_BF16, 512>(a, stream);
        case DG_NORM_BF16_RELU: return launch_dg<DG_NORM_BF16_RELU, 512>(a, stream);
        case DG_NORM_LOGITS: return launch_dg<DG_NORM_LOGITS, 512>(a, stream);
        default: return -1;
    }
}

int launch_dec_attention(bool self_attn, const DecAttnArgs& a, hipStream_t stream) {
    if (a.R <= 0) return 0;
    if (self_attn) dec_attn_kernel<true><<<a.R * a.H, 256, 0, stream>>>(a);
    else dec_attn_kernel<false><<<a.R * a.H, 256, 0, stream>>>(a);
    return 0;
}

int launch_argmax_embed(const ArgmaxArgs& a, hipStream_t stream) {
    if (a.R <= 0) return 0;
    argmax_embed_kernel<<<a.R, 256, 0, stream>>>(a);
    return 0;
}

int launch_decode_init(const ArgmaxArgs& a, int n_steps, int32_t* tokens_out, const int32_t* forced, float* logits_out,
                       hipStream_t stream) {
    if (a.R <= 0) return 0;
    decode_init_kernel<<<a.R, 256, 0, stream>>>(a, n_steps, tokens_out, forced, logits_out);
    return 0;
}
