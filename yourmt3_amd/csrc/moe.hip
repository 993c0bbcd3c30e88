// a11 of SURVEY.md section 8: mixture-of-experts decoder FFN (E experts, top-k = 2), decode-time form.
// The reference tree has no implementation and the container no oracle (SURVEY section 2.2 star-9): the
// arithmetic below is this build's own spec, restated on the CPU in oracle/ymt3_oracle.py::moe_ffn
// ("parity unpinned" with respect to the reference):
//     xn      = R( rmsnorm(h) * gain )                         bf16, shared by router and experts
//     logits  = xn . router^T                                  fp32, E values per row
//     (e0,e1) = the two largest logits, ties to the lower expert id; gates = softmax(l[e0], l[e1])
//     y_j     = R( relu(xn . wi[e_j]^T) ) . wo[e_j]^T           dense ReLU FFN of expert e_j
//     h      += g0*y_0 + g1*y_1                                 fp32, slot order fixed
//
// Three kernels (four launches per layer), no float atomics, every reduction in a fixed order (bitwise reproducible):
//   moe_router_kernel   one wave per row: norm from the carried sum(h^2) partials, router dots, top-2, gates
//   moe_gemm_kernel     grouped skinny GEMM, one workgroup per (expert, 16-column tile): it finds its expert's (row, slot)
//                       pairs itself -- a ballot scan of the 2R selections, pairs in ascending order, chunks of <= 16, exactly
//                       the work items a counting sort by expert would make (round 1 spent a launch on that sort) -- and walks
//                       the chunks with its weight tile loaded ONCE: same coalesced-load / wave-private-LDS-strip / 8-way
//                       split-K structure as dec_gemm_kernel, activation rows gathered by pair
//   moe_combine_kernel  one wave per row: h += y[2r] + y[2r+1]; sum(h^2) for the next norm
// Pair p = 2 * (row - row0) + slot indexes `hidden`, `y` and the gates directly: no sorted order is ever materialised.
// Expert weights are replicated on every GPU (8 x 2 x 2 MB per layer): the path stays pure data-parallel,
// no all-to-all.  bf16 MFMA by default; moe_fp8 = 1 selects the OCP-e4m3 MFMA form of the expert
// GEMMs that BASELINE configs[4] names (moe_gemm_fp8_kernel below).
#include "common.h"
#include "kernels.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int E_MAX = 16;

__device__ __forceinline__ float moe_dpp_sum8(float v) {      // sum over 8 consecutive lanes, DPP only
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    return v;
}

// one wave per row.  The normed row goes through a wave-private LDS strip so that lane group e (8 lanes) can take the
// WHOLE dot product of expert e: all E <= 8 router logits come out of one 3-step DPP reduction instead of E wave-wide
// ones (E > 8 falls back to a second pass over experts 8..15).
__global__ __launch_bounds__(512) void moe_router_kernel(const float* __restrict__ pH, const float* __restrict__ pGain, const float* __restrict__ pSsq,
                                                         const bf16_t* __restrict__ pRouter, bf16_t* __restrict__ pXn, int row0, int R, int ssq_stride, int E,
                                                         MoeArgs a) {      // leading scalars: kernarg preload (decode.hip, dec_gemm_kernel)
    __shared__ __attribute__((aligned(16))) float xrow[8][512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = row0 + blockIdx.x * 8 + wave;
    if (r >= row0 + R) return;
    constexpr int D = 512;
    float ss = 0.f;
    if (lane < SSQ_TILES) ss = pSsq[(size_t)lane * ssq_stride + r];
    const float4* x = reinterpret_cast<const float4*>(pH + (size_t)r * D);
    const float4* g = reinterpret_cast<const float4*>(pGain);
    const float4 v0 = x[lane], v1 = x[lane + 64], g0 = g[lane], g1 = g[lane + 64];
    // router rows for this lane's expert: lane group `grp` = expert, `sub` = which eighth of the row
    const int grp = lane >> 3, sub = lane & 7;
    uint4 wv[8];
    const bool has_e = grp < E;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        wv[i] = has_e ? *reinterpret_cast<const uint4*>(pRouter + (size_t)grp * D + sub * 64 + i * 8) : make_uint4(0, 0, 0, 0);
    ss = wave_sum(ss);                                          // fixed-order tree over the 32 partials
    const float sc = rsqrtf(ss / (float)D + a.eps);
    auto norm4 = [&](const float4& v, const float4& gg, int idx) {
        const bf16_t b0 = f2bf(v.x * sc * gg.x), b1 = f2bf(v.y * sc * gg.y), b2 = f2bf(v.z * sc * gg.z), b3 = f2bf(v.w * sc * gg.w);
        *reinterpret_cast<uint2*>(pXn + (size_t)r * D + idx * 4) =
            make_uint2((uint32_t)b0 | ((uint32_t)b1 << 16), (uint32_t)b2 | ((uint32_t)b3 << 16));
        *reinterpret_cast<float4*>(&xrow[wave][idx * 4]) = make_float4(bf2f(b0), bf2f(b1), bf2f(b2), bf2f(b3));
    };
    norm4(v0, g0, lane);
    norm4(v1, g1, lane + 64);
    // wave-private strip: LDS executes a wave's accesses in order, no barrier needed
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float4 xa = *reinterpret_cast<const float4*>(&xrow[wave][sub * 64 + i * 8]);
        const float4 xb = *reinterpret_cast<const float4*>(&xrow[wave][sub * 64 + i * 8 + 4]);
        s = fmaf(xa.x, __uint_as_float(wv[i].x << 16), s); s = fmaf(xa.y, __uint_as_float(wv[i].x & 0xffff0000u), s);
        s = fmaf(xa.z, __uint_as_float(wv[i].y << 16), s); s = fmaf(xa.w, __uint_as_float(wv[i].y & 0xffff0000u), s);
        s = fmaf(xb.x, __uint_as_float(wv[i].z << 16), s); s = fmaf(xb.y, __uint_as_float(wv[i].z & 0xffff0000u), s);
        s = fmaf(xb.z, __uint_as_float(wv[i].w << 16), s); s = fmaf(xb.w, __uint_as_float(wv[i].w & 0xffff0000u), s);
    }
    s = moe_dpp_sum8(s);                                        // every lane of group e now holds logit[e]
    float best = -3.4e38f, second = -3.4e38f;
    int e0 = 0, e1 = 0;
    const int ne = E < 8 ? E : 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        // logit[e] from lane 8e by v_readlane (a scalar broadcast), not eight dependent trips through the LDS crossbar
        const float le = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), e * 8));
        if (e < ne) {
            if (le > best) { second = best; e1 = e0; best = le; e0 = e; }
            else if (le > second) { second = le; e1 = e; }
        }
    }
    for (int e = 8; e < E; ++e) {                             // experts 8..15: plain wave-wide dots
        const bf16_t* w = pRouter + (size_t)e * D;
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint2 w2 = *reinterpret_cast<const uint2*>(w + (lane + 64 * i) * 4);
            const float4 xx = *reinterpret_cast<const float4*>(&xrow[wave][(lane + 64 * i) * 4]);
            t = fmaf(xx.x, __uint_as_float(w2.x << 16), t); t = fmaf(xx.y, __uint_as_float(w2.x & 0xffff0000u), t);
            t = fmaf(xx.z, __uint_as_float(w2.y << 16), t); t = fmaf(xx.w, __uint_as_float(w2.y & 0xffff0000u), t);
        }
        t = wave_sum(t);
        if (t > best) { second = best; e1 = e0; best = t; e0 = e; }
        else if (t > second) { second = t; e1 = e; }
    }
    if (lane == 0) {
        const float t = __expf(second - best);
        a.sel[2 * r] = e0; a.sel[2 * r + 1] = e1;
        a.gate[2 * r] = 1.0f / (1.0f + t); a.gate[2 * r + 1] = t / (1.0f + t);
        if (a.sel_trace) {                                      // debug hook only
            const int st = a.shared->step - a.shared->step0;
            if (st >= 0 && st < a.trace_steps && r < a.trace_rows) {
                int32_t* dst = a.sel_trace + (((size_t)st * a.n_layers + a.layer) * a.trace_rows + r) * 2;
                dst[0] = e0; dst[1] = e1;
            }
        }
    }
}

// The pairs of expert `e` among the P = 2R selections, ascending, into plist[]; returns their count (workgroup-uniform).
// 512 threads, one pair per thread and pass; ballots give the rank inside a wave, an 8-entry LDS table the rest.
__device__ __forceinline__ int moe_find_pairs(const int* __restrict__ sel, int P, int e, int* plist, int* wcnt) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int total = 0;
    for (int base = 0; base < P; base += 512) {
        const int p = base + tid;
        const bool mine = p < P && sel[p] == e;
        const unsigned long long m = __ballot(mine);
        if (lane == 0) wcnt[wave] = __popcll(m);
        __syncthreads();
        int before = total;
        for (int w = 0; w < wave; ++w) before += wcnt[w];
        if (mine) plist[before + __popcll(m & ((1ull << lane) - 1ull))] = p;
        int all = 0;
#pragma unroll
        for (int w = 0; w < 8; ++w) all += wcnt[w];
        total += all;
        __syncthreads();
    }
    return total;
}

// STAGE 0: hidden[p] = R(relu(xn[row(p)] . wi[e]^T))      (K = d_model, N = d_ff)
// STAGE 1: y[p]      = gate[p] * (hidden[p] . wo[e]^T)     (K = d_ff,    N = d_model)
template <int STAGE, int K>
__global__ __launch_bounds__(512) void moe_gemm_kernel(const void* __restrict__ pW, const bf16_t* __restrict__ pA, const int* __restrict__ pSel,
                                                       const float* __restrict__ pGate, int row0, int R, MoeArgs a) {
    // leading scalars (kernarg preload): pW = this stage's expert weights, pA = its activation rows (xn or hidden), the router's selections / gates
    constexpr int KW = K / 8, KS = KW / 32, PITCH = KW * 2 + 16, STRIP = 16 * PITCH;
    constexpr int LPR = KW * 2 / 16, RPI = 64 / LPR, NI = 16 / RPI;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);               // [8][16][16]
    int* wcnt = reinterpret_cast<int*>(red + 8 * 16 * 16);     // [8]
    char* strips = reinterpret_cast<char*>(wcnt + 8);          // [8 waves][A strip | W strip]
    int* plist = reinterpret_cast<int*>(strips + 8 * 2 * STRIP);   // [2R] pairs of this expert, ascending

    const int N = STAGE == 0 ? a.d_ff : a.d_model, n_nt = N / 16;
    const int e = blockIdx.x / n_nt, nt = blockIdx.x % n_nt;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
    const int n0 = nt * 16;
    const bf16_t* W = static_cast<const bf16_t*>(pW) + (size_t)e * N * K;

    char* sA = strips + wave * 2 * STRIP;
    char* sW = sA + STRIP;
    u32x4 wv[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {                              // the weight tile does not depend on the routing: in flight under the scan
        const int row = i * RPI + lane / LPR, ch = lane % LPR;
        wv[i] = *reinterpret_cast<const u32x4*>(W + (size_t)(n0 + row) * K + wave * KW + ch * 8);
    }
    const int cnt_all = moe_find_pairs(pSel + 2 * row0, 2 * R, e, plist, wcnt);
    if (cnt_all == 0) return;
#pragma unroll
    for (int i = 0; i < NI; ++i) *reinterpret_cast<u32x4*>(sW + (i * RPI + lane / LPR) * PITCH + (lane % LPR) * 16) = wv[i];
    for (int c0 = 0; c0 < cnt_all; c0 += 16) {
        const int cnt = min(16, cnt_all - c0);
        u32x4 av[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int row = i * RPI + lane / LPR, ch = lane % LPR;
            const int pp = plist[c0 + (row < cnt ? row : cnt - 1)];
            const bf16_t* arow = STAGE == 0 ? pA + (size_t)(row0 + (pp >> 1)) * K : pA + (size_t)(2 * row0 + pp) * K;
            av[i] = *reinterpret_cast<const u32x4*>(arow + wave * KW + ch * 8);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) *reinterpret_cast<u32x4*>(sA + (i * RPI + lane / LPR) * PITCH + (lane % LPR) * 16) = av[i];
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int o = li * PITCH + (ks * 32 + g * 8) * 2;
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(sW + o),
                                                          *reinterpret_cast<const bf16x8*>(sA + o), acc, 0, 0, 0);
        }
        *reinterpret_cast<float4*>(red + ((wave * 16 + li) * 16 + g * 4)) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        __syncthreads();
        if (tid < 128) {
            const int mr = tid >> 3, nq = (tid & 7) * 2;
            float2 s = *reinterpret_cast<const float2*>(red + (mr * 16 + nq));
#pragma unroll
            for (int w = 1; w < 8; ++w) {
                const float2 t = *reinterpret_cast<const float2*>(red + ((w * 16 + mr) * 16 + nq));
                s.x += t.x; s.y += t.y;
            }
            if (mr < cnt) {
                const int pp = 2 * row0 + plist[c0 + mr];
                if constexpr (STAGE == 0) {
                    *reinterpret_cast<uint32_t*>(a.hidden + (size_t)pp * a.d_ff + n0 + nq) = pack_bf16x2(fmaxf(s.x, 0.f), fmaxf(s.y, 0.f));
                } else {
                    const float gt = pGate[pp];
                    *reinterpret_cast<float2*>(a.y + (size_t)pp * a.d_model + n0 + nq) = make_float2(gt * s.x, gt * s.y);
                }
            }
        }
        __syncthreads();                                       // `red` and the activation strips are reused by the next chunk
    }
}

// fp8 (OCP e4m3) form of the grouped expert GEMM (BASELINE configs[4]): weights arrive pre-quantised with one fp32
// scale per (expert, matrix); the 16 activation rows of the work item are quantised here, per row, with a dynamic scale
// (amax / 448) -- every workgroup sees its rows over the full K, so the row maximum is an in-workgroup reduction
// (per-wave partial maxima -> LDS -> 8-way max).  v_mfma_f32_16x16x32_fp8_fp8 accumulates in fp32; the epilogue
// multiplies by row_scale * weight_scale.  Oracle: oracle/ymt3_oracle.py::moe_ffn (moe_fp8 branch).
template <int STAGE, int K>
__global__ __launch_bounds__(512) void moe_gemm_fp8_kernel(const void* __restrict__ pW, const bf16_t* __restrict__ pA, const int* __restrict__ pSel,
                                                           const float* __restrict__ pGate, int row0, int R, MoeArgs a) {
    // leading scalars (kernarg preload): as moe_gemm_kernel; pW = e4m3 weights
    constexpr int KW = K / 8, KS = KW / 32, PITCH = KW + 16, STRIP = 16 * PITCH;
    constexpr int LPR = KW * 2 / 16, RPI = 64 / LPR, NI = 16 / RPI;          // bf16 activation rows
    constexpr int LPRW = KW / 16, RPIW = 64 / LPRW, NIW = 16 / RPIW;         // fp8 weight rows
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);               // [8][16][16]
    int* wcnt = reinterpret_cast<int*>(red + 8 * 16 * 16);     // [8]
    unsigned* smax = reinterpret_cast<unsigned*>(wcnt + 8);    // [16] row maxima of |x| as float bits
    float* sinv = reinterpret_cast<float*>(smax + 16);         // [16] 448 / amax
    float* sxs = sinv + 16;                                    // [16] amax / 448
    char* strips = reinterpret_cast<char*>(sxs + 16);
    int* plist = reinterpret_cast<int*>(strips + 8 * 2 * STRIP);

    const int N = STAGE == 0 ? a.d_ff : a.d_model, n_nt = N / 16;
    const int e = blockIdx.x / n_nt, nt = blockIdx.x % n_nt;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
    const int n0 = nt * 16;
    const uint8_t* W = static_cast<const uint8_t*>(pW) + (size_t)e * N * K;

    char* sA = strips + wave * 2 * STRIP;
    char* sW = sA + STRIP;
    u32x4 wv[NIW];
#pragma unroll
    for (int i = 0; i < NIW; ++i) {
        const int row = i * RPIW + lane / LPRW, ch = lane % LPRW;
        wv[i] = *reinterpret_cast<const u32x4*>(W + (size_t)(n0 + row) * K + wave * KW + ch * 16);
    }
    const int cnt_all = moe_find_pairs(pSel + 2 * row0, 2 * R, e, plist, wcnt);
    if (cnt_all == 0) return;
    const float wscale = (STAGE == 0 ? a.wi_s : a.wo_s)[e];
#pragma unroll
    for (int i = 0; i < NIW; ++i)
        *reinterpret_cast<u32x4*>(sW + (i * RPIW + lane / LPRW) * PITCH + (lane % LPRW) * 16) = wv[i];
    for (int c0 = 0; c0 < cnt_all; c0 += 16) {
        const int cnt = min(16, cnt_all - c0);
        if (tid < 16) smax[tid] = 0u;
        __syncthreads();
        u32x4 av[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int row = i * RPI + lane / LPR, ch = lane % LPR;
            const int pp = plist[c0 + (row < cnt ? row : cnt - 1)];
            const bf16_t* arow = STAGE == 0 ? pA + (size_t)(row0 + (pp >> 1)) * K : pA + (size_t)(2 * row0 + pp) * K;
            av[i] = *reinterpret_cast<const u32x4*>(arow + wave * KW + ch * 8);
        }
        // row maxima of |x|: 8-lane DPP max, then one LDS integer max per (row, 8-lane group) -- |x| >= 0, so the float
        // bit patterns order like unsigned integers
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            float mx = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                mx = fmaxf(mx, fabsf(__uint_as_float(av[i][j] << 16)));
                mx = fmaxf(mx, fabsf(__uint_as_float(av[i][j] & 0xffff0000u)));
            }
            mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mx), 0xB1, 0xF, 0xF, true)));
            mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mx), 0x4E, 0xF, 0xF, true)));
            mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mx), 0x141, 0xF, 0xF, true)));
            if ((lane & 7) == 0) atomicMax(&smax[i * RPI + lane / LPR], __float_as_uint(mx));
        }
        __syncthreads();
        if (tid < 16) {
            const float mx = fmaxf(__uint_as_float(smax[tid]), 1e-12f);
            sinv[tid] = 448.0f / mx;
            sxs[tid] = mx / 448.0f;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int row = i * RPI + lane / LPR;
            const float inv = sinv[row];
            int lo = 0, hi = 0;
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(__uint_as_float(av[i][0] << 16) * inv, __uint_as_float(av[i][0] & 0xffff0000u) * inv, lo, false);
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(__uint_as_float(av[i][1] << 16) * inv, __uint_as_float(av[i][1] & 0xffff0000u) * inv, lo, true);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(__uint_as_float(av[i][2] << 16) * inv, __uint_as_float(av[i][2] & 0xffff0000u) * inv, hi, false);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(__uint_as_float(av[i][3] << 16) * inv, __uint_as_float(av[i][3] & 0xffff0000u) * inv, hi, true);
            *reinterpret_cast<int2*>(sA + row * PITCH + (lane % LPR) * 8) = make_int2(lo, hi);
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int o = li * PITCH + ks * 32 + g * 8;
            acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(*reinterpret_cast<const long*>(sW + o), *reinterpret_cast<const long*>(sA + o),
                                                             acc, 0, 0, 0);
        }
        *reinterpret_cast<float4*>(red + ((wave * 16 + li) * 16 + g * 4)) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        __syncthreads();
        if (tid < 128) {
            const int mr = tid >> 3, nq = (tid & 7) * 2;
            float2 s = *reinterpret_cast<const float2*>(red + (mr * 16 + nq));
#pragma unroll
            for (int w = 1; w < 8; ++w) {
                const float2 t = *reinterpret_cast<const float2*>(red + ((w * 16 + mr) * 16 + nq));
                s.x += t.x; s.y += t.y;
            }
            if (mr < cnt) {
                const int pp = 2 * row0 + plist[c0 + mr];
                const float sc = sxs[mr] * wscale;
                s.x *= sc; s.y *= sc;
                if constexpr (STAGE == 0) {
                    *reinterpret_cast<uint32_t*>(a.hidden + (size_t)pp * a.d_ff + n0 + nq) = pack_bf16x2(fmaxf(s.x, 0.f), fmaxf(s.y, 0.f));
                } else {
                    const float gt = pGate[pp];
                    *reinterpret_cast<float2*>(a.y + (size_t)pp * a.d_model + n0 + nq) = make_float2(gt * s.x, gt * s.y);
                }
            }
        }
        __syncthreads();                                       // `red`, the row scales and the activation strips are reused by the next chunk
    }
}

template <int K>
constexpr size_t moe_fp8_lds(int R) { return (size_t)(8 * 16 * 16 + 8 + 48) * 4 + (size_t)8 * 2 * 16 * (K / 8 + 16) + (size_t)2 * R * 4; }

__global__ __launch_bounds__(512) void moe_combine_kernel(float* __restrict__ pH, const float* __restrict__ pY, float* __restrict__ pSsq, int row0,
                                                          int R, int ssq_stride, MoeArgs a) {   // leading scalars: kernarg preload
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = row0 + blockIdx.x * 8 + wave;
    if (r >= row0 + R) return;
    constexpr int D = 512;
    const float4* y0 = reinterpret_cast<const float4*>(pY + (size_t)(2 * r) * D);
    const float4* y1 = reinterpret_cast<const float4*>(pY + (size_t)(2 * r + 1) * D);
    float4* h = reinterpret_cast<float4*>(pH + (size_t)r * D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float4 u = y0[lane + 64 * i], v = y1[lane + 64 * i];
        float4 o = h[lane + 64 * i];
        o.x += u.x + v.x; o.y += u.y + v.y; o.z += u.z + v.z; o.w += u.w + v.w;
        h[lane + 64 * i] = o;
        q += o.x * o.x + o.y * o.y + o.z * o.z + o.w * o.w;
    }
    q = wave_sum(q);
    if (lane < SSQ_TILES) pSsq[(size_t)lane * ssq_stride + r] = lane == 0 ? q : 0.f;
}

template <int STAGE, int K>
constexpr size_t moe_lds(int R) { return (size_t)(8 * 16 * 16 + 8) * 4 + (size_t)8 * 2 * 16 * (K / 8 * 2 + 16) + (size_t)2 * R * 4; }

}  // namespace

constexpr int MOE_MAX_ROWS = 1536;       // LDS pair list: 2 * rows * 4 B on top of the 135 KB of strips of the K = 2048 stage (160 KB per CU)

int init_moe_kernels() {
    const hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void*>(moe_gemm_kernel<0, 512>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)moe_lds<0, 512>(MOE_MAX_ROWS));
    const hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(moe_gemm_kernel<1, 2048>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)moe_lds<1, 2048>(MOE_MAX_ROWS));
    const hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(moe_gemm_fp8_kernel<0, 512>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)moe_fp8_lds<512>(MOE_MAX_ROWS));
    const hipError_t e3 = hipFuncSetAttribute(reinterpret_cast<const void*>(moe_gemm_fp8_kernel<1, 2048>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)moe_fp8_lds<2048>(MOE_MAX_ROWS));
    return (e0 == hipSuccess && e1 == hipSuccess && e2 == hipSuccess && e3 == hipSuccess) ? 0 : -2;
}

// stage: 0 router, 1 expert wi, 2 expert wo, 3 combine
int launch_moe_stage(int stage, const MoeArgs& a, hipStream_t stream) {
    if (a.R <= 0) return 0;
    if (a.d_model != 512 || a.d_ff != 2048 || a.E > E_MAX || a.top_k != 2 || a.R > MOE_MAX_ROWS) return -1;
    switch (stage) {
        case 0: moe_router_kernel<<<(a.R + 7) / 8, 512, 0, stream>>>(a.h, a.gain, a.ssq, a.router, a.xn, a.row0, a.R, a.ssq_stride, a.E, a); break;
        case 1:
            if (a.fp8) moe_gemm_fp8_kernel<0, 512><<<a.E * (a.d_ff / 16), 512, moe_fp8_lds<512>(a.R), stream>>>(a.wi_q8, a.xn, a.sel, a.gate, a.row0, a.R, a);
            else moe_gemm_kernel<0, 512><<<a.E * (a.d_ff / 16), 512, moe_lds<0, 512>(a.R), stream>>>(a.wi, a.xn, a.sel, a.gate, a.row0, a.R, a);
            break;
        case 2:
            if (a.fp8) moe_gemm_fp8_kernel<1, 2048><<<a.E * (a.d_model / 16), 512, moe_fp8_lds<2048>(a.R), stream>>>(a.wo_q8, a.hidden, a.sel, a.gate, a.row0, a.R, a);
            else moe_gemm_kernel<1, 2048><<<a.E * (a.d_model / 16), 512, moe_lds<1, 2048>(a.R), stream>>>(a.wo, a.hidden, a.sel, a.gate, a.row0, a.R, a);
            break;
        case 3: moe_combine_kernel<<<(a.R + 7) / 8, 512, 0, stream>>>(a.h, a.y, a.ssq, a.row0, a.R, a.ssq_stride, a); break;
        default: return -1;
    }
    return 0;
}
