// The MoE decoder layer's skinny-GEMM chain as ONE launch (a11 of SURVEY.md section 8; up to 64 rows, 8 experts, top-2), the counterpart of
// dec_chain.hip for BASELINE configs[4]:
//
//     stage 0  cross-attention O-projection   h   += attn . wo_c^T (+ the folded self-attention partials)            as dec_chain stage 0
//     stage 1  router                         xn   = R(rmsnorm(h) * g3); logits = xn . router^T; top-2; gates          moe.hip: moe_router_kernel
//     stage 2  expert FFN-in                  hidden[p] = R(relu(xn[row(p)] . wi[e]^T))   for the pairs p of expert e   moe.hip: moe_gemm_kernel<0>
//     stage 3  expert FFN-out                 y[p] = gate[p] * (hidden[p] . wo[e]^T)                                    moe.hip: moe_gemm_kernel<1>
//     stage 4  the NEXT layer's QKV projection + cache append (or lm_head) on x = h + (y[2r] + y[2r+1])                 decode.hip: dec_gemm_kernel<.., PEND>
//
// Round 2 ran these as five launches per layer (cross O, router, two grouped expert GEMMs, the next norm GEMM with the combine folded in):
// 5 x (~2 us gap + 4-8 us body), most of a body waiting for weights from beyond L2.  Here all 256 workgroups (one per CU) request the weight
// tiles of every stage at entry and hand 16-row tiles / expert pair sets to each other through arrival counters on 128-byte lines of their own and
// agent-scope loads / stores (dec_chain.hip's protocol).  Workgroup (mt, nt) -- dec_chain's mapping -- owns:
//     stage 0: row tile mt, 16-column tile nt < 32;  stage 1: rows 16 mt + 8 nt .. + 7 for nt < 2 (one wave per row);
//     stage 2: expert e = mt + 4 (nt >> 5), tile j = nt & 31: hidden columns 64 j .. + 63 for all of e's pairs (up to 32 per MFMA pass);
//     stage 3: the same expert; its 32 workgroups split the expert's PAIRS as well as the columns (fp8: 64 columns x a quarter of the pairs, bf16:
//              32 columns x half of them), so the most popular expert costs one pass and a hidden row is read by 8 / 16 workgroups instead of 32;
//     stage 4: row tile mt, 32-column tile nt of the next projection.
// The arithmetic is that of the five kernels operation for operation (same K-slices per wave, MFMA chains, fixed-order reductions, the fp8 form's
// per-row dynamic scale; a pair's outputs do not depend on which pairs share its pass or its workgroup): ids are bit-identical to the launches (tests).
// Every wait is bounded (1 s) behind the handle's sticky abort word; the grid must be resident at once (runtime.hip asks the occupancy API).
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace {

#include "dec_chain_body.h"

constexpr int MC_B0 = 0, MC_ROUTER = 32, MC_FFN_IN = 40, MC_FFN_OUT = 104;       // counter lines (x CHAIN_LINE words): [4][8], [8], [8 experts][8], [8 experts][8]
static_assert(MC_FFN_OUT + 64 <= CHAIN_COUNTERS, "the MoE chain's counters live in the chain's counter block (zeroed by the preceding cross-attention launch)");

__device__ __forceinline__ void mc_wait(unsigned* sync, int line0, unsigned target, unsigned* host_abort) {
    counter_wait(sync + (size_t)(line0 + (blockIdx.x & 7)) * CHAIN_LINE, target, sync + CHAIN_ABORT_WORD, host_abort);
}
// wait until EIGHT counters (lines line0 + 8 e + this workgroup's replica, e = 0..7) have all reached `target`: one poll instruction, eight lanes.
// (Stage 4 needs every expert's outputs.  One counter for all 256 producers made the slowest arrival visible ~2 us late: adds to one line are served
// one after the other, and the producers finish together -- profiles/r03_moe_chain_marks_parts.txt.)
__device__ __forceinline__ void mc_wait8(unsigned* sync, int line0, unsigned target, unsigned* host_abort) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const unsigned* cnt = sync + (size_t)(line0 + (lane & 7) * 8 + (blockIdx.x & 7)) * CHAIN_LINE;
        unsigned* abort_word = sync + CHAIN_ABORT_WORD;
        unsigned long long t0 = 0;
        unsigned polls = 0;
        for (;;) {
            unsigned v = target;
            if (lane < 8) v = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__ballot(v < target) == 0ull) break;
            YMT3_POLL_PAUSE;
            if ((++polls & 63u) == 0u) {
                const unsigned long long now = wall_clock64();
                if (t0 == 0) t0 = now;
                if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || now - t0 > SPIN_LIMIT) {
                    if (lane == 0) {
                        __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (host_abort) __hip_atomic_store(host_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    break;
                }
            }
        }
    }
    __syncthreads();
}
__device__ __forceinline__ void mc_signal(unsigned* sync, int line0) { counter_signal(sync + (size_t)line0 * CHAIN_LINE, 8, CHAIN_LINE); }

__device__ __forceinline__ float mc_dpp_sum8(float v) {      // moe.hip: moe_dpp_sum8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    return v;
}

// stage 1, one wave per row: moe_router_kernel's arithmetic; inputs (h, the sum(h^2) partials) and outputs (xn, sel, gate) at agent scope
__device__ __forceinline__ void router_row(const MoeChainArgs& c, int r, float* xrow) {
    constexpr int D = 512;
    const int lane = threadIdx.x & 63;
    float ss = 0.f;
    if (lane < SSQ_TILES) ss = ld_agent(c.ssq + (size_t)lane * c.ssq_stride + r);
    const __amdgpu_buffer_rsrc_t rh = raw_rsrc(c.h);
    const f32x4 v0 = __builtin_bit_cast(f32x4, ld16_agent(rh, (r * D + lane * 4) * 4)), v1 = __builtin_bit_cast(f32x4, ld16_agent(rh, (r * D + (lane + 64) * 4) * 4));
    const float4* g = reinterpret_cast<const float4*>(c.gain_r);
    const float4 g0 = g[lane], g1 = g[lane + 64];
    const int grp = lane >> 3, sub = lane & 7;
    uint4 wv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) wv[i] = *reinterpret_cast<const uint4*>(c.router + (size_t)grp * D + sub * 64 + i * 8);      // 8 experts: every lane group has one
    ss = wave_sum(ss);
    const float sc = rsqrtf(ss / (float)D + c.eps);
    auto norm4 = [&](const f32x4& v, const float4& gg, int idx) {
        const bf16_t b0 = f2bf(v[0] * sc * gg.x), b1 = f2bf(v[1] * sc * gg.y), b2 = f2bf(v[2] * sc * gg.z), b3 = f2bf(v[3] * sc * gg.w);
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(c.xn + (size_t)r * D + idx * 4),
                           (unsigned long long)((uint32_t)b0 | ((uint32_t)b1 << 16)) | ((unsigned long long)((uint32_t)b2 | ((uint32_t)b3 << 16)) << 32),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *reinterpret_cast<float4*>(xrow + idx * 4) = make_float4(bf2f(b0), bf2f(b1), bf2f(b2), bf2f(b3));
    };
    norm4(v0, g0, lane);
    norm4(v1, g1, lane + 64);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float4 xa = *reinterpret_cast<const float4*>(xrow + sub * 64 + i * 8);
        const float4 xb = *reinterpret_cast<const float4*>(xrow + sub * 64 + i * 8 + 4);
        s = fmaf(xa.x, __uint_as_float(wv[i].x << 16), s); s = fmaf(xa.y, __uint_as_float(wv[i].x & 0xffff0000u), s);
        s = fmaf(xa.z, __uint_as_float(wv[i].y << 16), s); s = fmaf(xa.w, __uint_as_float(wv[i].y & 0xffff0000u), s);
        s = fmaf(xb.x, __uint_as_float(wv[i].z << 16), s); s = fmaf(xb.y, __uint_as_float(wv[i].z & 0xffff0000u), s);
        s = fmaf(xb.z, __uint_as_float(wv[i].w << 16), s); s = fmaf(xb.w, __uint_as_float(wv[i].w & 0xffff0000u), s);
    }
    s = mc_dpp_sum8(s);
    float best = -3.4e38f, second = -3.4e38f;
    int e0 = 0, e1 = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float le = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), e * 8));
        if (le > best) { second = best; e1 = e0; best = le; e0 = e; }
        else if (le > second) { second = le; e1 = e; }
    }
    if (lane == 0) {
        const float t = __expf(second - best);
        __hip_atomic_store(c.sel + 2 * r, e0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(c.sel + 2 * r + 1, e1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        st_agent(c.gate + 2 * r, 1.0f / (1.0f + t));
        st_agent(c.gate + 2 * r + 1, t / (1.0f + t));
        if (c.sel_trace) {                                      // debug hook only (ymt3_debug_moe_trace)
            const int st = c.shared->step - c.shared->step0;
            if (st >= 0 && st < c.trace_steps && r < c.trace_rows) {
                int32_t* dst = c.sel_trace + (((size_t)st * c.n_layers + c.layer) * c.trace_rows + r) * 2;
                dst[0] = e0; dst[1] = e1;
            }
        }
    }
}

// moe.hip: moe_find_pairs, the selections read at agent scope (the router wrote them in this launch)
__device__ __forceinline__ int find_pairs(const int* sel, int P, int e, int* plist, int* wcnt) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int total = 0;
    for (int base = 0; base < P; base += 512) {
        const int p = base + tid;
        const bool mine = p < P && __hip_atomic_load(sel + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == e;
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

// a chunk's <= 16 activation rows (this wave's K-slice of each), gathered by pair at agent scope: STAGE 0 rows of xn (K = 512), STAGE 1 rows of hidden (K = 2048)
template <int STAGE>
__device__ __forceinline__ void gather_rows(const MoeChainArgs& c, const int* plist, int cnt_all, int c0, u32x4* av) {
    constexpr int K = STAGE == 0 ? 512 : 2048;
    using G = Geo<K, 1>;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cnt = min(16, cnt_all - c0);
    const __amdgpu_buffer_rsrc_t ra = raw_rsrc(STAGE == 0 ? static_cast<const void*>(c.xn) : static_cast<const void*>(c.hidden));
#pragma unroll
    for (int i = 0; i < G::NIA; ++i) {
        const int row = i * G::RPIW + lane / G::LPRW, ch = lane % G::LPRW;
        const int pp = plist[c0 + (row < cnt ? row : cnt - 1)];
        const int arow = STAGE == 0 ? (pp >> 1) : pp;
        av[i] = ld16_agent(ra, (arow * K + wave * G::KW + ch * 8) * 2);
    }
}

// LDS map of the kernel (bytes from the start of the dynamic block)
constexpr int L_PLIST = 0;                                 // the expert's pair list [128] + wcnt [8]
constexpr int MC_PASS_MAX = 64;                            // pairs a pass of the expert stages can take (an expert has at most R <= 64)
constexpr int L_FP8 = 576;                                 // smax [64] (uint), sinv [64], sxs [64]: per-row fp8 scales of a pass
constexpr int L_WPART = L_FP8 + 3 * MC_PASS_MAX * 4;                    // [8][16] floats (stage 4's sum(x^2) per wave)
constexpr int L_RED = L_WPART + 8 * 16 * 4;                // cross-wave reduction: [8][16][64] floats in stage 2, [8][16][32] + 16 scales in stage 4, [8][16][16] elsewhere
constexpr int L_STRIPS = L_RED + (8 * 16 * 64 + 16) * 4;   // operand strips of stages 0, 1 (router rows), 2 and 4
constexpr int l_strips3(int nt) { return L_RED + (8 * 16 * 16 * nt + 16) * 4; }   // stage 3 (K = 2048, 16 nt columns): its 32-row strips start behind 8 nt KB of reduction space
static_assert(L_RED % 16 == 0 && L_STRIPS % 16 == 0 && l_strips3(2) % 16 == 0 && l_strips3(4) % 16 == 0, "16-byte LDS accesses");
constexpr int L_STRIPS_ALIGNED = L_STRIPS;

// Stages 2 / 3 for ONE expert: moe_gemm_kernel<STAGE> / moe_gemm_fp8_kernel<STAGE> (moe.hip) for `cnt_all` of its pairs, in ascending pair order,
// with the same K-slices per wave, MFMA chains per output, reduction order and epilogues -- but up to 32 pairs per pass: both 16-row halves are
// gathered (and, fp8, quantised) together and share the weight fragments and the barriers; each half's outputs are what a 16-pair chunk of the
// launch form computes, bit for bit.  (128 pairs over 8 experts: an expert has more than 16 in almost every step, and the second chunk of the
// slowest expert sat on the chain's critical path: profiles/r03_moe_chain_marks_first.txt.)
// STAGE 0: hidden[p] = R(relu(xn[row(p)] . wi[e]^T)), 64 columns (NT = 4), K = 512, weights parked in LDS strips from `wreg` (row layout);
// STAGE 1: y[p] = gate[p] * (hidden[p] . wo[e]^T), 16 NT columns, K = 2048, weights as MFMA fragments in registers: wreg[tt * KS + ks] (bf16, 16 B)
//          or wreg8[tt * KS + ks] (fp8, 8 B).
template <int STAGE, bool FP8, int NT, int NH>
__device__ __forceinline__ void expert_pass(const MoeChainArgs& c, const int* plist, int cnt_all, int col0, const u32x4* wreg, const long* wreg8, float wscale, char* smem) {
    constexpr int K = STAGE == 0 ? 512 : 2048;
    static_assert(STAGE == 1 || NT == 4, "stage 2 takes 64 hidden columns");
    constexpr int KW = K / 8, KS = KW / 32;
    constexpr int PITCH = FP8 ? KW + 16 : KW * 2 + 16, STRIP = 16 * PITCH;
    constexpr int LPR = KW * 2 / 16, RPI = 64 / LPR, NI = 16 / RPI;                               // a gathered bf16 row slice: lanes per row, rows per instruction
    constexpr int LPRW = FP8 ? KW / 16 : KW * 2 / 16, RPIW = 64 / LPRW, NIW = 16 * NT / RPIW;     // the weight rows of the tile as loaded (whole lines)
    constexpr bool WREG = STAGE == 1;
    constexpr int NSTRIP = NH + (WREG ? 0 : NT);                                                  // per wave: NH activation strips of 16 rows, then the weight strips
    constexpr int PASS = 16 * NH;                                                                 // pairs per pass
    static_assert(PASS <= MC_PASS_MAX, "the per-row fp8 scales");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
    float* red = reinterpret_cast<float*>(smem + L_RED);
    unsigned* smax = reinterpret_cast<unsigned*>(smem + L_FP8);
    float* sinv = reinterpret_cast<float*>(smax + MC_PASS_MAX);
    float* sxs = sinv + MC_PASS_MAX;
    char* sA = smem + (STAGE == 0 ? L_STRIPS : l_strips3(NT)) + wave * NSTRIP * STRIP;          // half hh at sA + hh * STRIP
    char* sW = sA + NH * STRIP;
    if constexpr (!WREG) {
#pragma unroll
        for (int i = 0; i < NIW; ++i)
            *reinterpret_cast<u32x4*>(sW + (i * RPIW + lane / LPRW) * PITCH + (lane % LPRW) * 16) = wreg[i];
    }
    const bool epi = tid < 16 * 8 * NT;
    const int mr = tid / (8 * NT), nq = (tid % (8 * NT)) * 2;
    for (int c0 = 0; c0 < cnt_all; c0 += PASS) {
        const int cnt = min(PASS, cnt_all - c0);
        const int nh = (cnt + 15) >> 4;                         // halves in use: workgroup-uniform
        u32x4 av[NH][NI];
        float gt[NH];                                           // STAGE 1: the pairs' gates, requested with the rows instead of behind the reduction
#pragma unroll
        for (int hh = 0; hh < NH; ++hh) {
            gt[hh] = 0.f;
            if (hh < nh) {
                gather_rows<STAGE>(c, plist, cnt_all, c0 + 16 * hh, av[hh]);
                if constexpr (STAGE == 1) {
                    if (epi && 16 * hh + mr < cnt) gt[hh] = ld_agent(c.gate + plist[c0 + 16 * hh + mr]);
                }
            }
        }
        if constexpr (FP8) {
            // per-row dynamic scale (amax / 448): 8-lane DPP max, then one LDS integer max per (row, 8-lane group) -- |x| >= 0, so float bit patterns order like unsigned integers
            if (tid < PASS) smax[tid] = 0u;
            __syncthreads();
#pragma unroll
            for (int hh = 0; hh < NH; ++hh) {
                if (hh >= nh) continue;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    float mx = 0.f;
#pragma unroll
                    for (int jq = 0; jq < 4; ++jq) {
                        mx = fmaxf(mx, fabsf(__uint_as_float(av[hh][i][jq] << 16)));
                        mx = fmaxf(mx, fabsf(__uint_as_float(av[hh][i][jq] & 0xffff0000u)));
                    }
                    mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mx), 0xB1, 0xF, 0xF, true)));
                    mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mx), 0x4E, 0xF, 0xF, true)));
                    mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mx), 0x141, 0xF, 0xF, true)));
                    if ((lane & 7) == 0) atomicMax(&smax[16 * hh + i * RPI + lane / LPR], __float_as_uint(mx));
                }
            }
            __syncthreads();
            if (tid < PASS) {
                const float mx = fmaxf(__uint_as_float(smax[tid]), 1e-12f);
                sinv[tid] = 448.0f / mx;
                sxs[tid] = mx / 448.0f;
            }
            __syncthreads();
#pragma unroll
            for (int hh = 0; hh < NH; ++hh) {
                if (hh >= nh) continue;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int row = i * RPI + lane / LPR;
                    const float inv = sinv[16 * hh + row];
                    const u32x4 v = av[hh][i];
                    int lo = 0, hi = 0;
                    lo = __builtin_amdgcn_cvt_pk_fp8_f32(__uint_as_float(v[0] << 16) * inv, __uint_as_float(v[0] & 0xffff0000u) * inv, lo, false);
                    lo = __builtin_amdgcn_cvt_pk_fp8_f32(__uint_as_float(v[1] << 16) * inv, __uint_as_float(v[1] & 0xffff0000u) * inv, lo, true);
                    hi = __builtin_amdgcn_cvt_pk_fp8_f32(__uint_as_float(v[2] << 16) * inv, __uint_as_float(v[2] & 0xffff0000u) * inv, hi, false);
                    hi = __builtin_amdgcn_cvt_pk_fp8_f32(__uint_as_float(v[3] << 16) * inv, __uint_as_float(v[3] & 0xffff0000u) * inv, hi, true);
                    *reinterpret_cast<int2*>(sA + hh * STRIP + row * PITCH + (lane % LPR) * 8) = make_int2(lo, hi);
                }
            }
        } else {
#pragma unroll
            for (int hh = 0; hh < NH; ++hh) {
                if (hh >= nh) continue;
#pragma unroll
                for (int i = 0; i < NI; ++i)
                    *reinterpret_cast<u32x4*>(sA + hh * STRIP + (i * RPI + lane / LPR) * PITCH + (lane % LPR) * 16) = av[hh][i];
            }
        }
        // MFMA: one chain per (half, 16-column tile) over this wave's K-slice; a weight fragment is read once for all halves
        f32x4 acc[NH][NT];
#pragma unroll
        for (int hh = 0; hh < NH; ++hh)
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) acc[hh][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if constexpr (FP8) {
                const int o = li * PITCH + ks * 32 + g * 8;
                long af[NH];
#pragma unroll
                for (int hh = 0; hh < NH; ++hh) af[hh] = hh < nh ? *reinterpret_cast<const long*>(sA + hh * STRIP + o) : 0L;
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) {
                    long wf;
                    if constexpr (WREG) wf = wreg8[tt * KS + ks];
                    else wf = *reinterpret_cast<const long*>(sW + tt * STRIP + o);
#pragma unroll
                    for (int hh = 0; hh < NH; ++hh)
                        if (hh < nh) acc[hh][tt] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wf, af[hh], acc[hh][tt], 0, 0, 0);
                }
            } else {
                const int o = li * PITCH + (ks * 32 + g * 8) * 2;
                bf16x8 af[NH];
#pragma unroll
                for (int hh = 0; hh < NH; ++hh) af[hh] = *reinterpret_cast<const bf16x8*>(sA + (hh < nh ? hh : 0) * STRIP + o);
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) {
                    bf16x8 wf;
                    if constexpr (WREG) wf = __builtin_bit_cast(bf16x8, wreg[tt * KS + ks]);
                    else wf = *reinterpret_cast<const bf16x8*>(sW + tt * STRIP + o);
#pragma unroll
                    for (int hh = 0; hh < NH; ++hh)
                        if (hh < nh) acc[hh][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af[hh], acc[hh][tt], 0, 0, 0);
                }
            }
        }
        // fixed-order cross-wave reduction and epilogue, one half after the other (the reduction space holds one)
#pragma unroll
        for (int hh = 0; hh < NH; ++hh) {
            if (hh >= nh) continue;
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                const f32x4 a = acc[hh][tt];
                *reinterpret_cast<float4*>(red + ((wave * 16 + li) * (16 * NT) + tt * 16 + g * 4)) = make_float4(a[0], a[1], a[2], a[3]);
            }
            __syncthreads();
            if (epi) {
                float2 sv = *reinterpret_cast<const float2*>(red + (mr * (16 * NT) + nq));
#pragma unroll
                for (int w = 1; w < 8; ++w) {
                    const float2 t = *reinterpret_cast<const float2*>(red + ((w * 16 + mr) * (16 * NT) + nq));
                    sv.x += t.x; sv.y += t.y;
                }
                const int rr = 16 * hh + mr;
                if (rr < cnt) {
                    const int pp = plist[c0 + rr];
                    if constexpr (FP8) {
                        const float sc = sxs[rr] * wscale;
                        sv.x *= sc; sv.y *= sc;
                    }
                    if constexpr (STAGE == 0) {
                        st_agent(reinterpret_cast<uint32_t*>(c.hidden + (size_t)pp * 2048 + col0 + nq), pack_bf16x2(fmaxf(sv.x, 0.f), fmaxf(sv.y, 0.f)));
                    } else {
                        st2_agent(c.y + (size_t)pp * 512 + col0 + nq, make_float2(gt[hh] * sv.x, gt[hh] * sv.y));
                    }
                }
            }
            __syncthreads();                                   // the reduction space (and, after the last half, the strips and the scales) are reused
        }
    }
}

// stage 4: dec_gemm_kernel<MODE, 512, 2, PEND = true> for tile (mt, nt3): x = h + (y[2r] + y[2r+1]) formed while loading (agent scope), sum(x^2)
// per wave and row, the eight waves in order; column tile 0 publishes x to the other residual buffer (QKV mode)
template <int MODE>
__device__ __forceinline__ void pend_tile(const MoeChainArgs& c, int R, const f32x4 gv, int nt_idx, int mt_idx, int step, const u32x4 (&w3)[Geo<512, 2>::NIW], char* smem) {
    using G = Geo<512, 2>;
    constexpr int NT = 2, K = 512;
    float* red = reinterpret_cast<float*>(smem + L_RED);
    float* sscale = red + 8 * 16 * G::COLS;
    float* wpart = reinterpret_cast<float*>(smem + L_WPART);
    char* strips = smem + L_STRIPS_ALIGNED;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = nt_idx * G::COLS, m0 = mt_idx * 16, m_end = R;
    char* sA = strips + wave * (1 + NT) * G::STRIP;
    char* sW = sA + G::STRIP;
    const __amdgpu_buffer_rsrc_t rx = raw_rsrc(c.h), ry = raw_rsrc(c.y);
    f32x4 xv[G::NIX], y0v[G::NIX], y1v[G::NIX];
#pragma unroll
    for (int i = 0; i < G::NIX; ++i) {
        int mm = m0 + i * G::RPIX + lane / G::LPRX;
        mm = mm < m_end ? mm : m_end - 1;
        const int col = wave * G::KW + (lane % G::LPRX) * 4;
        xv[i] = __builtin_bit_cast(f32x4, ld16_agent(rx, (mm * K + col) * 4));
        y0v[i] = __builtin_bit_cast(f32x4, ld16_agent(ry, ((2 * mm) * K + col) * 4));
        y1v[i] = __builtin_bit_cast(f32x4, ld16_agent(ry, ((2 * mm + 1) * K + col) * 4));
    }
#pragma unroll
    for (int i = 0; i < G::NIX; ++i) {
        const int row = i * G::RPIX + lane / G::LPRX;
#pragma unroll
        for (int e = 0; e < 4; ++e) xv[i][e] += y0v[i][e] + y1v[i][e];          // h + (y0 + y1), as the combine kernel summed
        float q = (xv[i][0] * xv[i][0] + xv[i][1] * xv[i][1]) + (xv[i][2] * xv[i][2] + xv[i][3] * xv[i][3]);
        q = add_xor8(sum8(q));                           // the LPRX = 16 lanes that hold this row's K-slice
        if ((lane % G::LPRX) == 0) wpart[wave * 16 + row] = q;
        if constexpr (MODE == DG_NORM_QKV_CACHE) {
            if (nt_idx == 0 && m0 + row < m_end)         // column tile 0 publishes the completed residual row (read by the next launch)
                *reinterpret_cast<f32x4*>(c.h_out + (size_t)(m0 + row) * K + wave * G::KW + (lane % G::LPRX) * 4) = xv[i];
        }
    }
#pragma unroll
    for (int i = 0; i < G::NIW; ++i)
        *reinterpret_cast<u32x4*>(sW + (i * G::RPIW + lane / G::LPRW) * G::PITCH + (lane % G::LPRW) * 16) = w3[i];
    __syncthreads();
    if (tid < 16) {
        float t = wpart[tid];
#pragma unroll
        for (int w = 1; w < 8; ++w) t += wpart[w * 16 + tid];
        sscale[tid] = rsqrtf(t / (float)K + c.eps);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < G::NIX; ++i) {
        const int row = i * G::RPIX + lane / G::LPRX;
        const float sc = sscale[row];
        *reinterpret_cast<uint2*>(sA + row * G::PITCH + (lane % G::LPRX) * 8) =
            make_uint2(pack_bf16x2(xv[i][0] * sc * gv[0], xv[i][1] * sc * gv[1]), pack_bf16x2(xv[i][2] * sc * gv[2], xv[i][3] * sc * gv[3]));
    }
    float2 s = mfma_reduce<512, NT>(sA, sW, red);
    const bool epi = tid < 16 * 8 * NT;
    const int mr = tid / (8 * NT), nq = (tid % (8 * NT)) * 2;
    const int m = m0 + mr, n = n0 + nq;
    if (!(epi && m < m_end)) return;
    if constexpr (MODE == DG_NORM_LOGITS) {
        *reinterpret_cast<float2*>(c.logits + (size_t)m * c.N3 + n) = s;
    } else {
        const uint32_t pk = pack_bf16x2(s.x, s.y);
        const int inner = c.H * DKV;
        if (n < inner) {
            *reinterpret_cast<uint32_t*>(c.out_q + (size_t)m * inner + n) = pk;
        } else {
            const int nn = n - inner, kv = nn / inner, hh = (nn % inner) >> 6, dd = nn & 63;
            bf16_t* cache = kv ? c.vcache : c.kcache;
            *reinterpret_cast<uint32_t*>(cache + (((size_t)m * c.H + hh) * c.L + step) * DKV + dd) = pk;
        }
    }
}

template <int MODE3, bool FP8>
__global__ __launch_bounds__(512) void moe_chain_kernel(const bf16_t* __restrict__ pW0, const void* __restrict__ pWi, const void* __restrict__ pWo,
                                                        const bf16_t* __restrict__ pW3, const bf16_t* __restrict__ pAttn, float* pH, int R, int pad_,
                                                        MoeChainArgs c) {
    // leading scalar arguments: kernarg preload, as dec_chain_kernel
    using G0 = Geo<512, 1>;
    using G1 = Geo<512, 2>;
    using G2 = Geo<2048, 1>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_mt = (R + 15) / 16, m_end = R;
    const int t = blockIdx.x;
    const int mt = (t >> 3) & 3, nt = (t >> 5) * 8 + (t & 7);             // nt in [0, 64): dec_chain_kernel's mapping
    const int n_nt3 = c.N3 / 32;
    const bool has0 = mt < n_mt && nt < 32, has3 = mt < n_mt && nt < n_nt3;
    const int nt0 = nt & 31, nt3 = nt < n_nt3 ? nt : nt - n_nt3;
    const int e2 = mt + 4 * (nt >> 5), ot = nt & 31;          // the workgroup's expert and its tile index in stages 2 (64 hidden columns) and 3 (16 output columns)

    // ---- stage 0 operands first, then every later stage's weights
    u32x4 w0[G0::NIW], av[G0::NIA];
    load_w<512, 1>(pW0, nt0 * 16, w0);
    const int m0 = mt * 16;
#pragma unroll
    for (int i = 0; i < G0::NIA; ++i) {
        int mm = m0 + i * G0::RPIW + lane / G0::LPRW;
        mm = mm < m_end ? mm : m_end - 1;
        av[i] = *reinterpret_cast<const u32x4*>(pAttn + (size_t)mm * 512 + wave * G0::KW + (lane % G0::LPRW) * 8);
    }
    const bool epi0 = tid < 16 * 8;
    const int mr0 = tid / 8, nq0 = (tid % 8) * 2;
    const int mE = m0 + mr0, nE = nt0 * 16 + nq0;
    const bool live0 = has0 && epi0 && mE < m_end;
    float2 hold = make_float2(0.f, 0.f);
    float2 hp[8];
    if (live0) hold = *reinterpret_cast<const float2*>(pH + (size_t)mE * 512 + nE);
    if (c.part && live0) {
#pragma unroll
        for (int w = 0; w < 8; ++w) hp[w] = *reinterpret_cast<const float2*>(c.part + ((size_t)mE * 8 + w) * 512 + nE);
    }
    int step = 0;
    if constexpr (MODE3 == DG_NORM_QKV_CACHE) {
        const int mq = mt * 16 + tid / 16;
        step = c.row_pos ? c.row_pos[mq < m_end ? mq : m_end - 1] : c.shared->step;
    }
    __builtin_amdgcn_sched_barrier(0);
    // expert weights: stage 2 = hidden columns 64 ot .. + 63 of expert e2 (K = 512), row layout; stage 3 = output columns 16 NT3 ct .. of the same
    // expert (K = 2048) as MFMA fragments: lane (li, g) holds column li's k = 32 ks + 8 g .. + 7 of this wave's K-slice for every 16-column tile
    constexpr int NH2 = 2;                                    // stage 2: 16-pair halves per pass (4 -- all of an expert's <= 64 pairs in one pass, fp8 form -- was measured: no faster, 334-336 vs 332.5 ms)
    constexpr int NT3 = FP8 ? 4 : 2;                          // stage 3: 16 NT3 columns per workgroup, the expert's pairs split NT3 ways (32 workgroups per expert either way)
    const int ct = ot % (32 / NT3), part = ot / (32 / NT3);   // ot = ct + (32 / NT3) part: the NT3 parts of a column tile sit on one XCD (blockIdx & 7 == nt & 7) and share its weights in L2
    constexpr int NW_IN = FP8 ? 4 : Geo<512, 4>::NIW;
    u32x4 wi4[NW_IN], wo2[FP8 ? 16 : NT3 * G2::KS], w3[G1::NIW];   // fp8: wo2 = the 64 rows' K-slices in row layout (whole lines; fragments are made of them below)
    long wo8[FP8 ? NT3 * G2::KS : 1];
    // Requested as the stages come up, not all at entry (round 3, late: a load instruction waits for room in the CU's vector-memory queue, and
    // 300 KB of weight requests per workgroup at entry sat in front of the first stages' own loads -- dec_chain_body.h, profiles/r03_notes.md):
    // stage 2's tile behind stage 0, stage 3's behind the router's stage, stage 4's behind stage 2
    auto load_wi = [&]() {
        if constexpr (FP8) {
            const uint8_t* Wi = static_cast<const uint8_t*>(pWi);
            // fp8 rows: K bytes; this wave's K-slice = KW bytes, 16 bytes per lane: 4 lanes per row of the K = 512 tile
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
                wi4[tt] = *reinterpret_cast<const u32x4*>(Wi + ((size_t)e2 * 2048 + ot * 64 + tt * 16 + lane / 4) * 512 + wave * 64 + (lane % 4) * 16);
        } else {
            const bf16_t* Wi = static_cast<const bf16_t*>(pWi);
            load_w<512, 4>(Wi + (size_t)e2 * 2048 * 512, ot * 64, wi4);
        }
    };
    auto load_wo = [&]() {
        if constexpr (FP8) {
            const uint8_t* Wo = static_cast<const uint8_t*>(pWo);
#pragma unroll
            for (int i = 0; i < 16; ++i)
                wo2[i] = *reinterpret_cast<const u32x4*>(Wo + ((size_t)e2 * 512 + ct * 64 + i * 4 + lane / 16) * 2048 + wave * 256 + (lane % 16) * 16);
        } else {
            const bf16_t* Wo = static_cast<const bf16_t*>(pWo);
#pragma unroll
            for (int tt = 0; tt < NT3; ++tt)
                load_w_frag<2048>(Wo + (size_t)e2 * 512 * 2048, ct * 16 * NT3 + tt * 16, *reinterpret_cast<u32x4(*)[G2::KS]>(wo2 + tt * G2::KS));
        }
    };
    __builtin_amdgcn_sched_barrier(0);
    CH_STAMP_IN(c);

    // ---- stage 0: cross-attention O-projection (dec_chain_kernel's stage 0)
    if (has0) {
        float* red = reinterpret_cast<float*>(smem + L_RED);
        char* strips = smem + L_STRIPS_ALIGNED;
        char* sA = strips + wave * 2 * G0::STRIP;
        char* sW = sA + G0::STRIP;
#pragma unroll
        for (int i = 0; i < G0::NIW; ++i) {
            const int off = (i * G0::RPIW + lane / G0::LPRW) * G0::PITCH + (lane % G0::LPRW) * 16;
            *reinterpret_cast<u32x4*>(sW + off) = w0[i];
            *reinterpret_cast<u32x4*>(sA + off) = av[i];
        }
        const float2 s = mfma_reduce<512, 1>(sA, sW, red);
        if (c.part && live0) {
            float2 sp = hp[0];
#pragma unroll
            for (int w = 1; w < 8; ++w) { sp.x += hp[w].x; sp.y += hp[w].y; }
            hold.x += sp.x; hold.y += sp.y;
        }
        resid_out(hold, s, pH, c.ssq, c.ssq_stride, mE, nE, nt0, live0);
        mc_signal(c.sync, MC_B0 + mt * 8);
        CH_MARK(c, 0);
    }
    load_wi();
    __builtin_amdgcn_sched_barrier(0);
    // ---- stage 1: router, one wave per row: workgroups (mt, nt = 0 / 1) take rows 16 mt + 8 nt .. + 7
    if (has0 && nt0 < 2) {
        mc_wait(c.sync, MC_B0 + mt * 8, 32u, c.host_abort);
        CH_MARK(c, 1);
        const int r = m0 + nt0 * 8 + wave;
        float* xrow = reinterpret_cast<float*>(smem + L_STRIPS_ALIGNED) + wave * 512;       // wave-private: LDS executes a wave's accesses in order
        if (r < m_end) router_row(c, r, xrow);
        mc_signal(c.sync, MC_ROUTER);
        CH_MARK(c, 2);
    }
    // ---- stage 2: expert FFN-in, 64 hidden columns of expert e2 for its pairs (an expert nobody chose costs a scan)
    int* plist = reinterpret_cast<int*>(smem + L_PLIST);
    int* wcnt = plist + 128;
    load_wo();
    __builtin_amdgcn_sched_barrier(0);
    mc_wait(c.sync, MC_ROUTER, (unsigned)(2 * n_mt), c.host_abort);
    CH_MARK(c, 3);
    const int cnt = find_pairs(c.sel, 2 * R, e2, plist, wcnt);
    float ws_in = 1.f, ws_out = 1.f;
    if constexpr (FP8) { ws_in = c.wi_s[e2]; ws_out = c.wo_s[e2]; }
    if (cnt) expert_pass<0, FP8, 4, NH2>(c, plist, cnt, ot * 64, wi4, nullptr, ws_in, smem);
    mc_signal(c.sync, MC_FFN_IN + e2 * 8);
    CH_MARK(c, 4);
    load_w<512, 2>(pW3, nt3 * 32, w3);
    const f32x4 g3 = norm_gain(c.gain3);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (FP8) {
        // stage 3's weights from row layout to MFMA fragments through this wave's own strip space, 32 rows at a time (a wave's LDS accesses execute in
        // order: no barrier).  (Fragment-shaped 8-byte global loads at entry cost stage 0 ~2 us: 16 cache lines per instruction.)
        constexpr int WP = 256 + 16;
        char* wl = smem + L_STRIPS + wave * 32 * WP;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(wl + (i * 4 + lane / 16) * WP + (lane % 16) * 16) = wo2[hh * 8 + i];
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int ks = 0; ks < G2::KS; ++ks)
                    wo8[(hh * 2 + t2) * G2::KS + ks] = *reinterpret_cast<const long*>(wl + (t2 * 16 + (lane & 15)) * WP + ks * 32 + (lane >> 4) * 8);
        }
    }
    // ---- stage 3: expert FFN-out, 16 NT3 output columns of the same expert for part `part` of its pairs (ascending pair order within a part; a
    // pair's outputs do not depend on which pairs it is grouped with).  Splitting the pairs, not only the columns, keeps the most popular expert's
    // workgroups to one pass (<= 64 / NT3 pairs), and a hidden row is read by 32 / NT3 workgroups instead of 32
    {
        mc_wait(c.sync, MC_FFN_IN + e2 * 8, 32u, c.host_abort);
        CH_MARK(c, 5);
        const int per = (cnt + NT3 - 1) / NT3, first = min(cnt, part * per), mine = min(cnt, first + per) - first;
        if (mine > 0) expert_pass<1, FP8, NT3, 4 / NT3>(c, plist + first, mine, ct * 16 * NT3, wo2, wo8, ws_out, smem);     // a part has at most 64 / NT3 pairs: one pass
        mc_signal(c.sync, MC_FFN_OUT + e2 * 8);
        CH_MARK(c, 6);
    }
    // ---- stage 4: the next layer's QKV projection (or lm_head) on h + (y0 + y1)
    if (has3) {
        mc_wait8(c.sync, MC_FFN_OUT, 32u, c.host_abort);
        CH_MARK(c, 7);
        pend_tile<MODE3>(c, R, g3, nt3, mt, step, w3, smem);
    }
    CH_STAMP_OUT(c);
}

constexpr size_t MOE_CHAIN_LDS_BF16 = (size_t)l_strips3(2) + (size_t)8 * 2 * 16 * (2048 / 8 * 2 + 16);               // stage 3: two 16-row activation strips per wave (weights in registers)
constexpr size_t MOE_CHAIN_LDS_FP8 = (size_t)L_STRIPS + (size_t)8 * 32 * (256 + 16);                                // stage 3's weights on their way to fragments; stage 2: two activation + four weight strips per wave
static_assert(MOE_CHAIN_LDS_BF16 >= (size_t)L_STRIPS + (size_t)8 * 6 * 16 * (512 / 8 * 2 + 16), "stage 2 (bf16): two activation + four weight strips per wave");
static_assert(MOE_CHAIN_LDS_FP8 >= (size_t)l_strips3(4) + (size_t)8 * 1 * 16 * (2048 / 8 + 16), "stage 3 (fp8): one activation strip per wave");
static_assert(MOE_CHAIN_LDS_FP8 >= (size_t)L_STRIPS + (size_t)8 * 6 * 16 * (512 / 8 + 16), "stage 2 (fp8)");
static_assert(MOE_CHAIN_LDS_FP8 >= (size_t)L_STRIPS + (size_t)8 * 3 * 16 * (512 / 8 * 2 + 16), "stages 0 and 4 (bf16 strips) in the fp8 kernel");
static_assert(MOE_CHAIN_LDS_BF16 >= (size_t)L_STRIPS + 8 * 512 * 4 && MOE_CHAIN_LDS_FP8 >= (size_t)L_STRIPS + 8 * 512 * 4, "router rows");
static_assert(MOE_CHAIN_LDS_BF16 <= 160 * 1024 && MOE_CHAIN_LDS_FP8 <= 160 * 1024, "one workgroup per CU");

template <int MODE3, bool FP8>
int set_lds() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(moe_chain_kernel<MODE3, FP8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)(FP8 ? MOE_CHAIN_LDS_FP8 : MOE_CHAIN_LDS_BF16)) == hipSuccess ? 0 : -2;
}

}  // namespace

int init_moe_chain_kernels() {
    return set_lds<DG_NORM_QKV_CACHE, false>() | set_lds<DG_NORM_LOGITS, false>() | set_lds<DG_NORM_QKV_CACHE, true>() | set_lds<DG_NORM_LOGITS, true>();
}

// all 256 workgroups wait for each other: one per CU at the kernel's LDS / register footprint
bool moe_chain_fits(int n_cus, bool fp8) {
    int a = 0, b = 0;
    if (fp8) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, moe_chain_kernel<DG_NORM_QKV_CACHE, true>, 512, MOE_CHAIN_LDS_FP8) != hipSuccess) return false;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, moe_chain_kernel<DG_NORM_LOGITS, true>, 512, MOE_CHAIN_LDS_FP8) != hipSuccess) return false;
    } else {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, moe_chain_kernel<DG_NORM_QKV_CACHE, false>, 512, MOE_CHAIN_LDS_BF16) != hipSuccess) return false;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, moe_chain_kernel<DG_NORM_LOGITS, false>, 512, MOE_CHAIN_LDS_BF16) != hipSuccess) return false;
    }
    return (long long)(a < b ? a : b) * n_cus >= 256;
}

// 0 = launched; negative = not this kernel's shape (the caller launches the five kernels instead)
int launch_moe_chain(const MoeChainArgs& c, hipStream_t stream) {
    if (c.R <= 0) return 0;
    if (c.R > 64 || c.E != 8 || c.N3 % 32 || c.N3 / 32 < 32 || c.N3 / 32 > 64 || !c.sync || !c.part || (c.mode3 != DG_NORM_QKV_CACHE && c.mode3 != DG_NORM_LOGITS) ||
        (c.mode3 == DG_NORM_QKV_CACHE && !c.h_out))
        return -1;
#define MOE_CHAIN(M, F, LDS) moe_chain_kernel<M, F><<<256, 512, LDS, stream>>>(c.wo_c, c.wi, c.wo, c.w3, c.attn, c.h, c.R, 0, c)
    if (c.fp8) {
        if (c.mode3 == DG_NORM_QKV_CACHE) MOE_CHAIN(DG_NORM_QKV_CACHE, true, MOE_CHAIN_LDS_FP8);
        else MOE_CHAIN(DG_NORM_LOGITS, true, MOE_CHAIN_LDS_FP8);
    } else {
        if (c.mode3 == DG_NORM_QKV_CACHE) MOE_CHAIN(DG_NORM_QKV_CACHE, false, MOE_CHAIN_LDS_BF16);
        else MOE_CHAIN(DG_NORM_LOGITS, false, MOE_CHAIN_LDS_BF16);
    }
#undef MOE_CHAIN
    return 0;
}
