// One decode step's SIX decoder layers as ONE launch (a7 of SURVEY.md section 8; dense FFN, one channel, up to 64 rows).
//
// Round 2 left a step as 14 dependent launches: per layer an attention pair (dec_attn_pair_kernel: self-attention with its folded
// O-projection, then the fused cross-attention) and a GEMM chain (dec_chain_kernel: cross O-projection -> FFN-in -> FFN-out -> the next
// layer's QKV projection / lm_head).  A launch boundary costs ~2 us of stream time plus a dispatch ramp and a cold first load (~1.6 us),
// twelve times per step.  Here the two remaining kinds of boundary become in-kernel hand-offs of the kind the two merged kernels already
// use (agent-scope data, an arrival counter on a 128-byte line of its own, bounded polls):
//
//     attention -> chain     the row tile's 16 x 8 (row, head) workgroups arrive at `attn_done[layer][row tile]` (eight replicas); the
//                            tile's chain workgroups have requested every weight of their four stages before they wait for it
//     QKV -> attention       a stage-3 tile holds 32 columns of ONE head's q, k or v: it stores them at agent scope (the new cache line of
//                            a (row, head) has never been touched in this launch, so the attention's streaming loads find it in memory) and
//                            arrives at `qkv_done[layer + 1][row tile][head]` (six tiles); the sixteen (row, head) workgroups of that
//                            head wait for it, their head's 64 KB of wo already requested
//
// Workgroup b of the 512 = (row b / 8, head b % 8) in the attention halves; its row tile is b / 128, and workgroups b % 128 < 64 are that
// row tile's 64 chain tiles.  A row tile's 128 workgroups wait for nothing outside the tile: the four tiles run their six layers as four
// independent pipelines (one tile's latency-bound GEMM stages overlap another's K/V streams), which no sequence of launches can do.
//
// Arithmetic: attn_body's and chain_stages', operation for operation (same K-slices per wave, same MFMA chains, same fixed-order
// reductions): token ids are bit-identical to the 14-launch step, to the 38-launch step and to the round-1 digests.
// Residency: 512 workgroups of 512 threads, <= 128 VGPRs, 76 KB of LDS = two per CU on 256 CUs; the runtime asks the occupancy API.
// Counters are zeroed by the step's argmax kernel (the launch after this one) and by decode_init; every poll is bounded (1 s) behind the
// sticky abort word (runtime.hip: recovery through the separate launches).
#include <cstddef>
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace {

// Every read of the thread index is opaque to the optimiser (an empty volatile asm): the kernel body is a loop over the decoder layers whose
// phases each need close to all 128 VGPRs, and hipcc otherwise hoists the per-thread address arithmetic of ALL phases out of that loop and
// spills it (144 VGPRs of scratch, reloaded on the critical path of every stage).  Recomputing it per phase costs a few VALU operations.
__device__ __forceinline__ unsigned step_tid() {
    unsigned t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}
#define YMT3_TID step_tid()
#define YMT3_POLL_PAUSE __builtin_amdgcn_s_sleep(4)
#include "dec_chain_body.h"

constexpr int STEP_MARK0 = 1024, STEP_MARKS = 24;     // measurement: 4 marks per layer per workgroup behind the [512][2] entry / exit stamps (a stamp slot holds 16384 words)
constexpr int WO_BYTES = 8 * 64 * 128;          // the head's 64 k of all 512 wo rows, per wave 8 KB
constexpr int STEP_LDS = (8 * 16 * 16 + 16) * 4 + 8 * 16 * (2048 / 8 * 2 + 16);     // chain stage 2 (W2F) = 75 840 B; attention needs 64 KB + 4.4 KB
static_assert(STEP_LDS >= WO_BYTES + 5120, "attention's statics live behind the wo strip");
static_assert(STEP_LDS <= 80 * 1024, "two workgroups per CU");

// The argument struct reaches the kernel body through an opaque copy (see dec_step_kernel), so hipcc no longer knows that its pointers are
// global memory and would emit flat loads / stores (which also count against LDS waits, and in front of which it drains the store queue).
// Each pointer is re-made from its bits as a global-address-space pointer: the optimiser propagates that to every access.
template <typename T>
__device__ __forceinline__ T* as_global(T* p) {
    return (T*)(__attribute__((address_space(1))) T*)(unsigned long long)p;
}
__device__ __forceinline__ void globalize(StepArgs& s) {
    s.q = as_global(s.q); s.attn = as_global(s.attn); s.opart = as_global(s.opart); s.h = as_global(s.h); s.ssq = as_global(s.ssq);
    s.dff = as_global(s.dff); s.logits = as_global(s.logits); s.bias = as_global(s.bias); s.shared = as_global(s.shared);
    s.row_pos = as_global(s.row_pos); s.sync = as_global(s.sync); s.pair_rows = as_global(s.pair_rows); s.abort_word = as_global(s.abort_word);
    s.stamp = as_global(s.stamp);
    // (host_abort is pinned host memory: it stays a generic pointer)
}
__device__ __forceinline__ void globalize(StepLayer& L) {
    L.wo = as_global(L.wo); L.wq_c = as_global(L.wq_c); L.wo_c = as_global(L.wo_c); L.wi = as_global(L.wi); L.wo2 = as_global(L.wo2); L.w3 = as_global(L.w3);
    L.ln2 = as_global(L.ln2); L.ln3 = as_global(L.ln3); L.gain3 = as_global(L.gain3);
    L.kself = as_global(L.kself); L.vself = as_global(L.vself); L.kcross = as_global(L.kcross); L.vcross = as_global(L.vcross);
    L.knext = as_global(L.knext); L.vnext = as_global(L.vnext);
}

struct AttnLds {            // attention scratch inside the dynamic LDS block, behind the wo strip
    float* sm; float* sl; float* sacc; float* xs; bf16_t* qs; float* s_tile;
    char* wo;
};
__device__ __forceinline__ AttnLds attn_lds(char* smem) {
    AttnLds a;
    a.wo = smem;
    float* f = reinterpret_cast<float*>(smem + WO_BYTES);
    a.sm = f; a.sl = f + 8; a.sacc = f + 16; a.xs = f + 16 + 8 * DKV; a.s_tile = a.xs + 512;
    a.qs = reinterpret_cast<bf16_t*>(a.s_tile + SSQ_TILES);
    return a;
}

// the row's eight heads meet between the self- and the cross-attention half: decode.hip's pair_wait
__device__ __forceinline__ void row_wait(unsigned* rows, unsigned* abort_word, unsigned* host_abort, int r) {
    if (YMT3_TID == 0) {
        unsigned* arr = rows + (size_t)(2 * r) * CHAIN_LINE;
        unsigned* dep = arr + CHAIN_LINE;
        unsigned long long t0 = 0;
        unsigned polls = 0;
        bool ok = true;
        while (__hip_atomic_load(arr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 8u) {
            YMT3_POLL_PAUSE;
            if ((++polls & 63u) == 0u) {
                const unsigned long long now = wall_clock64();
                if (t0 == 0) t0 = now;
                if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || now - t0 > SPIN_LIMIT) {
                    __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (host_abort) __hip_atomic_store(host_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    ok = false;
                    break;
                }
            }
        }
        if (ok && __hip_atomic_fetch_add(dep, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 7u) {       // the last head to leave zeroes both
            __hip_atomic_store(arr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dep, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
}

// Online-softmax state of one lane: its 8-lane group's key, 8 of the 64 output dims
struct Soft { float m, l, acc[8]; };

// ---- self-attention half of (row r, head h): decode.hip attn_body<SELF, !FUSEQ, 8, OP, PAIR = 1>, statics in the dynamic LDS block.
// qkv_done != null: q and the newest cache line were written earlier in this launch (by the previous layer's QKV stage): wait for the head's
// six tiles after the wo request.  q is read at agent scope either way.  (ONE instantiation for all layers: with a second one for layer 0
// hipcc merged the two tails and drained the store queue -- s_waitcnt vmcnt(0) -- in front of the cross-attention half's counted wait.)
__device__ __forceinline__ void self_half(const StepArgs& s, const StepLayer& L, int r, int h, int n_keys, const AttnLds& lds, const unsigned* qkv_done,
                                          unsigned qkv_target, unsigned long long* mark) {
    constexpr int NW = 8, U = 6, H = 8;
    const int tid = YMT3_TID, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sub = lane & 7, kg = lane >> 3;
    const size_t slab = ((size_t)r * H + h) * s.L * DKV;
    const bf16_t* kb = L.kself + slab + sub * 8;
    const bf16_t* vb = L.vself + slab + sub * 8;
    const float* bias = s.bias + (size_t)h * s.L;
    {   // this wave's 64 rows of wo (output columns 64w..64w+63), the head's 64 k each: 8 DMAs of 8 rows x 128 B, source-swizzled
        const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds.wo + (unsigned)wave * 8192u;
        const bf16_t* src = L.wo + ((size_t)(wave * 64 + (lane >> 3)) * H + h) * DKV + ((lane & 7) ^ (lane >> 3)) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) glds16(src + (size_t)i * 8 * H * DKV, lds0 + (unsigned)i * 1024u);
    }
    if (qkv_done) counter_wait(qkv_done, qkv_target, s.abort_word, s.host_abort);
    if (mark && YMT3_TID == 0) mark[0] = wall_clock64();       // measurement: q is there
    const u32x4 qp = ld16_agent(raw_rsrc(s.q), (((r * H + h) * DKV) + sub * 8) * 2);
    float m = -1.0e30f, l = 0.f, acc[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) acc[d] = 0.f;

    auto load_block = [&](u32x4 (&ku)[U], u32x4 (&vu)[U], bool (&ok)[U], float (&bv)[U], int kw, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int k0 = kw + kg;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int key = k0 + 8 * NW * u;
            ok[u] = FULL || key < n_keys;
            const int kc = ok[u] ? key : kw;
            ku[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kb + (size_t)kc * DKV));
            vu[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(vb + (size_t)kc * DKV));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) bv[u] = bias[ok[u] ? n_keys - 1 - (k0 + 8 * NW * u) : 0];
        __builtin_amdgcn_sched_barrier(0);
    };
    auto compute_block = [&](u32x4 (&ku)[U], u32x4 (&vu)[U], bool (&ok)[U], float (&bv)[U]) {
        float sc[U];
        float mn = m;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float sv = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sv = fmaf(__uint_as_float(qp[j] << 16), __uint_as_float(ku[u][j] << 16), sv);
                sv = fmaf(__uint_as_float(qp[j] & 0xffff0000u), __uint_as_float(ku[u][j] & 0xffff0000u), sv);
            }
            sv = sum8(sv);
            sv += bv[u];
            sc[u] = sv;
            if (ok[u]) mn = fmaxf(mn, sv);
        }
        const float rescale = __expf(m - mn);
        l *= rescale;
#pragma unroll
        for (int d = 0; d < 8; ++d) acc[d] *= rescale;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float p = ok[u] ? __expf(sc[u] - mn) : 0.f;
            l += p;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[2 * j] = fmaf(p, __uint_as_float(vu[u][j] << 16), acc[2 * j]);
                acc[2 * j + 1] = fmaf(p, __uint_as_float(vu[u][j] & 0xffff0000u), acc[2 * j + 1]);
            }
        }
        m = mn;
    };
    {
        auto block = [&](int kw, auto full_tag) {
            u32x4 ku[U], vu[U];
            bool ok[U];
            float bv[U];
            load_block(ku, vu, ok, bv, kw, full_tag);
            compute_block(ku, vu, ok, bv);
        };
        for (int kw = wave * 8; kw < n_keys; kw += 8 * NW * U) {
            if (n_keys - kw >= 8 * NW * U) block(kw, std::true_type{});
            else block(kw, std::false_type{});
        }
    }
    // merge the 8 key groups of the wave, then the 8 waves through LDS (decode.hip, same order)
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) {
        const float mo = lane_xor(m, off), lo = lane_xor(l, off);
        const float mn = fmaxf(m, mo);
        const float sa = __expf(m - mn), sb = __expf(mo - mn);
        l = l * sa + lo * sb;
#pragma unroll
        for (int d = 0; d < 8; ++d) acc[d] = acc[d] * sa + lane_xor(acc[d], off) * sb;
        m = mn;
    }
    if (kg == 0) {
#pragma unroll
        for (int d = 0; d < 8; ++d) lds.sacc[wave * DKV + sub * 8 + d] = acc[d];
        if (sub == 0) { lds.sm[wave] = m; lds.sl[wave] = l; }
    }
    __syncthreads();
    if (tid < DKV) {
        float M = lds.sm[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) M = fmaxf(M, lds.sm[w]);
        float Lsum = 0.f, o = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float e = __expf(lds.sm[w] - M);
            Lsum += lds.sl[w] * e;
            o += lds.sacc[w * DKV + tid] * e;
        }
        lds.qs[tid] = f2bf(o / Lsum);
    }
    __syncthreads();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's own wo rows have landed
    {
        const int li = lane & 15, g = lane >> 4;
        const char* strip = lds.wo + wave * 8192;
        f32x4 pa[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) pa[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af = *reinterpret_cast<const bf16x8*>(lds.qs + ks * 32 + g * 8);
            if (li != 0) af = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int row = c * 16 + li;
                const bf16x8 wf = *reinterpret_cast<const bf16x8*>(strip + row * 128 + (((ks * 4 + g) ^ (row & 7)) * 16));
                pa[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af, pa[c], 0, 0, 0);
            }
        }
        if (li == 0) {       // read by the row's other heads, on other XCDs, later in this launch: agent scope
            const __amdgpu_buffer_rsrc_t ro = raw_rsrc(s.opart);
            const int off = (((r * H + h) * 512) + wave * 64 + g * 4) * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pa[c]), ro, off + c * 64, 0, AGENT);
        }
    }
}

// ---- cross-attention half: decode.hip attn_body<!SELF, FUSEQ, 8, OP, PAIR = 2> with T <= 256 * k keys; the residual row h comes in at agent
// scope (the previous layer's FFN-out stage wrote it in this launch), the 64 outputs leave as one agent-scope line.
__device__ __forceinline__ void cross_half(const StepArgs& s, const StepLayer& L, int r, int h, const AttnLds& lds) {
    constexpr int NW = 8, U = 4, H = 8;
    const int tid = YMT3_TID, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sub = lane & 7, kg = lane >> 3;
    const int n_keys = s.T;
    const size_t slab = ((size_t)r * H + h) * s.T * DKV;
    const bf16_t* kb = L.kcross + slab + sub * 8;
    const bf16_t* vb = L.vcross + slab + sub * 8;
    // operands of the fused projection: wave w owns outputs 8w..8w+7, lane l the k-chunk 8l..8l+7
    u32x4 wq_v[8];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj)
        wq_v[jj] = *reinterpret_cast<const u32x4*>(L.wq_c + ((size_t)h * DKV + wave * 8 + jj) * 512 + lane * 8);
    float x_v = ld_agent(s.h + (size_t)r * 512 + tid);
    const float g_v = L.ln2[tid];
    // The projection's own operands are in flight (10 loads per lane).  The self-attention half's partial stores are OLDER than these and
    // vector-memory operations retire in issue order, so a wait that leaves 10 outstanding has seen them acknowledged (decode.hip, PAIR == 2).
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(s.pair_rows + (size_t)(2 * r) * CHAIN_LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    row_wait(s.pair_rows, s.abort_word, s.host_abort, r);
    float pv[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) pv[w] = __hip_atomic_load(s.opart + ((size_t)r * H + w) * 512 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_sched_barrier(0);
    // first (for T <= 256: only) K/V block goes in flight now, straight-line code (decode.hip)
    u32x4 fku[U], fvu[U];
    bool fok[U];
    {
        const int k0 = wave * 8 + kg;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int key = k0 + 8 * NW * u;
            fok[u] = key < n_keys;
            const int kc = fok[u] ? key : 0;
            fku[u] = *reinterpret_cast<const u32x4*>(kb + (size_t)kc * DKV);
            fvu[u] = *reinterpret_cast<const u32x4*>(vb + (size_t)kc * DKV);
        }
    }
    // x = h + (p0 + p1 + ... + p7), then DG_RESID's sum(x^2) tree (mul_sep / add_sep: never an fma)
    {
        float sp = pv[0];
#pragma unroll
        for (int w = 1; w < 8; ++w) sp += pv[w];
        x_v += sp;
    }
    float q2 = mul_sep(x_v, x_v);
    q2 = add_sep(q2, DPP_F(q2, 0xB1));
    q2 = add_sep(q2, DPP_F(q2, 0x4E));
    q2 = add_sep(q2, DPP_F(q2, 0x141));
    q2 = add_sep(q2, DPP_F(q2, 0x128));
    if ((tid & 15) == 0) lds.s_tile[tid >> 4] = q2;
    __syncthreads();
    float ss = lane < SSQ_TILES ? lds.s_tile[lane] : 0.f;
    ss = wave_sum(ss);
    const float scale = rsqrtf(ss / 512.f + s.eps);
    lds.xs[tid] = bf2f(f2bf(x_v * scale * g_v));
    __syncthreads();
    float xn[8];
    *reinterpret_cast<float4*>(xn) = *reinterpret_cast<const float4*>(lds.xs + lane * 8);
    *reinterpret_cast<float4*>(xn + 4) = *reinterpret_cast<const float4*>(lds.xs + lane * 8 + 4);
    float dd[8];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            d = fmaf(xn[2 * j], __uint_as_float(wq_v[jj][j] << 16), d);
            d = fmaf(xn[2 * j + 1], __uint_as_float(wq_v[jj][j] & 0xffff0000u), d);
        }
        d = sum8(d);
        d += DPP_F(d, 0x128);
        dd[jj] = d;
    }
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) dd[jj] += DPP_ROWS(dd[jj], 0x142, 0xA);
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) dd[jj] += DPP_ROWS(dd[jj], 0x143, 0xC);
#pragma unroll
    for (int jj = 0; jj < 8; ++jj)
        if (lane == 56 + jj) lds.qs[wave * 8 + jj] = f2bf(dd[jj]);
    __syncthreads();
    const u32x4 qp = *reinterpret_cast<const u32x4*>(lds.qs + sub * 8);
    float m = -1.0e30f, l = 0.f, acc[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) acc[d] = 0.f;
    for (int kw = wave * 8; kw < n_keys; kw += 8 * NW * U) {
        if (kw != wave * 8) {
            const int k0 = kw + kg;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int key = k0 + 8 * NW * u;
                fok[u] = key < n_keys;
                const int kc = fok[u] ? key : kw;
                fku[u] = *reinterpret_cast<const u32x4*>(kb + (size_t)kc * DKV);
                fvu[u] = *reinterpret_cast<const u32x4*>(vb + (size_t)kc * DKV);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        float sc[U];
        float mn = m;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float sv = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sv = fmaf(__uint_as_float(qp[j] << 16), __uint_as_float(fku[u][j] << 16), sv);
                sv = fmaf(__uint_as_float(qp[j] & 0xffff0000u), __uint_as_float(fku[u][j] & 0xffff0000u), sv);
            }
            sv = sum8(sv);
            sc[u] = sv;
            if (fok[u]) mn = fmaxf(mn, sv);
        }
        const float rescale = __expf(m - mn);
        l *= rescale;
#pragma unroll
        for (int d = 0; d < 8; ++d) acc[d] *= rescale;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float p = fok[u] ? __expf(sc[u] - mn) : 0.f;
            l += p;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[2 * j] = fmaf(p, __uint_as_float(fvu[u][j] << 16), acc[2 * j]);
                acc[2 * j + 1] = fmaf(p, __uint_as_float(fvu[u][j] & 0xffff0000u), acc[2 * j + 1]);
            }
        }
        m = mn;
    }
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) {
        const float mo = lane_xor(m, off), lo = lane_xor(l, off);
        const float mn = fmaxf(m, mo);
        const float sa = __expf(m - mn), sb = __expf(mo - mn);
        l = l * sa + lo * sb;
#pragma unroll
        for (int d = 0; d < 8; ++d) acc[d] = acc[d] * sa + lane_xor(acc[d], off) * sb;
        m = mn;
    }
    if (kg == 0) {
#pragma unroll
        for (int d = 0; d < 8; ++d) lds.sacc[wave * DKV + sub * 8 + d] = acc[d];
        if (sub == 0) { lds.sm[wave] = m; lds.sl[wave] = l; }
    }
    __syncthreads();
    if (tid < DKV) {
        float M = lds.sm[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) M = fmaxf(M, lds.sm[w]);
        float Lsum = 0.f, o = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float e = __expf(lds.sm[w] - M);
            Lsum += lds.sl[w] * e;
            o += lds.sacc[w * DKV + tid] * e;
        }
        // the head's 64 outputs leave as ONE 128-byte line at agent scope (eight 16-byte stores of wave 0): the chain's O-projection stage reads
        // them from other XCDs later in this launch
        lds.qs[tid] = f2bf(o / Lsum);                   // (wave 0 only touches qs here; LDS executes a wave's accesses in order)
        if (tid < 8) {
            const u32x4 line = *reinterpret_cast<const u32x4*>(lds.qs + tid * 8);
            __builtin_amdgcn_raw_buffer_store_b128(line, raw_rsrc(s.attn), (((r * H + h) * DKV) + tid * 8) * 2, 0, AGENT);
        }
    }
}

template <int N_LAYERS_MAX>
__global__ __launch_bounds__(512, 4) void dec_step_kernel(const DecodeShared* __restrict__ pShared, const int* __restrict__ pRowPos, int R, int n_layers,
                                                          StepArgs s_in) {
    // leading scalar arguments: kernarg preload (the position and the geometry every workgroup needs first)
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const AttnLds lds = attn_lds(smem);
    const int b = blockIdx.x;
    const int r = b >> 3, h = b & 7;                    // attention role: (row, head)
    const int tile = b >> 7;                            // this workgroup's row tile, in both roles
    // Chain role: 64 of the tile's 128 workgroups.  Blocks are dealt round-robin over the 8 XCDs and, inside an XCD, over its 32 CUs before a
    // CU gets its second workgroup (observed placement, speed only): block b sits on CU (b >> 3) & 31 of XCD b & 7, so rows tiles 0 / 2 share
    // CUs and so do 1 / 3.  Taking the even (b >> 3) of tiles 0 and 1 and the odd ones of tiles 2 and 3 puts every chain tile on a CU of its
    // own (its 147 KB of weights are what a chain stage waits for); tiles with one column tile stay on one XCD, as in dec_chain_kernel.
    const int k16 = (b >> 3) & 15;                      // position among the tile's 16 blocks of this XCD
    const bool chain_role = (k16 & 1) == (tile >> 1);
    const int local = ((k16 >> 1) << 3) | (b & 7);      // the chain's column tile, 0..63
    const int n_mt = (R + 15) >> 4;
    if (tile >= n_mt) return;                           // (whole row tiles beyond R: nothing waits for them)
    const bool has_attn = r < R;
    const int rows_in_tile = min(16, R - 16 * tile);
    const int n_keys = has_attn ? (pRowPos ? pRowPos[r] : pShared->step) + 1 : 0;
    // chain tile index in dec_chain_kernel's encoding: row tile = (t >> 3) & 3, column tile = (t >> 5) * 8 + (t & 7)
    const int t_chain = ((local >> 3) << 5) | (tile << 3) | (local & 7);
    // The argument struct is read from the kernel-argument segment itself, per layer, through a pointer the optimiser cannot see through:
    // otherwise every field of every phase is loaded (and the address arithmetic on it done) before the loop and spilled.
    typedef const __attribute__((address_space(4))) char* kernarg_ptr;
    static_assert(alignof(StepArgs) == 8 && offsetof(StepArgs, layer) % 8 == 0 && sizeof(StepLayer) % 8 == 0, "kernel-argument layout");
    constexpr int ARGS_AT = 24;                          // (pShared, pRowPos, R, n_layers) come first: 8 + 8 + 4 + 4 bytes
    const kernarg_ptr ka = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    (void)s_in;
    for (int l = 0; l < n_layers; ++l) {
        kernarg_ptr kp = ka;
        asm volatile("" : "+s"(kp));
        StepArgs s;                                      // (only the common fields are copied; `layer` stays in the segment)
        {
            typedef const __attribute__((address_space(4))) unsigned long long* kq;
            unsigned long long* dst = reinterpret_cast<unsigned long long*>(&s);
            kq src = (kq)(kp + ARGS_AT);
#pragma unroll
            for (int i = 0; i < (int)(offsetof(StepArgs, layer) / 8); ++i) dst[i] = src[i];
        }
        StepLayer L;
        {
            typedef const __attribute__((address_space(4))) unsigned long long* kq;
            unsigned long long* dst = reinterpret_cast<unsigned long long*>(&L);
            kq src = (kq)(kp + ARGS_AT + offsetof(StepArgs, layer) + (size_t)l * sizeof(StepLayer));
#pragma unroll
            for (int i = 0; i < (int)(sizeof(StepLayer) / 8); ++i) dst[i] = src[i];
        }
        globalize(s);
        globalize(L);
        if (s.stamp && l == 0 && YMT3_TID == 0) s.stamp[2 * b] = wall_clock64();
        unsigned* sync_l = s.sync + (size_t)l * STEP_SYNC_LINES_PER_LAYER * CHAIN_LINE;
        // tiles_free: every row tile has counters of its own.  Else all tiles arrive at tile 0's lines and wait for everybody: the four tiles
        // move through the phases in step, as a sequence of launches would, and a phase's hand-offs are not slowed by another tile's streams
        unsigned* attn_done = sync_l + (size_t)(CHAIN_COUNTERS + (s.tiles_free ? tile * 8 : 0)) * CHAIN_LINE;    // 8 replicas per tile, or all 32 lines for everybody
        if (has_attn) {
            self_half(s, L, r, h, n_keys, lds, l == 0 ? nullptr : sync_l + (size_t)(CHAIN_COUNTERS + 32 + tile * 8 + h) * CHAIN_LINE, s.tiles_free ? 6u : 6u * n_mt,
                      s.stamp ? s.stamp + STEP_MARK0 + STEP_MARKS * b + 4 * l : nullptr);
            // the second half counts its loads against the first half's stores: nothing may cross this line
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            cross_half(s, L, r, h, lds);
            counter_signal(attn_done, s.tiles_free ? 8 : 32, CHAIN_LINE);       // (waits for the output line's acknowledgement, then a barrier: the LDS block is free)
            if (s.stamp && YMT3_TID == 0) s.stamp[STEP_MARK0 + STEP_MARKS * b + 4 * l + 1] = wall_clock64();          // mark: this layer's attention left
        }
        if (chain_role) {
            ChainArgs c{};
            c.part = s.opart; c.ssq = s.ssq; c.ssq_stride = s.ssq_stride; c.gain1 = L.ln3; c.gain3 = L.gain3; c.dff = s.dff; c.d_ff = 2048;
            c.N3 = L.N3; c.out_q = s.q; c.kcache = L.knext; c.vcache = L.vnext; c.logits = s.logits; c.H = 8; c.L = s.L;
            c.shared = pShared; c.row_pos = pRowPos; c.row0 = 0; c.R = R; c.eps = s.eps;
            c.sync = sync_l; c.host_abort = s.host_abort; c.sync_abort = s.abort_word; c.stamp = nullptr;
            ChainInLaunch in;
            in.attn_done = attn_done + (size_t)(s.tiles_free ? (b & 7) : ((b >> 3) & 31)) * CHAIN_LINE;
            in.attn_target = (unsigned)((s.tiles_free ? rows_in_tile : R) * 8);
            in.tiles_free = s.tiles_free;
            in.qkv_done = sync_l + (size_t)(STEP_SYNC_LINES_PER_LAYER + CHAIN_COUNTERS + 32) * CHAIN_LINE;      // the NEXT layer's lines
            in.mark = s.stamp ? s.stamp + STEP_MARK0 + STEP_MARKS * b + 4 * l + 2 : nullptr;
            if (L.last) chain_stages<DG_NORM_LOGITS, true, true>(L.wo_c, L.wi, L.wo2, L.w3, s.attn, s.h, 0, R, c, smem, t_chain, in);
            else chain_stages<DG_NORM_QKV_CACHE, true, true>(L.wo_c, L.wi, L.wo2, L.w3, s.attn, s.h, 0, R, c, smem, t_chain, in);
            __syncthreads();                                 // the LDS block goes back to the attention halves
            if (s.stamp && YMT3_TID == 0) s.stamp[STEP_MARK0 + STEP_MARKS * b + 4 * l + 3] = wall_clock64();      // mark: this layer's chain tile done
        }
        if (s.stamp && l + 1 == n_layers && (YMT3_TID & 63) == 0) atomicMax(s.stamp + 2 * b + 1, (unsigned long long)wall_clock64());
    }
}

}  // namespace

int init_step_kernel() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(dec_step_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, STEP_LDS) == hipSuccess ? 0 : -2;
}

// all 512 workgroups wait for each other: two per CU at the kernel's LDS / register footprint
bool dec_step_fits(int n_cus) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, dec_step_kernel<8>, 512, STEP_LDS) != hipSuccess) return false;
    return (long long)n * n_cus >= 512;
}

// 0 = launched; negative = not this kernel's shape
int launch_dec_step(const StepArgs& s, hipStream_t stream) {
    if (s.R <= 0) return 0;
    if (s.R > 64 || s.n_layers < 1 || s.n_layers > 8 || !s.sync || !s.pair_rows || !s.abort_word || s.T < 1 || s.T > 0xfff || s.L < 1) return -1;
    for (int l = 0; l < s.n_layers; ++l)
        if (s.layer[l].N3 % 32 || s.layer[l].N3 / 32 < 32 || s.layer[l].N3 / 32 > 64) return -1;
    const int grid = ((s.R + 15) / 16) * 128;
    dec_step_kernel<8><<<grid, 512, STEP_LDS, stream>>>(s.shared, s.row_pos, s.R, s.n_layers, s);
    return 0;
}
