// Device code of the decoder step's skinny-GEMM chain (dec_chain.hip: the chain as its own launch; dec_step.hip: inside the per-step
// kernel).  Included inside an anonymous namespace by both translation units.
#pragma once

// The thread index as the including file wants it read: dec_step.hip runs this code inside a loop over the decoder layers and makes every read
// opaque to the optimiser, so that nothing derived from it is hoisted out of that loop (and spilled: the loop body needs all 128 VGPRs).
#ifndef YMT3_TID
#define YMT3_TID threadIdx.x
#endif
// What a poll loop does between two looks at its counter: nothing in the launch form of the chain (its workgroups have nothing else on their
// CU); dec_step.hip sleeps a little, because its pollers share CUs and the memory side with workgroups that stream K/V
#ifndef YMT3_POLL_PAUSE
#define YMT3_POLL_PAUSE do { } while (0)
#endif

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int DKV = 64;
constexpr int AGENT = 16;                       // buffer-instruction cache policy bit 4 = sc1 (agent scope) on gfx94x / gfx950
constexpr unsigned long long SPIN_LIMIT = 100000000ull;   // 1 s of the 100 MHz wall clock (a stage hands over in ~1 us; this only has to outlast a time-sliced GPU)

#define CH_STAMP_IN(c) do { if ((c).stamp && YMT3_TID == 0) (c).stamp[2 * blockIdx.x] = wall_clock64(); } while (0)
#define CH_MARK(c, k) do { if ((c).stamp && YMT3_TID == 0) (c).stamp[512 + 8 * blockIdx.x + (k)] = wall_clock64(); } while (0)   // measurement: stage marks
#define CH_STAMP_OUT(c) do { if ((c).stamp && (YMT3_TID & 63) == 0) atomicMax((c).stamp + 2 * blockIdx.x + 1, (unsigned long long)wall_clock64()); } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t raw_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ u32x4 ld16_agent(__amdgpu_buffer_rsrc_t r, int byte_off) { return __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, AGENT); }
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st2_agent(float* p, float2 v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int K, int NT>
struct Geo {
    static constexpr int KW = K / 8, KS = KW / 32, COLS = 16 * NT, PITCH = KW * 2 + 16, STRIP = 16 * PITCH;
    static constexpr int LPRW = KW * 2 / 16, RPIW = 64 / LPRW, NIA = 16 / RPIW, NIW = NIA * NT;
    static constexpr int LPRX = KW * 4 / 16, RPIX = 64 / LPRX, NIX = 16 / RPIX;
};

// this wave's K-slice of the tile's COLS weight rows: whole-line coalesced 16-byte loads (dec_gemm_kernel's operand path)
template <int K, int NT>
__device__ __forceinline__ void load_w(const bf16_t* W, int n0, u32x4 (&wv)[Geo<K, NT>::NIW]) {
    using G = Geo<K, NT>;
    const int lane = YMT3_TID & 63, wave = YMT3_TID >> 6;
#pragma unroll
    for (int i = 0; i < G::NIW; ++i)
        wv[i] = *reinterpret_cast<const u32x4*>(W + (size_t)(n0 + i * G::RPIW + lane / G::LPRW) * K + wave * G::KW + (lane % G::LPRW) * 8);
}

// The same K-slice of the tile's 16 weight rows in MFMA fragment order, straight into registers: lane (li, g) holds, per 32-wide k-step, the 8 bf16 of
// row n0 + li at k = wave * KW + ks * 32 + g * 8 -- exactly what `sW + li * PITCH + (ks * 32 + g * 8) * 2` holds after load_w + the strip store, so the
// MFMA operands (and the results) are the same bits.  Fragment-shaped loads are 3-4x slower per byte than whole-line ones (profiles/r01_notes.md);
// the chain requests them at entry, two to three stages before their use.  What this buys: stage 2 (K = 2048) needs no weight strip in LDS --
// 76 KB instead of 143 KB per workgroup, two workgroups per CU.
template <int K>
__device__ __forceinline__ void load_w_frag(const bf16_t* W, int n0, u32x4 (&wf)[Geo<K, 1>::KS]) {
    using G = Geo<K, 1>;
    const int lane = YMT3_TID & 63, wave = YMT3_TID >> 6, li = lane & 15, g = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks)
        wf[ks] = *reinterpret_cast<const u32x4*>(W + (size_t)(n0 + li) * K + wave * G::KW + ks * 32 + g * 8);
}

// Arrival counters.  Requests to ONE line are served one after the other at the memory side (~15 ns each, profiles/r01_barrier_probe.txt),
// so 64 workgroups polling the counter their 32-64 producers add to would see an arrival ~7 us late (measured: the first version of
// this kernel).  Hence: every counter on its own 128-byte line, eight replicas per (boundary, row tile) -- a producer adds to all
// eight (eight lanes, one instruction), a consumer polls the replica of its own XCD slot (8 pollers per line) -- and the poll loop
// touches nothing else: the clock is read every 64 polls and the abort word only then.
__device__ __forceinline__ unsigned* chain_counter(unsigned* sync, int boundary, int mt, int replica) {
    return sync + ((boundary * CHAIN_TILES_MAX + mt) * 8 + replica) * CHAIN_LINE;
}
// With <= 4 row tiles (the merged regime: up to 64 rows) a (boundary, row tile) counter is split in FOUR by the producer's column tile (nt & 3), at
// tile slots mt + 4 sub: adds to one line are served one after the other, and a stage's 32-64 producers finish together -- a quarter of them per
// line makes the last arrival visible sooner.  The consumer polls its replica of all four with four lanes of one instruction.
__device__ __forceinline__ void chain_wait(unsigned* sync, int boundary, int mt, unsigned target, unsigned* host_abort, unsigned* abort_at = nullptr, int nsub = 1) {
    if (YMT3_TID < 64) {
        const int lane = YMT3_TID;
        const unsigned* cnt = chain_counter(sync, boundary, mt + 4 * (lane < nsub ? lane : 0), blockIdx.x & 7);
        const unsigned each = target / (unsigned)nsub;
        unsigned* abort_word = abort_at ? abort_at : sync + CHAIN_ABORT_WORD;      // (dec_step.hip keeps one abort word for its per-layer counter sets)
        unsigned long long t0 = 0;
        unsigned polls = 0;
        for (;;) {
            unsigned v = each;
            if (lane < nsub) v = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__ballot(v < each) == 0ull) break;
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
// the same wait on an explicit counter line (dec_step.hip's attention -> chain and QKV -> attention hand-offs)
__device__ __forceinline__ void counter_wait(const unsigned* cnt, unsigned target, unsigned* abort_word, unsigned* host_abort) {
    if (YMT3_TID == 0) {
        unsigned long long t0 = 0;
        unsigned polls = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            YMT3_POLL_PAUSE;
            if ((++polls & 63u) == 0u) {
                const unsigned long long now = wall_clock64();
                if (t0 == 0) t0 = now;
                if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || now - t0 > SPIN_LIMIT) {
                    __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (host_abort) __hip_atomic_store(host_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
            }
        }
    }
    __syncthreads();
}
// every thread's agent-scope stores acknowledged -> workgroup barrier -> `n` lanes add 1 to `n` counter lines (replicas) `stride` words apart
__device__ __forceinline__ void counter_signal(unsigned* cnt, int n, int stride) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if ((int)YMT3_TID < n) __hip_atomic_fetch_add(cnt + YMT3_TID * stride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void chain_signal(unsigned* sync, int boundary, int mt, int sub = 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this thread's agent-scope stores have been acknowledged
    __syncthreads();
    if (YMT3_TID < 8) __hip_atomic_fetch_add(chain_counter(sync, boundary, mt + 4 * sub, YMT3_TID), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// strips -> MFMA -> fixed-order cross-wave reduction; returns this thread's two outputs (dec_gemm_kernel, from "fragment order read-back")
template <int K, int NT>
__device__ __forceinline__ float2 mfma_reduce(char* sA, char* sW, float* red) {
    using G = Geo<K, NT>;
    const int tid = YMT3_TID, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
    f32x4 acc[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
        const int off = li * G::PITCH + (ks * 32 + g * 8) * 2;
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(sA + off);
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            const bf16x8 wf = *reinterpret_cast<const bf16x8*>(sW + c * G::STRIP + off);
            acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af, acc[c], 0, 0, 0);
        }
    }
#pragma unroll
    for (int c = 0; c < NT; ++c)
        *reinterpret_cast<float4*>(red + ((wave * 16 + li) * G::COLS + c * 16 + g * 4)) = make_float4(acc[c][0], acc[c][1], acc[c][2], acc[c][3]);
    __syncthreads();
    float2 s = make_float2(0.f, 0.f);
    if (tid < 16 * 8 * NT) {
        const int mr = tid / (8 * NT), nq = (tid % (8 * NT)) * 2;
        s = *reinterpret_cast<const float2*>(red + (mr * G::COLS + nq));
#pragma unroll
        for (int w = 1; w < 8; ++w) {
            const float2 t = *reinterpret_cast<const float2*>(red + ((w * 16 + mr) * G::COLS + nq));
            s.x += t.x; s.y += t.y;
        }
    }
    return s;
}

// the same with the weight fragments already in registers (load_w_frag): NT = 1
template <int K>
__device__ __forceinline__ float2 mfma_reduce_wfrag(char* sA, const u32x4 (&wfr)[Geo<K, 1>::KS], float* red) {
    using G = Geo<K, 1>;
    const int tid = YMT3_TID, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(sA + li * G::PITCH + (ks * 32 + g * 8) * 2);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wfr[ks]), af, acc, 0, 0, 0);
    }
    *reinterpret_cast<float4*>(red + ((wave * 16 + li) * 16 + g * 4)) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    __syncthreads();
    float2 s = make_float2(0.f, 0.f);
    if (tid < 16 * 8) {
        const int mr = tid / 8, nq = (tid % 8) * 2;
        s = *reinterpret_cast<const float2*>(red + (mr * 16 + nq));
#pragma unroll
        for (int w = 1; w < 8; ++w) {
            const float2 t = *reinterpret_cast<const float2*>(red + ((w * 16 + mr) * 16 + nq));
            s.x += t.x; s.y += t.y;
        }
    }
    return s;
}

// the RESID epilogue: h tile and its sum(h^2) partial out at agent scope (the next stage reads them from another XCD)
__device__ __forceinline__ float2 resid_out(float2 hold, float2 s, float* pH, float* pSsq, int ssq_stride, int m, int n, int nt_idx, bool live) {
    float2 o = make_float2(0.f, 0.f);
    if (live) {
        o = make_float2(hold.x + s.x, hold.y + s.y);
        st2_agent(pH + (size_t)m * 512 + n, o);
    }
    // two products and a sum, as dec_gemm_kernel's epilogue compiles (v_pk_mul_f32, v_add_f32): hipcc would contract this copy into an fma
    float q = add_sep(mul_sep(o.x, o.x), mul_sep(o.y, o.y));
    q = sum8(q);
    if (live && (YMT3_TID & 7) == 0) st_agent(pSsq + (size_t)nt_idx * ssq_stride + m, q);
    return o;
}

// a NORM-mode tile (K = 512, 32 columns): x rows and sum(h^2) partials in at agent scope, weights already in registers
// What a NORM-mode tile can do BEFORE the stage in front of it has finished: park its weight slices in the LDS strips (the caller has
// just passed a workgroup barrier: nobody reads the previous stage's LDS any more).  Its slice of the norm gain is fetched at kernel
// entry with the weights: a load issued here would sit in front of the poll loop's loads, which return in issue order.
__device__ __forceinline__ f32x4 norm_gain(const float* gain) {
    using G = Geo<512, 2>;
    return *reinterpret_cast<const f32x4*>(gain + (YMT3_TID >> 6) * G::KW + ((YMT3_TID & 63) % G::LPRX) * 4);
}
template <int NT>
__device__ __forceinline__ void norm_park(char* smem, u32x4 (&wv)[Geo<512, NT>::NIW]) {   // (hipcc mis-parses the array reference as a first parameter)
    using G = Geo<512, NT>;
    const int lane = YMT3_TID & 63, wave = YMT3_TID >> 6;
    char* sW = smem + (8 * 16 * G::COLS + 16) * 4 + wave * (1 + NT) * G::STRIP + G::STRIP;
#pragma unroll
    for (int i = 0; i < G::NIW; ++i)
        *reinterpret_cast<u32x4*>(sW + (i * G::RPIW + lane / G::LPRW) * G::PITCH + (lane % G::LPRW) * 16) = wv[i];
}

// INL (dec_step.hip): the QKV outputs are consumed later in the SAME launch, by attention workgroups on other XCDs: agent-scope stores
template <int MODE, bool INL = false>
__device__ __forceinline__ void norm_tile(const ChainArgs& c, const float* pH, int row0, int R, const f32x4 gv, int N, int nt_idx, int mt_idx,
                                          int step, char* smem) {
    using G = Geo<512, 2>;
    constexpr int NT = 2;
    float* red = reinterpret_cast<float*>(smem);
    float* sscale = red + 8 * 16 * G::COLS;
    char* strips = smem + (8 * 16 * G::COLS + 16) * 4;
    const int tid = YMT3_TID, lane = tid & 63, wave = tid >> 6;
    const int n0 = nt_idx * G::COLS, m0 = row0 + mt_idx * 16, m_end = row0 + R;
    char* sA = strips + wave * (1 + NT) * G::STRIP;
    char* sW = sA + G::STRIP;
    const __amdgpu_buffer_rsrc_t rx = raw_rsrc(pH);
    f32x4 xv[G::NIX];
#pragma unroll
    for (int i = 0; i < G::NIX; ++i) {
        int mm = m0 + i * G::RPIX + lane / G::LPRX;
        mm = mm < m_end ? mm : m_end - 1;
        xv[i] = __builtin_bit_cast(f32x4, ld16_agent(rx, (mm * 512 + wave * G::KW + (lane % G::LPRX) * 4) * 4));
    }
    float ss = 0.f;
    if (tid < 16 * 8) {
        const int mm = m0 + (tid >> 3) < m_end ? m0 + (tid >> 3) : m_end - 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) ss += ld_agent(c.ssq + (size_t)((tid & 7) * 4 + j) * c.ssq_stride + mm);
    }
    ss = sum8(ss);
    if (tid < 16 * 8 && (tid & 7) == 0) sscale[tid >> 3] = rsqrtf(ss / 512.f + c.eps);
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
        *reinterpret_cast<float2*>(c.logits + (size_t)m * N + n) = s;
    } else if constexpr (MODE == DG_NORM_BF16_RELU) {
        s.x = fmaxf(s.x, 0.f); s.y = fmaxf(s.y, 0.f);
        st_agent(reinterpret_cast<uint32_t*>(c.dff + (size_t)m * N + n), pack_bf16x2(s.x, s.y));     // stage 2 reads it from other XCDs
    } else {
        const uint32_t pk = pack_bf16x2(s.x, s.y);
        const int inner = c.H * DKV;
        if (n < inner) {
            if constexpr (INL) st_agent(reinterpret_cast<uint32_t*>(c.out_q + (size_t)m * inner + n), pk);
            else *reinterpret_cast<uint32_t*>(c.out_q + (size_t)m * inner + n) = pk;
        } else {
            const int nn = n - inner, kv = nn / inner, hh = (nn % inner) >> 6, dd = nn & 63;
            bf16_t* cache = kv ? c.vcache : c.kcache;
            if constexpr (INL) st_agent(reinterpret_cast<uint32_t*>(cache + (((size_t)m * c.H + hh) * c.L + step) * DKV + dd), pk);
            else *reinterpret_cast<uint32_t*>(cache + (((size_t)m * c.H + hh) * c.L + step) * DKV + dd) = pk;
        }
    }
}

// The chain's four stages for workgroup-tile `t` (0..255).  W2F: stage 2's weights as register-resident MFMA fragments (load_w_frag) instead of a
// 67 KB LDS strip -- 76 KB of LDS per workgroup instead of 143 KB, same bits.
// INL (dec_step.hip: the chain inside the per-step kernel): its inputs -- the cross-attention output, the O-projection partials, the residual
// stream -- were written earlier in the SAME launch, so they are read at agent scope, and only after the row tile's attention workgroups have
// all arrived (`in.attn_done`, `in.attn_target` arrivals); the weights are requested before that wait.  The QKV stage stores at agent scope and
// signals `in.qkv_done` (one line per (row tile, head): six 32-column tiles each) for the next layer's attention.
struct ChainInLaunch {
    const unsigned* attn_done;      // this workgroup's replica of its row tile's arrival counter
    unsigned attn_target;
    unsigned* qkv_done;             // [4 row tiles][8 heads] lines of CHAIN_LINE words; null after the last layer
    unsigned long long* mark;       // measurement or null: [0] the row tile's attention has arrived
    int tiles_free;                 // 0: every row tile signals the lines of row tile 0 (dec_step.hip: the tiles move in step)
};
template <int MODE3, bool W2F, bool INL = false>
__device__ __forceinline__ void chain_stages(const bf16_t* __restrict__ pW0, const bf16_t* __restrict__ pW1, const bf16_t* __restrict__ pW2,
                                             const bf16_t* __restrict__ pW3, const bf16_t* __restrict__ pAttn, float* pH, int row0, int R,
                                             const ChainArgs& c, char* smem, int t, const ChainInLaunch in = ChainInLaunch{}) {
    using G0 = Geo<512, 1>;
    using G1 = Geo<512, 2>;
    using G2 = Geo<2048, 1>;
    const int tid = YMT3_TID, lane = tid & 63, wave = tid >> 6;
    const int n_mt = (R + 15) / 16, m_end = row0 + R;
    // counters split by the producer's column tile (chain_wait), per boundary
    const int nsub_all = c.nsub ? c.nsub : 0x444;
    const int nsub0 = n_mt <= 4 ? nsub_all & 15 : 1, nsub1 = n_mt <= 4 ? (nsub_all >> 4) & 15 : 1, nsub2 = n_mt <= 4 ? (nsub_all >> 8) & 15 : 1;
    // workgroup -> tile, the same in every stage, no divisions in front of the first loads: XCD = t & 7 (round-robin placement; speed
    // only), row tile = (t >> 3) & 3, column tile = (t >> 5) * 8 + XCD -- the workgroups that share a weight tile sit on one XCD, as
    // in dec_gemm_kernel.  A workgroup without a tile in a stage (row tile beyond R, column tile beyond the stage's N) still requests
    // (and drops) a valid tile's weights: the loads below stay straight-line code, so hipcc's wait counts for stage 0's operands do
    // not fall back to "everything outstanding".
    // More than four row tiles (65..256 rows): row-tile-major, workgroups 64 mt .. 64 mt + 63 are row tile mt.  Such a grid need NOT be resident
    // as a whole: a tile's 64 workgroups wait only for each other and are consecutive in dispatch order, so with workgroups dispatched in
    // index order (per XCD) every row tile below the slowest XCD's dispatch frontier is complete, finishes and frees its slots (observed
    // order, not a HIP promise: every wait is bounded and runtime.hip recovers through the separate launches).
    const int mt = n_mt <= 4 ? (t >> 3) & 3 : t >> 6, nt = n_mt <= 4 ? (t >> 5) * 8 + (t & 7) : t & 63;             // nt in [0, 64)
    const int n_nt3 = c.N3 / 32;                                           // 32..64 (launcher)
    const bool has0 = mt < n_mt && nt < 32, has1 = mt < n_mt, has3 = mt < n_mt && nt < n_nt3;
    const int nt0 = nt & 31, mt0 = mt, nt1 = nt, mt1 = mt, nt3 = nt < n_nt3 ? nt : nt - n_nt3, mt3 = mt;

    // ---- stage 0 operands first (they are this kernel's critical path), then stage 1's weights; stages 2 and 3 request theirs below
    u32x4 w0[G0::NIW], av[G0::NIA];
    load_w<512, 1>(pW0, nt0 * 16, w0);
    const int m0 = row0 + mt0 * 16;
    const bool epi0 = tid < 16 * 8;
    const int mr0 = tid / 8, nq0 = (tid % 8) * 2;
    const int mE = m0 + mr0, nE = nt0 * 16 + nq0;
    const bool live0 = has0 && epi0 && mE < m_end;
    float2 hold = make_float2(0.f, 0.f);
    float2 hp[8];
    if constexpr (!INL) {
#pragma unroll
        for (int i = 0; i < G0::NIA; ++i) {
            int mm = m0 + i * G0::RPIW + lane / G0::LPRW;
            mm = mm < m_end ? mm : m_end - 1;
            av[i] = *reinterpret_cast<const u32x4*>(pAttn + (size_t)mm * 512 + wave * G0::KW + (lane % G0::LPRW) * 8);
        }
        if (live0) hold = *reinterpret_cast<const float2*>(pH + (size_t)mE * 512 + nE);
        if (c.part && live0) {
#pragma unroll
            for (int w = 0; w < 8; ++w) hp[w] = *reinterpret_cast<const float2*>(c.part + ((size_t)mE * 8 + w) * 512 + nE);
        }
    }
    int step = 0;
    if constexpr (MODE3 == DG_NORM_QKV_CACHE) {
        const int mq = row0 + mt3 * 16 + tid / 16;
        step = c.row_pos ? c.row_pos[mq < m_end ? mq : m_end - 1] : c.shared->step;
    }
    if constexpr (INL) {
        // Stage 0's weights are in flight; now the row tile's attention must be complete, then its outputs come in at agent scope, and only
        // then the later stages' weights are requested (as in the launch form: stage 0's operands first).  (Requesting all weights before the
        // wait put 147 KB per workgroup in front of the poll -- vector-memory data returns in issue order -- on the critical path of the
        // workgroup whose own attention half was the tile's last.)
        if (mt < n_mt) counter_wait(in.attn_done, in.attn_target, c.sync_abort, c.host_abort);
        if (in.mark && YMT3_TID == 0) in.mark[0] = wall_clock64();
        if (has0) {
            const __amdgpu_buffer_rsrc_t rat = raw_rsrc(pAttn);
#pragma unroll
            for (int i = 0; i < G0::NIA; ++i) {
                int mm = m0 + i * G0::RPIW + lane / G0::LPRW;
                mm = mm < m_end ? mm : m_end - 1;
                av[i] = ld16_agent(rat, (mm * 512 + wave * G0::KW + (lane % G0::LPRW) * 8) * 2);
            }
            if (live0) {
                hold = __builtin_bit_cast(float2, __hip_atomic_load(reinterpret_cast<const unsigned long long*>(pH + (size_t)mE * 512 + nE), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#pragma unroll
                for (int w = 0; w < 8; ++w)
                    hp[w] = __builtin_bit_cast(float2, __hip_atomic_load(reinterpret_cast<const unsigned long long*>(c.part + ((size_t)mE * 8 + w) * 512 + nE),
                                                                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    u32x4 w1[G1::NIW], w2[W2F ? G2::KS : G2::NIW], w3[G1::NIW];
    load_w<512, 2>(pW1, nt1 * 32, w1);
    const f32x4 g1 = norm_gain(c.gain1);
    __builtin_amdgcn_sched_barrier(0);
    CH_STAMP_IN(c);

    float2 o0 = make_float2(0.f, 0.f);
    if (has0) {
        float* red = reinterpret_cast<float*>(smem);
        char* strips = smem + (8 * 16 * 16 + 16) * 4;
        char* sA = strips + wave * 2 * G0::STRIP;
        char* sW = sA + G0::STRIP;
#pragma unroll
        for (int i = 0; i < G0::NIW; ++i) {
            const int off = (i * G0::RPIW + lane / G0::LPRW) * G0::PITCH + (lane % G0::LPRW) * 16;
            *reinterpret_cast<u32x4*>(sW + off) = w0[i];
            *reinterpret_cast<u32x4*>(sA + off) = av[i];
        }
        const float2 s = mfma_reduce<512, 1>(sA, sW, red);
        if (c.part && live0) {                   // h + (p0 + ... + p7): what the separate O-projection launch left in h
            float2 sp = hp[0];
#pragma unroll
            for (int w = 1; w < 8; ++w) { sp.x += hp[w].x; sp.y += hp[w].y; }
            hold.x += sp.x; hold.y += sp.y;
        }
        o0 = resid_out(hold, s, pH, c.ssq, c.ssq_stride, mE, nE, nt0, live0);
        CH_MARK(c, 0);
        chain_signal(c.sync, 0, mt0, nt0 & (nsub0 - 1));
        CH_MARK(c, 1);
    }
    // The later stages' weights are requested as the stages come up, not all at entry as in round 2 (stage 2's here, stage 3's behind stage 1): a load
    // instruction waits for room in the CU's vector-memory queue, and 112 KB of weight requests per workgroup at entry sat in front of the first
    // stages' own loads and polls -- 240.0 -> 234.9 ms per batch, same box, two builds interleaved (profiles/r03_notes.md); each still has a stage's
    // worth of time (2.5-4 us) to arrive
    if constexpr (W2F) load_w_frag<2048>(pW2, nt0 * 16, w2);
    else load_w<2048, 1>(pW2, nt0 * 16, w2);
    __builtin_amdgcn_sched_barrier(0);
    // ---- stage 1: FFN-in
    if (has1) {
        norm_park<2>(smem, w1);
        chain_wait(c.sync, 0, mt1, 32u, c.host_abort, c.sync_abort, nsub0);
        CH_MARK(c, 2);
        norm_tile<DG_NORM_BF16_RELU>(c, pH, row0, R, g1, c.d_ff, nt1, mt1, 0, smem);
        CH_MARK(c, 3);
        chain_signal(c.sync, 1, mt1, nt1 & (nsub1 - 1));
        CH_MARK(c, 4);
    }
    load_w<512, 2>(pW3, nt3 * 32, w3);
    const f32x4 g3 = norm_gain(c.gain3);
    __builtin_amdgcn_sched_barrier(0);
    // ---- stage 2: FFN-out (the tile of stage 0 again: its h values are still in registers)
    if (has0) {
        float* red = reinterpret_cast<float*>(smem);
        char* strips = smem + (8 * 16 * 16 + 16) * 4;
        char* sA = strips + wave * (W2F ? 1 : 2) * G2::STRIP;
        if constexpr (!W2F) {
            char* sW = sA + G2::STRIP;
#pragma unroll
            for (int i = 0; i < G2::NIW; ++i)          // the weight slices are parked while the FFN-in stage is still running
                *reinterpret_cast<u32x4*>(sW + (i * G2::RPIW + lane / G2::LPRW) * G2::PITCH + (lane % G2::LPRW) * 16) = w2[i];
        }
        chain_wait(c.sync, 1, mt0, (unsigned)(c.d_ff / 32), c.host_abort, c.sync_abort, nsub1);
        CH_MARK(c, 5);
        const __amdgpu_buffer_rsrc_t ra = raw_rsrc(c.dff);
        u32x4 dv[G2::NIA];
#pragma unroll
        for (int i = 0; i < G2::NIA; ++i) {
            int mm = m0 + i * G2::RPIW + lane / G2::LPRW;
            mm = mm < m_end ? mm : m_end - 1;
            dv[i] = ld16_agent(ra, (mm * 2048 + wave * G2::KW + (lane % G2::LPRW) * 8) * 2);
        }
#pragma unroll
        for (int i = 0; i < G2::NIA; ++i)
            *reinterpret_cast<u32x4*>(sA + (i * G2::RPIW + lane / G2::LPRW) * G2::PITCH + (lane % G2::LPRW) * 16) = dv[i];
        float2 s;
        if constexpr (W2F) s = mfma_reduce_wfrag<2048>(sA, w2, red);
        else s = mfma_reduce<2048, 1>(sA, sA + G2::STRIP, red);
        resid_out(o0, s, pH, c.ssq, c.ssq_stride, mE, nE, nt0, live0);
        CH_MARK(c, 6);
        chain_signal(c.sync, 2, mt0, nt0 & (nsub2 - 1));
    }
    // ---- stage 3: the next layer's QKV projection, or lm_head
    if (has3) {
        norm_park<2>(smem, w3);
        chain_wait(c.sync, 2, mt3, 32u, c.host_abort, c.sync_abort, nsub2);
        CH_MARK(c, 7);
        norm_tile<MODE3, INL>(c, pH, row0, R, g3, c.N3, nt3, mt3, step, smem);
    }
    if constexpr (INL && MODE3 == DG_NORM_QKV_CACHE) {
        // the tile's q / k / v columns are one head's: 32-column tiles 2h, 2h + 1 of each of the three 512-column blocks
        if (mt < n_mt) {                                                  // (workgroup-uniform: counter_signal holds a barrier)
            // (tiles in step: the head's line of EVERY row tile gets the arrival -- four lanes, lines 8 apart -- so that a line has 16 pollers, not 64)
            if (has3) counter_signal(in.qkv_done + (size_t)((in.tiles_free ? mt3 * 8 : 0) + ((nt3 & 15) >> 1)) * CHAIN_LINE, in.tiles_free ? 1 : 4, 8 * CHAIN_LINE);
        }
    }
    CH_STAMP_OUT(c);
}

