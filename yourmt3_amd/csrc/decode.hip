// Autoregressive decoder step kernels (a7, a8, a10 of SURVEY.md section 8).
//
// Every kernel reads the decode position from DecodeShared in device memory, so ONE captured
// hipGraph of a step can be replayed for every position with no host round trip (SURVEY section 3.4:
// the per-step `.max()==0` host sync of TP: generation/utils.py:2936-2937 becomes device state).
//
//  dec_gemm_kernel      skinny GEMM, R = segments x channels rows (up to 511) against a [N][K] bf16
//                       weight: one workgroup per 16 rows x 16 output columns, eight waves split K, each
//                       wave pulls its K-slice of both operands with whole-line coalesced loads (all in
//                       flight at once), parks them in a wave-private LDS strip and reads MFMA fragments
//                       back; partial tiles are summed through LDS in a fixed order (bitwise reproducible;
//                       no float atomics).  NORM modes apply the T5 RMS norm (TP: modeling_t5.py:50-72)
//                       with the row's sum(h^2) carried between kernels as partials; the epilogues fuse
//                       KV-cache append (TP: cache_utils.py:144-145), ReLU, residual add + sum(h^2) partials.
//  dec_gemm_mid_kernel  the same GEMMs on BM x 64 tiles with a K loop, from 512 rows on (the 13-channel decoder).
//  dec_attn_kernel      one (row, head) per workgroup; K/V slabs streamed HBM -> registers with 16-byte
//                       coalesced loads, 8-12 in flight per lane; online softmax per 8-lane group (DPP sums),
//                       merged by shuffles and one LDS pass.  Self-attention adds the unidirectional relative
//                       position bias by distance (TP: modeling_t5.py:264-279) and, up to 96 rows, ends with its
//                       head's share of the output projection (OP); cross-attention has no bias and
//                       computes its own query projection (FUSEQ) while its K/V loads are in flight.
//  argmax_embed_kernel  fp32 argmax (first index wins ties, TP: utils.py:2925), EOS -> PAD fill
//                       (:2928-2929), token store, next-token embedding gather into the residual
//                       stream, and the step counter advance by the last workgroup to finish.
//
// Oracle: oracle/ymt3_oracle.py::decoder_step / greedy_decode.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int DKV = 64;
constexpr int WO_LDS_BYTES = 8 * 64 * 128;     // folded O-projection: the head's 64 k of all 512 wo rows
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Measurement only (YMT3_STAMP=1 at ymt3_create; the pointer is null otherwise): constant-rate wall clock (100 MHz) at the
// first instruction of a workgroup and at the end of its last wave, read back by ymt3_debug_step_stamps to split a decode
// step into launch gaps, dispatch ramps and kernel bodies.
#ifndef YMT3_STAMP_PHASE
#define YMT3_STAMP_PHASE 0           // 1 / 2: timing-only builds that move dec_gemm_kernel's entry stamp to a later phase (tools, not product)
#endif
#define STAMP_IN(a) do { if ((a).stamp && threadIdx.x == 0) (a).stamp[2 * blockIdx.x] = wall_clock64(); } while (0)
// attention pair only: stage marks behind the [grid][2] stamps of the kernel's slot (1024 words for its 512 workgroups), 8 per workgroup
#define PAIR_MARK(a, k, v) do { if ((a).stamp && threadIdx.x == 0) (a).stamp[1024 + 8 * blockIdx.x + (k)] = (v); } while (0)
#define STAMP_OUT(a) do { if ((a).stamp && (threadIdx.x & 63) == 0) atomicMax((a).stamp + 2 * blockIdx.x + 1, (unsigned long long)wall_clock64()); } while (0)

// ------------------------------------------------------------------------------------------------
// 512 threads = 8 waves; one workgroup = a 16 rows x 16 columns output tile over the full K, each wave
// owning K/8 of the reduction.  Small tiles keep every workgroup's operand traffic small (the per-CU load
// path is what bounds these kernels, not HBM); workgroups that share a weight tile are placed on one XCD
// (same blockIdx % 8) so its re-reads hit that XCD's L2.
//
// Operand path: every wave pulls ITS K-slice of the 16 activation rows and 16 weight rows
// with fully coalesced 16-byte loads (8..32 consecutive lanes per row = whole 128-byte lines), all
// issued up front (one memory round trip), parks them in a wave-private LDS strip, and reads them
// back in MFMA fragment order with ds_read_b128.  Fragment-shaped global loads (16 rows x 64 B per
// instruction) measured 3-4x slower for the same bytes (profiles/r01_notes.md).  No workgroup barrier
// is needed for the strip: only the owning wave touches it and LDS executes a wave's ops in order.
// (A 64-row-tile variant with direct fragment loads was measured 13-15 % slower at 256 and 832 rows and removed.)
//
// The RMS norm needs sum(x^2) over the FULL row, which no single wave sees: it is carried between
// kernels as per-row partial sums `ssq[tile][row]` written by whoever last wrote the residual stream
// (the 32 column tiles of the RESID epilogue, or the embedding gather) and summed in a fixed order.
// NT = 16-column tiles per workgroup (1 or 2): two halve the workgroup count of the wide projections (QKV, FFN-in, lm_head)
// to about one per CU -- at two per CU the second one's operands queue behind the first's and the kernel's last exit came
// 1.1 us after its first (profiles/r01_step_stamps.txt) -- and the activation strip is read once for both.
// PEND (NORM modes, MoE decoder): the residual stream still lacks the previous layer's expert outputs -- x = h + (y[2r] + y[2r+1]),
// the sum the MoE combine launch used to store.  The kernel forms x while loading, takes sum(x^2) itself (16-lane row sums, then
// the eight waves in order), and the workgroups of column tile 0 write x to a.h_out, the OTHER residual buffer (the rest of the
// layer reads that one; writing in place would race with the other column tiles still reading h): one launch less per MoE layer.
template <int MODE, int K, int NT, bool PEND = false>
__global__ __launch_bounds__(512) void dec_gemm_kernel(const bf16_t* __restrict__ pW, const void* __restrict__ pX, const float* __restrict__ pGain,
                                                       float* pSsq, float* pOut, int row0, int R, int N, int ssq_stride, DecGemmArgs a) {
    // The operand pointers and the tile geometry are LEADING SCALAR kernel arguments (14 dwords): built with
    // -mllvm -amdgpu-kernarg-preload-count=16 they arrive in SGPRs with the dispatch instead of through a scalar load at the head
    // of the kernel's critical path (every global load's address depends on them).  `a` carries the rest, read when needed.
    constexpr bool NORM = (MODE != DG_RESID);
    constexpr int KW = K / 8;            // K slice per wave
    constexpr int KS = KW / 32;          // MFMA k-steps per wave
    constexpr int ROWS = 16, COLS = 16 * NT;
    constexpr int PITCH = KW * 2 + 16;   // bytes per strip row (bf16 slice + 16 B pad against bank conflicts)
    constexpr int STRIP = 16 * PITCH;    // one operand strip (16 rows) of one wave
    static_assert(MODE != DG_RESID || NT == 1, "the RESID epilogue writes one sum(h^2) partial per 16-column tile");
    static_assert(!PEND || NORM, "a pending combine is folded into the norm prologue");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);                    // [8][ROWS][COLS]
    float* sscale = red + 8 * ROWS * COLS;                          // [ROWS]
    float* wpart = sscale + ROWS;                                   // PEND: [8 waves][ROWS] sum(x^2) of each wave's K-slice
    char* strips = smem + (8 * ROWS * COLS + ROWS + (PEND ? 8 * ROWS : 0)) * 4;   // [8 waves][A strip | NT W strips]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    // block -> (n tile, m tile): blocks with equal blockIdx % 8 (one XCD under round-robin placement;
    // speed only) walk the m tiles of one n tile back to back
    const int n_mt = (R + ROWS - 1) / ROWS, n_nt = N / COLS;
    int nt_idx, mt_idx;
    if ((n_nt & 7) == 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        mt_idx = slot % n_mt;
        nt_idx = (slot / n_mt) * 8 + xcd;
    } else {
        mt_idx = blockIdx.x % n_mt;
        nt_idx = blockIdx.x / n_mt;
    }
    const int n0 = nt_idx * COLS, m0 = row0 + mt_idx * ROWS, m_end = row0 + R;

    // epilogue ownership: thread -> (row mr, 2 columns nq)
    const bool epi = tid < ROWS * 8 * NT;
    const int mr = tid / (8 * NT), nq = (tid % (8 * NT)) * 2;
    const int m = m0 + mr, n = n0 + nq;
    const bool live = epi && m < m_end;
    float2 hold = make_float2(0.f, 0.f);
    float2 hp[8];                                // DG_RESID with a.part: the eight per-head partials of the folded O-projection
    if constexpr (MODE == DG_RESID) {
        if (live) hold = *reinterpret_cast<const float2*>(pOut + (size_t)m * N + n);   // prefetch the RMW operand
        if (a.part && live) {
#pragma unroll
            for (int w = 0; w < 8; ++w) hp[w] = *reinterpret_cast<const float2*>(a.part + ((size_t)m * 8 + w) * N + n);
        }
    }
    int step = 0;                                // cache position of the KV append: requested now, not as a round trip in the epilogue
    if constexpr (MODE == DG_NORM_QKV_CACHE) step = a.row_pos ? a.row_pos[m < m_end ? m : m_end - 1] : a.shared->step;

    f32x4 acc[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};

    {
        char* sA = strips + wave * (1 + NT) * STRIP;
        char* sW = sA + STRIP;
        // weight slice: LPRW lanes cover one row's KW bf16
        constexpr int LPRW = KW * 2 / 16, RPIW = 64 / LPRW, NIA = 16 / RPIW, NIW = NIA * NT;
        u32x4 wv[NIW];
#pragma unroll
        for (int i = 0; i < NIW; ++i) {
            const int row = i * RPIW + lane / LPRW, ch = lane % LPRW;
            wv[i] = *reinterpret_cast<const u32x4*>(pW + (size_t)(n0 + row) * K + wave * KW + ch * 8);
        }
        if constexpr (NORM) {
            constexpr int LPRX = KW * 4 / 16, RPIX = 64 / LPRX, NIX = 16 / RPIX;   // fp32 rows
            f32x4 xv[NIX];
#pragma unroll
            for (int i = 0; i < NIX; ++i) {
                int mm = m0 + i * RPIX + lane / LPRX;
                mm = mm < m_end ? mm : m_end - 1;
                xv[i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(pX) + (size_t)mm * K + wave * KW + (lane % LPRX) * 4);
            }
            const f32x4 gv = *reinterpret_cast<const f32x4*>(pGain + wave * KW + (lane % LPRX) * 4);
            f32x4 y0v[PEND ? NIX : 1], y1v[PEND ? NIX : 1];
            float ss = 0.f;
            if constexpr (PEND) {
#pragma unroll
                for (int i = 0; i < NIX; ++i) {
                    int mm = m0 + i * RPIX + lane / LPRX;
                    mm = mm < m_end ? mm : m_end - 1;
                    const float* yr = a.pend_y + (size_t)(2 * mm) * K + wave * KW + (lane % LPRX) * 4;
                    y0v[i] = *reinterpret_cast<const f32x4*>(yr);
                    y1v[i] = *reinterpret_cast<const f32x4*>(yr + K);
                }
            } else if (tid < ROWS * 8) {                             // 8 threads per row, 4 of the 32 partials each
                const int mm = m0 + (tid >> 3) < m_end ? m0 + (tid >> 3) : m_end - 1;
#pragma unroll
                for (int j = 0; j < 4; ++j) ss += pSsq[(size_t)((tid & 7) * 4 + j) * ssq_stride + mm];
            }
            __builtin_amdgcn_sched_barrier(0);   // every operand load is in flight before anything waits
            STAMP_IN(a);                         // (reads a.stamp from the kernarg segment: after the loads, not in front of them)
            if constexpr (PEND) {
#pragma unroll
                for (int i = 0; i < NIX; ++i) {
                    const int row = i * RPIX + lane / LPRX;
#pragma unroll
                    for (int e = 0; e < 4; ++e) xv[i][e] += y0v[i][e] + y1v[i][e];          // h + (y0 + y1), as the combine kernel summed
                    float q = (xv[i][0] * xv[i][0] + xv[i][1] * xv[i][1]) + (xv[i][2] * xv[i][2] + xv[i][3] * xv[i][3]);
                    q = add_xor8(sum8(q));                           // the LPRX = 16 lanes that hold this row's K-slice (lanes ^1, ^2, ^4, ^8)
                    if ((lane % LPRX) == 0) wpart[wave * ROWS + row] = q;
                    if constexpr (MODE == DG_NORM_QKV_CACHE) {       // (lm_head is the last reader of the stream: nothing to publish)
                        if (nt_idx == 0 && m0 + row < m_end)         // column tile 0 publishes the completed residual row
                            *reinterpret_cast<f32x4*>(a.h_out + (size_t)(m0 + row) * K + wave * KW + (lane % LPRX) * 4) = xv[i];
                    }
                }
            } else {
                ss = sum8(ss);
                if (tid < ROWS * 8 && (tid & 7) == 0) sscale[tid >> 3] = rsqrtf(ss / (float)K + a.eps);
            }
#pragma unroll
            for (int i = 0; i < NIW; ++i)
                *reinterpret_cast<u32x4*>(sW + (i * RPIW + lane / LPRW) * PITCH + (lane % LPRW) * 16) = wv[i];
            __syncthreads();
            if constexpr (PEND) {
                if (tid < ROWS) {
                    float t = wpart[tid];
#pragma unroll
                    for (int w = 1; w < 8; ++w) t += wpart[w * ROWS + tid];
                    sscale[tid] = rsqrtf(t / (float)K + a.eps);
                }
                __syncthreads();
            }
#pragma unroll
            for (int i = 0; i < NIX; ++i) {
                const int row = i * RPIX + lane / LPRX;
                const float sc = sscale[row];
                *reinterpret_cast<uint2*>(sA + row * PITCH + (lane % LPRX) * 8) =
                    make_uint2(pack_bf16x2(xv[i][0] * sc * gv[0], xv[i][1] * sc * gv[1]),
                               pack_bf16x2(xv[i][2] * sc * gv[2], xv[i][3] * sc * gv[3]));
            }
        } else {
            u32x4 av[NIA];
#pragma unroll
            for (int i = 0; i < NIA; ++i) {
                int mm = m0 + i * RPIW + lane / LPRW;
                mm = mm < m_end ? mm : m_end - 1;
                av[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(pX) + (size_t)mm * K + wave * KW + (lane % LPRW) * 8);
            }
            __builtin_amdgcn_sched_barrier(0);   // every operand load is in flight before anything waits
            STAMP_IN(a);
#pragma unroll
            for (int i = 0; i < NIW; ++i) {
                const int off = (i * RPIW + lane / LPRW) * PITCH + (lane % LPRW) * 16;
                *reinterpret_cast<u32x4*>(sW + off) = wv[i];
                if (i < NIA) *reinterpret_cast<u32x4*>(sA + off) = av[i];
            }
        }
        // fragment order read-back (wave-private strips: in-order LDS, no barrier)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int off = li * PITCH + (ks * 32 + g * 8) * 2;
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(sA + off);
#pragma unroll
            for (int c = 0; c < NT; ++c) {
                const bf16x8 wf = *reinterpret_cast<const bf16x8*>(sW + c * STRIP + off);
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af, acc[c], 0, 0, 0);
            }
        }
    }

#if YMT3_STAMP_PHASE == 1
    STAMP_IN(a);             // timing-only build: "entry" = operands have arrived and the MFMAs are done
#endif
    // fixed-order cross-wave reduction: red[wave][row][n (COLS)]
#pragma unroll
    for (int c = 0; c < NT; ++c)
        *reinterpret_cast<float4*>(red + ((wave * ROWS + li) * COLS + c * 16 + g * 4)) =
            make_float4(acc[c][0], acc[c][1], acc[c][2], acc[c][3]);
    __syncthreads();
#if YMT3_STAMP_PHASE == 2
    STAMP_IN(a);             // timing-only build: "entry" = partial tiles reduced across waves, epilogue next
#endif
    float2 s = make_float2(0.f, 0.f);
    if (epi) {
        s = *reinterpret_cast<const float2*>(red + (mr * COLS + nq));
#pragma unroll
        for (int w = 1; w < 8; ++w) {
            const float2 t = *reinterpret_cast<const float2*>(red + ((w * ROWS + mr) * COLS + nq));
            s.x += t.x; s.y += t.y;
        }
    }

    if constexpr (MODE == DG_RESID) {
        float2 o = make_float2(0.f, 0.f);
        if (live) {
            if (a.part) {                         // h + (p0 + ... + p7): what the separate O-projection launch left in h
                float2 sp = hp[0];
#pragma unroll
                for (int w = 1; w < 8; ++w) { sp.x += hp[w].x; sp.y += hp[w].y; }
                hold.x += sp.x; hold.y += sp.y;
            }
            o = make_float2(hold.x + s.x, hold.y + s.y);
            *reinterpret_cast<float2*>(pOut + (size_t)m * N + n) = o;
        }
        // this tile's share of sum(h^2) for the next norm
        float q = o.x * o.x + o.y * o.y;
        q = sum8(q);
        if (live && (tid & 7) == 0) pSsq[(size_t)nt_idx * ssq_stride + m] = q;
    } else if constexpr (MODE == DG_NORM_LOGITS) {
        if (live) *reinterpret_cast<float2*>(pOut + (size_t)m * N + n) = s;
    } else if (live) {
        if constexpr (MODE == DG_NORM_BF16_RELU) { s.x = fmaxf(s.x, 0.f); s.y = fmaxf(s.y, 0.f); }
        const uint32_t pk = pack_bf16x2(s.x, s.y);
        if constexpr (MODE == DG_NORM_QKV_CACHE) {
            const int inner = a.H * DKV;
            if (n < inner) {
                *reinterpret_cast<uint32_t*>(a.out_bf16 + (size_t)m * inner + n) = pk;
            } else {
                const int nn = n - inner, kv = nn / inner, hh = (nn % inner) >> 6, dd = nn & 63;
                bf16_t* cache = kv ? a.vcache : a.kcache;
                *reinterpret_cast<uint32_t*>(cache + (((size_t)m * a.H + hh) * a.L + step) * DKV + dd) = pk;
            }
        } else {
            *reinterpret_cast<uint32_t*>(a.out_bf16 + (size_t)m * N + n) = pk;
        }
    }
    STAMP_OUT(a);
}

// ------------------------------------------------------------------------------------------------
// Mid-size tile form of the same GEMMs for MANY rows (R >= 256: the 13-channel decoder's 832 rows, batches of 256+).
// The 16-row kernel above launches (R / 16) x (N / 16..32) workgroups of 143 KB LDS each -- 3328 for the FFN-in projection at
// 832 rows, thirteen rounds over the chip -- and re-reads the weights once per 16 rows.  Here one 256-thread workgroup owns a
// BM x 64 output tile (BM = 64, or 32 where N = 512 would leave the chip half empty), walks K in 64-wide steps through two LDS
// stages (operands staged through registers one step ahead; the RMS norm -- row scale from the carried sum(h^2) partials,
// gain per k -- is applied while the fp32 rows are converted), wave w owning output columns [16w, 16w + 16) of every 16-row
// block.  Same epilogues as above (KV-cache append, ReLU, residual add + sum(h^2) partials per 16-column tile, logits).
// K is accumulated in ONE MFMA chain per output (no 8-way split), so results differ from the 16-row kernel in the last
// bits: the launcher switches on R alone, and rows stay independent of their batch inside either regime.
template <int MODE, int K, int BM, int PF, int NTH>
__global__ __launch_bounds__(NTH) void dec_gemm_mid_kernel(const bf16_t* __restrict__ pW, const void* __restrict__ pX, const float* __restrict__ pGain,
                                                           float* pSsq, float* pOut, int row0, int R, int N, int ssq_stride, DecGemmArgs a) {
    constexpr bool NORM = (MODE != DG_RESID);
    constexpr int BN = 64, BK = 64, MT = BM / 16, PITCH = BK * 2 + 16, NKT = K / BK;
    // NTH = 512: two waves per SIMD, each with half of the row tiles of its 16 columns (wave & 3: columns, wave >> 2: row tiles) -- with one wave per
    // SIMD every LDS read, MFMA and barrier of a K-step was exposed (K = 2048 at 832 rows: 8.5 -> ... us, profiles/r03_notes.md)
    constexpr int RG = NTH / 256, MTW = MT / RG;
    static_assert(NTH == 256 || NTH == 512, "");
    static_assert(MT % RG == 0, "row tiles split between the two wave groups");
    constexpr int A_STAGE = BM * PITCH, W_STAGE = BN * PITCH;
    __shared__ __attribute__((aligned(16))) char sA[2][A_STAGE];
    __shared__ __attribute__((aligned(16))) char sW[2][W_STAGE];
    __shared__ float sscale[BM];
    __shared__ __attribute__((aligned(16))) float sgain[NORM ? K : 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
    const int wc = wave & 3, rg = wave >> 2;
    const int n_mt = (R + BM - 1) / BM, n_nt = N / BN;
    int nt_idx, mt_idx;
    if ((n_nt & 7) == 0) {                       // blocks with equal blockIdx % 8 (one XCD; speed only) walk the row tiles of one column tile
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        mt_idx = slot % n_mt;
        nt_idx = (slot / n_mt) * 8 + xcd;
    } else {
        mt_idx = blockIdx.x % n_mt;
        nt_idx = blockIdx.x / n_mt;
    }
    const int n0 = nt_idx * BN, m0 = row0 + mt_idx * BM, m_end = row0 + R;

    // staging registers, a ring of PF K-steps: W tile 64 x 128 B = 2 x 16 B per thread; A tile BM x 128 B (bf16) or BM x 256 B (fp32).
    // One step ahead left every step exposed to a full L2 / HBM round trip (22.6 us for K = 2048 at 832 rows, profiles/r02_notes.md)
    constexpr int NWV = 512 / NTH, NAB = (BM * 8 + NTH - 1) / NTH, NAF = BM * 16 / NTH;
    static_assert(BM * 16 % NTH == 0, "");
    static_assert(NKT % PF == 0, "the K loop is unrolled by the prefetch depth");
    u32x4 wvr[PF][NWV];
    u32x4 abr[PF][NORM ? 1 : NAB];
    f32x4 afr[PF][NORM ? NAF : 1];
    auto load_regs = [&](int kt, u32x4 (&wv)[NWV], u32x4 (&ab)[NORM ? 1 : NAB], f32x4 (&af)[NORM ? NAF : 1]) {
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int idx = tid + i * NTH, row = idx >> 3, ch = idx & 7;
            wv[i] = *reinterpret_cast<const u32x4*>(pW + (size_t)(n0 + row) * K + kt * BK + ch * 8);
        }
        if constexpr (NORM) {
#pragma unroll
            for (int i = 0; i < NAF; ++i) {
                const int idx = tid + i * NTH, row = idx >> 4, ch = idx & 15;
                int mm = m0 + row;
                mm = mm < m_end ? mm : m_end - 1;
                af[i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(pX) + (size_t)mm * K + kt * BK + ch * 4);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NAB; ++i) {
                const int idx = tid + i * NTH, row = idx >> 3, ch = idx & 7;
                int mm = m0 + row;
                mm = mm < m_end ? mm : m_end - 1;
                if (BM * 8 % NTH == 0 || idx < BM * 8)  // (BM = 32 with 512 threads: half of them carry a chunk)
                    ab[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(pX) + (size_t)mm * K + kt * BK + ch * 8);
            }
        }
    };
    auto store_lds = [&](int kt, int buf, const u32x4 (&wv)[NWV], const u32x4 (&ab)[NORM ? 1 : NAB], const f32x4 (&af)[NORM ? NAF : 1]) {
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int idx = tid + i * NTH, row = idx >> 3, ch = idx & 7;
            *reinterpret_cast<u32x4*>(sW[buf] + row * PITCH + ch * 16) = wv[i];
        }
        if constexpr (NORM) {
#pragma unroll
            for (int i = 0; i < NAF; ++i) {
                const int idx = tid + i * NTH, row = idx >> 4, ch = idx & 15;
                const float sc = sscale[row];
                const f32x4 gv = *reinterpret_cast<const f32x4*>(sgain + kt * BK + ch * 4);
                *reinterpret_cast<uint2*>(sA[buf] + row * PITCH + ch * 8) =
                    make_uint2(pack_bf16x2(af[i][0] * sc * gv[0], af[i][1] * sc * gv[1]), pack_bf16x2(af[i][2] * sc * gv[2], af[i][3] * sc * gv[3]));
            }
        } else {
#pragma unroll
            for (int i = 0; i < NAB; ++i) {
                const int idx = tid + i * NTH, row = idx >> 3, ch = idx & 7;
                if (BM * 8 % NTH == 0 || idx < BM * 8) *reinterpret_cast<u32x4*>(sA[buf] + row * PITCH + ch * 16) = ab[i];
            }
        }
    };

    // requested in the order of first use (vector loads return in order): the norm's inputs -- the sum(h^2) partials of this thread's row, the gain
    // vector -- then the operand ring
    float ss = 0.f;
    f32x4 gv = {0.f, 0.f, 0.f, 0.f};
    if constexpr (NORM) {
        static_assert(K / 4 <= NTH, "one float4 of the gain vector per thread");
        if (tid < K / 4) gv = *reinterpret_cast<const f32x4*>(pGain + tid * 4);
        if (tid < BM * 4) {
            const int row = tid >> 2;
            const int mm = m0 + row < m_end ? m0 + row : m_end - 1;
#pragma unroll
            for (int j = 0; j < SSQ_TILES / 4; ++j) ss += pSsq[(size_t)((tid & 3) * (SSQ_TILES / 4) + j) * ssq_stride + mm];
        }
    }
#pragma unroll
    for (int p = 0; p < PF; ++p) load_regs(p, wvr[p], abr[p], afr[p]);
    int step = 0;
    if constexpr (MODE == DG_NORM_QKV_CACHE) step = a.row_pos ? 0 : a.shared->step;       // per-row positions are read in the epilogue
    if constexpr (NORM) {
        // row scales from the carried partials (4 threads per row, 8 partials each, then the 4-lane sum) and the gain vector
        if (tid < K / 4) *reinterpret_cast<f32x4*>(sgain + tid * 4) = gv;
        if (tid < BM * 4) {
            ss = add_xor2(add_xor1(ss));
            if ((tid & 3) == 0) sscale[tid >> 2] = rsqrtf(ss / (float)K + a.eps);
        }
        __syncthreads();
    }
    // DG_RESID: the residual values this lane will add to are requested now, behind the operand ring, not in the epilogue (a dependent round trip there)
    float4 resid[MODE == DG_RESID ? MTW : 1];
    if constexpr (MODE == DG_RESID) {
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
            const int m = m0 + (rg * MTW + mt) * 16 + li;
            resid[mt] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < m_end) resid[mt] = *reinterpret_cast<const float4*>(pOut + (size_t)m * N + n0 + wc * 16 + g * 4);
        }
    }
    STAMP_IN(a);
    store_lds(0, 0, wvr[0], abr[0], afr[0]);
    __syncthreads();

    f32x4 acc[MTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kt0 = 0; kt0 < NKT; kt0 += PF) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {             // ring slot j holds step kt0 + j; after its LDS store it is refilled with step kt + PF
            const int kt = kt0 + j, buf = kt & 1;
            if (kt + PF < NKT) load_regs(kt + PF, wvr[j], abr[j], afr[j]);     // slot j went to LDS one step ago (or before the loop)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 wf = *reinterpret_cast<const bf16x8*>(sW[buf] + (wc * 16 + li) * PITCH + (ks * 32 + g * 8) * 2);
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) {
                    const bf16x8 xf = *reinterpret_cast<const bf16x8*>(sA[buf] + ((rg * MTW + mt) * 16 + li) * PITCH + (ks * 32 + g * 8) * 2);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf, acc[mt], 0, 0, 0);
                }
            }
            if (kt + 1 < NKT) {
                constexpr int PFm = PF - 1;
                const int jn = (j + 1) & PFm;
                // the other LDS stage: its last readers passed the barrier at the end of step kt - 1
                store_lds(kt + 1, buf ^ 1, wvr[jn], abr[jn], afr[jn]);          // (j is a constant after unrolling)
                __syncthreads();
            }
        }
    }

    // epilogue: lane (li, g) of wave w holds row m0 + mt*16 + li, columns n0 + 16w + 4g .. +3
    const int n = n0 + wc * 16 + g * 4;
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        const int m = m0 + (rg * MTW + mt) * 16 + li;
        const bool live = m < m_end;
        f32x4 s = acc[mt];
        if constexpr (MODE == DG_RESID) {
            float q = 0.f;
            if (live) {
                float4* dst = reinterpret_cast<float4*>(pOut + (size_t)m * N + n);
                float4 o = resid[mt];
                o.x += s[0]; o.y += s[1]; o.z += s[2]; o.w += s[3];
                *dst = o;
                q = (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w);
            }
            q += lane_xor16(q);                   // the four lanes (g) that hold this row's 16 columns of the tile
            q += lane_xor32(q);
            if (live && g == 0) pSsq[(size_t)(n0 / 16 + wc) * ssq_stride + m] = q;
        } else if constexpr (MODE == DG_NORM_LOGITS) {
            if (live) *reinterpret_cast<float4*>(pOut + (size_t)m * N + n) = make_float4(s[0], s[1], s[2], s[3]);
        } else if (live) {
            if constexpr (MODE == DG_NORM_BF16_RELU) { s[0] = fmaxf(s[0], 0.f); s[1] = fmaxf(s[1], 0.f); s[2] = fmaxf(s[2], 0.f); s[3] = fmaxf(s[3], 0.f); }
            const uint2 pk = make_uint2(pack_bf16x2(s[0], s[1]), pack_bf16x2(s[2], s[3]));
            if constexpr (MODE == DG_NORM_QKV_CACHE) {
                const int inner = a.H * DKV;
                if (n < inner) {
                    *reinterpret_cast<uint2*>(a.out_bf16 + (size_t)m * inner + n) = pk;
                } else {
                    const int nn = n - inner, kv = nn / inner, hh = (nn % inner) >> 6, dd = nn & 63;
                    const int pos = a.row_pos ? a.row_pos[m] : step;
                    bf16_t* cache = kv ? a.vcache : a.kcache;
                    *reinterpret_cast<uint2*>(cache + (((size_t)m * a.H + hh) * a.L + pos) * DKV + dd) = pk;
                }
            } else {
                *reinterpret_cast<uint2*>(a.out_bf16 + (size_t)m * N + n) = pk;
            }
        }
    }
    STAMP_OUT(a);
}

// ------------------------------------------------------------------------------------------------
// (sum8: the sum over the 8 lanes that share one key -- lane & 7 = 16-byte chunk of the 128-byte row -- by DPP moves: common.h)

// FUSEQ (cross-attention only): the query projection of the block,  q = R( R(rmsnorm(h_r) * gain) . Wq[head]^T ),  is
// computed inside this kernel instead of by a separate skinny GEMM launch.  Its operands (64 weight rows of this
// head, the residual row, the sum(h^2) partials) are independent of the K/V stream, so they are issued first and
// the whole projection runs while the K/V loads are in flight: one kernel boundary (~5 us in situ) less per layer.
// NW = waves per (row, head): 8 for up to ~2k workgroups (16 waves per CU keep > 12 MB in flight chip-wide); 2 when there
// are many more (row, head) pairs than CUs (multi-channel / large batches), where 512-thread workgroups with a few keys
// each only add dispatch rounds.
// (Two rows per workgroup sharing the head's projection weights -- half the weight reads from L2 -- was measured twice and is
// slower: with 512 threads, 10.2 vs 8.7 us, eight waves per CU keep too little of the K/V stream in flight
// (profiles/r01_step_stamps.txt); with 1024 threads, one workgroup per CU puts both rows' projections and streams in lockstep behind
// the same barriers, +1.4 us per cross-attention launch, and the folded self-attention gains nothing end to end (profiles/r02_two_rows*.txt).)
// OP (O-projection partials; 8 waves only).  SELF: the kernel ends with its head's share of the output projection,
// opart[r][h][:] = R(o) . wo[:, 64h..64h+64)^T -- the same two chained MFMAs over the same 64 k as wave h of the DG_RESID kernel, so
// the values are that kernel's split-K partials bit for bit; the 64 weight rows x 128 B of every wave come in by LDS DMA issued
// before the K/V stream (L2 hits, no registers) and sit in LDS until the tail.  FUSEQ cross-attention: the residual row is
// h + (p0 + ... + p7), summed in wave order as DG_RESID's reduction does, and sum(x^2) is rebuilt with DG_RESID's tree
// (pairs, then the 16-column tiles) followed by the 32-tile wave sum the ssq consumers use: one launch less per layer, same bits.
// PAIR (dec_attn_pair_kernel below: a layer's self- and cross-attention as one launch): 1 = the self-attention half -- its
// O-projection partials leave at agent scope, the caller signals the row; 2 = the cross-attention half -- it requests everything that
// does not depend on the self-attention (its head's wq, the residual row, its K/V block), THEN waits for the row's eight heads and
// reads their partials at agent scope.
struct PairSync {
    unsigned* rows;             // [R][2] lines of CHAIN_LINE words: arrivals, departures (the eighth to leave zeroes both)
    unsigned* abort_word;       // the handle's sticky abort word (shared with the GEMM chain)
    unsigned* host_abort;
};
// Returns (thread 0) the number of heads that had left the row before this one, or ~0u after an abort: the departure count is REQUESTED here, behind
// the barrier, and looked at only at the end of the kernel (pair_leave) -- as a fetch-add whose result thread 0 waited for in front of the barrier it
// put a memory round trip (~0.7 us) on every workgroup's way out of the hand-off.
__device__ __forceinline__ unsigned pair_wait(const PairSync& ps, int r) {
    bool ok = true;
    if (threadIdx.x == 0) {
        unsigned* arr = ps.rows + (size_t)(2 * r) * CHAIN_LINE;
        unsigned long long t0 = 0;
        unsigned polls = 0;
        while (__hip_atomic_load(arr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 8u) {
            if ((++polls & 63u) == 0u) {
                const unsigned long long now = wall_clock64();
                if (t0 == 0) t0 = now;
                if (__hip_atomic_load(ps.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || now - t0 > 100000000ull) {      // 1 s
                    __hip_atomic_store(ps.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (ps.host_abort) __hip_atomic_store(ps.host_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    ok = false;
                    break;
                }
            }
        }
    }
    __syncthreads();
    unsigned before = ~0u;
    if (threadIdx.x == 0 && ok) before = __hip_atomic_fetch_add(ps.rows + (size_t)(2 * r + 1) * CHAIN_LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return before;
}
// the last of the row's eight heads to have left leaves both counters at zero for the next launch
__device__ __forceinline__ void pair_leave(const PairSync& ps, int r, unsigned before) {
    if (threadIdx.x == 0 && before == 7u) {
        unsigned* arr = ps.rows + (size_t)(2 * r) * CHAIN_LINE;
        __hip_atomic_store(arr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(arr + CHAIN_LINE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <bool SELF, bool FUSEQ, int NW, bool OP, int PAIR>
__device__ __forceinline__ void attn_body(const bf16_t* __restrict__ pK, const bf16_t* __restrict__ pV, const bf16_t* __restrict__ pQ,
                                          const float* __restrict__ pF, const void* __restrict__ p4, const void* __restrict__ p5,
                                          unsigned geom0, unsigned geom1, const DecAttnArgs& a, const PairSync& ps) {
    // Leading scalar arguments (14 dwords; kernarg preload, see dec_gemm_kernel) -- everything the first loads' addresses depend
    // on: K / V slab bases, pQ = q (wq when FUSEQ), pF = the position-bias table (SELF) or the residual stream (FUSEQ),
    // p4 / p5 = loop state and per-row positions (SELF) or norm gain and sum(h^2) partials / O-projection partials (FUSEQ),
    // geom0 = row0 | rows_per_kv << 16 | H << 24, geom1 = slab_keys | n_keys_const << 20.
    unsigned long long t_entry = 0;
    if constexpr (PAIR == 1) t_entry = wall_clock64();          // measurement: stored with the entry stamp, behind the first loads
    const int row0 = geom0 & 0xffff, rows_per_kv = (geom0 >> 16) & 0xff, H = geom0 >> 24;
    const int slab_keys = geom1 & 0xfffff, n_keys_const = geom1 >> 20;
    const DecodeShared* pShared = static_cast<const DecodeShared*>(p4);
    const int* pRowPos = static_cast<const int*>(p5);
    const float* pGain = static_cast<const float*>(p4);
    const float* pPart = static_cast<const float*>(p5);
    static_assert(!OP || NW == 8, "the folded O-projection maps wave w to output columns [64w, 64w + 64)");
    __shared__ float sm[NW], sl[NW], sacc[NW][DKV];
    __shared__ __attribute__((aligned(16))) float xs[FUSEQ ? 512 : 4];
    __shared__ __attribute__((aligned(16))) bf16_t qs[DKV];
    __shared__ float s_scale;
    __shared__ float s_tile[OP && FUSEQ ? SSQ_TILES : 1];
    extern __shared__ __attribute__((aligned(1024))) char wo_lds[];      // OP && SELF: [8 waves][64 rows][128 B], chunks XOR-swizzled by row
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sub = lane & 7, kg = lane >> 3;
    const int r = row0 + blockIdx.x / H, h = blockIdx.x % H;      // (a row's eight heads on one XCD instead -- block -> row 8 * (b / 64) + b % 8 -- measured 1 % slower)
    const int n_keys = SELF ? (pRowPos ? pRowPos[r] : pShared->step) + 1 : n_keys_const;
    const int kv_row = r / rows_per_kv;
    const size_t slab = ((size_t)kv_row * H + h) * slab_keys * DKV;
    const bf16_t* kb = pK + slab + sub * 8;
    const bf16_t* vb = pV + slab + sub * 8;
    const float* bias = SELF ? pF + (size_t)h * slab_keys : nullptr;      // the bias table's row pitch is the cache length (launcher checks)
    // OP && SELF: this wave's 64 rows of wo (output columns 64w..64w+63), the head's 64 k each: 8 DMAs of 8 rows x 128 B.  LDS slot
    // (row, pos) holds source chunk pos ^ (row & 7): the swizzle sits on the source address, the DMA writes linearly.
    // Issued BEHIND the wave's first K/V block (below): at entry, 512 workgroups' 64 KB pulls backed the vector-memory queues up and a
    // wave's first K/V request left up to 5.9 us after entry (profiles/r02_attn_pair_marks.txt); requested when the stream has been consumed
    // instead, they land under the merge and lengthen the tail by more than the head gains (259.5 against 253.9 ms per batch, same file).
    auto issue_wo = [&]() {
        if constexpr (OP && SELF) {
            const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)wo_lds + (unsigned)wave * 8192u;
            const bf16_t* src = a.wo + ((size_t)(wave * 64 + (lane >> 3)) * H + h) * DKV + ((lane & 7) ^ (lane >> 3)) * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) glds16(src + (size_t)i * 8 * H * DKV, lds0 + (unsigned)i * 1024u);
        }
    };

    // q stays packed (4 x bf16x2); halves are widened to fp32 at use
    u32x4 qp;
    u32x4 wq_v[FUSEQ ? 8 : 1];
    float x_v = 0.f, g_v = 0.f, ss = 0.f;
    float pv[OP && FUSEQ ? 8 : 1];
    unsigned left_before = ~0u;                 // PAIR == 2: pair_wait's departure count (thread 0)
    auto load_proj = [&]() {
        // operands of the fused projection: wave w owns outputs 8w..8w+7, lane l the k-chunk 8l..8l+7
#pragma unroll
        for (int jj = 0; jj < 8; ++jj)
            wq_v[jj] = *reinterpret_cast<const u32x4*>(pQ + ((size_t)h * DKV + wave * 8 + jj) * 512 + lane * 8);
        x_v = pF[(size_t)r * 512 + tid];
        g_v = pGain[tid];
    };
    if constexpr (FUSEQ) {
        if constexpr (PAIR != 2) load_proj();
        if constexpr (OP) {
            if constexpr (PAIR != 2) {
#pragma unroll
                for (int w = 0; w < 8; ++w) pv[w] = pPart[((size_t)r * H + w) * 512 + tid];
            }
        } else {
            if (tid < SSQ_TILES) ss = pPart[(size_t)tid * a.ssq_stride + r];
        }
    } else {
        qp = *reinterpret_cast<const u32x4*>(pQ + ((size_t)r * H + h) * DKV + sub * 8);
    }

    float m = -1.0e30f, l = 0.f, acc[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) acc[d] = 0.f;
    if constexpr (PAIR != 2) STAMP_IN(a);      // (reads a.stamp from the kernarg segment: behind the first loads, not in front of them)
    if constexpr (PAIR == 1) PAIR_MARK(a, 0, t_entry);

    // The self-attention cache (up to 805 MB) is read exactly once per step: non-temporal loads keep it
    // from evicting the weights (42 MB) and the cross-attention K/V (201 MB at 64 segments), both
    // re-read every step, out of the 256 MB Infinity Cache.  Up to 12 x 16-byte loads in flight per lane.
    constexpr int U = SELF ? 6 : 4;            // self: 12 loads in flight per lane (8 would spill at 128 VGPRs); cross: 8 x 8 x 4 = 256 frames in one shot
    // Every load of a block -- K, V and the position-bias values -- is issued in STRAIGHT-LINE code, keys beyond n_keys
    // clamped to the wave's first key and masked in the math: with the loads behind `if (u < nblk)` branches hipcc's
    // wait-count pass merges the paths and waits for the oldest load with the count of the path that issued the fewest
    // (vmcnt(1): the whole block), and each bias value became its own load + vmcnt(0) round trip in the score loop.
    auto load_block = [&](u32x4 (&ku)[U], u32x4 (&vu)[U], bool (&ok)[U], float (&bv)[U], int kw, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;         // FULL: all U blocks valid, no masks
        const int k0 = kw + kg;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int key = k0 + 8 * NW * u;
            ok[u] = FULL || key < n_keys;
            const int kc = ok[u] ? key : kw;
            if constexpr (SELF) {
                ku[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kb + (size_t)kc * DKV));
                vu[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(vb + (size_t)kc * DKV));
            } else {
                ku[u] = *reinterpret_cast<const u32x4*>(kb + (size_t)kc * DKV);
                vu[u] = *reinterpret_cast<const u32x4*>(vb + (size_t)kc * DKV);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if constexpr (SELF) bv[u] = bias[ok[u] ? n_keys - 1 - (k0 + 8 * NW * u) : 0];
            else bv[u] = 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);                       // every load of the block is in flight before any math waits
    };
    auto compute_block = [&](u32x4 (&ku)[U], u32x4 (&vu)[U], bool (&ok)[U], float (&bv)[U], auto full_tag) {
        float sc[U];
        float mn = m;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            {
                float s = 0.f;
                // plain fp32 FMAs on the unpacked halves: v_dot2c_f32_bf16 chains gave O(1) wrong scores on
                // gfx950 / ROCm 7.2 in this kernel (bisected on hardware), so the packed dot is not used
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s = fmaf(__uint_as_float(qp[j] << 16), __uint_as_float(ku[u][j] << 16), s);
                    s = fmaf(__uint_as_float(qp[j] & 0xffff0000u), __uint_as_float(ku[u][j] & 0xffff0000u), s);
                }
                s = sum8(s);
                if constexpr (SELF) s += bv[u];
                sc[u] = s;
                if (ok[u]) mn = fmaxf(mn, s);
            }
        }
        const float rescale = __expf(m - mn);
        l *= rescale;
#pragma unroll
        for (int d = 0; d < 8; ++d) acc[d] *= rescale;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            {
                const float p = ok[u] ? __expf(sc[u] - mn) : 0.f;
                l += p;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[2 * j] = fmaf(p, __uint_as_float(vu[u][j] << 16), acc[2 * j]);
                    acc[2 * j + 1] = fmaf(p, __uint_as_float(vu[u][j] & 0xffff0000u), acc[2 * j + 1]);
                }
            }
        }
        m = mn;
    };
    u32x4 fku[FUSEQ ? U : 1], fvu[FUSEQ ? U : 1];             // FUSEQ: the first K/V block lives across the projection
    bool fok[FUSEQ ? U : 1];
    float fbv[FUSEQ ? U : 1];
    if constexpr (!FUSEQ) {
        auto block = [&](int kw, auto full_tag) {                // staging registers scoped to one block
            u32x4 ku[U], vu[U];
            bool ok[U];
            float bv[U];
            load_block(ku, vu, ok, bv, kw, full_tag);
            compute_block(ku, vu, ok, bv, full_tag);
        };
        int kw = wave * 8;
        if constexpr (OP && SELF) {
            // the first block's K/V requests, then the wo DMAs, then its math (the inline-asm DMAs are invisible to hipcc's wait counts: its waits
            // for the block's loads, which are older, do not wait for them until the block's last load is needed)
            if (kw < n_keys) {
                u32x4 ku[U], vu[U];
                bool ok[U];
                float bv[U];
                if (n_keys - kw >= 8 * NW * U) {
                    load_block(ku, vu, ok, bv, kw, std::true_type{});
                    issue_wo();
                    compute_block(ku, vu, ok, bv, std::true_type{});
                } else {
                    load_block(ku, vu, ok, bv, kw, std::false_type{});
                    issue_wo();
                    compute_block(ku, vu, ok, bv, std::false_type{});
                }
                kw += 8 * NW * U;
            } else {
                issue_wo();
            }
        }
        for (; kw < n_keys; kw += 8 * NW * U) {      // wave-uniform trip count: waves with no key skip everything
            if (n_keys - kw >= 8 * NW * U) block(kw, std::true_type{});
            else block(kw, std::false_type{});
        }
    } else {
        // every wave's projection operands are requested before ANY wave's K/V: the CU returns vector-memory data in request
        // order, so weight lines (L2 hits) queued behind another wave's K/V lines (HBM) reached the projection only after
        // most of the stream had arrived, and the projection ran exposed at the end (+2.7 us, profiles/r01_step_stamps.txt)
        if constexpr (PAIR == 2) {
            // The self-attention half's partial stores are acknowledged (vmcnt(0)), the row is signalled, and only THEN are the projection's own
            // operands requested (10 loads per lane: 8 wq, x, gain).  Round 2 issued them first and signalled behind `s_waitcnt vmcnt(10)` (the
            // stores are older; vector-memory operations retire in issue order) so that the acknowledgement ran under the requests -- but a load
            // instruction waits for room in the CU's vector-memory queue, which the other workgroups' K/V streams keep full, and that wait sat
            // in front of the signal: 245.5 -> 241.6 ms per batch with the order turned round (same box, two builds interleaved,
            // profiles/r03_notes.md).  The K/V block is requested after the hand-off.  Requested before it, a poll (and the partials) came
            // back behind it, in issue
            // order, i.e. only once the whole block had arrived (254.5 ms per batch against 253.9); requested by the self-attention half
            // as soon as its own stream was consumed, it slowed the other workgroups' self-attention streams by more than it gained
            // (263 ms; profiles/r02_attn_pair_marks.txt).  (The pair kernel keeps hipcc from moving memory operations across the
            // boundary between the halves.)
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(ps.rows + (size_t)(2 * r) * CHAIN_LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            PAIR_MARK(a, 3, wall_clock64());     // row signalled
            load_proj();
            __builtin_amdgcn_sched_barrier(0);
            left_before = pair_wait(ps, r);
            PAIR_MARK(a, 4, wall_clock64());     // the row's eight heads have arrived
#pragma unroll
            for (int w = 0; w < 8; ++w) pv[w] = __hip_atomic_load(pPart + ((size_t)r * H + w) * 512 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            __builtin_amdgcn_s_barrier();
        }
    }
    if constexpr (FUSEQ) {
        // first (for T <= 256: only) K/V block goes in flight now; its math waits for the projection below.  STRAIGHT-LINE
        // code, keys beyond n_keys clamped to key 0 of the slab: behind a branch hipcc merges the two paths' load counts and
        // waits for the projection operands with the count of the path that issued no K/V loads -- i.e. for (almost) the
        // whole K/V stream, which put the projection after the stream instead of under it
        const int k0 = wave * 8 + kg;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int key = k0 + 8 * NW * u;
            fok[u] = key < n_keys;
            fbv[u] = 0.f;
            const int kc = fok[u] ? key : 0;
            fku[u] = *reinterpret_cast<const u32x4*>(kb + (size_t)kc * DKV);
            fvu[u] = *reinterpret_cast<const u32x4*>(vb + (size_t)kc * DKV);
        }
    }
    if constexpr (FUSEQ) {
        if constexpr (OP) {
            // x = h + (p0 + p1 + ... + p7), then DG_RESID's sum(x^2) tree: (x_even^2 + x_odd^2), the 8 pairs of a 16-column
            // tile by xor 1, 2, 4 -- mul_sep / add_sep: the epilogue this mirrors multiplies (v_pk_mul) and adds, never an fma
            float sp = pv[0];
#pragma unroll
            for (int w = 1; w < 8; ++w) sp += pv[w];
            x_v += sp;
            float q2 = mul_sep(x_v, x_v);
            q2 = add_sep(q2, DPP_F(q2, 0xB1));        // lane ^ 1
            q2 = add_sep(q2, DPP_F(q2, 0x4E));        // lane ^ 2
            q2 = add_sep(q2, DPP_F(q2, 0x141));       // row_half_mirror: a lane of the other quad, which holds what lane ^ 4 holds
            q2 = add_sep(q2, DPP_F(q2, 0x128));       // row_ror:8 = lane ^ 8 within the row of 16
            if ((tid & 15) == 0) s_tile[tid >> 4] = q2;
            __syncthreads();
            ss = lane < SSQ_TILES ? s_tile[lane] : 0.f;              // every wave sums the 32 tiles itself: no broadcast of the scale, one barrier less
        }
        // norm scale (fixed-order tree over the 32 partials), normed row -> LDS as bf16-rounded floats
        ss = wave_sum(ss);
        float scale;
        if constexpr (OP) {
            scale = rsqrtf(ss / 512.f + a.eps);
        } else {                                                     // the carried partials were loaded by the first 32 threads only
            if (tid == 0) s_scale = rsqrtf(ss / 512.f + a.eps);
            __syncthreads();
            scale = s_scale;
        }
        xs[tid] = bf2f(f2bf(x_v * scale * g_v));
        __syncthreads();
        float xn[8];
        *reinterpret_cast<float4*>(xn) = *reinterpret_cast<const float4*>(xs + lane * 8);
        *reinterpret_cast<float4*>(xn + 4) = *reinterpret_cast<const float4*>(xs + lane * 8 + 4);
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
            d += DPP_F(d, 0x128);                                 // row_ror:8 -> lane ^ 8 within the row of 16
            dd[jj] = d;
        }
        // Every lane of a 16-lane row now holds its row's sum r0..r3.  The wave total is (r0 + r1) + (r2 + r3) -- what two butterfly steps
        // (lane ^ 16, lane ^ 32) gave every lane, through the LDS crossbar, sixteen dependent round trips for the eight outputs (0.8 us on
        // the critical path).  Row broadcasts (DPP) add the same pairs with the operands swapped -- IEEE addition commutes, same bits --
        // and leave the total in row 3.
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) dd[jj] += DPP_ROWS(dd[jj], 0x142, 0xA);       // row_bcast:15 -> rows 1 and 3: r1 + r0, r3 + r2
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) dd[jj] += DPP_ROWS(dd[jj], 0x143, 0xC);       // row_bcast:31 -> row 3: (r3 + r2) + (r1 + r0)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj)
            if (lane == 56 + jj) qs[wave * 8 + jj] = f2bf(dd[jj]);
        __syncthreads();
        qp = *reinterpret_cast<const u32x4*>(qs + sub * 8);
        // the block loaded above, then the remaining ones
        for (int kw = wave * 8; kw < n_keys; kw += 8 * NW * U) {
            if (kw != wave * 8) load_block(fku, fvu, fok, fbv, kw, std::false_type{});
            compute_block(fku, fvu, fok, fbv, std::false_type{});
        }
    }
    if constexpr (PAIR == 1) PAIR_MARK(a, 1, wall_clock64());     // self-attention stream consumed
    if constexpr (PAIR == 2) PAIR_MARK(a, 5, wall_clock64());     // cross-attention block consumed
    // merge the 8 key groups of the wave (lanes with equal `sub`)
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) {
        const float mo = lane_xor(m, off), lo = lane_xor(l, off);      // (register moves, not the LDS crossbar: common.h)
        const float mn = fmaxf(m, mo);
        const float sa = __expf(m - mn), sb = __expf(mo - mn);
        l = l * sa + lo * sb;
#pragma unroll
        for (int d = 0; d < 8; ++d) acc[d] = acc[d] * sa + lane_xor(acc[d], off) * sb;
        m = mn;
    }
    if (kg == 0) {
#pragma unroll
        for (int d = 0; d < 8; ++d) sacc[wave][sub * 8 + d] = acc[d];
        if (sub == 0) { sm[wave] = m; sl[wave] = l; }
    }
    __syncthreads();
    if (tid < DKV) {
        float M = sm[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) M = fmaxf(M, sm[w]);
        float L = 0.f, o = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float e = __expf(sm[w] - M);
            L += sl[w] * e;
            o += sacc[w][tid] * e;
        }
        if constexpr (OP && SELF) qs[tid] = f2bf(o / L);        // the projection's operand; nobody reads a.out in this mode
        else a.out[((size_t)r * H + h) * DKV + tid] = f2bf(o / L);
    }
    if constexpr (OP && SELF) {
        __syncthreads();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's own wo rows have landed (nobody else reads them)
        const int li = lane & 15, g = lane >> 4;
        const char* strip = wo_lds + wave * 8192;
        f32x4 pa[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) pa[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            // activation operand: row 0 of the 16-row tile is (r, h)'s output, rows 1..15 are zero (their results are not stored)
            bf16x8 af = *reinterpret_cast<const bf16x8*>(qs + ks * 32 + g * 8);
            if (li != 0) af = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int row = c * 16 + li;
                const bf16x8 wf = *reinterpret_cast<const bf16x8*>(strip + row * 128 + (((ks * 4 + g) ^ (row & 7)) * 16));
                pa[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af, pa[c], 0, 0, 0);
            }
        }
        if (li == 0) {
            float* dst = a.opart + ((size_t)r * H + h) * 512 + wave * 64 + g * 4;
            if constexpr (PAIR == 1) {           // read by the row's other heads, on other XCDs, later in this launch: agent scope (aux 16 = sc1)
                const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(a.opart, 0, 0x7fffffff, 0x00020000);
                const int off = (((r * H + h) * 512) + wave * 64 + g * 4) * 4;
#pragma unroll
                for (int c = 0; c < 4; ++c) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pa[c]), ro, off + c * 64, 0, 16);
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) *reinterpret_cast<float4*>(dst + c * 16) = make_float4(pa[c][0], pa[c][1], pa[c][2], pa[c][3]);
            }
        }
    }
    if constexpr (FUSEQ) {
        if (a.chain_sync && blockIdx.x == 0 && tid < CHAIN_COUNTERS) a.chain_sync[tid * CHAIN_LINE] = 0u;      // the next launch's arrival counters (dec_chain.hip)
    }
    if constexpr (PAIR == 1) PAIR_MARK(a, 2, wall_clock64());     // O-projection partial stored
    if constexpr (PAIR == 2) pair_leave(ps, r, left_before);
    if constexpr (PAIR != 1) STAMP_OUT(a);
}

template <bool SELF, bool FUSEQ, int NW, bool OP = false>
__global__ __launch_bounds__(64 * NW, 4) void dec_attn_kernel(const bf16_t* __restrict__ pK, const bf16_t* __restrict__ pV, const bf16_t* __restrict__ pQ,
                                                              const float* __restrict__ pF, const void* __restrict__ p4, const void* __restrict__ p5,
                                                              unsigned geom0, unsigned geom1, DecAttnArgs a) {   // 4 waves / SIMD -> <= 128 VGPRs
    attn_body<SELF, FUSEQ, NW, OP, 0>(pK, pV, pQ, pF, p4, p5, geom0, geom1, a, PairSync{});
}

// A layer's self-attention and cross-attention as ONE launch (up to 64 rows: 512 workgroups, two per CU, all resident).  Workgroup
// (row, head) streams its self-attention slab and leaves its O-projection partial (the folded form above), signals the row, requests
// its cross-attention operands, waits for the row's eight heads and goes on as the fused cross-attention.  What that saves is the
// launch gap and the cross-attention's start-up: its K/V stream is requested ~1 us after the self-attention's last bytes instead of
// after a kernel boundary and a dispatch ramp.  Same arithmetic as the two launches, bit for bit.
__global__ __launch_bounds__(512, 4) void dec_attn_pair_kernel(const bf16_t* __restrict__ pK, const bf16_t* __restrict__ pV, const bf16_t* __restrict__ pQ,
                                                               const float* __restrict__ pF, const void* __restrict__ p4, const void* __restrict__ p5,
                                                               unsigned geom0, unsigned geom1, DecAttnArgs a, DecAttnArgs b, unsigned gb0, unsigned gb1,
                                                               PairSync ps) {
    attn_body<true, false, 8, true, 1>(pK, pV, pQ, pF, p4, p5, geom0, geom1, a, ps);
    // the second half counts its loads against the first half's stores (see PAIR == 2): nothing may cross this line
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    attn_body<false, true, 8, true, 2>(b.k, b.v, b.wq, b.x_f32, b.gain, b.ipart, gb0, gb1, b, ps);
}

// ------------------------------------------------------------------------------------------------
// h[r] = embed row (+ channel row); sum(h^2) goes to ssq tile 0 of the row, tiles 1.. are zeroed
__device__ __forceinline__ void embed_row(const ArgmaxArgs& a, int r, const bf16_t* e, const bf16_t* c, float* scratch4) {
    const int tid = threadIdx.x;
    float q = 0.f;
    for (int i = tid; i < a.d; i += 256) {
        const float v = bf2f(e[i]) + (c ? bf2f(c[i]) : 0.f);
        a.h[(size_t)r * a.d + i] = v;
        q += v * v;
    }
    q = wave_sum(q);
    __syncthreads();                       // scratch4 may still be read by the caller's previous phase
    if ((tid & 63) == 0) scratch4[tid >> 6] = q;
    __syncthreads();
    if (tid < SSQ_TILES) a.ssq[(size_t)tid * a.ssq_stride + r] = tid == 0 ? (scratch4[0] + scratch4[1]) + (scratch4[2] + scratch4[3]) : 0.f;
}

__global__ __launch_bounds__(256) void argmax_embed_kernel(const float* __restrict__ pLogits, DecodeShared* pShared, int* pFinished, int* pRowPos,
                                                           const long long* __restrict__ pRowOut, const bf16_t* __restrict__ pEmbed, int row0, int V,
                                                           ArgmaxArgs a) {
    // leading scalar arguments: kernarg preload (see dec_gemm_kernel)
    __shared__ float sv[4];
    __shared__ int si[4];
    __shared__ int s_feed;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = row0 + blockIdx.x;
    DecodeShared* sh = pShared;
    const int t = sh->step, n_steps = sh->n_steps, col = t - sh->step0;   // col: index within this call
    const float* row = pLogits + (size_t)r * V;
    // everything thread 0 needs after the argmax is requested now (workgroup-uniform addresses), under the logits loads,
    // instead of as a chain of round trips behind them
    int32_t* const tokens_out = sh->tokens_out;
    const int32_t* const forced = sh->forced;
    float* const logits_out = sh->logits_out;
    const int was_finished = pFinished[r];
    const int forced_tok = forced ? forced[(size_t)r * n_steps + col] : 0;
    const int pos0 = pRowPos ? pRowPos[r] : 0;
    const long long out0 = pRowPos ? pRowOut[r] : 0;

    STAMP_IN(a);
    float bv = -3.4e38f;
    int bi = 0x7fffffff;
    for (int i = tid; i < V; i += 256) {
        const float v = row[i];
        if (v > bv) { bv = v; bi = i; }       // ascending i: the first maximum is kept
    }
    // wave-wide (max value, lowest index): the selection is commutative and associative, so any reduction order gives the same pair.
    // Rotations inside the 16-lane rows, then row broadcasts (all DPP; twelve dependent ds_bpermute round trips before): row 3 ends up
    // with the wave's pair.
#define ARGMAX_TAKE(ctrl, rows, cond)                                                                                      \
    do {                                                                                                                  \
        const float ov = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, bv), (ctrl), (rows), 0xF, false)); \
        const int oi = __builtin_amdgcn_update_dpp(0, bi, (ctrl), (rows), 0xF, false);                                    \
        if ((cond) && (ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }                                           \
    } while (0)
    ARGMAX_TAKE(0x128, 0xF, true);            // row_ror:8
    ARGMAX_TAKE(0x124, 0xF, true);            // row_ror:4
    ARGMAX_TAKE(0x122, 0xF, true);            // row_ror:2
    ARGMAX_TAKE(0x121, 0xF, true);            // row_ror:1: every lane holds its row's pair
    ARGMAX_TAKE(0x142, 0xA, (lane & 16) != 0);    // row_bcast:15 -> rows 1 and 3
    ARGMAX_TAKE(0x143, 0xC, lane >= 32);          // row_bcast:31 -> (rows 2 and) 3
#undef ARGMAX_TAKE
    if (lane == 63) { sv[wave] = bv; si[wave] = bi; }
    __syncthreads();
    if (tid == 0) { bv = sv[0]; bi = si[0]; }      // (the wave's pair sits in its last row)
    if (tid == 0 && pRowPos) {
        // slot mode: this row's own position; a stopped row writes nothing and stays where it is (its slot is refilled
        // by the host), a live one stops after EOS or its n_steps-th token
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
        int feed = a.pad_id;
        if (!was_finished) {
            const int p = pos0;
            tokens_out[out0 + p] = bi;
            if ((a.eos_id >= 0 && bi == a.eos_id) || p + 1 >= n_steps) pFinished[r] = 1;
            else pRowPos[r] = p + 1;
            feed = bi;
        }
        s_feed = feed;
    } else if (tid == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
        int tok = bi;
        if (a.eos_id >= 0) {
            if (was_finished) tok = a.pad_id;
            else if (tok == a.eos_id) pFinished[r] = 1;
        }
        tokens_out[(size_t)r * n_steps + col] = tok;
        int feed = forced ? forced_tok : tok;
        s_feed = feed < 0 ? 0 : (feed >= V ? V - 1 : feed);       // caller-supplied ids must not index outside the table
    }
    __syncthreads();
    const int feed = s_feed;
    const bf16_t* e = pEmbed + (size_t)feed * a.d;
    const bf16_t* c = a.chan_embed ? a.chan_embed + (size_t)(r % a.n_channels) * a.d : nullptr;
    embed_row(a, r, e, c, sv);
    if (logits_out && !pRowPos) {
        float* dst = logits_out + ((size_t)r * n_steps + col) * V;
        for (int i = tid; i < V; i += 256) dst[i] = row[i];
    }
    // the per-step kernel's arrival counters (dec_step.hip), left at zero for the next step's launch
    if (a.zero_sync)
        for (int i = blockIdx.x * 256 + tid; i < a.zero_lines; i += gridDim.x * 256) a.zero_sync[(size_t)i * CHAIN_LINE] = 0u;
    // the last workgroup to finish advances the position; every workgroup has read `t` by then
    __syncthreads();
    if (tid == 0) {
        if (a.eos_id >= 0) __threadfence();      // only the flags the last arriver counts below need to be visible to it
        // Who is last?  One atomic per workgroup on ONE word serialises at the memory side (~12 ns each: 10 us of the 13-channel decoder's
        // 832 workgroups); beyond 64 workgroups groups of 32 count on lines of their own and only each group's last one takes the shared ticket.
        bool last;
        if (a.ticket && gridDim.x > 64) {
            const int grp = blockIdx.x >> 5, n_grp = ((int)gridDim.x + 31) >> 5, in_grp = min(32, (int)gridDim.x - 32 * grp);
            last = false;
            if (atomicAdd(a.ticket + (size_t)grp * CHAIN_LINE, 1u) == (unsigned)(in_grp - 1)) {
                a.ticket[(size_t)grp * CHAIN_LINE] = 0u;
                if (a.eos_id >= 0) __threadfence();
                last = atomicAdd(&sh->done_count, 1) == n_grp - 1;
            }
        } else {
            last = atomicAdd(&sh->done_count, 1) == (int)gridDim.x - 1;
        }
        if (last) {
            sh->done_count = 0;
            sh->step = t + 1;
            if (a.eos_id >= 0) {             // the fence above makes every row's flag visible to this last arriver
                int n = 0;
                for (int i = 0; i < (int)gridDim.x; ++i) n += pFinished[row0 + i] ? 0 : 1;
                sh->n_unfinished = n;
            }
        }
    }
    STAMP_OUT(a);
}

__global__ __launch_bounds__(256) void decode_init_kernel(ArgmaxArgs a, int n_chains, int n_steps, int step0, int32_t* tokens_out,
                                                          const int32_t* forced, float* logits_out) {
    const int r = blockIdx.x, tid = threadIdx.x;
    const bf16_t* e = a.embed + (size_t)a.pad_id * a.d;
    const bf16_t* c = a.chan_embed ? a.chan_embed + (size_t)(r % a.n_channels) * a.d : nullptr;
    __shared__ float sv[4];
    embed_row(a, r, e, c, sv);
    if (tid == 0) a.finished[r] = 0;
    if (r == 0 && tid < n_chains) {          // a.shared is the array of per-chain loop states
        DecodeShared* sh = a.shared + tid;
        sh->step = step0;
        sh->step0 = step0;
        sh->done_count = 0;
        sh->n_unfinished = a.R;
        sh->n_steps = n_steps;
        sh->tokens_out = tokens_out;
        sh->forced = forced;
        sh->logits_out = logits_out;
    }
}

__global__ __launch_bounds__(256) void slot_start_kernel(ArgmaxArgs a, int row0, long long first_out, int n_steps, long long* row_out) {
    const int r = row0 + blockIdx.x, tid = threadIdx.x;
    const bf16_t* e = a.embed + (size_t)a.pad_id * a.d;
    const bf16_t* c = a.chan_embed ? a.chan_embed + (size_t)(r % a.n_channels) * a.d : nullptr;
    __shared__ float sv[4];
    embed_row(a, r, e, c, sv);
    if (tid == 0) {
        a.finished[r] = 0;
        a.row_pos[r] = 0;
        row_out[r] = first_out + (long long)blockIdx.x * n_steps;
    }
}

__global__ void slot_retire_kernel(ArgmaxArgs a, int row0, int n_steps, int32_t* tokens_out) {
    const int r = row0 + blockIdx.x;
    int32_t* row = tokens_out + a.row_out[r];
    for (int i = a.row_pos[r] + 1 + threadIdx.x; i < n_steps; i += blockDim.x) row[i] = a.pad_id;
}

__global__ void pad_tail_kernel(int32_t* tokens_out, int row0, int n_steps, int from, int pad_id) {
    int32_t* row = tokens_out + (size_t)(row0 + blockIdx.x) * n_steps;
    for (int i = from + threadIdx.x; i < n_steps; i += blockDim.x) row[i] = pad_id;
}

template <int MODE, int K, int NT, bool PEND = false>
constexpr size_t dg_lds_bytes() {
    return (size_t)(8 * 16 * 16 * NT + 16 + (PEND ? 8 * 16 : 0)) * 4 + (size_t)8 * (1 + NT) * 16 * (K / 8 * 2 + 16);
}

template <int MODE, int K, int NT = 1, bool PEND = false>
int launch_dg(const DecGemmArgs& a, hipStream_t stream) {
    if (a.W == nullptr)     // attribute-only call from init_decode_kernels(): > 64 KB of dynamic LDS needs opting in
        return hipFuncSetAttribute(reinterpret_cast<const void*>(dec_gemm_kernel<MODE, K, NT, PEND>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)dg_lds_bytes<MODE, K, NT, PEND>()) == hipSuccess ? 0 : -2;
    if (MODE == DG_RESID && a.N != 16 * SSQ_TILES) return -3;   // the norm consumers sum exactly SSQ_TILES partials
    if (PEND && (!a.pend_y || (MODE == DG_NORM_QKV_CACHE && !a.h_out))) return -4;
    dec_gemm_kernel<MODE, K, NT, PEND><<<(a.N / (16 * NT)) * ((a.R + 15) / 16), 512, dg_lds_bytes<MODE, K, NT, PEND>(), stream>>>(
        a.W, MODE == DG_RESID ? static_cast<const void*>(a.a_bf16) : static_cast<const void*>(a.x_f32), a.gain, a.ssq, a.out_f32, a.row0, a.R, a.N,
        a.ssq_stride, a);
    return 0;
}

// wide projections: 32 columns per workgroup once 16-column tiles would put more than one workgroup on a CU
template <int MODE, bool PEND = false>
int launch_dg_wide(const DecGemmArgs& a, hipStream_t stream) {
    static const bool narrow = getenv("YMT3_DEC_GEMM_NARROW") != nullptr;      // A/B timing only
    const int wgs16 = (a.N / 16) * ((a.R + 15) / 16);
    if (!narrow && a.N % 32 == 0 && wgs16 > 320) return launch_dg<MODE, 512, 2, PEND>(a, stream);
    return launch_dg<MODE, 512, 1, PEND>(a, stream);
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
    rc |= launch_dg<DG_NORM_QKV_CACHE, 512, 2>(z, nullptr);
    rc |= launch_dg<DG_NORM_BF16, 512, 2>(z, nullptr);
    rc |= launch_dg<DG_NORM_BF16_RELU, 512, 2>(z, nullptr);
    rc |= launch_dg<DG_NORM_LOGITS, 512, 2>(z, nullptr);
    rc |= launch_dg<DG_NORM_QKV_CACHE, 512, 1, true>(z, nullptr);
    rc |= launch_dg<DG_NORM_QKV_CACHE, 512, 2, true>(z, nullptr);
    rc |= launch_dg<DG_NORM_LOGITS, 512, 1, true>(z, nullptr);
    rc |= launch_dg<DG_NORM_LOGITS, 512, 2, true>(z, nullptr);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(dec_attn_pair_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, WO_LDS_BYTES) != hipSuccess) rc |= -2;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(dec_attn_kernel<true, false, 8, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            WO_LDS_BYTES) != hipSuccess) rc |= -2;
    return rc;
}

template <int MODE, int K, int BM>
int launch_dg_mid(const DecGemmArgs& a, hipStream_t stream) {
    if (MODE == DG_RESID && a.N != 16 * SSQ_TILES) return -3;
    constexpr int PF = 4;     // K-steps of operands in flight per thread (8 = all of K = 512 and 2 were measured: profiles/r03_notes.md)
    constexpr int NTH = 256;  // (512 -- two waves per SIMD, half the row tiles each -- was measured: K = 512 forms 0.5 us slower, K = 2048 twice as slow: profiles/r03_notes.md)
    dec_gemm_mid_kernel<MODE, K, BM, PF, NTH><<<(a.N / 64) * ((a.R + BM - 1) / BM), NTH, 0, stream>>>(
        a.W, MODE == DG_RESID ? static_cast<const void*>(a.a_bf16) : static_cast<const void*>(a.x_f32), a.gain, a.ssq, a.out_f32, a.row0, a.R, a.N,
        a.ssq_stride, a);
    return 0;
}

// many rows: the mid-size tile kernel from DEC_GEMM_MID_ROWS rows on (kernels.h; a handle created under YMT3_DEC_GEMM_MID_ROWS=n passes its own
// threshold in DecGemmArgs::mid_rows, 0 = never: A/B timing, and the late-position parity test of the many-row kernels at a few rows)
static int launch_dec_gemm_mid(int mode, const DecGemmArgs& a, hipStream_t stream) {
    if (a.N % 64 || a.part) return -1;
    switch (mode) {
        case DG_RESID:
            if (a.K == 512) return launch_dg_mid<DG_RESID, 512, 32>(a, stream);
            if (a.K == 2048) return launch_dg_mid<DG_RESID, 2048, 32>(a, stream);
            return -1;
        // (32-row tiles for the wide projections too: 154.6 vs 156.4 ms per configs[3] batch -- no difference; 64 stays)
        case DG_NORM_QKV_CACHE: return a.K == 512 ? launch_dg_mid<DG_NORM_QKV_CACHE, 512, 64>(a, stream) : -1;
        case DG_NORM_BF16: return a.K == 512 ? launch_dg_mid<DG_NORM_BF16, 512, 32>(a, stream) : -1;
        case DG_NORM_BF16_RELU: return a.K == 512 ? launch_dg_mid<DG_NORM_BF16_RELU, 512, 64>(a, stream) : -1;
        case DG_NORM_LOGITS: return a.K == 512 ? launch_dg_mid<DG_NORM_LOGITS, 512, 64>(a, stream) : -1;
        default: return -1;
    }
}

int launch_dec_gemm(int mode, const DecGemmArgs& a, hipStream_t stream) {
    if (a.R <= 0) return 0;
    if (a.N % 16) return -1;
    const int mid_rows = a.mid_rows >= 0 ? a.mid_rows : DEC_GEMM_MID_ROWS;
    if (mid_rows > 0 && a.R >= mid_rows && a.N % 64 == 0 && !a.part && !a.pend_y && (a.K == 512 || (a.K == 2048 && mode == DG_RESID)))
        return launch_dec_gemm_mid(mode, a, stream);
    if (mode == DG_RESID) {
        if (a.K == 512) return launch_dg<DG_RESID, 512>(a, stream);
        if (a.K == 2048) return launch_dg<DG_RESID, 2048>(a, stream);
        if (a.K == 1024) return launch_dg<DG_RESID, 1024>(a, stream);
        return -1;
    }
    if (a.K != 512) return -1;
    if (a.pend_y) {           // a pending MoE combine: only the two kernels that can follow an MoE FFN take it
        if (mode == DG_NORM_QKV_CACHE) return launch_dg_wide<DG_NORM_QKV_CACHE, true>(a, stream);
        if (mode == DG_NORM_LOGITS) return launch_dg_wide<DG_NORM_LOGITS, true>(a, stream);
        return -1;
    }
    switch (mode) {
        case DG_NORM_QKV_CACHE: return launch_dg_wide<DG_NORM_QKV_CACHE>(a, stream);
        case DG_NORM_BF16: return launch_dg_wide<DG_NORM_BF16>(a, stream);
        case DG_NORM_BF16_RELU: return launch_dg_wide<DG_NORM_BF16_RELU>(a, stream);
        case DG_NORM_LOGITS: return launch_dg_wide<DG_NORM_LOGITS>(a, stream);
        default: return -1;
    }
}

int launch_dec_attention(bool self_attn, const DecAttnArgs& a, hipStream_t stream) {
    if (a.R <= 0) return 0;
    const bool many = a.force_many || a.R * a.H > 2048;
    if (a.chain_sync && (self_attn || !a.wq)) return -1;                              // only the fused cross-attention zeroes the chain counters
    if ((a.wo || a.ipart) && (many || a.H != 8 || (a.wo && !a.opart) || (a.ipart && !a.wq))) return -1;   // folded O-projection: 8 heads, 8-wave kernels
    if (a.row0 < 0 || a.row0 > 0xffff || a.rows_per_kv < 1 || a.rows_per_kv > 0xff || a.H < 1 || a.H > 0xff || a.slab_keys < 1 || a.slab_keys > 0xfffff ||
        a.n_keys_const < 0 || a.n_keys_const > 0xfff || (self_attn && a.bias_stride != a.slab_keys))
        return -1;                                                                     // the packed geometry words
    const unsigned g0 = (unsigned)a.row0 | (unsigned)a.rows_per_kv << 16 | (unsigned)a.H << 24, g1 = (unsigned)a.slab_keys | (unsigned)a.n_keys_const << 20;
    const int grid = a.R * a.H;
#define ATTN(S, F, W, O, THREADS, LDS, PQ, PF, P4, P5) dec_attn_kernel<S, F, W, O><<<grid, THREADS, LDS, stream>>>(a.k, a.v, PQ, PF, P4, P5, g0, g1, a)
    if (self_attn) {
        if (a.wo) ATTN(true, false, 8, true, 512, WO_LDS_BYTES, a.q, a.bias, a.shared, a.row_pos);
        else if (many) ATTN(true, false, 2, false, 128, 0, a.q, a.bias, a.shared, a.row_pos);
        else ATTN(true, false, 8, false, 512, 0, a.q, a.bias, a.shared, a.row_pos);
    } else if (a.wq) {
        // the fused projection maps one thread per d_model element
        if (a.ipart) ATTN(false, true, 8, true, 512, 0, a.wq, a.x_f32, a.gain, a.ipart);
        else ATTN(false, true, 8, false, 512, 0, a.wq, a.x_f32, a.gain, a.ssq);
    } else {
        if (many) ATTN(false, false, 2, false, 128, 0, a.q, nullptr, nullptr, nullptr);
        else ATTN(false, false, 8, false, 512, 0, a.q, nullptr, nullptr, nullptr);
    }
#undef ATTN
    return 0;
}

// the attention pair's 8 x R <= 512 workgroups must all be resident at once: two per CU at its LDS / register footprint
bool dec_attention_pair_fits(int n_cus) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, dec_attn_pair_kernel, 512, WO_LDS_BYTES) != hipSuccess) return false;
    return (long long)n * n_cus >= 512;
}

// self-attention (folded O-projection) + fused cross-attention of one layer as one launch; < 0: not this kernel's shape
int launch_dec_attention_pair(const DecAttnArgs& a, const DecAttnArgs& b, unsigned* pair_rows, unsigned* abort_word, unsigned* host_abort,
                              hipStream_t stream) {
    if (a.R <= 0) return 0;
    if (a.R > 16 * CHAIN_TILES_MAX || a.R != b.R || a.row0 != b.row0 || a.H != 8 || b.H != 8 || !a.wo || !a.opart || !b.wq || b.ipart != a.opart || !pair_rows || !abort_word ||
        a.rows_per_kv != 1 || b.rows_per_kv < 1 || b.rows_per_kv > 0xff || a.row0 < 0 || a.row0 > 0xffff || a.slab_keys < 1 || a.slab_keys > 0xfffff ||
        b.slab_keys < 1 || b.slab_keys > 0xfffff || b.n_keys_const < 1 || b.n_keys_const > 0xfff || a.bias_stride != a.slab_keys)
        return -1;
    const unsigned g0 = (unsigned)a.row0 | 1u << 16 | 8u << 24, g1 = (unsigned)a.slab_keys;
    const unsigned h0 = (unsigned)b.row0 | (unsigned)b.rows_per_kv << 16 | 8u << 24, h1 = (unsigned)b.slab_keys | (unsigned)b.n_keys_const << 20;
    dec_attn_pair_kernel<<<a.R * 8, 512, WO_LDS_BYTES, stream>>>(a.k, a.v, a.q, a.bias, a.shared, a.row_pos, g0, g1, a, b, h0, h1,
                                                                 PairSync{pair_rows, abort_word, host_abort});
    return 0;
}

int launch_argmax_embed(const ArgmaxArgs& a, hipStream_t stream) {
    if (a.R <= 0) return 0;
    argmax_embed_kernel<<<a.R, 256, 0, stream>>>(a.logits, a.shared, a.finished, a.row_pos, a.row_out, a.embed, a.row0, a.V, a);
    return 0;
}

int launch_decode_init(const ArgmaxArgs& a, int n_chains, int n_steps, int step0, int32_t* tokens_out, const int32_t* forced,
                       float* logits_out, hipStream_t stream) {
    if (a.R <= 0) return 0;
    decode_init_kernel<<<a.R, 256, 0, stream>>>(a, n_chains, n_steps, step0, tokens_out, forced, logits_out);
    return 0;
}

int launch_slot_start(const ArgmaxArgs& a, int row0, long long first_out, int n_steps, long long* row_out, hipStream_t stream) {
    if (!a.row_pos || !row_out || a.n_channels <= 0) return -1;
    slot_start_kernel<<<a.n_channels, 256, 0, stream>>>(a, row0, first_out, n_steps, row_out);
    return 0;
}

int launch_slot_retire(const ArgmaxArgs& a, int row0, int n_rows, int n_steps, int32_t* tokens_out, hipStream_t stream) {
    if (!a.row_pos || !a.row_out || n_rows <= 0) return -1;
    slot_retire_kernel<<<n_rows, 256, 0, stream>>>(a, row0, n_steps, tokens_out);
    return 0;
}

int launch_pad_tail(int32_t* tokens_out, int row0, int R, int n_steps, int from, int pad_id, hipStream_t stream) {
    if (R <= 0 || from >= n_steps) return 0;
    pad_tail_kernel<<<R, 256, 0, stream>>>(tokens_out, row0, n_steps, from, pad_id);
    return 0;
}
