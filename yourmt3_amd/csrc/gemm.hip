// Dense bf16 MFMA GEMM for the encoder-side projections (a3, a4, a6 of SURVEY.md section 8):
//   C[M][N] (+)= A[M][K] * W[N][K]^T, fp32 accumulate, M = segments x frames (thousands of rows).
//
// 128 x 128 x 64 tile, 256 threads = 4 waves in a 2 x 2 grid, each wave a 64 x 64 sub-tile as
// 4 x 4 v_mfma_f32_16x16x32_bf16 accumulators.  Operand roles are swapped (W fragment as the MFMA
// A operand, activation fragment as B) so every lane ends up with 4 CONSECUTIVE output columns of
// one row: the epilogue stores 8/16 bytes per lane instead of 2-byte scatters.  Tiles go global -> LDS
// directly (global_load_lds_dwordx4, 16-byte chunks XOR-swizzled by row through the SOURCE address so that
// ds_read_b128 spreads over the banks), double buffered with the next tile's DMA in flight under the MFMAs.
// (A register-staged variant of the same loop measured 2 % slower and 28 VGPRs fatter; a three-tile register
// prefetch changed nothing: the loop is bound by its LDS-read / MFMA / barrier structure at 2 workgroups per CU.)
// Fused epilogues: bias, ReLU, bf16 rounding, residual add, head-major cross-KV scatter.
//
// Oracle: oracle/ymt3_oracle.py (input_projection / encoder_t5 / cross_kv);
// arithmetic TP: transformers/models/t5/modeling_t5.py:75-94 (FFN), :304-326 (q/k/v/o linears).
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;

__device__ __forceinline__ int swz(int row, int chunk) { return row * 8 + (chunk ^ (row & 7)); }  // 16-byte units

// epilogue of one wave's 64 x 64 sub-tile at (mw, nw): lane holds rows m = mw + mt*16 + (lane & 15), columns
// n = nw + nt*16 + (lane >> 4) * 4 + {0..3}
template <int EPI>
__device__ __forceinline__ void store_tile(const GemmArgs& g, const f32x4 (&acc)[4][4], int mw, int nw, int lane) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = mw + mt * 16 + (lane & 15);
        if (m >= g.M) continue;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = nw + nt * 16 + (lane >> 4) * 4;
            f32x4 v = acc[nt][mt];
            if constexpr (EPI == EPI_F32) {
                if (g.bias) {
                    const float4 b = *reinterpret_cast<const float4*>(g.bias + n);
                    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                }
                *reinterpret_cast<float4*>(static_cast<float*>(g.out) + (size_t)m * g.ldc + n) =
                    make_float4(v[0], v[1], v[2], v[3]);
            } else if constexpr (EPI == EPI_RESID) {
                float4* p = reinterpret_cast<float4*>(static_cast<float*>(g.out) + (size_t)m * g.ldc + n);
                float4 o = *p;
                o.x += v[0]; o.y += v[1]; o.z += v[2]; o.w += v[3];
                *p = o;
            } else {
                if constexpr (EPI == EPI_BF16_RELU) {
                    v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
                }
                const uint2 pk = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                size_t off;
                if constexpr (EPI == EPI_KV_HEADMAJOR) {
                    const int hd = g.H * 64;
                    const int slab = n / hd, h = (n % hd) >> 6, dd = n & 63;
                    const int seg = m / g.T, t = m % g.T;
                    off = ((((size_t)slab * g.n_seg + seg) * g.H + h) * g.T + t) * 64 + dd;
                } else {
                    off = (size_t)m * g.ldc + n;
                }
                *reinterpret_cast<uint2*>(static_cast<bf16_t*>(g.out) + off) = pk;
            }
        }
    }
}

// The same sub-tile for the bf16 epilogues, turned through a wave-private 4 KB LDS region so that every store
// instruction writes 8 whole 128-byte rows (16 B per lane) instead of 16 rows x 32 B: the direct form costs a third of a
// K = 512 GEMM (tools/gemm_probe.cpp).  Two halves of 32 rows; 16-byte chunks XOR-swizzled by row (writes 2-way, reads
// conflict free).  Only the owning wave touches the region and LDS executes a wave's operations in order: no barrier.
template <int EPI>
__device__ __forceinline__ void store_tile_lds(const GemmArgs& g, const f32x4 (&acc)[4][4], int mw, int nw, int lane, char* region) {
    static_assert(EPI == EPI_BF16 || EPI == EPI_BF16_RELU || EPI == EPI_KV_HEADMAJOR, "bf16 epilogues only");
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int mh = 0; mh < 2; ++mh) {
            const int r = mh * 16 + (lane & 15);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                f32x4 v = acc[nt][half * 2 + mh];
                if constexpr (EPI == EPI_BF16_RELU) {
                    v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
                }
                const int p = nt * 4 + (lane >> 4);                      // 8-byte piece of the 128-byte row
                *reinterpret_cast<uint2*>(region + r * 128 + (((p >> 1) ^ (r & 7)) << 4) + ((p & 1) << 3)) =
                    make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = i * 8 + (lane >> 3), c = lane & 7;
            const u32x4 val = *reinterpret_cast<const u32x4*>(region + r * 128 + ((c ^ (r & 7)) << 4));
            const int m = mw + half * 32 + r;
            if (m >= g.M) continue;
            size_t off;
            if constexpr (EPI == EPI_KV_HEADMAJOR) {
                const int hd = g.H * 64;
                const int slab = nw / hd, h = (nw % hd) >> 6;             // nw is a multiple of 64: the wave's columns are one head
                const int seg = m / g.T, t = m % g.T;
                off = ((((size_t)slab * g.n_seg + seg) * g.H + h) * g.T + t) * 64 + c * 8;
            } else {
                off = (size_t)m * g.ldc + nw + c * 8;
            }
            *reinterpret_cast<u32x4*>(static_cast<bf16_t*>(g.out) + off) = val;
        }
    }
}

// Residual epilogue (out fp32 += acc) of the same sub-tile: the residual rows were prefetched at the start of the tile in
// ROW-MAJOR register order (hold[q*4+i]: row q*16 + i*4 + (lane>>4), 16-byte chunk lane&15 of the wave's 256-byte row
// span: whole lines, in flight under the K loop); the accumulators are turned into that order through the wave-private
// 4 KB region a quarter (16 rows) at a time, added, and stored 4 rows x 256 B per instruction.
__device__ __forceinline__ void load_resid_rows(const GemmArgs& g, f32x4 (&hold)[16], int mw, int nw, int lane) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        int m = mw + j * 4 + (lane >> 4);
        m = m < g.M ? m : g.M - 1;
        hold[j] = *reinterpret_cast<const f32x4*>(static_cast<const float*>(g.out) + (size_t)m * g.ldc + nw + (lane & 15) * 4);
    }
}
__device__ __forceinline__ void store_tile_resid_lds(const GemmArgs& g, const f32x4 (&acc)[4][4], const f32x4 (&hold)[16], int mw, int nw,
                                                     int lane, char* region) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = lane & 15;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int c = nt * 4 + (lane >> 4);                          // 16-byte chunk of the 256-byte row
            *reinterpret_cast<f32x4*>(region + r * 256 + ((c ^ r) << 4)) = acc[nt][q];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rr = i * 4 + (lane >> 4), c = lane & 15;
            f32x4 v = *reinterpret_cast<const f32x4*>(region + rr * 256 + ((c ^ rr) << 4));
            const int m = mw + q * 16 + rr;
            if (m >= g.M) continue;
            v += hold[q * 4 + i];
            *reinterpret_cast<f32x4*>(static_cast<float*>(g.out) + (size_t)m * g.ldc + nw + c * 4) = v;
        }
    }
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // first-class vectors: HIP's uint4 struct arrays went to scratch
    __shared__ u32x4 sA[2][BM * 8];
    __shared__ u32x4 sW[2][BN * 8];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order: consecutive block ids round-robin over the 8 XCDs, so give each XCD a
    // contiguous run of tiles that share W panels (speed only; any order is correct).
    const int nbn = g.N / BN, nbm = (g.M + BM - 1) / BM, nwg = nbn * nbm;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    }
    const int m0 = (bid / nbn) * BM, n0 = (bid % nbn) * BN;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = g.K / BK;
    {
        // direct global -> LDS (global_load_lds_dwordx4): no staging registers, no ds_write pass.  One wave-instruction
        // writes 1 KiB linearly = 8 tile rows x 128 B; the XOR swizzle therefore sits on the per-lane SOURCE address
        // (lane -> row l>>3, LDS chunk l&7 holds global chunk (l&7)^(row&7)) and on the fragment reads -- never on the
        // LDS destination.  Each wave stages rows [32w, 32w+32) of both operands.
        auto issue = [&](int kt, int buf) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row8 = wave * 32 + i * 8, row = row8 + (lane >> 3), ch = (lane & 7) ^ (lane >> 3);
                int am = m0 + row;
                am = am < g.M ? am : g.M - 1;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g.A + (size_t)am * g.lda + kt * BK + ch * 8),
                                                 (__attribute__((address_space(3))) void*)&sA[buf][row8 * 8], 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g.W + (size_t)(n0 + row) * g.ldw + kt * BK + ch * 8),
                                                 (__attribute__((address_space(3))) void*)&sW[buf][row8 * 8], 16, 0, 0);
            }
        };
        issue(0, 0);
        __syncthreads();                    // hipcc drains the LDS-DMA (vmcnt(0)) in front of the barrier
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) issue(kt + 1, buf ^ 1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 fa[4], fw[4];
                const int ch = ks * 4 + (lane >> 4);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int ar = wm * 64 + t * 16 + (lane & 15);
                    const int wr = wn * 64 + t * 16 + (lane & 15);
                    fa[t] = __builtin_bit_cast(bf16x8, sA[buf][swz(ar, ch)]);
                    fw[t] = __builtin_bit_cast(bf16x8, sW[buf][swz(wr, ch)]);
                }
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[nt], fa[mt], acc[nt][mt], 0, 0, 0);
            }
            __syncthreads();
        }
    }

#ifdef YMT3_PROBE_NO_STORE
    if (acc[0][0][0] != 12345.678f) return;                              // timing-only build: keep the math, drop the stores
#endif
    store_tile<EPI>(g, acc, m0 + wm * 64, n0 + wn * 64, lane);
}


// ------------------------------------------------------------------------------------------------
// Large-M variant: 256 x 128 x 64 tile, 512 threads = 8 waves as 4 (M) x 2 (N), same 64 x 64 wave sub-tile and
// fragment order as above, ONE persistent workgroup per CU walking its share of the tiles, with a ring of three LDS
// stages (3 x 48 KB) that runs straight through tile boundaries: the DMAs of K-step s+2 -- of the NEXT tile during a
// tile's last two steps -- are issued at the top of step s, so the epilogue's stores (a third of the time of a
// K = 512 GEMM when nothing overlaps them: tools/gemm_probe.cpp) drain under the next tile's loads and MFMAs.
// The stage DMAs stay in flight ACROSS the per-step barrier: the global_load_lds instructions are inline asm (hipcc,
// not seeing an LDS write it must order, inserts no vmcnt(0) in front of the fragment reads or the barrier),
// completion is a counted `s_waitcnt vmcnt(6)` (6 DMAs per wave and stage: the newest stage stays in flight; stores
// issued before it are older and so retired by the same count), visibility to the other waves the raw s_barrier that
// follows, and a stage is read only in the step AFTER that wait + barrier.  A stage is overwritten two steps after
// its last read, behind a barrier every wave passed with its LDS reads retired (lgkmcnt(0)).
constexpr int TM = 256, TN = 128, NST = 3;
constexpr int STAGE_BYTES = (TM + TN) * BK * 2;      // 48 KB: A rows, then W rows, 128 B each

#define WAIT_DMA_AND_BARRIER(n) asm volatile("s_waitcnt vmcnt(" #n ")\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <int EPI>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_big_kernel(GemmArgs g) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    // tiles of this workgroup: blocks with equal blockIdx % 8 (one XCD under round-robin placement; speed only) share a
    // contiguous run of tiles, n fastest, and walk it together, so an XCD's L2 holds one or two A row-tiles and W
    const int nbn = g.N / TN, nbm = (g.M + TM - 1) / TM, n_tiles = nbn * nbm, G = gridDim.x;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tq = n_tiles / 8, tr = n_tiles % 8;
    const int t_first = (xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq) + slot;
    const int t_end = (xcd < tr ? (xcd + 1) * (tq + 1) : tr * (tq + 1) + (xcd + 1 - tr) * tq);
    const int t_stride = G / 8 + (xcd < G % 8 ? 1 : 0);
    const int n_my = t_first < t_end ? (t_end - t_first + t_stride - 1) / t_stride : 0;
    const int nk = g.K / BK, total = n_my * nk;
    if (total == 0) return;

    // loader state: runs two K-steps ahead of the MFMAs, through tile boundaries.  This wave's DMA pieces (1 KB = 8 rows
    // x 128 B each): A pieces 4w..4w+3, W pieces 2w, 2w+1; per lane source row l>>3 of the piece, source chunk
    // (l&7) ^ (row&7) = (l&7) ^ (l>>3): the XOR swizzle sits on the source address
    const bf16_t* srcA[4];
    const bf16_t* srcW[2];
    const int sch = ((lane & 7) ^ (lane >> 3)) * 8;
    auto point_at = [&](int tile) {
        const int m0 = (tile / nbn) * TM, n0 = (tile % nbn) * TN;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int am = m0 + (wave * 4 + i) * 8 + (lane >> 3);
            am = am < g.M ? am : g.M - 1;
            srcA[i] = g.A + (size_t)am * g.lda + sch;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) srcW[i] = g.W + (size_t)(n0 + (wave * 2 + i) * 8 + (lane >> 3)) * g.ldw + sch;
    };
    int l_tile = t_first, l_kt = 0, l_st = 0, l_left = total;
    point_at(l_tile);
    auto issue_next = [&]() {
        if (l_left == 0) return;
        const unsigned base = lds0 + (unsigned)l_st * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(srcA[i] + l_kt * BK, base + (unsigned)(wave * 4 + i) * 1024u);
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(srcW[i] + l_kt * BK, base + (unsigned)(TM * 128) + (unsigned)(wave * 2 + i) * 1024u);
        --l_left;
        l_st = l_st == NST - 1 ? 0 : l_st + 1;
        if (++l_kt == nk) {
            l_kt = 0;
            l_tile += t_stride;
            if (l_left) point_at(l_tile);
        }
    };

    issue_next();
    issue_next();
    if (total > 1) WAIT_DMA_AND_BARRIER(6); else WAIT_DMA_AND_BARRIER(0);
    int st = 0, left = total;                                            // compute side: stage being read, steps not yet computed
    // The fragments of a K-step's two 32-wide halves are read one half AHEAD of the MFMAs that use them: half 1 of a step under the MFMAs of its
    // half 0, half 0 of the NEXT step (its stage visible since the barrier in the middle of this one) under the MFMAs of half 1 -- also through
    // tile boundaries and epilogues.  (hipcc's own schedule kept 24 fragment registers and waited for an LDS read in front of every group of
    // four MFMAs: 0.28-0.30 MFMA utilisation in situ, profiles/r03_pmc_encoder.json.)
    auto read_frags = [&](int stage, int ks, bf16x8 (&fa)[4], bf16x8 (&fw)[4]) {
        const u32x4* sA = reinterpret_cast<const u32x4*>(smem + stage * STAGE_BYTES);
        const u32x4* sW = sA + TM * 8;
        const int ch = ks * 4 + (lane >> 4);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            fa[t] = __builtin_bit_cast(bf16x8, sA[swz(wm * 64 + t * 16 + (lane & 15), ch)]);
            fw[t] = __builtin_bit_cast(bf16x8, sW[swz(wn * 64 + t * 16 + (lane & 15), ch)]);
        }
    };
    bf16x8 fa0[4], fw0[4];
    read_frags(0, 0, fa0, fw0);
    for (int tile = t_first; tile < t_end; tile += t_stride) {
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 hold[EPI == EPI_RESID ? 16 : 1];
        if constexpr (EPI == EPI_RESID) load_resid_rows(g, hold, (tile / nbn) * TM + wm * 64, (tile % nbn) * TN + wn * 64, lane);
        for (int kt = 0; kt < nk; ++kt) {
#ifndef YMT3_PROBE_NO_DMA
            issue_next();                                                // step + 2: into the stage read one step ago
#endif
            bf16x8 fa1[4], fw1[4];
            read_frags(st, 1, fa1, fw1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw0[nt], fa0[mt], acc[nt][mt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            --left;
            if (left > 0) {                                              // the next step's stage: landed (mine) and visible (barrier)
                if (left > 1) WAIT_DMA_AND_BARRIER(6); else WAIT_DMA_AND_BARRIER(0);
            }
            st = st == NST - 1 ? 0 : st + 1;
            if (left > 0) read_frags(st, 0, fa0, fw0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw1[nt], fa1[mt], acc[nt][mt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#ifdef YMT3_PROBE_NO_STORE
        if (acc[0][0][0] != 12345.678f) continue;                        // timing-only build: keep the math, drop the stores
#endif
        if constexpr (EPI == EPI_RESID) {
            if (left == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const int prev = st == 0 ? NST - 1 : st - 1;
            store_tile_resid_lds(g, acc, hold, (tile / nbn) * TM + wm * 64, (tile % nbn) * TN + wn * 64, lane,
                                 smem + prev * STAGE_BYTES + wave * 4096);
        } else if constexpr (EPI == EPI_BF16 || EPI == EPI_BF16_RELU || EPI == EPI_KV_HEADMAJOR) {
            // the stage read in the tile's last step is free until the loader's next issue, and this wave's own A pieces
            // of it (4 KB) are bytes no other wave's DMA ever writes: the staging region of the turned store.  Every wave
            // has finished reading that stage at the barrier that closed the step (added here when the step was the last)
            if (left == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const int prev = st == 0 ? NST - 1 : st - 1;
            store_tile_lds<EPI>(g, acc, (tile / nbn) * TM + wm * 64, (tile % nbn) * TN + wn * 64, lane,
                                smem + prev * STAGE_BYTES + wave * 4096);
        } else {
            store_tile<EPI>(g, acc, (tile / nbn) * TM + wm * 64, (tile % nbn) * TN + wn * 64, lane);
        }
    }
}

}  // namespace

int init_gemm_kernels() {
    int rc = 0;
#define GEMM_ATTR(E) rc |= hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_big_kernel<E>), hipFuncAttributeMaxDynamicSharedMemorySize, NST * STAGE_BYTES) == hipSuccess ? 0 : -2
    GEMM_ATTR(EPI_F32); GEMM_ATTR(EPI_BF16); GEMM_ATTR(EPI_BF16_RELU); GEMM_ATTR(EPI_RESID); GEMM_ATTR(EPI_KV_HEADMAJOR);
#undef GEMM_ATTR
    return rc;
}

int launch_gemm(int epilogue, const GemmArgs& a, hipStream_t stream) {
    if (a.M <= 0) return 0;
    if (a.N % BN != 0 || a.K % BK != 0 || (a.lda % 8) || (a.ldw % 8)) return -1;
    // large M: the pipelined 256 x 128 kernel once its tiles can fill half the chip; YMT3_GEMM_SMALL=1 forces the 128 x 128 one
    static const bool force_small = getenv("YMT3_GEMM_SMALL") != nullptr;
    const int big_grid = (a.N / TN) * ((a.M + TM - 1) / TM);
    const bool big = !force_small && big_grid >= 128;
    const int grid = (a.N / BN) * ((a.M + BM - 1) / BM);
#define GEMM_LAUNCH(E) do { if (big) gemm_big_kernel<E><<<big_grid < 256 ? big_grid : 256, 512, NST * STAGE_BYTES, stream>>>(a); else gemm_kernel<E><<<grid, 256, 0, stream>>>(a); } while (0)
    switch (epilogue) {
        case EPI_F32: GEMM_LAUNCH(EPI_F32); break;
        case EPI_BF16: GEMM_LAUNCH(EPI_BF16); break;
        case EPI_BF16_RELU: GEMM_LAUNCH(EPI_BF16_RELU); break;
        case EPI_RESID: GEMM_LAUNCH(EPI_RESID); break;
        case EPI_KV_HEADMAJOR: GEMM_LAUNCH(EPI_KV_HEADMAJOR); break;
        default: return -1;
    }
#undef GEMM_LAUNCH
    return 0;
}
