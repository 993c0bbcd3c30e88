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
#include "common.h"
#include "kernels.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;

__device__ __forceinline__ int swz(int row, int chunk) { return row * 8 + (chunk ^ (row & 7)); }  // 16-byte units

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

    // epilogue: lane holds rows m = .. + (lane & 15), columns n = .. + (lane >> 4) * 4 + {0..3}
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = m0 + wm * 64 + mt * 16 + (lane & 15);
        if (m >= g.M) continue;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = n0 + wn * 64 + nt * 16 + (lane >> 4) * 4;
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

}  // namespace

int launch_gemm(int epilogue, const GemmArgs& a, hipStream_t stream) {
    if (a.M <= 0) return 0;
    if (a.N % BN != 0 || a.K % BK != 0 || (a.lda % 8) || (a.ldw % 8)) return -1;
    const int grid = (a.N / BN) * ((a.M + BM - 1) / BM);
#define GEMM_LAUNCH(E) gemm_kernel<E><<<grid, 256, 0, stream>>>(a)
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
