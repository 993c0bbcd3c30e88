// The decoder step's skinny-GEMM chain as ONE launch (a7 of SURVEY.md section 8; dense FFN, up to 64 rows):
//
//     stage 0  cross-attention O-projection   h   += attn . wo_c^T  (+ the folded self-attention partials)     DG_RESID, K = 512
//     stage 1  FFN-in                         dff  = relu(R(rmsnorm(h) * g3) . wi^T)                          DG_NORM_BF16_RELU
//     stage 2  FFN-out                        h   += dff . wo2^T                                              DG_RESID, K = 2048
//     stage 3  the NEXT layer's QKV projection + KV-cache append, or lm_head after the last layer             DG_NORM_QKV_CACHE / DG_NORM_LOGITS
//
// Four dependent launches of dec_gemm_kernel cost 4 x (1.6-2.1 us gap + 1.4-2.1 us body), and most of a body is waiting for the
// weights to arrive from beyond L2 (the XCDs' L2s do not keep them across a kernel boundary; profiles/r02_weight_warmup.txt).  Here
// every workgroup requests the weight tiles of ALL its stages at entry, behind stage 0's operands, so stages 1-3 find theirs in
// registers; a stage boundary is a per-row-tile arrival counter (tile (mt, *) of a stage needs only the 16 complete rows of row tile
// mt from the stage before), and what crosses it -- h, the sum(h^2) partials, the FFN hidden -- is written and read at agent scope
// (sc1: the L2s of the eight XCDs are not coherent with each other).  tools/chain_probe.cpp measured that boundary at 2.4 us against
// 2.25 for a launch *without* the weight wait.  The arithmetic is dec_gemm_kernel's, operation for operation (same K-slices per wave,
// same MFMA chains, same fixed-order reductions, same epilogues), so ids are bit-identical to the four launches.
//
// All workgroups must be co-resident (a stage spins on the one before): grid <= 256 workgroups of 143 KB LDS = one per CU, launched
// on an otherwise idle stream position.  Every spin is bounded by a wall-clock limit and a sticky abort word, so the grid always drains;
// an abort poisons nothing silently -- runtime.hip checks the word when the decode call ends.
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace {

#include "dec_chain_body.h"

template <int MODE3, bool W2F>
__global__ __launch_bounds__(512) void dec_chain_kernel(const bf16_t* __restrict__ pW0, const bf16_t* __restrict__ pW1, const bf16_t* __restrict__ pW2,
                                                        const bf16_t* __restrict__ pW3, const bf16_t* __restrict__ pAttn, float* pH, int row0, int R,
                                                        ChainArgs c) {
    // leading scalar arguments (14 dwords): kernarg preload, as dec_gemm_kernel
    extern __shared__ __attribute__((aligned(16))) char smem[];
    chain_stages<MODE3, W2F>(pW0, pW1, pW2, pW3, pAttn, pH, row0, R, c, smem, blockIdx.x);
}

__global__ void chain_poison_kernel(const unsigned* sync, int32_t* tokens, long long n) {
    if (sync[CHAIN_ABORT_WORD] == 0u) return;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) tokens[i] = INT32_MIN;
}

constexpr size_t CHAIN_LDS = (size_t)(8 * 16 * 16 + 16) * 4 + (size_t)8 * 2 * 16 * (2048 / 8 * 2 + 16);     // stage 2's; the NORM stages need less
constexpr size_t CHAIN_LDS_W2F = (size_t)(8 * 16 * 16 + 16) * 4 + (size_t)8 * 16 * (2048 / 8 * 2 + 16);        // stage 2 without its weight strip
constexpr size_t CHAIN_LDS_NORM = (size_t)(8 * 16 * 32 + 16) * 4 + (size_t)8 * 3 * 16 * (512 / 8 * 2 + 16);
static_assert(CHAIN_LDS >= CHAIN_LDS_NORM && CHAIN_LDS_W2F >= CHAIN_LDS_NORM, "NORM stage LDS");
static_assert(CHAIN_LDS_W2F <= 80 * 1024, "two workgroups per CU");

}  // namespace

int init_chain_kernels() {
    int rc = 0;
    rc |= hipFuncSetAttribute(reinterpret_cast<const void*>(dec_chain_kernel<DG_NORM_QKV_CACHE, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CHAIN_LDS) != hipSuccess;
    rc |= hipFuncSetAttribute(reinterpret_cast<const void*>(dec_chain_kernel<DG_NORM_LOGITS, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CHAIN_LDS) != hipSuccess;
    rc |= hipFuncSetAttribute(reinterpret_cast<const void*>(dec_chain_kernel<DG_NORM_QKV_CACHE, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CHAIN_LDS_W2F) != hipSuccess;
    rc |= hipFuncSetAttribute(reinterpret_cast<const void*>(dec_chain_kernel<DG_NORM_LOGITS, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CHAIN_LDS_W2F) != hipSuccess;
    return rc ? -2 : 0;
}

// A/B switch (YMT3_CHAIN_W2F=1 at the first launch): stage 2's weights as register fragments, 76 KB of LDS (dec_step.hip's form) instead of 143 KB
static bool chain_w2f() {
    static const bool v = [] { const char* e = getenv("YMT3_CHAIN_W2F"); return e && e[0] == '1'; }();
    return v;
}

// the chain's 256 workgroups must all be resident at once: at least one per CU at its LDS / register footprint
bool dec_chain_fits(int n_cus) {
    int a = 0, b = 0;
    if (chain_w2f()) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, dec_chain_kernel<DG_NORM_QKV_CACHE, true>, 512, CHAIN_LDS_W2F) != hipSuccess) return false;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, dec_chain_kernel<DG_NORM_LOGITS, true>, 512, CHAIN_LDS_W2F) != hipSuccess) return false;
    } else {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, dec_chain_kernel<DG_NORM_QKV_CACHE, false>, 512, CHAIN_LDS) != hipSuccess) return false;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, dec_chain_kernel<DG_NORM_LOGITS, false>, 512, CHAIN_LDS) != hipSuccess) return false;
    }
    return (long long)(a < b ? a : b) * n_cus >= 256;
}

// 0 = launched; negative = this shape is not the chain's (the caller launches the four GEMMs instead)
int launch_dec_chain(const ChainArgs& c, hipStream_t stream) {
    if (c.R <= 0) return 0;
    if (c.R > 16 * CHAIN_TILES_MAX || c.d_ff != 2048 || c.N3 % 32 || c.N3 / 32 < 32 || c.N3 / 32 > 64 || !c.sync || (c.mode3 != DG_NORM_QKV_CACHE && c.mode3 != DG_NORM_LOGITS))
        return -1;
    const int n_mt = (c.R + 15) / 16;
    const int grid = n_mt <= 4 ? 256 : 64 * n_mt;  // 64 column tiles (stage 1: 2048 / 32) x 4 row-tile slots, one workgroup per CU; beyond 64 rows 64 per row tile
    if (chain_w2f()) {
        if (c.mode3 == DG_NORM_QKV_CACHE)
            dec_chain_kernel<DG_NORM_QKV_CACHE, true><<<grid, 512, CHAIN_LDS_W2F, stream>>>(c.w0, c.w1, c.w2, c.w3, c.attn, c.h, c.row0, c.R, c);
        else
            dec_chain_kernel<DG_NORM_LOGITS, true><<<grid, 512, CHAIN_LDS_W2F, stream>>>(c.w0, c.w1, c.w2, c.w3, c.attn, c.h, c.row0, c.R, c);
    } else if (c.mode3 == DG_NORM_QKV_CACHE)
        dec_chain_kernel<DG_NORM_QKV_CACHE, false><<<grid, 512, CHAIN_LDS, stream>>>(c.w0, c.w1, c.w2, c.w3, c.attn, c.h, c.row0, c.R, c);
    else
        dec_chain_kernel<DG_NORM_LOGITS, false><<<grid, 512, CHAIN_LDS, stream>>>(c.w0, c.w1, c.w2, c.w3, c.attn, c.h, c.row0, c.R, c);
    return 0;
}

int launch_chain_poison(const unsigned* sync, int32_t* tokens, long long n, hipStream_t stream) {
    if (!sync || !tokens || n <= 0) return 0;
    chain_poison_kernel<<<64, 256, 0, stream>>>(sync, tokens, n);
    return 0;
}
