// C-ABI runtime of the hot path: handle, weight blob, workspace, kernel orchestration, hipGraph
// replay of the decode step.  Declarations and ownership rules: include/ymt3.h.
//
// Host logic only -- every FLOP is in frontend.hip / gemm.hip / norm.hip / enc_attn.hip /
// decode.hip.  There is no CPU fallback: a missing device, tensor or unsupported shape is an error.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <map>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ymt3.h"
#include "common.h"
#include "kernels.h"

// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void ymt3_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
#define FAIL(code, ...)              \
    do {                             \
        ymt3_set_error(__VA_ARGS__); \
        return (code);               \
    } while (0)
#define LAUNCH(expr)                                                             \
    do {                                                                         \
        int _rc = (expr);                                                        \
        if (_rc != 0) FAIL(YMT3_ERR_UNSUPPORTED, "%s rejected its shape (rc=%d)", #expr, _rc); \
    } while (0)

struct Tensor {
    void* dev = nullptr;
    uint32_t dtype = 0, ndim = 0, shape[4] = {1, 1, 1, 1};
    size_t nbytes = 0;
};

struct StepGraph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    bool merged = false;            // the captured steps contain merged kernels (GEMM chain / attention pair): their abort word must be looked at
};

struct ymt3_ctx {
    ymt3_config cfg{};
    int device = 0;
    int T = 0, inner = 0, maxB = 0, maxR = 0;
    char* blob_dev = nullptr;
    size_t blob_bytes = 0;
    std::map<std::string, Tensor> tensors;
    std::vector<void*> allocs;
    size_t dev_bytes = 0;
    FrontendTables fe{};
    // encoder workspace
    float* mel = nullptr;
    bf16_t* mel_bf = nullptr;
    float* h_enc = nullptr;
    bf16_t *xn = nullptr, *qkv = nullptr, *attn = nullptr, *ff = nullptr, *enc_out = nullptr;
    // Perceiver-TF encoder workspace (a9): N1 = B*T*F' spectral tokens, N2 = B*T*K latent rows, D = ptf_d
    bf16_t *p_xs = nullptr, *p_kvs = nullptr;      // [N1][D] normed spectral tokens, [N1][2D] their K/V of one block
    float* p_z = nullptr;                          // [N2][D] fp32 latent residual stream, layout [b][t][k][:]
    bf16_t *p_zn = nullptr, *p_qkv = nullptr, *p_att = nullptr, *p_ff = nullptr;   // [N2][D], [N2][3D], [N2][D], [N2][ptf_dff]
    // decoder workspace
    bf16_t* wkv_all = nullptr;          // [n_dec*2*inner][d]
    bf16_t* ckv = nullptr;              // [n_dec*2][B][H][T][64]
    bf16_t *kcache = nullptr, *vcache = nullptr;   // [n_dec][maxR][H][L][64]
    float* h_dec = nullptr;
    float* h_dec2 = nullptr;            // MoE decoder: the second residual buffer of the folded combine (layers alternate between the two)
    bf16_t *dq = nullptr, *dattn = nullptr, *dff = nullptr;
    float* logits = nullptr;
    float* ssq = nullptr;               // [SSQ_TILES][maxR]
    float* opart = nullptr;             // [maxR][H][d]: per-head O-projection partials of the self-attention kernel (fold_o)
    MoeArgs moe{};                      // scratch pointers of the MoE FFN (dec_ffn == YMT3_FFN_MOE)
    int* finished = nullptr;
    // slot mode (ymt3_transcribe_stream): per-row positions and output offsets; launch_step wires them in while set
    int* row_pos = nullptr;             // [maxR]
    long long* row_out = nullptr;       // [maxR]
    int* host_rows = nullptr;           // pinned [maxR]: copy of `finished` for the host's retire/admit decisions
    bool slot_mode = false;
    DecodeShared* shared = nullptr;     // [MAX_CHAINS] per-chain loop state
    hipStream_t cap_stream = nullptr;
    // Decode rows are independent, so a batch CAN be cut into `n_chains` contiguous row ranges whose step graphs replay
    // concurrently on separate HIP streams.  Measured on MI355X in both rounds (profiles/r01_chain_sweep.txt with one host thread
    // feeding all chains, profiles/r02_chain_sweep_threads.txt with one launcher thread per chain): it loses -- 2 chains 289.8 ms
    // per batch against 279.8 for one, 3 / 4 chains 474 / 486 ms -- so the chains do not overlap on the device either.
    // Default 1; YMT3_CHAINS overrides (kept as a tested option: any row split must give identical ids).
    int n_chains = 1;
    // Round 3, many rows: with 168-256 rows of one channel the attention kernels are bandwidth-bound and the GEMMs between them latency-bound, and
    // two chains of 84-128 rows do overlap: 626 against 690 ms per batch of 256, -6 % at 176 and 192 (profiles/r03_chains_many_rows.txt; -1 % at 160, slower from
    // 512).  `auto_chains` (YMT3_CHAINS unset) takes two chains exactly there -- both halves and the whole stay in the same kernel regime (8-wave
    // attention, 16-row-tile GEMMs, no folded O-projection), so the ids do not depend on the choice (tested).
    bool auto_chains = true;
    int last_chains = 1;
    bool chain_threads = true;              // one launcher thread per chain (YMT3_CHAIN_THREADS=0: the caller's thread feeds all)
    hipStream_t chain_stream[8] = {};
    hipEvent_t fork_ev = nullptr, join_ev[8] = {};
    std::map<long, StepGraph> step_graphs;  // keyed by (B, n_chains_used, chain)
    bool use_graph = true;
    int graph_steps = 16;                   // decode steps per replayed graph (YMT3_GRAPH_STEPS): a graph launch costs ~7 us of stream time on top of its kernels
    bool fuse_q = true;                     // cross-attention computes its own query projection
    bool moe_fold_combine = true;           // MoE: h += y0 + y1 is done by the next norm GEMM's prologue (YMT3_MOE_COMBINE_LAUNCH=1: own launch)
    bool fold_o = true;                     // self-attention ends with its head's O-projection partial; no separate O-projection launch
    bool gemm_chain = true;                 // cross O -> FFN-in -> FFN-out -> next QKV / lm_head as one launch (dec_chain.hip; YMT3_NO_GEMM_CHAIN=1: four launches)
    unsigned* chain_sync = nullptr;         // [CHAIN_SYNC_WORDS] device: the chain kernel's arrival counters + sticky abort word
    unsigned* chain_host_abort = nullptr;   // pinned: set by the chain kernel together with the abort word; checked at every call
    bool step_merged = false;               // set by launch_step: the step it just captured / launched contains merged kernels
    bool forced_abort = false;              // ymt3_debug_force_stage_abort: raise the host word after the next decode call, as a kernel would during it
    // What happens when a merged kernel gives up waiting (ymt3_set_abort_recovery): 1 (default) = every decode call that ran merged kernels ends
    // by waiting for its stream and looking at the abort word; an abort re-runs the call through the separate launches (same bits) and the
    // handle stays on them.  0 = fully asynchronous calls: an aborted call's ids are INT32_MIN and the NEXT call on the handle switches over.
    int abort_recovery = 1;
    int fallback_count = 0;                 // how often this handle fell back from the merged kernels to the separate launches (0 or 1)
    int mid_rows = -1;                      // YMT3_DEC_GEMM_MID_ROWS at create: decode GEMMs take mid-size tiles from this many rows on (-1: DEC_GEMM_MID_ROWS, 0: never)
    bool force_2wave = false;               // YMT3_SELF_ATTN_2WAVE=1 at create (test knob): the many-row 2-wave self-attention at any row count
    bool attn_pair = true;                  // a layer's self- and cross-attention as one launch (decode.hip: dec_attn_pair_kernel; YMT3_NO_ATTN_PAIR=1: two)
    unsigned* pair_rows = nullptr;          // [maxR <= 64][2] counter lines of that kernel (zero between launches)
    bool step_kernel = false;               // a step's six layers as ONE launch (dec_step.hip): YMT3_STEP_KERNEL=1; default: attention pair + GEMM chain per layer
    bool moe_chain = false;                 // MoE decoder: a layer's five skinny launches as one (moe_chain.hip; YMT3_NO_MOE_CHAIN=1: separate launches)
    int merged_max_rows = 64;               // YMT3_MERGED_MAX_ROWS: the attention pair / GEMM chain are taken up to this many rows (<= 256)
    bool step_tiles_free = false;           // YMT3_STEP_TILES_FREE=1 (A/B): the step kernel's four row tiles as independent pipelines instead of in step
    unsigned* ticket = nullptr;             // [maxR / 32 + 1] lines: the argmax kernel's two-level ticket (many rows)
    unsigned* step_sync = nullptr;          // [STEP_SYNC_LINES] counter lines of that kernel (zeroed by the step's argmax kernel / before a decode call)
    // sampled per-kernel-class timing (ymt3_profile_decode): events bracket single launches
    bool prof_on = false;
    size_t prof_span_idx = 0;
    bool prof_span_open = false;
    int last_steps = 0;                     // steps launched by the last decode call (ymt3_last_decode_steps)
    int early_stop_interval = 0;            // ymt3_set_early_stop: host checks `n_unfinished` every N steps (0 = never)
    int* host_flag = nullptr;               // pinned, for that check
    bool debug_hooks = false;               // YMT3_DEBUG_HOOKS=1 at create: ymt3_debug_decode_start is accepted
    int prof_step0 = 0;                     // ymt3_debug_decode_start: the next decode call begins at this position (one shot)
    int32_t* moe_trace = nullptr;           // ymt3_debug_moe_trace: caller's [steps][layers][rows][2] buffer the MoE router records its choices in
    int moe_trace_steps = 0, moe_trace_rows = 0;
    std::vector<hipEvent_t> prof_ev;        // pairs
    std::vector<int> prof_cls;
    // measurement (YMT3_STAMP=1): per-workgroup wall-clock stamps of the decode-step kernels, slot = launch order in the step
    unsigned long long* stamp_buf = nullptr;   // [STAMP_NODES][STAMP_WGS][2]
    int stamp_n = 0, stamp_cls[64] = {}, stamp_grid[64] = {};
    // ymt3_ingest: polyphase low-pass per (up, down), built on first use
    struct Resampler { float* taps = nullptr; int up = 1, down = 1, J = 1, Jp = 4, window = 0; long long r = 0; };
    std::map<std::pair<int, int>, Resampler> resamplers;
};

constexpr int STAMP_NODES = 64, STAMP_WGS = 8192;
static unsigned long long* next_stamp(ymt3_ctx* c, int cls, int grid) {
    if (!c->stamp_buf || c->stamp_n >= STAMP_NODES || grid > STAMP_WGS) return nullptr;
    const int i = c->stamp_n++;
    c->stamp_cls[i] = cls;
    c->stamp_grid[i] = grid;
    return c->stamp_buf + (size_t)i * STAMP_WGS * 2;
}

enum { PC_QKV = 0, PC_SELF_ATTN, PC_SELF_O, PC_CROSS_Q, PC_CROSS_ATTN, PC_CROSS_O, PC_FFN_WI, PC_FFN_WO, PC_LM_HEAD, PC_ARGMAX, PC_SPAN, PC_CHAIN, PC_ATTN_PAIR, PC_STEP, PC_COUNT };

struct ProfScope {
    ymt3_ctx* c; hipStream_t s; bool on;
    ProfScope(ymt3_ctx* c_, int cls, hipStream_t s_) : c(c_), s(s_), on(c_->prof_on) {
        if (!on) return;
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
        c->prof_ev.push_back(a); c->prof_ev.push_back(b); c->prof_cls.push_back(cls);
        (void)hipEventRecord(a, s);
    }
    ~ProfScope() { if (on) (void)hipEventRecord(c->prof_ev.back(), s); }
};
#define PLAUNCH(cls, expr) do { ProfScope _ps(h, cls, s); LAUNCH(expr); } while (0)

static int dev_alloc(ymt3_ctx* c, void** p, size_t bytes) {
    HIP_TRY(hipMalloc(p, bytes ? bytes : 16));
    c->allocs.push_back(*p);
    c->dev_bytes += bytes;
    return 0;
}

template <typename T>
static int get(ymt3_ctx* c, const std::string& name, uint32_t dtype, T** out, size_t min_elems = 0) {
    auto it = c->tensors.find(name);
    if (it == c->tensors.end()) FAIL(YMT3_ERR_BLOB, "weight blob has no tensor '%s'", name.c_str());
    if (it->second.dtype != dtype) FAIL(YMT3_ERR_BLOB, "tensor '%s' has dtype %u, expected %u", name.c_str(), it->second.dtype, dtype);
    const size_t esz = dtype == 1 ? 2 : (dtype == 3 ? 1 : 4);
    if (it->second.nbytes < min_elems * esz)
        FAIL(YMT3_ERR_BLOB, "tensor '%s' holds %zu bytes, expected at least %zu", name.c_str(), it->second.nbytes, min_elems * esz);
    *out = reinterpret_cast<T*>(it->second.dev);
    return 0;
}
#define GET(...)                     \
    do {                             \
        int _rc = get(__VA_ARGS__);  \
        if (_rc) return _rc;         \
    } while (0)

#pragma pack(push, 1)
struct BlobEntry {
    char name[48];
    uint32_t dtype, ndim, shape[4];
    uint64_t offset, nbytes;
};
#pragma pack(pop)
static_assert(sizeof(BlobEntry) == 88, "blob entry layout");

static int parse_blob(ymt3_ctx* c, const void* blob, size_t nbytes) {
    const char* b = static_cast<const char*>(blob);
    if (nbytes < 16 || memcmp(b, "YMT3BLOB", 8) != 0) FAIL(YMT3_ERR_BLOB, "bad weight blob magic");
    uint32_t ver, n;
    memcpy(&ver, b + 8, 4);
    memcpy(&n, b + 12, 4);
    if (ver != 1) FAIL(YMT3_ERR_BLOB, "unsupported blob version %u", ver);
    if (16 + (size_t)n * sizeof(BlobEntry) > nbytes) FAIL(YMT3_ERR_BLOB, "truncated blob header");
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->blob_dev), nbytes));
    c->allocs.push_back(c->blob_dev);
    c->dev_bytes += nbytes;
    c->blob_bytes = nbytes;
    HIP_TRY(hipMemcpy(c->blob_dev, blob, nbytes, hipMemcpyHostToDevice));
    for (uint32_t i = 0; i < n; ++i) {
        BlobEntry e;
        memcpy(&e, b + 16 + (size_t)i * sizeof(BlobEntry), sizeof(e));
        const size_t header_end = 16 + (size_t)n * sizeof(BlobEntry);
        if (e.offset % 16 || e.offset < header_end || e.offset > nbytes || e.nbytes > nbytes - e.offset)     // no wrap-around
            FAIL(YMT3_ERR_BLOB, "tensor %u out of bounds / misaligned", i);
        e.name[47] = 0;
        Tensor t;
        t.dev = c->blob_dev + e.offset;
        t.dtype = e.dtype;
        t.ndim = e.ndim;
        memcpy(t.shape, e.shape, sizeof(t.shape));
        t.nbytes = e.nbytes;
        c->tensors[e.name] = t;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
extern "C" int ymt3_abi_version(void) { return YMT3_ABI_VERSION; }
extern "C" const char* ymt3_last_error(void) { return g_err; }

extern "C" void ymt3_destroy(ymt3_handle h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    for (auto& kv : h->step_graphs) {
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
        if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
    }
    if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
    if (h->host_flag) (void)hipHostFree(h->host_flag);
    if (h->chain_host_abort) (void)hipHostFree(h->chain_host_abort);
    if (h->host_rows) (void)hipHostFree(h->host_rows);
    for (int i = 0; i < 8; ++i) {
        if (h->chain_stream[i]) (void)hipStreamDestroy(h->chain_stream[i]);
        if (h->join_ev[i]) (void)hipEventDestroy(h->join_ev[i]);
    }
    if (h->fork_ev) (void)hipEventDestroy(h->fork_ev);
    for (void* p : h->allocs) (void)hipFree(p);
    delete h;
}

extern "C" size_t ymt3_device_bytes(ymt3_handle h) { return h ? h->dev_bytes : 0; }

static int create_impl(ymt3_ctx* c, const ymt3_config* cfg, const void* blob, size_t nbytes) {
    const ymt3_config& k = c->cfg;
    if (k.d_kv != 64) FAIL(YMT3_ERR_UNSUPPORTED, "d_kv must be 64 (got %d)", k.d_kv);
    if (k.d_model != 16 * SSQ_TILES) FAIL(YMT3_ERR_UNSUPPORTED, "d_model must be 512 (got %d)", k.d_model);
    if (k.n_heads * k.d_kv != 512) FAIL(YMT3_ERR_UNSUPPORTED, "n_heads*d_kv must be 512");
    if (k.encoder_type != YMT3_ENC_T5 && k.encoder_type != YMT3_ENC_PERCEIVER_TF) FAIL(YMT3_ERR_UNSUPPORTED, "unknown encoder_type %d", k.encoder_type);
    if (k.dec_ffn == YMT3_FFN_MOE && (k.moe_top_k != 2 || k.n_experts < 2 || k.n_experts > 16 || k.d_ff != 2048))
        FAIL(YMT3_ERR_UNSUPPORTED, "MoE FFN needs top_k = 2, 2..16 experts, d_ff = 2048");
    if (k.max_batch <= 0 || k.n_channels <= 0 || k.max_decode_len <= 0) FAIL(YMT3_ERR_ARG, "bad max_batch / n_channels / max_decode_len");
    if (k.hop <= 0 || k.sample_rate <= 0 || k.segment_samples <= 0 || k.n_fft <= 0 || k.n_mels <= 0 || k.vocab <= 0 || k.d_ff <= 0 ||
        k.n_enc_layers < 0 || k.n_dec_layers <= 0 || k.n_enc_layers > 64 || k.n_dec_layers > 64 || k.n_channels > 64 ||
        k.max_decode_len > 65536 || (long long)k.max_batch * k.n_channels > 65536)
        FAIL(YMT3_ERR_ARG, "a size field of the config is zero, negative or absurd");
    c->T = 1 + k.segment_samples / k.hop;
    c->inner = k.n_heads * k.d_kv;
    c->maxB = k.max_batch;
    c->maxR = k.max_batch * k.n_channels;
    if (c->T % 64) FAIL(YMT3_ERR_UNSUPPORTED, "n_frames must be a multiple of 64 (got %d)", c->T);
    if (k.segment_samples <= k.n_fft / 2) FAIL(YMT3_ERR_UNSUPPORTED, "segment_samples must exceed n_fft/2 (reflect padding)");
    if (k.pad_id < 0 || k.pad_id >= k.vocab || k.eos_id >= k.vocab) FAIL(YMT3_ERR_ARG, "pad_id / eos_id outside the vocabulary");
    if (k.vocab % 16 || k.d_ff % 128) FAIL(YMT3_ERR_UNSUPPORTED, "vocab %% 16 and d_ff %% 128 must be 0");

    HIP_TRY(hipSetDevice(c->device));
    int rc = parse_blob(c, blob, nbytes);
    if (rc) return rc;
    if (init_gemm_kernels() || init_enc_attn_kernels() || init_decode_kernels() || init_moe_kernels() || init_mc_cross_kernels()) FAIL(YMT3_ERR_HIP, "hipFuncSetAttribute(max dynamic LDS) failed");

    // front-end tables (built by yourmt3_amd/tables.py, carried in the blob)
    const int nfft = k.n_fft;
    FrontendTables& fe = c->fe;
    GET(c, "fe.window", 0u, const_cast<float**>(&fe.window), (size_t)nfft);
    GET(c, "fe.tw", 0u, reinterpret_cast<float**>(const_cast<float2**>(&fe.tw)), (size_t)nfft);
    GET(c, "fe.untw", 0u, reinterpret_cast<float**>(const_cast<float2**>(&fe.untw)), (size_t)nfft + 2);
    GET(c, "fe.mel_start", 2u, const_cast<int**>(&fe.mel_start), (size_t)k.n_mels);
    GET(c, "fe.mel_len", 2u, const_cast<int**>(&fe.mel_len), (size_t)k.n_mels);
    GET(c, "fe.mel_off", 2u, const_cast<int**>(&fe.mel_off), (size_t)k.n_mels);
    GET(c, "fe.mel_w", 0u, const_cast<float**>(&fe.mel_w), 1);
    fe.n_mel_w = (int)(c->tensors["fe.mel_w"].nbytes / 4);
    fe.n_fft = nfft; fe.hop = k.hop; fe.n_mels = k.n_mels; fe.n_samples = k.segment_samples;
    fe.n_frames = c->T; fe.log_floor = k.log_floor;
    if (nfft != 2048 && nfft != 512) FAIL(YMT3_ERR_UNSUPPORTED, "n_fft must be 2048 or 512");
    if (k.n_mels % 64) FAIL(YMT3_ERR_UNSUPPORTED, "n_mels must be a multiple of 64");

    const size_t BT = (size_t)c->maxB * c->T, d = k.d_model, R = c->maxR;
    if (dev_alloc(c, (void**)&c->mel, BT * k.n_mels * 4)) return YMT3_ERR_HIP;
    if (dev_alloc(c, (void**)&c->mel_bf, BT * k.n_mels * 2)) return YMT3_ERR_HIP;
    if (dev_alloc(c, (void**)&c->h_enc, BT * d * 4)) return YMT3_ERR_HIP;
    if (dev_alloc(c, (void**)&c->xn, BT * d * 2)) return YMT3_ERR_HIP;
    if (dev_alloc(c, (void**)&c->qkv, BT * 3 * c->inner * 2)) return YMT3_ERR_HIP;
    if (dev_alloc(c, (void**)&c->attn, BT * c->inner * 2)) return YMT3_ERR_HIP;
    if (dev_alloc(c, (void**)&c->ff, BT * k.d_ff * 2)) return YMT3_ERR_HIP;
    if (dev_alloc(c, (void**)&c->enc_out, BT * d * 2)) return YMT3_ERR_HIP;
    if (k.encoder_type == YMT3_ENC_PERCEIVER_TF) {
        const int D = k.ptf_d, K = k.n_latents;
        if (D <= 0 || D % 64 || D > 256) FAIL(YMT3_ERR_UNSUPPORTED, "ptf_d must be 64, 128, 192 or 256 (got %d)", D);
        if (K < 16 || K > 64 || K % 16) FAIL(YMT3_ERR_UNSUPPORTED, "n_latents must be 16, 32, 48 or 64 latents per frame (got %d)", K);
        if (k.ptf_blocks < 1 || k.ptf_blocks > 16 || k.ptf_dff <= 0 || k.ptf_dff % 128) FAIL(YMT3_ERR_UNSUPPORTED, "ptf_blocks must be 1..16 and ptf_dff a multiple of 128");
        if (D % 128 || (3 * D) % 128) FAIL(YMT3_ERR_UNSUPPORTED, "ptf_d must be a multiple of 128 (GEMM column tiles)");
        if (c->T > 256 || (k.n_mels != 64 && k.n_mels != 128 && k.n_mels != 256)) FAIL(YMT3_ERR_UNSUPPORTED, "the Perceiver-TF encoder needs n_frames <= 256 and n_mels in {64, 128, 256}");
        if (K != 32 && K != 64) FAIL(YMT3_ERR_UNSUPPORTED, "n_latents must be 32 or 64 (latent self-attention key tiles)");
        const size_t N1 = (size_t)c->maxB * c->T * k.n_mels, N2 = (size_t)c->maxB * c->T * K;
        if (N1 > 0x7fffffffULL / 2) FAIL(YMT3_ERR_UNSUPPORTED, "max_batch too large for the Perceiver-TF encoder workspace");
        if (dev_alloc(c, (void**)&c->p_xs, N1 * D * 2) || dev_alloc(c, (void**)&c->p_kvs, N1 * 2 * D * 2) || dev_alloc(c, (void**)&c->p_z, N2 * D * 4) ||
            dev_alloc(c, (void**)&c->p_zn, N2 * D * 2) || dev_alloc(c, (void**)&c->p_qkv, N2 * 3 * D * 2) || dev_alloc(c, (void**)&c->p_att, N2 * D * 2) ||
            dev_alloc(c, (void**)&c->p_ff, N2 * k.ptf_dff * 2))
            return YMT3_ERR_HIP;
        std::vector<std::string> names = {"spec_w", "spec_pos", "ln_x", "latents", "bias_off", "ln_out", "out_w"};
        for (int b = 0; b < k.ptf_blocks; ++b) {
            const std::string p = std::to_string(b) + ".";
            for (const char* n : {"s.ln_q", "s.wq", "s.wkv", "s.wo", "l.ln1", "l.wqkv", "l.wo", "t.ln1", "t.wqkv", "t.wo"}) names.push_back(p + n);
            for (const char* sub : {"s.", "l.", "t."})
                for (const char* n : {"ln_ff", "wi", "wo2"}) names.push_back(p + sub + n);
        }
        for (const std::string& n : names)
            if (!c->tensors.count("ptf." + n)) FAIL(YMT3_ERR_BLOB, "missing ptf.%s", n.c_str());
    }

    const int nd = k.n_dec_layers;
    const size_t wkv_elems = (size_t)2 * c->inner * d;
    if (dev_alloc(c, (void**)&c->wkv_all, nd * wkv_elems * 2)) return YMT3_ERR_HIP;
    for (int l = 0; l < nd; ++l) {
        bf16_t* src;
        GET(c, "dec." + std::to_string(l) + ".wkv_c", 1u, &src, wkv_elems);
        HIP_TRY(hipMemcpy(c->wkv_all + l * wkv_elems, src, wkv_elems * 2, hipMemcpyDeviceToDevice));
    }
    if (dev_alloc(c, (void**)&c->ckv, (size_t)nd * 2 * BT * c->inner * 2)) return YMT3_ERR_HIP;
    const size_t cache_elems = (size_t)nd * R * k.n_heads * k.max_decode_len * 64;
    if (dev_alloc(c, (void**)&c->kcache, cache_elems * 2)) return YMT3_ERR_HIP;
    if (dev_alloc(c, (void**)&c->vcache, cache_elems * 2)) return YMT3_ERR_HIP;
    if (dev_alloc(c, (void**)&c->h_dec, R * d * 4)) return YMT3_ERR_HIP;
    if (k.dec_ffn == YMT3_FFN_MOE && dev_alloc(c, (void**)&c->h_dec2, R * d * 4)) return YMT3_ERR_HIP;
    if (dev_alloc(c, (void**)&c->dq, R * c->inner * 2)) return YMT3_ERR_HIP;
    if (dev_alloc(c, (void**)&c->dattn, R * c->inner * 2)) return YMT3_ERR_HIP;
    if (dev_alloc(c, (void**)&c->dff, R * k.d_ff * 2)) return YMT3_ERR_HIP;
    if (dev_alloc(c, (void**)&c->logits, R * k.vocab * 4)) return YMT3_ERR_HIP;
    if (dev_alloc(c, (void**)&c->finished, R * 4)) return YMT3_ERR_HIP;
    if (dev_alloc(c, (void**)&c->row_pos, R * 4)) return YMT3_ERR_HIP;
    if (dev_alloc(c, (void**)&c->row_out, R * 8)) return YMT3_ERR_HIP;
    if (dev_alloc(c, (void**)&c->ssq, (size_t)SSQ_TILES * R * 4)) return YMT3_ERR_HIP;
    if (dev_alloc(c, (void**)&c->opart, R * k.n_heads * d * 4)) return YMT3_ERR_HIP;
    if (dev_alloc(c, (void**)&c->ticket, (R / 32 + 1) * CHAIN_LINE * sizeof(unsigned))) return YMT3_ERR_HIP;
    HIP_TRY(hipMemset(c->ticket, 0, (R / 32 + 1) * CHAIN_LINE * sizeof(unsigned)));
    {   // The merged decode kernels (GEMM chain: 256 workgroups of 143 KB LDS; attention pair: 512 of 72 KB) wait for each other inside one
        // launch, so every workgroup of their grids must be resident at once: a kernel is enabled only if the switch allows it AND the
        // runtime's occupancy answer x CUs covers its grid.  YMT3_TEST_CHAIN_UNFIT / YMT3_TEST_PAIR_UNFIT = 1 force "does not fit" for
        // one kernel (tests: a partition on which only one of the two fits must take the separate launches for the other).
        auto env1 = [](const char* n) { const char* v = getenv(n); return v && v[0] == '1'; };
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, c->device));
        const bool init_ok = init_chain_kernels() == 0;
        c->gemm_chain = !env1("YMT3_NO_GEMM_CHAIN") && init_ok && !env1("YMT3_TEST_CHAIN_UNFIT") && dec_chain_fits(prop.multiProcessorCount);
        c->attn_pair = !env1("YMT3_NO_ATTN_PAIR") && init_ok && !env1("YMT3_TEST_PAIR_UNFIT") && dec_attention_pair_fits(prop.multiProcessorCount);
        // ... and the per-step kernel (all six layers in one launch: 512 workgroups of 76 KB) where both of the above are in use
        // -- measured SLOWER than the per-layer launches (271-296 ms per batch against 247: profiles/r03_step_kernel.md), so it is an option
        // (YMT3_STEP_KERNEL=1), bit-identical and tested, not the default
        c->step_kernel = c->gemm_chain && c->attn_pair && env1("YMT3_STEP_KERNEL") && !env1("YMT3_TEST_STEP_UNFIT") && init_step_kernel() == 0 &&
                         dec_step_fits(prop.multiProcessorCount);
        c->moe_chain = k.dec_ffn == YMT3_FFN_MOE && k.n_experts == 8 && c->attn_pair && !env1("YMT3_NO_MOE_CHAIN") && !env1("YMT3_NO_GEMM_CHAIN") &&
                       !env1("YMT3_TEST_CHAIN_UNFIT") && init_moe_chain_kernels() == 0 && moe_chain_fits(prop.multiProcessorCount, k.moe_fp8 != 0);
        if (c->step_kernel) {
            if (dev_alloc(c, (void**)&c->step_sync, (size_t)STEP_SYNC_LINES * CHAIN_LINE * sizeof(unsigned))) return YMT3_ERR_HIP;
            HIP_TRY(hipMemset(c->step_sync, 0, (size_t)STEP_SYNC_LINES * CHAIN_LINE * sizeof(unsigned)));
        }
        if (c->gemm_chain || c->attn_pair || c->moe_chain) {
            if (dev_alloc(c, (void**)&c->chain_sync, CHAIN_SYNC_WORDS * sizeof(unsigned))) return YMT3_ERR_HIP;
            HIP_TRY(hipMemset(c->chain_sync, 0, CHAIN_SYNC_WORDS * sizeof(unsigned)));
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->chain_host_abort), sizeof(unsigned), hipHostMallocDefault));
            *c->chain_host_abort = 0u;
            if (dev_alloc(c, (void**)&c->pair_rows, (size_t)16 * CHAIN_TILES_MAX * 2 * CHAIN_LINE * sizeof(unsigned))) return YMT3_ERR_HIP;
            HIP_TRY(hipMemset(c->pair_rows, 0, (size_t)16 * CHAIN_TILES_MAX * 2 * CHAIN_LINE * sizeof(unsigned)));
        }
    }
    if (k.dec_ffn == YMT3_FFN_MOE) {
        MoeArgs& m = c->moe;
        const size_t P = 2 * R;
        if (R > 1536) FAIL(YMT3_ERR_UNSUPPORTED, "the MoE FFN holds its pair list in LDS: at most 1536 decoder rows (got %zu)", R);
        if (dev_alloc(c, (void**)&m.xn, R * d * 2) || dev_alloc(c, (void**)&m.sel, P * 4) || dev_alloc(c, (void**)&m.gate, P * 4) ||
            dev_alloc(c, (void**)&m.hidden, P * k.d_ff * 2) || dev_alloc(c, (void**)&m.y, P * d * 4))
            return YMT3_ERR_HIP;
        m.E = k.n_experts; m.top_k = k.moe_top_k; m.d_model = d; m.d_ff = k.d_ff; m.eps = k.ln_eps;
    }
    if (dev_alloc(c, (void**)&c->shared, 8 * sizeof(DecodeShared))) return YMT3_ERR_HIP;
    HIP_TRY(hipMemset(c->shared, 0, 8 * sizeof(DecodeShared)));
    if (getenv("YMT3_STAMP")) {
        if (dev_alloc(c, (void**)&c->stamp_buf, (size_t)STAMP_NODES * STAMP_WGS * 2 * sizeof(unsigned long long))) return YMT3_ERR_HIP;
        HIP_TRY(hipMemset(c->stamp_buf, 0, (size_t)STAMP_NODES * STAMP_WGS * 2 * sizeof(unsigned long long)));
    }
    HIP_TRY(hipStreamCreateWithFlags(&c->cap_stream, hipStreamNonBlocking));
    const char* dh = getenv("YMT3_DEBUG_HOOKS");
    c->debug_hooks = dh && dh[0] == '1';
    const char* ng = getenv("YMT3_NO_GRAPH");
    c->use_graph = !(ng && ng[0] == '1');
    if (const char* gs = getenv("YMT3_GRAPH_STEPS")) { const int v = atoi(gs); if (v >= 1 && v <= 64) c->graph_steps = v; }
    const char* nf = getenv("YMT3_NO_FUSEQ");
    c->fuse_q = !(nf && nf[0] == '1');
    const char* mcl = getenv("YMT3_MOE_COMBINE_LAUNCH");
    c->moe_fold_combine = !(mcl && mcl[0] == '1');
    const char* nfo = getenv("YMT3_NO_FOLD_O");          // A/B: keep the separate self-attention O-projection launch
    c->fold_o = !(nfo && nfo[0] == '1');
    // (YMT3_NO_GEMM_CHAIN / YMT3_NO_ATTN_PAIR = 1 -- the skinny GEMMs as four launches, the two attentions as two -- are read above, with the fit decisions)
    if (const char* mr = getenv("YMT3_DEC_GEMM_MID_ROWS")) c->mid_rows = atoi(mr) < 0 ? -1 : atoi(mr);
    const char* f2 = getenv("YMT3_SELF_ATTN_2WAVE");
    c->force_2wave = f2 && f2[0] == '1';
    if (const char* mm = getenv("YMT3_MERGED_MAX_ROWS")) { const int v = atoi(mm); if (v >= 16 && v <= 16 * CHAIN_TILES_MAX) c->merged_max_rows = v; }
    { const char* tf = getenv("YMT3_STEP_TILES_FREE"); c->step_tiles_free = tf && tf[0] == '1'; }
    if (const char* ar = getenv("YMT3_ABORT_RECOVERY")) c->abort_recovery = ar[0] == '0' ? 0 : 1;
    const char* nc = getenv("YMT3_CHAINS");
    if (nc && atoi(nc) >= 1) { c->n_chains = atoi(nc) > 8 ? 8 : atoi(nc); c->auto_chains = false; }
    const char* ct = getenv("YMT3_CHAIN_THREADS");
    c->chain_threads = !(ct && ct[0] == '0');
    for (int i = 0; i < std::max(c->n_chains, 2); ++i) {
        HIP_TRY(hipStreamCreateWithFlags(&c->chain_stream[i], hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&c->join_ev[i], hipEventDisableTiming));
    }
    HIP_TRY(hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming));

    // every tensor the kernels will ask for must be present now, not at the first call
    const char* enc_names[] = {"ln1", "wqkv", "wo", "ln2", "wi", "wo2"};
    for (int l = 0; l < k.n_enc_layers; ++l)
        for (const char* n : enc_names)
            if (!c->tensors.count("enc." + std::to_string(l) + "." + n)) FAIL(YMT3_ERR_BLOB, "missing enc.%d.%s", l, n);
    const bool fp8_experts = k.dec_ffn == YMT3_FFN_MOE && k.moe_fp8;
    std::vector<const char*> dec_names = {"ln1", "wqkv", "wo", "ln2", "wq_c", "wo_c", "ln3"};
    if (fp8_experts) { dec_names.push_back("wi_q8"); dec_names.push_back("wo2_q8"); dec_names.push_back("wi_s"); dec_names.push_back("wo2_s"); }
    else { dec_names.push_back("wi"); dec_names.push_back("wo2"); }
    if (k.dec_ffn == YMT3_FFN_MOE) dec_names.push_back("router");
    for (int l = 0; l < nd; ++l)
        for (const char* n : dec_names)
            if (!c->tensors.count("dec." + std::to_string(l) + "." + n)) FAIL(YMT3_ERR_BLOB, "missing dec.%d.%s", l, n);
    HIP_TRY(hipDeviceSynchronize());
    return YMT3_OK;
}

extern "C" int ymt3_create(const ymt3_config* cfg, const void* blob, size_t nbytes, int device, ymt3_handle* out) {
    if (!cfg || !blob || !out) FAIL(YMT3_ERR_ARG, "null argument to ymt3_create");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) FAIL(YMT3_ERR_HIP, "no HIP device visible: the HIP path cannot run (there is no CPU fallback)");
    if (device < 0 || device >= ndev) FAIL(YMT3_ERR_ARG, "device %d out of range (%d visible)", device, ndev);
    ymt3_ctx* c = new ymt3_ctx();
    c->cfg = *cfg;
    c->device = device;
    int rc = create_impl(c, cfg, blob, nbytes);
    if (rc != YMT3_OK) {
        ymt3_destroy(c);
        return rc;
    }
    *out = c;
    return YMT3_OK;
}

// ------------------------------------------------------------------------------------------------
// A merged decode kernel gave up waiting (its sticky abort word is raised): leave the merged kernels for good.  The caller has made sure
// nothing of this handle is still running.  Counters, abort words and the cached step graphs (they hold merged launches) are reset; the
// separate launches compute the same bits, so the handle goes on working.
static int merged_fallback(ymt3_ctx* h) {
    h->gemm_chain = h->attn_pair = h->step_kernel = h->moe_chain = false;
    ++h->fallback_count;
    h->forced_abort = false;
    for (auto& kv : h->step_graphs) {
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
        if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
    }
    h->step_graphs.clear();
    if (h->chain_sync) HIP_TRY(hipMemset(h->chain_sync, 0, CHAIN_SYNC_WORDS * sizeof(unsigned)));
    if (h->pair_rows) HIP_TRY(hipMemset(h->pair_rows, 0, (size_t)16 * CHAIN_TILES_MAX * 2 * CHAIN_LINE * sizeof(unsigned)));
    if (h->step_sync) HIP_TRY(hipMemset(h->step_sync, 0, (size_t)STEP_SYNC_LINES * CHAIN_LINE * sizeof(unsigned)));
    HIP_TRY(hipDeviceSynchronize());
    if (h->chain_host_abort) *h->chain_host_abort = 0u;
    return YMT3_OK;
}

static int check_call(ymt3_handle h, int B) {
    if (!h) FAIL(YMT3_ERR_ARG, "null handle");
    if (B < 0 || B > h->maxB) FAIL(YMT3_ERR_ARG, "B=%d outside [0, max_batch=%d]", B, h->maxB);
    HIP_TRY(hipSetDevice(h->device));
    if (h->chain_host_abort && *static_cast<volatile unsigned*>(h->chain_host_abort)) {
        // an earlier call's merged kernel gave up (> 1 s without its co-resident workgroups: were all CUs available to it?) and nobody has
        // dealt with it yet (ymt3_set_abort_recovery(h, 0), or a measurement call): that call's ids are INT32_MIN; this and every later
        // call run the separate launches
        HIP_TRY(hipDeviceSynchronize());
        int rc = merged_fallback(h);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int ymt3_logmel(ymt3_handle h, const float* audio_dev, int B, float* mel_dev, void* stream) {
    int rc = check_call(h, B);
    if (rc) return rc;
    if (B == 0) return YMT3_OK;
    if (!audio_dev || !mel_dev) FAIL(YMT3_ERR_ARG, "null buffer");
    LAUNCH(launch_logmel(h->fe, audio_dev, mel_dev, B, (hipStream_t)stream));
    HIP_TRY(hipGetLastError());
    return YMT3_OK;
}

// a9: Perceiver-TF encoder (build-defined spec: oracle/perceiver_oracle.py).  Every FLOP is in the GEMM,
// norm and sequence-attention kernels the T5 encoder uses; this is their orchestration over the (B, T, F', C) spectral tokens
// and the (B, T, K, D) latent array.
static int encode_ptf(ymt3_handle h, const float* mel, int B, bf16_t* enc_out, hipStream_t s) {
    const ymt3_config& k = h->cfg;
    const int T = h->T, F = k.n_mels, K = k.n_latents, D = k.ptf_d, Hs = D / 64, dff = k.ptf_dff, d = k.d_model;
    const int N1 = B * T * F, N2 = B * T * K;
    bf16_t *w, *lat, *pos;
    float *f, *specw;
    const float* tbias;
    GET(h, "ptf.spec_w", 0u, &specw, (size_t)D);
    GET(h, "ptf.spec_pos", 1u, &pos, (size_t)F * D);
    GET(h, "ptf.ln_x", 0u, &f, (size_t)D);
    LAUNCH(launch_spec_embed(mel, specw, pos, f, h->p_xs, (long long)N1, F, D, k.ln_eps, s));
    GET(h, "ptf.latents", 1u, &lat, (size_t)K * D);
    LAUNCH(launch_broadcast_bf16(lat, h->p_z, B * T, (size_t)K * D, s));
    GET(h, "ptf.bias_off", 0u, const_cast<float**>(&tbias), (size_t)Hs * (2 * T - 1));
    auto gemm = [&](int epi, const bf16_t* A, const bf16_t* Wt, void* out, int M, int N, int Kd) -> int {
        GemmArgs g{A, Wt, out, nullptr, M, N, Kd, Kd, Kd, N, 0, 0, 0};
        LAUNCH(launch_gemm(epi, g, s));
        return YMT3_OK;
    };
    auto ffn = [&](const std::string& p) -> int {
        GET(h, p + "ln_ff", 0u, &f, (size_t)D);
        LAUNCH(launch_rmsnorm(h->p_z, f, h->p_zn, N2, D, k.ln_eps, s));
        GET(h, p + "wi", 1u, &w, (size_t)dff * D);
        int rc = gemm(EPI_BF16_RELU, h->p_zn, w, h->p_ff, N2, dff, D);
        if (rc) return rc;
        GET(h, p + "wo2", 1u, &w, (size_t)D * dff);
        return gemm(EPI_RESID, h->p_ff, w, h->p_z, N2, D, dff);
    };
    for (int blk = 0; blk < k.ptf_blocks; ++blk) {
        const std::string p = "ptf." + std::to_string(blk) + ".";
        int rc;
        // spectral cross-attention: one sequence per (segment, frame), K latent queries over the frame's F' spectral tokens
        GET(h, p + "s.wkv", 1u, &w, (size_t)2 * D * D);
        if ((rc = gemm(EPI_BF16, h->p_xs, w, h->p_kvs, N1, 2 * D, D))) return rc;
        GET(h, p + "s.ln_q", 0u, &f, (size_t)D);
        LAUNCH(launch_rmsnorm(h->p_z, f, h->p_zn, N2, D, k.ln_eps, s));
        GET(h, p + "s.wq", 1u, &w, (size_t)D * D);
        if ((rc = gemm(EPI_BF16, h->p_zn, w, h->p_qkv, N2, D, D))) return rc;
        {
            SeqAttnArgs a{};
            a.q = h->p_qkv; a.k = h->p_kvs; a.v = h->p_kvs + D; a.out = h->p_att; a.bias_off = nullptr;
            a.n_seq = B * T; a.H = Hs; a.Tq = K; a.Tk = F; a.inner_n = 1;
            a.q_outer = (long long)K * D; a.q_step = D; a.kv_outer = (long long)F * 2 * D; a.kv_step = 2 * D; a.o_outer = (long long)K * D; a.o_step = D;
            LAUNCH(launch_seq_attention(a, s));
        }
        GET(h, p + "s.wo", 1u, &w, (size_t)D * D);
        if ((rc = gemm(EPI_RESID, h->p_att, w, h->p_z, N2, D, D))) return rc;
        if ((rc = ffn(p + "s."))) return rc;
        // latent transformer: the same sequences, self-attention among the K latents
        GET(h, p + "l.ln1", 0u, &f, (size_t)D);
        LAUNCH(launch_rmsnorm(h->p_z, f, h->p_zn, N2, D, k.ln_eps, s));
        GET(h, p + "l.wqkv", 1u, &w, (size_t)3 * D * D);
        if ((rc = gemm(EPI_BF16, h->p_zn, w, h->p_qkv, N2, 3 * D, D))) return rc;
        {
            SeqAttnArgs a{};
            a.q = h->p_qkv; a.k = h->p_qkv + D; a.v = h->p_qkv + 2 * D; a.out = h->p_att; a.bias_off = nullptr;
            a.n_seq = B * T; a.H = Hs; a.Tq = K; a.Tk = K; a.inner_n = 1;
            a.q_outer = a.kv_outer = (long long)K * 3 * D; a.q_step = a.kv_step = 3 * D; a.o_outer = (long long)K * D; a.o_step = D;
            LAUNCH(launch_seq_attention(a, s));
        }
        GET(h, p + "l.wo", 1u, &w, (size_t)D * D);
        if ((rc = gemm(EPI_RESID, h->p_att, w, h->p_z, N2, D, D))) return rc;
        if ((rc = ffn(p + "l."))) return rc;
        // temporal transformer: one sequence per (segment, latent), positions = the T frames (stride K rows), T5 relative bias
        GET(h, p + "t.ln1", 0u, &f, (size_t)D);
        LAUNCH(launch_rmsnorm(h->p_z, f, h->p_zn, N2, D, k.ln_eps, s));
        GET(h, p + "t.wqkv", 1u, &w, (size_t)3 * D * D);
        if ((rc = gemm(EPI_BF16, h->p_zn, w, h->p_qkv, N2, 3 * D, D))) return rc;
        {
            SeqAttnArgs a{};
            a.q = h->p_qkv; a.k = h->p_qkv + D; a.v = h->p_qkv + 2 * D; a.out = h->p_att; a.bias_off = tbias;
            a.n_seq = B * K; a.H = Hs; a.Tq = T; a.Tk = T; a.inner_n = K;
            a.q_outer = a.kv_outer = (long long)T * K * 3 * D; a.q_inner = a.kv_inner = 3 * D; a.q_step = a.kv_step = (long long)K * 3 * D;
            a.o_outer = (long long)T * K * D; a.o_inner = D; a.o_step = (long long)K * D;
            LAUNCH(launch_seq_attention(a, s));
        }
        GET(h, p + "t.wo", 1u, &w, (size_t)D * D);
        if ((rc = gemm(EPI_RESID, h->p_att, w, h->p_z, N2, D, D))) return rc;
        if ((rc = ffn(p + "t."))) return rc;
    }
    // (B, T, K, D) -> per-latent norm -> the K latents of a frame side by side -> d_model
    GET(h, "ptf.ln_out", 0u, &f, (size_t)D);
    LAUNCH(launch_rmsnorm(h->p_z, f, h->p_zn, N2, D, k.ln_eps, s));
    GET(h, "ptf.out_w", 1u, &w, (size_t)d * K * D);
    int rc = gemm(EPI_F32, h->p_zn, w, h->h_enc, B * T, d, K * D);
    if (rc) return rc;
    GET(h, "enc.ln_f", 0u, &f, (size_t)d);
    LAUNCH(launch_rmsnorm(h->h_enc, f, enc_out, B * T, d, k.ln_eps, s));
    HIP_TRY(hipGetLastError());
    return YMT3_OK;
}

static int encode_impl(ymt3_handle h, const float* mel, int B, bf16_t* enc_out, hipStream_t s) {
    const ymt3_config& k = h->cfg;
    if (k.encoder_type == YMT3_ENC_PERCEIVER_TF) return encode_ptf(h, mel, B, enc_out, s);
    const int M = B * h->T, d = k.d_model, inner = h->inner;
    LAUNCH(launch_f32_to_bf16(mel, h->mel_bf, (size_t)M * k.n_mels, s));
    bf16_t* w;
    float* f;
    {
        float* bias;
        GET(h, "in_proj.w", 1u, &w, (size_t)d * k.n_mels);
        GET(h, "in_proj.b", 0u, &bias, (size_t)d);
        GemmArgs g{h->mel_bf, w, h->h_enc, bias, M, d, k.n_mels, k.n_mels, k.n_mels, d, 0, 0, 0};
        LAUNCH(launch_gemm(EPI_F32, g, s));
    }
    const float* bias_off;
    GET(h, "enc.bias_off", 0u, const_cast<float**>(&bias_off), (size_t)k.n_heads * (2 * h->T - 1));
    for (int l = 0; l < k.n_enc_layers; ++l) {
        const std::string p = "enc." + std::to_string(l) + ".";
        GET(h, p + "ln1", 0u, &f, (size_t)d);
        LAUNCH(launch_rmsnorm(h->h_enc, f, h->xn, M, d, k.ln_eps, s));
        GET(h, p + "wqkv", 1u, &w, (size_t)3 * inner * d);
        { GemmArgs g{h->xn, w, h->qkv, nullptr, M, 3 * inner, d, d, d, 3 * inner, 0, 0, 0}; LAUNCH(launch_gemm(EPI_BF16, g, s)); }
        LAUNCH(launch_enc_attention(h->qkv, bias_off, h->attn, B, h->T, k.n_heads, s));
        GET(h, p + "wo", 1u, &w, (size_t)d * inner);
        { GemmArgs g{h->attn, w, h->h_enc, nullptr, M, d, inner, inner, inner, d, 0, 0, 0}; LAUNCH(launch_gemm(EPI_RESID, g, s)); }
        GET(h, p + "ln2", 0u, &f, (size_t)d);
        LAUNCH(launch_rmsnorm(h->h_enc, f, h->xn, M, d, k.ln_eps, s));
        GET(h, p + "wi", 1u, &w, (size_t)k.d_ff * d);
        { GemmArgs g{h->xn, w, h->ff, nullptr, M, k.d_ff, d, d, d, k.d_ff, 0, 0, 0}; LAUNCH(launch_gemm(EPI_BF16_RELU, g, s)); }
        GET(h, p + "wo2", 1u, &w, (size_t)d * k.d_ff);
        { GemmArgs g{h->ff, w, h->h_enc, nullptr, M, d, k.d_ff, k.d_ff, k.d_ff, d, 0, 0, 0}; LAUNCH(launch_gemm(EPI_RESID, g, s)); }
    }
    GET(h, "enc.ln_f", 0u, &f, (size_t)d);
    LAUNCH(launch_rmsnorm(h->h_enc, f, enc_out, M, d, k.ln_eps, s));
    HIP_TRY(hipGetLastError());
    return YMT3_OK;
}

extern "C" int ymt3_encode(ymt3_handle h, const float* mel_dev, int B, void* enc_dev, void* stream) {
    int rc = check_call(h, B);
    if (rc) return rc;
    if (B == 0) return YMT3_OK;
    if (!mel_dev || !enc_dev) FAIL(YMT3_ERR_ARG, "null buffer");
    return encode_impl(h, mel_dev, B, static_cast<bf16_t*>(enc_dev), (hipStream_t)stream);
}

// one decoder step of rows [row0, row0 + R) = 8 kernels per layer + lm_head + argmax, all reading the
// position from the chain's DecodeShared
static int launch_step(ymt3_handle h, int B, int row0, int R, DecodeShared* shared, hipStream_t s, bool solo = true) {
    const ymt3_config& k = h->cfg;
    const int d = k.d_model, inner = h->inner, H = k.n_heads, L = k.max_decode_len;
    const size_t layer_cache = (size_t)h->maxR * H * L * 64;
    const size_t slab = (size_t)B * H * h->T * 64;
    const float* bias_dist;
    GET(h, "dec.bias_dist", 0u, const_cast<float**>(&bias_dist), (size_t)H * L);
    bf16_t* w;
    float* f;
    // every weight of the step up front
    struct LayerW { float *ln1, *ln2, *ln3; bf16_t *wqkv, *wo, *wq_c, *wo_c, *wi, *wo2, *router; uint8_t *wi_q8, *wo_q8; float *wi_s, *wo_s; };
    std::vector<LayerW> LW(k.n_dec_layers);
    for (int l = 0; l < k.n_dec_layers; ++l) {
        const std::string p = "dec." + std::to_string(l) + ".";
        GET(h, p + "ln1", 0u, &LW[l].ln1, (size_t)d);
        GET(h, p + "ln2", 0u, &LW[l].ln2, (size_t)d);
        GET(h, p + "ln3", 0u, &LW[l].ln3, (size_t)d);
        GET(h, p + "wqkv", 1u, &LW[l].wqkv, (size_t)3 * inner * d);
        GET(h, p + "wo", 1u, &LW[l].wo, (size_t)d * inner);
        GET(h, p + "wq_c", 1u, &LW[l].wq_c, (size_t)inner * d);
        GET(h, p + "wo_c", 1u, &LW[l].wo_c, (size_t)d * inner);
        const size_t ne = k.dec_ffn == YMT3_FFN_MOE ? (size_t)k.n_experts : 1;
        LW[l].wi = LW[l].wo2 = nullptr; LW[l].wi_q8 = LW[l].wo_q8 = nullptr; LW[l].wi_s = LW[l].wo_s = nullptr;
        if (k.dec_ffn == YMT3_FFN_MOE && k.moe_fp8) {
            GET(h, p + "wi_q8", 3u, &LW[l].wi_q8, ne * k.d_ff * d);
            GET(h, p + "wo2_q8", 3u, &LW[l].wo_q8, ne * d * k.d_ff);
            GET(h, p + "wi_s", 0u, &LW[l].wi_s, ne);
            GET(h, p + "wo2_s", 0u, &LW[l].wo_s, ne);
        } else {
            GET(h, p + "wi", 1u, &LW[l].wi, ne * k.d_ff * d);
            GET(h, p + "wo2", 1u, &LW[l].wo2, ne * d * k.d_ff);
        }
        LW[l].router = nullptr;
        if (k.dec_ffn == YMT3_FFN_MOE) GET(h, p + "router", 1u, &LW[l].router, (size_t)k.n_experts * d);
    }
    bf16_t* lm_head;
    GET(h, "dec.lm_head", 1u, &lm_head, (size_t)k.vocab * d);
    (void)w;
    h->stamp_n = 0;
    const int mtiles = (R + 15) / 16;
    // MoE with the combine folded away: after an MoE FFN the residual stream is hcur + (y[2r] + y[2r+1]) until the next norm GEMM
    // (the next layer's QKV projection, or lm_head) has formed it; that QKV kernel stores it to the other buffer, which the rest of
    // its layer then uses.  Needs the 16-row decode GEMMs (below the mid-size tile threshold).
    const bool fold_combine = k.dec_ffn == YMT3_FFN_MOE && h->moe_fold_combine && R < (h->mid_rows > 0 ? h->mid_rows : (h->mid_rows < 0 ? DEC_GEMM_MID_ROWS : 1 << 30));
    float* hcur = h->h_dec;
    const float* pend = nullptr;
    // all channels of a segment share its cross-attention K/V: one workgroup per (segment, head) serves them together (mc_cross_attn.hip)
    const bool mc = h->fuse_q && k.n_channels >= 2 && k.n_channels <= 16 && (h->T == 128 || h->T == 256 || h->T == 512) &&
                    row0 % k.n_channels == 0 && R % k.n_channels == 0;
    // fold_o: the self-attention kernel leaves per-head O-projection partials; the fused cross-attention and the cross
    // O-projection's residual read sum them (one launch less per layer, same bits).  Needs the 8-wave attention kernels
    // and the per-row fused cross-attention
    // and pays only while the per-(row, head) pull of wo (64 KB each) stays small against the launch it removes: +0.3 % at 64 rows,
    // -1.4 % at 128, -3.5 % at 256 (profiles/r02_b256_fold_fuseq_variants.txt); same bits either way
    const bool merged_rows = solo && k.n_channels == 1 && row0 == 0 && R <= h->merged_max_rows && h->attn_pair && h->pair_rows;     // (the pair kernel keeps the fold worthwhile beyond 96 rows)
    // (not for one of several concurrent chains: its 64 KB weight pulls per (row, head) share the chip badly -- two 96-row halves with it are no
    // faster than one 192-row chain, without it 6 % faster: profiles/r03_chains_many_rows.txt)
    const bool fold = h->fold_o && h->fuse_q && !mc && !h->force_2wave && H == 8 && d == 512 && ((R <= 96 && solo) || merged_rows);
    // The merged kernels' regime: one channel, up to 64 rows, and this step the only decode stream of the handle (`solo`: with YMT3_CHAINS > 1
    // other row ranges replay on other streams, and the merged kernels need every CU for their own workgroups while they run).
    const bool merged_regime = fold && solo && k.n_channels == 1 && R <= h->merged_max_rows && row0 == 0;
    // GEMM chain (dec_chain.hip): after a layer's cross-attention, ONE launch does the cross O-projection, the FFN and the NEXT
    // layer's QKV projection (or lm_head) -- decided per step shape, same bits as the four launches.
    const bool chain = h->gemm_chain && h->chain_sync && merged_regime && k.dec_ffn != YMT3_FFN_MOE && inner == 512 && k.d_ff == 2048 &&
                       k.vocab % 32 == 0 && k.vocab / 32 >= 32 && k.vocab / 32 <= 64;
    // attention pair (decode.hip: dec_attn_pair_kernel): a layer's two attention kernels as one launch wherever the folded
    // O-projection and the fused query projection apply to one channel of up to 64 rows (dense or MoE FFN alike)
    const bool pair_ok = h->attn_pair && h->pair_rows && merged_regime;
    // the per-step kernel (dec_step.hip): layer 0's QKV projection, then ALL layers' attention pairs and GEMM chains as one launch
    // MoE chain (moe_chain.hip): cross O-projection -> router -> expert FFN-in -> expert FFN-out -> the next QKV projection / lm_head as one launch
    const bool moe_chain = h->moe_chain && h->chain_sync && pair_ok && fold_combine && k.dec_ffn == YMT3_FFN_MOE && k.n_experts == 8 && inner == 512 && k.d_ff == 2048 &&
                           k.vocab % 32 == 0 && k.vocab / 32 >= 32 && k.vocab / 32 <= 64 && h->h_dec2;
    const bool stepk = h->step_kernel && h->step_sync && chain && pair_ok && R <= 64 && k.n_dec_layers <= 8 && h->T <= 0xfff;
    h->step_merged = chain || pair_ok || moe_chain;
    bool qkv_done = false, lm_done = false;         // the previous layer's chain launch already did this layer's QKV / the lm_head
    for (int l = 0; l < k.n_dec_layers; ++l) {
        const LayerW& W = LW[l];
        DecGemmArgs a{};
        a.row0 = row0; a.R = R; a.eps = k.ln_eps; a.H = H; a.L = L; a.shared = shared; a.ssq = h->ssq; a.ssq_stride = h->maxR;
        a.row_pos = h->slot_mode ? h->row_pos : nullptr;
        a.mid_rows = h->mid_rows;
        // self-attention block
        a.x_f32 = hcur; a.gain = W.ln1; a.W = W.wqkv; a.N = 3 * inner; a.K = d; a.out_bf16 = h->dq;
        if (pend) {                                  // the previous layer's expert outputs are still to be added: x -> the other buffer
            a.pend_y = pend;
            a.h_out = hcur == h->h_dec ? h->h_dec2 : h->h_dec;
        }
        a.kcache = h->kcache + l * layer_cache; a.vcache = h->vcache + l * layer_cache;
        if (!qkv_done) {
            a.stamp = next_stamp(h, PC_QKV, a.N / 16 * mtiles);
            PLAUNCH(PC_QKV, launch_dec_gemm(DG_NORM_QKV_CACHE, a, s));
        }
        qkv_done = false;
        if (pend) { hcur = a.h_out; pend = nullptr; a.pend_y = nullptr; a.h_out = nullptr; }
        if (stepk) {
            StepArgs sa{};
            sa.n_layers = k.n_dec_layers; sa.R = R; sa.T = h->T; sa.L = L; sa.ssq_stride = h->maxR; sa.eps = k.ln_eps;
            sa.tiles_free = h->step_tiles_free ? 1 : 0;
            sa.q = h->dq; sa.attn = h->dattn; sa.opart = h->opart; sa.h = hcur; sa.ssq = h->ssq; sa.dff = h->dff; sa.logits = h->logits;
            sa.bias = bias_dist; sa.shared = shared; sa.row_pos = a.row_pos;
            sa.sync = h->step_sync; sa.pair_rows = h->pair_rows; sa.abort_word = h->chain_sync + CHAIN_ABORT_WORD; sa.host_abort = h->chain_host_abort;
            const float* ln_f;
            GET(h, "dec.ln_f", 0u, const_cast<float**>(&ln_f), (size_t)d);
            for (int j = 0; j < k.n_dec_layers; ++j) {
                const bool last = j + 1 == k.n_dec_layers;
                StepLayer& SL = sa.layer[j];
                SL.wo = LW[j].wo; SL.wq_c = LW[j].wq_c; SL.wo_c = LW[j].wo_c; SL.wi = LW[j].wi; SL.wo2 = LW[j].wo2;
                SL.w3 = last ? lm_head : LW[j + 1].wqkv;
                SL.ln2 = LW[j].ln2; SL.ln3 = LW[j].ln3; SL.gain3 = last ? ln_f : LW[j + 1].ln1;
                SL.kself = h->kcache + (size_t)j * layer_cache; SL.vself = h->vcache + (size_t)j * layer_cache;
                SL.kcross = h->ckv + (size_t)(2 * j) * slab; SL.vcross = h->ckv + (size_t)(2 * j + 1) * slab;
                SL.knext = last ? nullptr : h->kcache + (size_t)(j + 1) * layer_cache; SL.vnext = last ? nullptr : h->vcache + (size_t)(j + 1) * layer_cache;
                SL.N3 = last ? k.vocab : 3 * inner; SL.last = last ? 1 : 0;
            }
            sa.stamp = next_stamp(h, PC_STEP, ((R + 15) / 16) * 128);
            PLAUNCH(PC_STEP, launch_dec_step(sa, s));
            lm_done = true;
            break;
        }
        DecAttnArgs t{};
        t.q = h->dq; t.k = a.kcache; t.v = a.vcache; t.out = h->dattn; t.bias = bias_dist; t.shared = shared; t.row0 = row0;
        t.n_keys_const = 0; t.slab_keys = L; t.rows_per_kv = 1; t.R = R; t.H = H; t.bias_stride = L;
        t.row_pos = a.row_pos; t.force_many = h->force_2wave ? 1 : 0;
        if (fold) { t.wo = W.wo; t.opart = h->opart; }
        const bool pair = pair_ok && fold;
        DecAttnArgs ts = t;                          // the self-attention half
        if (!pair) {
            t.stamp = next_stamp(h, PC_SELF_ATTN, R * H);
            PLAUNCH(PC_SELF_ATTN, launch_dec_attention(true, t, s));
        }
        t.wo = nullptr; t.opart = nullptr;
        a.a_bf16 = h->dattn; a.W = W.wo; a.N = d; a.K = inner; a.out_f32 = hcur;
        if (!fold) {
            a.stamp = next_stamp(h, PC_SELF_O, a.N / 16 * mtiles);
            PLAUNCH(PC_SELF_O, launch_dec_gemm(DG_RESID, a, s));
        }
        // cross-attention block: the query projection is fused into the attention kernel (YMT3_NO_FUSEQ=1 keeps
        // the separate skinny GEMM, for A/B measurements)
        t.k = h->ckv + (size_t)(2 * l) * slab; t.v = h->ckv + (size_t)(2 * l + 1) * slab; t.bias = nullptr;
        t.n_keys_const = h->T; t.slab_keys = h->T; t.rows_per_kv = k.n_channels;
        if (fold) t.ipart = h->opart;
        if (mc) {
            // all channels of a segment share its K/V: one workgroup per (segment, head) serves them together
            McCrossArgs mcx{};
            mcx.x_f32 = hcur; mcx.gain = W.ln2; mcx.ssq = h->ssq; mcx.ssq_stride = h->maxR; mcx.eps = k.ln_eps;
            mcx.wq = W.wq_c; mcx.k = t.k + (size_t)(row0 / k.n_channels) * H * h->T * 64; mcx.v = t.v + (size_t)(row0 / k.n_channels) * H * h->T * 64;
            mcx.out = h->dattn; mcx.row0 = row0; mcx.n_seg = R / k.n_channels; mcx.n_channels = k.n_channels; mcx.H = H; mcx.T = h->T;
            PLAUNCH(PC_CROSS_ATTN, launch_mc_cross_attention(mcx, s));
        } else if (h->fuse_q) {
            t.wq = W.wq_c; t.x_f32 = hcur; t.gain = W.ln2; t.ssq = h->ssq; t.ssq_stride = h->maxR; t.eps = k.ln_eps;
        } else {
            a.gain = W.ln2; a.W = W.wq_c; a.N = inner; a.K = d; a.out_bf16 = h->dq;
            a.stamp = next_stamp(h, PC_CROSS_Q, a.N / 16 * mtiles);
            PLAUNCH(PC_CROSS_Q, launch_dec_gemm(DG_NORM_BF16, a, s));
        }
        if (pair) {
            if (chain || moe_chain) t.chain_sync = h->chain_sync;
            ts.stamp = t.stamp = next_stamp(h, PC_ATTN_PAIR, R * H);
            PLAUNCH(PC_ATTN_PAIR, launch_dec_attention_pair(ts, t, h->pair_rows, h->chain_sync + CHAIN_ABORT_WORD, h->chain_host_abort, s));
        } else if (!mc) {
            if (chain) t.chain_sync = h->chain_sync;
            t.stamp = next_stamp(h, PC_CROSS_ATTN, R * H);
            PLAUNCH(PC_CROSS_ATTN, launch_dec_attention(false, t, s));
        }
        if (chain) {
            const bool last = l + 1 == k.n_dec_layers;
            ChainArgs cg{};
            cg.w0 = W.wo_c; cg.w1 = W.wi; cg.w2 = W.wo2; cg.w3 = last ? lm_head : LW[l + 1].wqkv;
            cg.attn = h->dattn; cg.part = h->opart; cg.h = hcur; cg.ssq = h->ssq; cg.ssq_stride = h->maxR;
            cg.gain1 = W.ln3;
            if (last) GET(h, "dec.ln_f", 0u, const_cast<float**>(&cg.gain3), (size_t)d);
            else cg.gain3 = LW[l + 1].ln1;
            cg.dff = h->dff; cg.d_ff = k.d_ff;
            cg.mode3 = last ? DG_NORM_LOGITS : DG_NORM_QKV_CACHE; cg.N3 = last ? k.vocab : 3 * inner;
            cg.out_q = h->dq; cg.kcache = h->kcache + (size_t)(l + 1) * layer_cache; cg.vcache = h->vcache + (size_t)(l + 1) * layer_cache;
            cg.logits = h->logits; cg.H = H; cg.L = L; cg.shared = shared; cg.row_pos = a.row_pos; cg.row0 = row0; cg.R = R; cg.eps = k.ln_eps;
            cg.sync = h->chain_sync; cg.host_abort = h->chain_host_abort;
            { static const int nsub_env = [] { const char* e = getenv("YMT3_CHAIN_NSUB"); return e ? (int)strtol(e, nullptr, 16) : 0; }(); cg.nsub = nsub_env; }
            cg.stamp = next_stamp(h, PC_CHAIN, 256);
            PLAUNCH(PC_CHAIN, launch_dec_chain(cg, s));
            if (last) lm_done = true; else qkv_done = true;
            continue;
        }
        if (moe_chain) {
            const bool last = l + 1 == k.n_dec_layers;
            MoeChainArgs mc2{};
            mc2.wo_c = W.wo_c; mc2.w3 = last ? lm_head : LW[l + 1].wqkv;
            if (k.moe_fp8) { mc2.wi = W.wi_q8; mc2.wo = W.wo_q8; mc2.wi_s = W.wi_s; mc2.wo_s = W.wo_s; }
            else { mc2.wi = W.wi; mc2.wo = W.wo2; }
            mc2.attn = h->dattn; mc2.h = hcur; mc2.part = h->opart; mc2.ssq = h->ssq; mc2.ssq_stride = h->maxR;
            mc2.gain_r = W.ln3; mc2.router = W.router; mc2.xn = h->moe.xn; mc2.sel = h->moe.sel; mc2.gate = h->moe.gate; mc2.hidden = h->moe.hidden; mc2.y = h->moe.y;
            if (last) GET(h, "dec.ln_f", 0u, const_cast<float**>(&mc2.gain3), (size_t)d);
            else mc2.gain3 = LW[l + 1].ln1;
            mc2.mode3 = last ? DG_NORM_LOGITS : DG_NORM_QKV_CACHE; mc2.N3 = last ? k.vocab : 3 * inner;
            mc2.h_out = last ? nullptr : (hcur == h->h_dec ? h->h_dec2 : h->h_dec);
            mc2.out_q = h->dq; mc2.kcache = h->kcache + (size_t)(l + 1) * layer_cache; mc2.vcache = h->vcache + (size_t)(l + 1) * layer_cache;
            mc2.logits = h->logits; mc2.H = H; mc2.L = L; mc2.shared = shared; mc2.row_pos = a.row_pos;
            mc2.R = R; mc2.E = k.n_experts; mc2.fp8 = k.moe_fp8; mc2.eps = k.ln_eps;
            mc2.sync = h->chain_sync; mc2.host_abort = h->chain_host_abort;
            mc2.sel_trace = h->slot_mode ? nullptr : h->moe_trace; mc2.layer = l; mc2.n_layers = k.n_dec_layers;
            mc2.trace_rows = h->moe_trace_rows; mc2.trace_steps = h->moe_trace_steps;
            mc2.stamp = next_stamp(h, PC_CHAIN, 256);
            PLAUNCH(PC_CHAIN, launch_moe_chain(mc2, s));
            if (last) lm_done = true;
            else { qkv_done = true; hcur = mc2.h_out; }
            continue;
        }
        a.a_bf16 = h->dattn; a.W = W.wo_c; a.N = d; a.K = inner;
        a.part = fold ? h->opart : nullptr;
        a.stamp = next_stamp(h, PC_CROSS_O, a.N / 16 * mtiles);
        PLAUNCH(PC_CROSS_O, launch_dec_gemm(DG_RESID, a, s));
        a.part = nullptr;
        // feed-forward block
        if (k.dec_ffn == YMT3_FFN_MOE) {
            MoeArgs mo = h->moe;
            mo.h = hcur; mo.gain = W.ln3; mo.ssq = h->ssq; mo.ssq_stride = h->maxR;
            mo.router = W.router; mo.wi = W.wi; mo.wo = W.wo2; mo.row0 = row0; mo.R = R;
            mo.wi_q8 = W.wi_q8; mo.wo_q8 = W.wo_q8; mo.wi_s = W.wi_s; mo.wo_s = W.wo_s; mo.fp8 = k.moe_fp8;
            mo.sel_trace = h->slot_mode ? nullptr : h->moe_trace; mo.shared = shared; mo.layer = l; mo.n_layers = k.n_dec_layers;
            mo.trace_rows = h->moe_trace_rows; mo.trace_steps = h->moe_trace_steps;
            { ProfScope _ps(h, PC_FFN_WI, s); LAUNCH(launch_moe_stage(0, mo, s)); LAUNCH(launch_moe_stage(1, mo, s)); }
            { ProfScope _ps(h, PC_FFN_WO, s); LAUNCH(launch_moe_stage(2, mo, s)); if (!fold_combine) LAUNCH(launch_moe_stage(3, mo, s)); }
            if (fold_combine) pend = mo.y;
        } else {
            a.gain = W.ln3; a.W = W.wi; a.N = k.d_ff; a.K = d; a.out_bf16 = h->dff;
            a.stamp = next_stamp(h, PC_FFN_WI, a.N / 16 * mtiles);
            PLAUNCH(PC_FFN_WI, launch_dec_gemm(DG_NORM_BF16_RELU, a, s));
            a.a_bf16 = h->dff; a.W = W.wo2; a.N = d; a.K = k.d_ff;
            a.stamp = next_stamp(h, PC_FFN_WO, a.N / 16 * mtiles);
            PLAUNCH(PC_FFN_WO, launch_dec_gemm(DG_RESID, a, s));
        }
    }
    DecGemmArgs a{};
    a.row0 = row0; a.R = R; a.eps = k.ln_eps; a.H = H; a.L = L; a.shared = shared; a.ssq = h->ssq; a.ssq_stride = h->maxR;
    a.mid_rows = h->mid_rows;
    GET(h, "dec.ln_f", 0u, &f, (size_t)d);
    a.x_f32 = hcur; a.gain = f; a.W = lm_head; a.N = k.vocab; a.K = d; a.out_f32 = h->logits;
    a.pend_y = pend;                                 // the last layer's expert outputs, if their combine was folded away
    if (!lm_done) {
        a.stamp = next_stamp(h, PC_LM_HEAD, a.N / 16 * mtiles);
        PLAUNCH(PC_LM_HEAD, launch_dec_gemm(DG_NORM_LOGITS, a, s));
    }
    ArgmaxArgs g{};
    g.logits = h->logits; g.h = h->h_dec; g.shared = shared; g.finished = h->finished; g.ssq = h->ssq; g.ssq_stride = h->maxR; g.row0 = row0;
    g.R = R; g.V = k.vocab; g.d = d; g.n_channels = k.n_channels; g.eos_id = k.eos_id; g.pad_id = k.pad_id;
    GET(h, "dec.embed", 1u, const_cast<bf16_t**>(&g.embed), (size_t)k.vocab * d);
    if (k.n_channels > 1) GET(h, "dec.chan_embed", 1u, const_cast<bf16_t**>(&g.chan_embed), (size_t)k.n_channels * d);
    if (h->slot_mode) { g.row_pos = h->row_pos; g.row_out = h->row_out; }
    if (solo) g.ticket = h->ticket;                  // (row ranges of several chains would share groups)
    if (stepk) { g.zero_sync = h->step_sync; g.zero_lines = k.n_dec_layers * STEP_SYNC_LINES_PER_LAYER; }
    g.stamp = next_stamp(h, PC_ARGMAX, R);
    PLAUNCH(PC_ARGMAX, launch_argmax_embed(g, s));
    return YMT3_OK;
}

static int decode_run(ymt3_handle h, const bf16_t* enc, int B, int n_steps, int32_t* tokens, const int32_t* forced,
                      float* logits_out, hipStream_t s, int prof_stride, int step0);

static int decode_impl(ymt3_handle h, const bf16_t* enc, int B, int n_steps, int32_t* tokens, const int32_t* forced,
                       float* logits_out, hipStream_t s, int prof_stride = 0) {
    const int step0 = h->prof_step0;          // one shot (debug hook): consumed by this call whatever its outcome
    h->prof_step0 = 0;
    return decode_run(h, enc, B, n_steps, tokens, forced, logits_out, s, prof_stride, step0);
}

static int decode_run(ymt3_handle h, const bf16_t* enc, int B, int n_steps, int32_t* tokens, const int32_t* forced,
                      float* logits_out, hipStream_t s, int prof_stride, int step0) {
    const ymt3_config& k = h->cfg;
    bool merged = false;                      // some step of this call ran merged kernels
    if (n_steps <= 0 || step0 + n_steps > k.max_decode_len) FAIL(YMT3_ERR_ARG, "n_steps=%d outside [1, max_decode_len=%d]", n_steps, k.max_decode_len - step0);
    const int d = k.d_model, R = B * k.n_channels;
    // a6: cross-attention K/V of every decoder layer in one GEMM, stored as per-(segment, head) slabs
    GemmArgs g{enc, h->wkv_all, h->ckv, nullptr, B * h->T, k.n_dec_layers * 2 * h->inner, d, d, d, 0, h->T, k.n_heads, B};
    LAUNCH(launch_gemm(EPI_KV_HEADMAJOR, g, s));

    ArgmaxArgs a{};
    a.h = h->h_dec; a.shared = h->shared; a.finished = h->finished; a.ssq = h->ssq; a.ssq_stride = h->maxR;
    a.R = R; a.V = k.vocab; a.d = d; a.n_channels = k.n_channels; a.eos_id = k.eos_id; a.pad_id = k.pad_id;
    GET(h, "dec.embed", 1u, const_cast<bf16_t**>(&a.embed), (size_t)k.vocab * d);
    if (k.n_channels > 1) GET(h, "dec.chan_embed", 1u, const_cast<bf16_t**>(&a.chan_embed), (size_t)k.n_channels * d);
    // chains: contiguous, near-equal row ranges
    int n_chains = (!h->use_graph || prof_stride > 0 || k.dec_ffn == YMT3_FFN_MOE) ? 1 : h->n_chains;   // MoE pair tables are per handle
    if (h->auto_chains && n_chains == 1 && h->use_graph && prof_stride == 0 && k.dec_ffn != YMT3_FFN_MOE && k.n_channels == 1 && R >= 168 && R <= 256 &&
        !(h->early_stop_interval > 0 && k.eos_id >= 0 && !forced))
        n_chains = 2;
    if (n_chains > R) n_chains = R;
    h->last_chains = n_chains;
    if (h->step_kernel && h->step_sync) HIP_TRY(hipMemsetAsync(h->step_sync, 0, (size_t)STEP_SYNC_LINES * CHAIN_LINE * sizeof(unsigned), s));
    LAUNCH(launch_decode_init(a, n_chains, n_steps, step0, tokens, forced, logits_out, s));
    h->last_steps = n_steps;
    int row0[9];
    row0[0] = 0;
    for (int c = 0; c < n_chains; ++c) row0[c + 1] = row0[c] + R / n_chains + (c < R % n_chains ? 1 : 0);

    if (!h->use_graph || prof_stride > 0) {
        // sampled steps (mid-stride: unbiased mean position) bracket every launch; the stride-1 unsampled steps that
        // follow each of them are bracketed as ONE span, which gives the true step time the per-launch brackets are
        // calibrated against (an event pair adds stream time of its own)
        for (int t = 0; t < n_steps; ++t) {
            const bool sampled = prof_stride > 0 && (t % prof_stride) == prof_stride / 2;
            const bool span_begin = prof_stride > 1 && (t % prof_stride) == prof_stride / 2 + 1 && t + prof_stride - 1 <= n_steps;
            const bool span_end = prof_stride > 1 && t >= prof_stride && (t % prof_stride) == prof_stride / 2 && !h->prof_ev.empty() && h->prof_span_open;
            if (span_end) { (void)hipEventRecord(h->prof_ev[h->prof_span_idx], s); h->prof_span_open = false; }
            if (span_begin) {
                hipEvent_t ea, eb;
                if (hipEventCreate(&ea) == hipSuccess && hipEventCreate(&eb) == hipSuccess) {
                    h->prof_ev.push_back(ea); h->prof_ev.push_back(eb); h->prof_cls.push_back(PC_SPAN);
                    h->prof_span_idx = h->prof_ev.size() - 1; h->prof_span_open = true;
                    (void)hipEventRecord(ea, s);
                }
            }
            h->prof_on = sampled;
            int rc = launch_step(h, B, 0, R, h->shared, s);
            h->prof_on = false;
            if (rc) return rc;
            merged = merged || h->step_merged;
        }
        if (h->prof_span_open) {            // a span the loop never closed: drop it (its end event was never recorded)
            (void)hipEventRecord(h->prof_ev[h->prof_span_idx], s);
            h->prof_cls[h->prof_span_idx / 2] = -1;
            h->prof_span_open = false;
        }
    } else {
        hipGraphExec_t exec[8];
        for (int c = 0; c < n_chains; ++c) {
            StepGraph& sg = h->step_graphs[((long)B * 16 + n_chains) * 16 + c];
            if (!sg.exec) {
                HIP_TRY(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
                int rc = launch_step(h, B, row0[c], row0[c + 1] - row0[c], h->shared + c, h->cap_stream, n_chains == 1);
                hipError_t e = hipStreamEndCapture(h->cap_stream, &sg.graph);
                if (rc) return rc;
                if (e != hipSuccess) FAIL(YMT3_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
                HIP_TRY(hipGraphInstantiate(&sg.exec, sg.graph, nullptr, nullptr, 0));
                sg.merged = h->step_merged;
            }
            exec[c] = sg.exec;
            merged = merged || sg.merged;
        }
        if (n_chains == 1 && h->early_stop_interval > 0 && k.eos_id >= 0 && !forced) {
            // opt-in (ymt3_set_early_stop): every `interval` steps the host reads how many rows are still decoding and
            // stops launching once none is; the rest of every row is PAD by the EOS fill rule.  This path synchronises
            // the stream (the only one that does); with forced tokens the trajectory is fixed, so it is not used.
            int t = 0;
            while (t < n_steps) {
                const int chunk = std::min(h->early_stop_interval, n_steps - t);
                for (int i = 0; i < chunk; ++i) HIP_TRY(hipGraphLaunch(exec[0], s));
                t += chunk;
                if (t >= n_steps) break;
                HIP_TRY(hipMemcpyAsync(h->host_flag, &h->shared->n_unfinished, sizeof(int), hipMemcpyDeviceToHost, s));
                HIP_TRY(hipStreamSynchronize(s));
                if (*h->host_flag == 0) break;
            }
            h->last_steps = t;
            LAUNCH(launch_pad_tail(tokens, 0, R, n_steps, t, k.pad_id, s));
        } else if (n_chains == 1) {
            // graph_steps consecutive steps are ONE replayed graph (every kernel reads the position from device memory, so a graph of G
            // steps is the step's kernels G times): the boundary between two graph launches costs ~7 us of stream time that a kernel
            // boundary inside a graph does not (eager launches ran 2.9 % faster than one-step graphs, profiles/r02_graph_steps.txt)
            const int G = h->graph_steps;
            int t = 0;
            if (G > 1 && n_steps >= G) {
                StepGraph& mg = h->step_graphs[(((long)B * 16 + 1) * 16) | (1L << 40) | ((long)G << 32)];
                if (!mg.exec) {
                    HIP_TRY(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
                    int rc = 0;
                    for (int i = 0; i < G && !rc; ++i) rc = launch_step(h, B, 0, R, h->shared, h->cap_stream);
                    hipError_t e = hipStreamEndCapture(h->cap_stream, &mg.graph);
                    if (rc) return rc;
                    if (e != hipSuccess) FAIL(YMT3_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
                    HIP_TRY(hipGraphInstantiate(&mg.exec, mg.graph, nullptr, nullptr, 0));
                    mg.merged = h->step_merged;
                }
                merged = merged || mg.merged;
                for (; t + G <= n_steps; t += G) HIP_TRY(hipGraphLaunch(mg.exec, s));
            }
            for (; t < n_steps; ++t) HIP_TRY(hipGraphLaunch(exec[0], s));
        } else {
            // fork: every chain stream waits for the cross-KV GEMM + init on the caller's stream
            HIP_TRY(hipEventRecord(h->fork_ev, s));
            for (int c = 0; c < n_chains; ++c) HIP_TRY(hipStreamWaitEvent(h->chain_stream[c], h->fork_ev, 0));
            // as for one chain: graph_steps consecutive steps of a chain are ONE replayed graph (the boundary between two graph launches costs ~7 us
            // of stream time that a kernel boundary inside a graph does not)
            const int G = h->graph_steps;
            hipGraphExec_t mexec[8] = {};
            if (G > 1 && n_steps >= G) {
                for (int c = 0; c < n_chains; ++c) {
                    StepGraph& mg = h->step_graphs[((((long)B * 16 + n_chains) * 16 + c)) | (1L << 40) | ((long)G << 32)];
                    if (!mg.exec) {
                        HIP_TRY(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
                        int rc = 0;
                        for (int i = 0; i < G && !rc; ++i) rc = launch_step(h, B, row0[c], row0[c + 1] - row0[c], h->shared + c, h->cap_stream, false);
                        hipError_t e = hipStreamEndCapture(h->cap_stream, &mg.graph);
                        if (rc) return rc;
                        if (e != hipSuccess) FAIL(YMT3_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
                        HIP_TRY(hipGraphInstantiate(&mg.exec, mg.graph, nullptr, nullptr, 0));
                        mg.merged = h->step_merged;
                    }
                    mexec[c] = mg.exec;
                }
            }
            auto feed = [&](int c) -> bool {
                int t = 0;
                if (mexec[c])
                    for (; t + G <= n_steps; t += G)
                        if (hipGraphLaunch(mexec[c], h->chain_stream[c]) != hipSuccess) return false;
                for (; t < n_steps; ++t)
                    if (hipGraphLaunch(exec[c], h->chain_stream[c]) != hipSuccess) return false;
                return true;
            };
            if (h->chain_threads) {
                // one host thread per chain: a hipGraphLaunch of the ~44-node step costs the host 60-150 us, so ONE thread
                // feeding n chains is host-bound as soon as a chain's step is shorter than n launches
                std::atomic<int> bad{0};
                std::vector<std::thread> th;
                for (int c = 1; c < n_chains; ++c)
                    th.emplace_back([&, c] {
                        if (hipSetDevice(h->device) != hipSuccess) { bad = 1; return; }
                        if (!feed(c)) bad = 1;
                    });
                if (!feed(0)) bad = 1;
                for (auto& t : th) t.join();
                if (bad) FAIL(YMT3_ERR_HIP, "hipGraphLaunch failed on a decode chain: %s", hipGetErrorString(hipGetLastError()));
            } else {
                // (the caller's thread feeds all chains: step by step, so that no chain runs ahead of the others' launches)
                for (int t = 0; t < n_steps; ++t)
                    for (int c = 0; c < n_chains; ++c) HIP_TRY(hipGraphLaunch(exec[c], h->chain_stream[c]));
            }
            // join: the caller's stream continues only after every chain has emitted its last token
            for (int c = 0; c < n_chains; ++c) {
                HIP_TRY(hipEventRecord(h->join_ev[c], h->chain_stream[c]));
                HIP_TRY(hipStreamWaitEvent(s, h->join_ev[c], 0));
            }
        }
    }
    if (merged) {
        // a merged launch that gave up on a stage (dec_chain.hip, dec_attn_pair_kernel) must not leave plausible ids behind
        LAUNCH(launch_chain_poison(h->chain_sync, tokens, (long long)R * n_steps, s));
        HIP_TRY(hipGetLastError());
        if (h->abort_recovery && prof_stride == 0) {
            // Recovery (ymt3_set_abort_recovery, default on): wait for the call's own work and look at the abort word.  Raised -- a
            // stage waited > 1 s for workgroups that were not resident: another kernel held CUs -- the call is run again through the
            // separate launches (fresh launches in this process, the same arithmetic bit for bit) and the handle stays on them.
            HIP_TRY(hipStreamSynchronize(s));
            if (h->forced_abort && h->chain_host_abort) *h->chain_host_abort = 1u;      // debug hook: as a kernel would have during the call
            if (h->chain_host_abort && *static_cast<volatile unsigned*>(h->chain_host_abort)) {
                int rc = merged_fallback(h);
                if (rc) return rc;
                return decode_run(h, enc, B, n_steps, tokens, forced, logits_out, s, prof_stride, step0);
            }
        } else if (h->forced_abort && h->chain_host_abort) {
            *h->chain_host_abort = 1u;            // asynchronous mode: the next call on the handle finds the word (check_call)
            h->forced_abort = false;
        }
    }
    HIP_TRY(hipGetLastError());
    return YMT3_OK;
}

extern "C" int ymt3_decode_greedy(ymt3_handle h, const void* enc_dev, int B, int n_steps, int32_t* tokens_dev,
                                  const int32_t* forced_dev, float* logits_dev, void* stream) {
    int rc = check_call(h, B);
    if (rc) return rc;
    if (B == 0) return YMT3_OK;
    if (!enc_dev || !tokens_dev) FAIL(YMT3_ERR_ARG, "null buffer");
    return decode_impl(h, static_cast<const bf16_t*>(enc_dev), B, n_steps, tokens_dev, forced_dev, logits_dev, (hipStream_t)stream);
}

extern "C" int ymt3_transcribe_segments(ymt3_handle h, const float* audio_dev, int B, int n_steps, int32_t* tokens_dev,
                                        void* stream) {
    int rc = check_call(h, B);
    if (rc) return rc;
    if (B == 0) return YMT3_OK;
    if (!audio_dev || !tokens_dev) FAIL(YMT3_ERR_ARG, "null buffer");
    hipStream_t s = (hipStream_t)stream;
    LAUNCH(launch_logmel(h->fe, audio_dev, h->mel, B, s));
    rc = encode_impl(h, h->mel, B, h->enc_out, s);
    if (rc) return rc;
    return decode_impl(h, h->enc_out, B, n_steps, tokens_dev, nullptr, nullptr, s);
}

// continuous batching.  `slots` decoder slots are kept busy from a queue of segments: each row
// decodes at its own position (slot mode of the step kernels), the host looks at the per-row `finished` flags every
// `interval` steps, pads and retires segments whose rows have all stopped, and encodes the next pending segments straight
// into the freed slots (log-mel + encoder batched over the admissions, cross-K/V written into each slot's slabs).  Rows are
// independent in every kernel, so the ids equal those of lock-step batches bit for bit.
extern "C" int ymt3_transcribe_stream(ymt3_handle h, const float* audio_dev, int n_segments, int n_steps, int32_t* tokens_dev,
                                      int slots, int interval, void* stream) {
    int rc = check_call(h, 0);
    if (rc) return rc;
    if (n_segments < 0) FAIL(YMT3_ERR_ARG, "n_segments=%d", n_segments);
    if (n_segments == 0) return YMT3_OK;
    if (!audio_dev || !tokens_dev) FAIL(YMT3_ERR_ARG, "null buffer");
    const ymt3_config& k = h->cfg;
    if (n_steps <= 0 || n_steps > k.max_decode_len) FAIL(YMT3_ERR_ARG, "n_steps=%d outside [1, max_decode_len=%d]", n_steps, k.max_decode_len);
    if (interval < 0) FAIL(YMT3_ERR_ARG, "interval=%d", interval);
    if (h->prof_step0) FAIL(YMT3_ERR_UNSUPPORTED, "ymt3_debug_decode_start is pending: it applies to lock-step decode calls only");
    if (interval == 0) interval = 8;
    if (slots <= 0 || slots > h->maxB) slots = h->maxB;
    if (slots > n_segments) slots = n_segments;
    hipStream_t s = (hipStream_t)stream;
    if (!h->host_rows) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->host_rows), (size_t)h->maxR * sizeof(int), hipHostMallocDefault));
    const int K = k.n_channels, R = slots * K, d = k.d_model, T = h->T, H = k.n_heads;
    const size_t S = (size_t)k.segment_samples;

    ArgmaxArgs a{};
    a.h = h->h_dec; a.shared = h->shared; a.finished = h->finished; a.ssq = h->ssq; a.ssq_stride = h->maxR;
    a.R = R; a.V = k.vocab; a.d = d; a.n_channels = K; a.eos_id = k.eos_id; a.pad_id = k.pad_id;
    a.row_pos = h->row_pos; a.row_out = h->row_out;
    GET(h, "dec.embed", 1u, const_cast<bf16_t**>(&a.embed), (size_t)k.vocab * d);
    if (K > 1) GET(h, "dec.chan_embed", 1u, const_cast<bf16_t**>(&a.chan_embed), (size_t)K * d);

    struct ModeGuard { ymt3_ctx* c; ~ModeGuard() { c->slot_mode = false; } } guard{h};
    h->slot_mode = true;
    // loop state: every row starts stopped; admissions start them
    if (h->step_kernel && h->step_sync) HIP_TRY(hipMemsetAsync(h->step_sync, 0, (size_t)STEP_SYNC_LINES * CHAIN_LINE * sizeof(unsigned), s));
    LAUNCH(launch_decode_init(a, 1, n_steps, 0, tokens_dev, nullptr, nullptr, s));
    HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(h->finished), 1, (size_t)R, s));
    HIP_TRY(hipMemsetAsync(h->row_pos, 0, (size_t)R * sizeof(int), s));
    HIP_TRY(hipMemsetAsync(h->row_out, 0, (size_t)R * sizeof(long long), s));

    hipGraphExec_t exec = nullptr;
    // one replayed graph per round of `interval` steps when that is at most 64 steps (a graph launch costs ~7 us of stream time on
    // top of its kernels, see decode_impl); else one per step
    const int per_graph = interval <= 64 ? interval : 1;
    if (h->use_graph) {
        StepGraph& sg = h->step_graphs[(((long)slots * 16 + 15) * 16) | ((long)per_graph << 32)];      // 15: slot-mode graph of `slots` segments
        if (!sg.exec) {
            HIP_TRY(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
            int rcs = 0;
            for (int i = 0; i < per_graph && !rcs; ++i) rcs = launch_step(h, slots, 0, R, h->shared, h->cap_stream);
            hipError_t e = hipStreamEndCapture(h->cap_stream, &sg.graph);
            if (rcs) return rcs;
            if (e != hipSuccess) FAIL(YMT3_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
            HIP_TRY(hipGraphInstantiate(&sg.exec, sg.graph, nullptr, nullptr, 0));
            sg.merged = h->step_merged;
        }
        exec = sg.exec;
    }

    std::vector<int> slot_seg((size_t)slots, -1), free_slots;
    auto admit = [&](int first_seg, int nb) -> int {
        LAUNCH(launch_logmel(h->fe, audio_dev + (size_t)first_seg * S, h->mel, nb, s));
        int rce = encode_impl(h, h->mel, nb, h->enc_out, s);
        if (rce) return rce;
        for (int i = 0; i < nb; ++i) {
            const int slot = free_slots[(size_t)i];
            GemmArgs g{h->enc_out + (size_t)i * T * d, h->wkv_all, h->ckv + (size_t)slot * H * T * 64, nullptr,
                       T, k.n_dec_layers * 2 * h->inner, d, d, d, 0, T, H, slots};
            LAUNCH(launch_gemm(EPI_KV_HEADMAJOR, g, s));
            LAUNCH(launch_slot_start(a, slot * K, (long long)(first_seg + i) * K * n_steps, n_steps, h->row_out, s));
            slot_seg[(size_t)slot] = first_seg + i;
        }
        return YMT3_OK;
    };
    for (int i = 0; i < slots; ++i) free_slots.push_back(i);
    rc = admit(0, slots);
    if (rc) return rc;
    int next = slots, live = slots;
    // every live segment stops within n_steps steps, so the loop is bounded; the guard only catches a logic error
    const long max_rounds = ((long)n_segments / slots + 2) * ((n_steps + interval - 1) / interval + 1);
    h->last_steps = 0;
    for (long round = 0; live > 0; ++round) {
        h->last_steps += interval;
        if (round > max_rounds) FAIL(YMT3_ERR_HIP, "slot scheduler made no progress (%d live, %d admitted of %d)", live, next, n_segments);
        for (int i = 0; i < interval; i += exec ? per_graph : 1) {
            if (exec) HIP_TRY(hipGraphLaunch(exec, s));
            else { int rcs = launch_step(h, slots, 0, R, h->shared, s); if (rcs) return rcs; }
        }
        HIP_TRY(hipMemcpyAsync(h->host_rows, h->finished, (size_t)R * sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (h->forced_abort && h->chain_host_abort) *h->chain_host_abort = 1u;          // debug hook: as a kernel would have during the round
        if (h->chain_host_abort && *static_cast<volatile unsigned*>(h->chain_host_abort)) {
            // a merged kernel gave up (the stream is idle here): start the queue again on the separate launches -- same ids
            rc = merged_fallback(h);
            if (rc) return rc;
            return ymt3_transcribe_stream(h, audio_dev, n_segments, n_steps, tokens_dev, slots, interval, stream);
        }
        free_slots.clear();
        for (int slot = 0; slot < slots; ++slot) {
            if (slot_seg[(size_t)slot] < 0) continue;
            bool all = true;
            for (int c = 0; c < K; ++c) all = all && h->host_rows[slot * K + c] != 0;
            if (!all) continue;
            LAUNCH(launch_slot_retire(a, slot * K, K, n_steps, tokens_dev, s));
            slot_seg[(size_t)slot] = -1;
            --live;
            free_slots.push_back(slot);
        }
        const int nb = std::min((int)free_slots.size(), n_segments - next);
        if (nb > 0) {
            rc = admit(next, nb);
            if (rc) return rc;
            next += nb;
            live += nb;
        }
    }
    HIP_TRY(hipGetLastError());
    return YMT3_OK;
}

extern "C" int ymt3_test_gemm(ymt3_handle h, const void* a_dev, const void* w_dev, float* c_dev, int M, int N, int K, void* stream) {
    int rc = check_call(h, 0);
    if (rc) return rc;
    GemmArgs g{static_cast<const bf16_t*>(a_dev), static_cast<const bf16_t*>(w_dev), c_dev, nullptr, M, N, K, K, K, N, 0, 0, 0};
    LAUNCH(launch_gemm(EPI_F32, g, (hipStream_t)stream));
    HIP_TRY(hipGetLastError());
    return YMT3_OK;
}

extern "C" int ymt3_profile_decode(ymt3_handle h, const void* enc_dev, int B, int n_steps, int stride, int32_t* tokens_dev,
                                   float* ms_by_class, int32_t* launches_by_class, void* stream) {
    int rc = check_call(h, B);
    if (rc) return rc;
    if (!enc_dev || !tokens_dev || !ms_by_class || !launches_by_class || stride <= 0 || B == 0) FAIL(YMT3_ERR_ARG, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    h->prof_ev.clear();
    h->prof_cls.clear();
    rc = decode_impl(h, static_cast<const bf16_t*>(enc_dev), B, n_steps, tokens_dev, nullptr, nullptr, s, stride);
    hipError_t e = hipStreamSynchronize(s);
    for (int i = 0; i < YMT3_PROFILE_CLASSES; ++i) { ms_by_class[i] = 0.f; launches_by_class[i] = 0; }
    for (size_t i = 0; i < h->prof_cls.size(); ++i) {
        float ms = 0.f;
        if (rc == 0 && e == hipSuccess && h->prof_cls[i] >= 0 &&
            hipEventElapsedTime(&ms, h->prof_ev[2 * i], h->prof_ev[2 * i + 1]) == hipSuccess) {
            ms_by_class[h->prof_cls[i]] += ms;
            launches_by_class[h->prof_cls[i]] += 1;
        }
        (void)hipEventDestroy(h->prof_ev[2 * i]);
        (void)hipEventDestroy(h->prof_ev[2 * i + 1]);
    }
    h->prof_ev.clear();
    h->prof_cls.clear();
    if (rc) return rc;
    if (e != hipSuccess) FAIL(YMT3_ERR_HIP, "hipStreamSynchronize: %s", hipGetErrorString(e));
    return YMT3_OK;
}
static_assert(PC_COUNT <= YMT3_PROFILE_CLASSES, "profile class table");

extern "C" int ymt3_debug_step_stamps(ymt3_handle h, int32_t* cls, int32_t* grid, uint64_t* stats, int* n_kernels) {
    if (!h || !cls || !grid || !stats || !n_kernels) FAIL(YMT3_ERR_ARG, "null argument");
    if (!h->stamp_buf) FAIL(YMT3_ERR_UNSUPPORTED, "stamps are recorded only by a handle created with YMT3_STAMP=1 in the environment");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    std::vector<unsigned long long> host((size_t)STAMP_WGS * 2);
    *n_kernels = h->stamp_n;
    for (int i = 0; i < h->stamp_n; ++i) {
        const int gsz = h->stamp_grid[i];
        HIP_TRY(hipMemcpy(host.data(), h->stamp_buf + (size_t)i * STAMP_WGS * 2, (size_t)gsz * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long in_min = ~0ull, in_max = 0, out_min = ~0ull, out_max = 0;
        for (int w = 0; w < gsz; ++w) {
            if (host[2 * w] == 0) continue;                          // a launcher may use fewer workgroups than the slot reserves
            in_min = std::min(in_min, host[2 * w]); in_max = std::max(in_max, host[2 * w]);
            out_min = std::min(out_min, host[2 * w + 1]); out_max = std::max(out_max, host[2 * w + 1]);
        }
        cls[i] = h->stamp_cls[i]; grid[i] = gsz;
        stats[4 * i] = in_min; stats[4 * i + 1] = in_max; stats[4 * i + 2] = out_min; stats[4 * i + 3] = out_max;
    }
    return YMT3_OK;
}

extern "C" int ymt3_debug_kernel_stamps(ymt3_handle h, int kernel, uint64_t* stamps, int capacity_wgs) {
    if (!h || !stamps) FAIL(YMT3_ERR_ARG, "null argument");
    if (!h->stamp_buf) FAIL(YMT3_ERR_UNSUPPORTED, "stamps are recorded only by a handle created with YMT3_STAMP=1 in the environment");
    if (kernel < 0 || kernel >= h->stamp_n || capacity_wgs < h->stamp_grid[kernel]) FAIL(YMT3_ERR_ARG, "kernel=%d of %d, capacity %d", kernel, h->stamp_n, capacity_wgs);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    // the caller's capacity decides how much of the kernel's slot comes back (the GEMM chain keeps stage marks behind its [grid][2] stamps)
    const size_t n_wgs = capacity_wgs < STAMP_WGS ? (size_t)capacity_wgs : (size_t)STAMP_WGS;
    HIP_TRY(hipMemcpy(stamps, h->stamp_buf + (size_t)kernel * STAMP_WGS * 2, n_wgs * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return YMT3_OK;
}

extern "C" int ymt3_debug_force_stage_abort(ymt3_handle h) {
    if (!h) FAIL(YMT3_ERR_ARG, "null handle");
    if (!h->debug_hooks) FAIL(YMT3_ERR_UNSUPPORTED, "debug hooks are accepted only by a handle created with YMT3_DEBUG_HOOKS=1 in the environment");
    if (!h->chain_sync || !h->chain_host_abort || !(h->gemm_chain || h->attn_pair || h->moe_chain)) FAIL(YMT3_ERR_UNSUPPORTED, "this handle does not run the merged decode kernels");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    const unsigned one = 1u;
    // the device word now (what the kernels and the poison pass look at); the host word is what a kernel would set with it --
    // left for the poison pass's caller to observe: set it only after the next decode call, as the kernel would during that call
    HIP_TRY(hipMemcpy(h->chain_sync + CHAIN_ABORT_WORD, &one, sizeof(one), hipMemcpyHostToDevice));
    h->forced_abort = true;
    return YMT3_OK;
}

extern "C" int ymt3_debug_moe_trace(ymt3_handle h, int32_t* trace_dev, int n_steps, int n_rows) {
    if (!h) FAIL(YMT3_ERR_ARG, "null handle");
    if (!h->debug_hooks) FAIL(YMT3_ERR_UNSUPPORTED, "debug hooks are accepted only by a handle created with YMT3_DEBUG_HOOKS=1 in the environment");
    if (h->cfg.dec_ffn != YMT3_FFN_MOE) FAIL(YMT3_ERR_UNSUPPORTED, "this handle has no MoE router");
    if (trace_dev && (n_steps <= 0 || n_rows <= 0)) FAIL(YMT3_ERR_ARG, "n_steps=%d n_rows=%d", n_steps, n_rows);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    for (auto& kv : h->step_graphs) {        // cached step graphs carry the old pointer in their kernel arguments
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
        if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
    }
    h->step_graphs.clear();
    h->moe_trace = trace_dev;
    h->moe_trace_steps = trace_dev ? n_steps : 0;
    h->moe_trace_rows = trace_dev ? n_rows : 0;
    return YMT3_OK;
}

extern "C" int ymt3_debug_decode_start(ymt3_handle h, int step0) {
    if (!h) FAIL(YMT3_ERR_ARG, "null handle");
    if (!h->debug_hooks) FAIL(YMT3_ERR_UNSUPPORTED, "debug hooks are accepted only by a handle created with YMT3_DEBUG_HOOKS=1 in the environment");
    if (step0 < 0 || step0 >= h->cfg.max_decode_len) FAIL(YMT3_ERR_ARG, "step0=%d outside [0, %d)", step0, h->cfg.max_decode_len);
    HIP_TRY(hipSetDevice(h->device));
    if (step0 > 0) {
        // positions [0, step0) of every (layer, row, head) slab: defined (zero) keys and values instead of whatever hipMalloc left
        const ymt3_config& k = h->cfg;
        const size_t slabs = (size_t)k.n_dec_layers * h->maxR * k.n_heads, pitch = (size_t)k.max_decode_len * 64 * sizeof(bf16_t);
        HIP_TRY(hipMemset2D(h->kcache, pitch, 0, (size_t)step0 * 64 * sizeof(bf16_t), slabs));
        HIP_TRY(hipMemset2D(h->vcache, pitch, 0, (size_t)step0 * 64 * sizeof(bf16_t), slabs));
        HIP_TRY(hipDeviceSynchronize());
    }
    h->prof_step0 = step0;
    return YMT3_OK;
}

// ------------------------------------------------------------------------------------------------
// Audio ingest.  Filter design on the host, in double: the Kaiser(5.0) windowed-sinc
// low-pass and the alignment of TP: scipy/signal/_signaltools.py resample_poly (restated in oracle/ingest_oracle.py).
static double bessel_i0(double x) {
    double sum = 1.0, term = 1.0;
    for (int k = 1; k < 500; ++k) {
        term *= x / (2.0 * k);
        const double t2 = term * term;
        sum += t2;
        if (t2 < 1e-20 * sum) break;
    }
    return sum;
}

static long long gcd_ll(long long a, long long b) { while (b) { const long long t = a % b; a = b; b = t; } return a; }

static int get_resampler(ymt3_ctx* c, int sr_in, const ymt3_ctx::Resampler** out) {
    const long long g = gcd_ll(sr_in, c->cfg.sample_rate);
    const int up = (int)(c->cfg.sample_rate / g), down = (int)(sr_in / g);
    auto it = c->resamplers.find({up, down});
    if (it != c->resamplers.end()) { *out = &it->second; return YMT3_OK; }
    const int max_rate = std::max(up, down);
    if (max_rate > 16384) FAIL(YMT3_ERR_UNSUPPORTED, "resampling ratio %d/%d needs a %d-tap filter; unsupported", up, down, 20 * max_rate + 1);
    ymt3_ctx::Resampler rs;
    rs.up = up; rs.down = down;
    std::vector<double> hp;
    if (up == 1 && down == 1) {
        hp.assign(1, 1.0);
        rs.r = 0;
    } else {
        const int half_len = 10 * max_rate, n = 2 * half_len + 1;
        const int n_pre_pad = down - half_len % down;
        rs.r = (half_len + n_pre_pad) / down;
        hp.assign((size_t)n_pre_pad + n, 0.0);
        const double fc = 1.0 / max_rate, i0b = bessel_i0(5.0), pi = 3.14159265358979323846;
        double sum = 0.0;
        for (int i = 0; i < n; ++i) {
            const double m = (double)(i - half_len), xx = pi * fc * m;
            const double sinc = m == 0.0 ? 1.0 : std::sin(xx) / xx;
            const double rel = m / half_len;
            const double w = bessel_i0(5.0 * std::sqrt(std::max(0.0, 1.0 - rel * rel))) / i0b;
            hp[(size_t)n_pre_pad + i] = fc * sinc * w;
            sum += hp[(size_t)n_pre_pad + i];
        }
        for (int i = 0; i < n; ++i) hp[(size_t)n_pre_pad + i] *= (double)up / sum;
    }
    rs.J = (int)((hp.size() + up - 1) / up);
    rs.Jp = (rs.J + 3) / 4 * 4;
    rs.window = (int)((255LL * down) / up) + rs.J + 2;
    if ((size_t)rs.window * sizeof(float) > 64 * 1024) FAIL(YMT3_ERR_UNSUPPORTED, "resampling ratio %d/%d needs a %d-sample LDS window; unsupported", up, down, rs.window);
    std::vector<float> P((size_t)up * rs.Jp, 0.f);
    for (size_t i = 0; i < hp.size(); ++i) P[(i % up) * rs.Jp + i / up] = (float)hp[i];
    void* dev = nullptr;
    int rc = dev_alloc(c, &dev, P.size() * sizeof(float));
    if (rc) return rc;
    HIP_TRY(hipMemcpy(dev, P.data(), P.size() * sizeof(float), hipMemcpyHostToDevice));
    rs.taps = static_cast<float*>(dev);
    *out = &(c->resamplers[{up, down}] = rs);
    return YMT3_OK;
}

extern "C" int ymt3_ingest_plan(ymt3_handle h, int64_t n_frames, int sample_rate_in, int64_t* n_samples_out, int* n_segments) {
    if (!h) FAIL(YMT3_ERR_ARG, "null handle");
    if (n_frames < 0 || sample_rate_in <= 0) FAIL(YMT3_ERR_ARG, "n_frames=%lld sample_rate_in=%d", (long long)n_frames, sample_rate_in);
    const long long g = gcd_ll(sample_rate_in, h->cfg.sample_rate);
    const long long up = h->cfg.sample_rate / g, down = sample_rate_in / g;
    const long long n_out = (n_frames * up + down - 1) / down;
    const long long S = h->cfg.segment_samples;
    const long long n_seg = std::max(1LL, (n_out + S - 1) / S);
    if (n_seg > 0x7fffffffLL) FAIL(YMT3_ERR_ARG, "too many segments");
    if (n_samples_out) *n_samples_out = n_out;
    if (n_segments) *n_segments = (int)n_seg;
    return YMT3_OK;
}

extern "C" int ymt3_ingest(ymt3_handle h, const void* pcm_dev, int pcm_format, int64_t n_frames, int n_channels,
                           int sample_rate_in, float* segments_dev, int n_segments, void* stream) {
    if (!h) FAIL(YMT3_ERR_ARG, "null handle");
    if (pcm_format != YMT3_PCM_S16 && pcm_format != YMT3_PCM_F32) FAIL(YMT3_ERR_ARG, "pcm_format=%d", pcm_format);
    if (n_frames < 0 || n_channels < 1 || n_channels > 64 || sample_rate_in <= 0 || n_segments < 1)
        FAIL(YMT3_ERR_ARG, "n_frames=%lld n_channels=%d sample_rate_in=%d n_segments=%d", (long long)n_frames, n_channels, sample_rate_in, n_segments);
    if (!segments_dev || (n_frames > 0 && !pcm_dev)) FAIL(YMT3_ERR_ARG, "null buffer");
    int64_t n_out = 0;
    int need = 0;
    int rc = ymt3_ingest_plan(h, n_frames, sample_rate_in, &n_out, &need);
    if (rc) return rc;
    if (n_segments < need) FAIL(YMT3_ERR_ARG, "n_segments=%d but %lld resampled samples need %d", n_segments, (long long)n_out, need);
    HIP_TRY(hipSetDevice(h->device));
    const ymt3_ctx::Resampler* rs = nullptr;
    rc = get_resampler(h, sample_rate_in, &rs);
    if (rc) return rc;
    IngestArgs a{};
    a.pcm = pcm_dev; a.taps = rs->taps; a.out = segments_dev;
    a.n_in = n_frames; a.n_out = n_out; a.n_total = (long long)n_segments * h->cfg.segment_samples; a.r = rs->r;
    a.up = rs->up; a.down = rs->down; a.J = rs->J; a.Jp = rs->Jp; a.n_channels = n_channels; a.s16 = pcm_format == YMT3_PCM_S16; a.window = rs->window;
    LAUNCH(launch_ingest(a, (hipStream_t)stream));
    HIP_TRY(hipGetLastError());
    return YMT3_OK;
}

extern "C" int ymt3_last_decode_steps(ymt3_handle h) { return h ? h->last_steps : 0; }

extern "C" int ymt3_merged_fallbacks(ymt3_handle h) { return h ? h->fallback_count : 0; }
extern "C" int ymt3_last_decode_chains(ymt3_handle h) { return h ? h->last_chains : 0; }

extern "C" int ymt3_set_abort_recovery(ymt3_handle h, int mode) {
    if (!h) FAIL(YMT3_ERR_ARG, "null handle");
    if (mode != 0 && mode != 1) FAIL(YMT3_ERR_ARG, "mode must be 0 (asynchronous, lazy) or 1 (verify every merged decode call)");
    h->abort_recovery = mode;
    return YMT3_OK;
}

extern "C" int ymt3_set_early_stop(ymt3_handle h, int interval) {
    if (!h) FAIL(YMT3_ERR_ARG, "null handle");
    if (interval < 0) FAIL(YMT3_ERR_ARG, "interval must be >= 0");
    HIP_TRY(hipSetDevice(h->device));
    if (interval > 0 && !h->host_flag) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->host_flag), sizeof(int), hipHostMallocDefault));
    h->early_stop_interval = interval;
    return YMT3_OK;
}
