// Launcher declarations shared between the kernel translation units and runtime.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"

// ---------------------------------------------------------------- front-end (frontend.hip)
struct FrontendTables {
    const float* window;      // [n_fft] periodic Hann
    const float2* tw;         // [n_fft/2]   exp(-2 pi i m / (n_fft/2))
    const float2* untw;       // [n_fft/2+1] exp(-2 pi i k / n_fft)
    const int* mel_start;     // [n_mels] first bin of each triangle
    const int* mel_len;       // [n_mels] bins in each triangle
    const int* mel_off;       // [n_mels] offset into mel_w
    const float* mel_w;       // concatenated triangle weights
    int n_fft, hop, n_mels, n_samples, n_frames, n_mel_w;
    float log_floor;
};
int launch_logmel(const FrontendTables& t, const float* audio, float* mel, int B, hipStream_t stream);

// ---------------------------------------------------------------- audio ingest (ingest.hip)
struct IngestArgs {
    const void* pcm;          // [n_in][n_channels] interleaved int16 or fp32
    const float* taps;        // [up][Jp] polyphase rows of the low-pass, zero padded beyond J
    float* out;               // [n_total] = (n_seg, segment_samples), zero beyond n_out
    long long n_in, n_out, n_total, r;   // r: output alignment offset (n_pre_remove of resample_poly)
    int up, down, J, Jp, n_channels, s16, window;   // window: LDS floats per workgroup
};
int launch_ingest(const IngestArgs& a, hipStream_t stream);

// ---------------------------------------------------------------- dense GEMM (gemm.hip)
// C[M][N] (+)= A[M][K] (bf16, row stride lda) * W[N][K]^T (bf16, row stride ldw), fp32 accumulate.
enum GemmEpilogue {
    EPI_F32 = 0,          // out f32 [M][ldc] = acc (+ bias[n])
    EPI_BF16 = 1,         // out bf16 [M][ldc] = R(acc)
    EPI_BF16_RELU = 2,    // out bf16 [M][ldc] = R(max(acc, 0))
    EPI_RESID = 3,        // out f32 [M][ldc] += acc
    EPI_KV_HEADMAJOR = 4, // out bf16 [n / (H*64)][m / T][h][m % T][64]  (cross-attention K/V slabs)
};
struct GemmArgs {
    const bf16_t* A; const bf16_t* W; void* out; const float* bias;
    int M, N, K, lda, ldw, ldc;
    int T, H, n_seg;      // EPI_KV_HEADMAJOR only: frames per segment, heads, segments
};
int launch_gemm(int epilogue, const GemmArgs& a, hipStream_t stream);
int init_gemm_kernels();

// ---------------------------------------------------------------- norm / casts (norm.hip)
// out bf16 [M][d] = R(x * rsqrt(mean(x^2) + eps) * gain)
int launch_rmsnorm(const float* x, const float* gain, bf16_t* out, int M, int d, float eps, hipStream_t stream);
int launch_f32_to_bf16(const float* x, bf16_t* out, size_t n, hipStream_t stream);
// out f32 [B][n] = src bf16 [n] repeated for every b (the latent array as the initial residual stream)
int launch_broadcast_bf16(const bf16_t* src, float* out, int B, size_t n, hipStream_t stream);

// ---------------------------------------------------------------- encoder attention (enc_attn.hip)
// qkv bf16 [B*T][3*H*64] -> out bf16 [B*T][H*64]; bias_off f32 [H][2T-1] indexed by key - query + T-1
int init_enc_attn_kernels();
// general form: queries and keys/values from separate buffers (latent cross-attention, a9)
int launch_enc_attention_qkv(const bf16_t* q, int ldq, const bf16_t* k, const bf16_t* v, int ldkv, const float* bias_off,
                             bf16_t* out, int B, int T, int H, hipStream_t stream);
int launch_enc_attention(const bf16_t* qkv, const float* bias_off, bf16_t* out, int B, int T, int H, hipStream_t stream);

// Batched small-sequence attention for the Perceiver-TF encoder (a9): n_seq independent sequences, heads of 64, Tq queries
// over Tk keys per sequence.  Sequence s starts at element (s / inner_n) * outer + (s % inner_n) * inner of each buffer and
// its consecutive positions are `step` elements apart -- so the same kernel serves sequences that are contiguous runs of rows
// (spectral cross-attention and latent self-attention: one sequence per (segment, frame)) and sequences strided through
// them (temporal self-attention: one sequence per (segment, latent), positions = frames).
struct SeqAttnArgs {
    const bf16_t* q; const bf16_t* k; const bf16_t* v; bf16_t* out;
    const float* bias_off;        // [H][2*Tk-1] by key - query + Tk - 1 (needs Tq == Tk), or null: no bias
    int n_seq, H, Tq, Tk, inner_n;
    long long q_outer, q_inner, q_step, kv_outer, kv_inner, kv_step, o_outer, o_inner, o_step;
};
int launch_seq_attention(const SeqAttnArgs& a, hipStream_t stream);
// spectral tokens of the Perceiver-TF encoder: out bf16 [n_rows][d] = R(rmsnorm(mel[row] * w + pos[row % F]) * gain)
int launch_spec_embed(const float* mel, const float* w, const bf16_t* pos, const float* gain, bf16_t* out, long long n_rows, int F, int d,
                      float eps, hipStream_t stream);

// ---------------------------------------------------------------- decoder step (decode.hip)
struct DecodeShared {           // device-resident loop state, read by every decode kernel
    int step;                   // position being decoded (tokens already in the cache)
    int done_count;             // ticket counter of the argmax kernel
    int n_steps;                // row stride of tokens_out / forced / logits_out
    int step0;                  // first position of this call (0 except under the debug hook ymt3_debug_decode_start)
    int n_unfinished;           // rows of this chain that have not emitted EOS yet (maintained when eos_id >= 0)
    int pad1;
    int32_t* tokens_out;        // [R][n_steps]
    const int32_t* forced;      // [R][n_steps] or null
    float* logits_out;          // [R][n_steps][V] or null
};

constexpr int SSQ_TILES = 32;    // sum(h^2) partials per row = d_model / 16 column tiles of the RESID epilogue

struct DecGemmArgs {
    const float* x_f32;         // NORM variants: residual stream [R][K] fp32
    const float* gain;          // NORM variants: [K]
    const bf16_t* a_bf16;       // plain variants: [R][K] bf16
    const bf16_t* W;            // [N][K] bf16
    int row0, R, N, K;          // rows [row0, row0 + R) of every row-indexed buffer
    float eps;
    // outputs (by mode)
    bf16_t* out_bf16;           // [R][N] (MODE_BF16, MODE_BF16_RELU, q part of MODE_QKV_CACHE)
    float* out_f32;             // [R][N] (MODE_RESID: +=, MODE_LOGITS: =)
    bf16_t* kcache;             // MODE_QKV_CACHE: [R][H][L][64]
    bf16_t* vcache;
    int H, L;                   // cache geometry
    const DecodeShared* shared; // MODE_QKV_CACHE reads shared->step
    const int* row_pos;         // slot mode (ymt3_transcribe_stream): per-row positions replace shared->step; else null
    unsigned long long* stamp;  // measurement (YMT3_STAMP=1): [grid][2] wall-clock entry / exit per workgroup; else null
    float* ssq;                 // [SSQ_TILES][ssq_stride] per-row partial sums of h^2 (read by NORM, written by RESID)
    int ssq_stride;
    // MODE_RESID only, or null: per-head O-projection partials [R][H][N] left by the self-attention kernel; the residual
    // operand becomes h + (p0 + p1 + ... + p7) -- the sum the separate O-projection launch would have stored in h
    const float* part;
    // NORM modes after an MoE FFN whose combine launch was folded away, or null: y [2R][K] gate-scaled expert outputs by pair;
    // the residual row is x_f32 + (y[2r] + y[2r+1]) and the workgroups of column tile 0 store it to h_out (QKV mode)
    const float* pend_y;
    float* h_out;
    int mid_rows;               // row count from which the mid-size tile kernel is taken; < 0: DEC_GEMM_MID_ROWS; 0: never (handle-level: YMT3_DEC_GEMM_MID_ROWS at create)
};
constexpr int DEC_GEMM_MID_ROWS = 512;
// The four skinny GEMMs between a layer's cross-attention and the next layer's self-attention as one launch (dec_chain.hip):
// cross O-projection -> FFN-in -> FFN-out -> next QKV projection (or lm_head); dense FFN, d_model = 512, d_ff = 2048, R <= 64.
struct ChainArgs {
    const bf16_t *w0, *w1, *w2, *w3;   // wo_c [512][512], wi [d_ff][512], wo2 [512][d_ff], next wqkv [3*512][512] or lm_head [V][512]
    const bf16_t* attn;         // [R][512] cross-attention output
    const float* part;          // folded self-attention O-projection partials [R][8][512], or null (DecGemmArgs::part)
    float* h;                   // [R][512] residual stream (read, += twice)
    float* ssq; int ssq_stride; // sum(h^2) partials, as DecGemmArgs
    const float *gain1, *gain3; // ln3 of this layer; ln1 of the next layer or ln_f
    bf16_t* dff;                // [R][d_ff] FFN hidden (scratch)
    int d_ff;
    int mode3, N3;              // DG_NORM_QKV_CACHE (N3 = 3*512) or DG_NORM_LOGITS (N3 = vocab)
    bf16_t* out_q; bf16_t* kcache; bf16_t* vcache;   // stage 3, QKV mode (next layer's cache slabs)
    float* logits;              // stage 3, lm_head mode
    int H, L;
    const DecodeShared* shared; const int* row_pos;
    int row0, R;
    float eps;
    unsigned* sync;             // CHAIN_SYNC_WORDS: arrival counters [3 boundaries][CHAIN_TILES_MAX row tiles][8 replicas], one 128-byte line each (zeroed by the
                                // preceding cross-attention launch), then the sticky abort word on a line of its own
    unsigned* host_abort;       // pinned host word, set with the abort (the host refuses further calls)
    unsigned long long* stamp;
    unsigned* sync_abort;       // dec_step.hip only: the sticky abort word (the chain launch finds it at sync + CHAIN_ABORT_WORD)
    int nsub;                   // measurement only: counters per boundary (1 / 2 / 4) in bits 0-3, 4-7, 8-11; 0 = the built-in choice
};
constexpr int CHAIN_LINE = 32;                                  // 32-bit words per 128-byte line
constexpr int CHAIN_TILES_MAX = 16;                             // row tiles of 16 rows a chain launch can hold (256 rows)
constexpr int CHAIN_COUNTERS = 3 * CHAIN_TILES_MAX * 8;
constexpr int CHAIN_ABORT_WORD = CHAIN_COUNTERS * CHAIN_LINE;
constexpr int CHAIN_SYNC_WORDS = (CHAIN_COUNTERS + 1) * CHAIN_LINE;
int init_chain_kernels();
bool dec_chain_fits(int n_cus);                 // occupancy x CUs covers the chain's grid (all its workgroups wait for each other)
bool dec_attention_pair_fits(int n_cus);        // the same for the attention pair at 64 rows
int launch_dec_chain(const ChainArgs& c, hipStream_t stream);       // 0 launched, < 0: not this kernel's shape
int launch_chain_poison(const unsigned* sync, int32_t* tokens, long long n, hipStream_t stream);   // tokens = INT32_MIN if the chain aborted

// One decode step's layers as ONE launch (dec_step.hip): per layer the attention pair and the GEMM chain, handed over inside the kernel.
struct StepLayer {
    const bf16_t *wo, *wq_c, *wo_c, *wi, *wo2, *w3;    // w3: the next layer's wqkv, or lm_head after the last layer
    const float *ln2, *ln3, *gain3;                      // gain3: the next layer's ln1, or ln_f
    const bf16_t *kself, *vself;                         // this layer's self-attention cache [R][H][L][64]
    const bf16_t *kcross, *vcross;                       // its cross-attention K/V [B][H][T][64]
    bf16_t *knext, *vnext;                               // the next layer's cache (the QKV stage appends to it)
    int N3, last;                                        // 3 * 512, or the vocabulary after the last layer
};
struct StepArgs {
    // (dec_step_kernel reads this struct straight from its kernel-argument segment, per layer: common fields first, layers last)
    int n_layers, R, T, L, ssq_stride;
    float eps;
    int tiles_free, pad_;                                // 1: a row tile waits only for itself (four independent pipelines); 0: the tiles move in step
    bf16_t* q; bf16_t* attn; float* opart; float* h; float* ssq; bf16_t* dff; float* logits;
    const float* bias;                                   // [H][L] self-attention bias by distance
    const DecodeShared* shared; const int* row_pos;
    unsigned* sync;                                      // [(n_layers + 1)][STEP_SYNC_LINES_PER_LAYER] counter lines, zero at entry (the argmax kernel zeroes them)
    unsigned* pair_rows;                                 // [R][2] self-resetting row counters (as the attention pair's)
    unsigned* abort_word; unsigned* host_abort;
    unsigned long long* stamp;                           // measurement (YMT3_STAMP=1) or null: [grid][2] entry / exit clocks, then 16 marks per workgroup from word 1024
    StepLayer layer[8];
};
// per layer: the chain's counter lines, then attn_done [4 row tiles][8 replicas], then qkv_done [4 row tiles][8 heads]
constexpr int STEP_SYNC_LINES_PER_LAYER = CHAIN_COUNTERS + 32 + 32;
constexpr int STEP_SYNC_LINES = 9 * STEP_SYNC_LINES_PER_LAYER;
int init_step_kernel();
bool dec_step_fits(int n_cus);
int launch_dec_step(const StepArgs& s, hipStream_t stream);

enum DecGemmMode { DG_NORM_QKV_CACHE = 0, DG_NORM_BF16 = 1, DG_NORM_BF16_RELU = 2, DG_NORM_LOGITS = 3, DG_RESID = 4 };
int init_decode_kernels();
int launch_dec_gemm(int mode, const DecGemmArgs& a, hipStream_t stream);

struct DecAttnArgs {
    const bf16_t* q;            // [R][H*64]
    const bf16_t* k;            // slab base; slab (kv_row, h) at ((kv_row*H + h) * slab_keys) * 64
    const bf16_t* v;
    bf16_t* out;                // [R][H*64]
    const float* bias;          // [H][L] by distance (self) or null (cross)
    const DecodeShared* shared; // self: n_keys = shared->step + 1
    const int* row_pos;         // slot mode: n_keys = row_pos[r] + 1; else null
    unsigned long long* stamp;  // measurement: as DecGemmArgs::stamp
    int n_keys_const;           // cross: fixed key count
    int slab_keys;              // keys allocated per (row, head) slab (L for self, T for cross)
    int rows_per_kv;            // 1 for self; n_channels for cross (row r reads segment r / n_channels)
    int row0, R, H, bias_stride;
    // fused query projection (cross-attention; wq == nullptr -> q is read from `q`): q = R(R(norm(x_r)*gain) . wq[head]^T)
    const bf16_t* wq;           // [H*64][512]
    const float* x_f32;         // [R][512] residual stream
    const float* gain;          // [512]
    const float* ssq; int ssq_stride;
    float eps;
    // O-projection folded into the self-attention kernel (wo != nullptr; self, 8 waves per (row, head)): the kernel ends with
    // its head's share of the projection, opart[r][h][0..512) = R(o_h) . wo[:, 64h..64h+64)^T, exactly the wave-h split-K partial
    // of the DG_RESID kernel; the fused cross-attention (ipart != nullptr) and the cross O-projection's residual read
    // (DecGemmArgs::part) sum the eight partials in wave order instead of reading an updated h
    const bf16_t* wo;           // [512][H*64]
    float* opart;               // [R][H][512]
    const float* ipart;         // [R][H][512]
    unsigned* chain_sync;       // fused cross-attention, or null: the arrival counters of the GEMM chain launched next (dec_chain.hip), zeroed here
    int force_many;             // test knob (YMT3_SELF_ATTN_2WAVE=1 at create): take the 2-waves-per-(row, head) form whatever the row count
};
int launch_dec_attention(bool self_attn, const DecAttnArgs& a, hipStream_t stream);
// one layer's self-attention (folded O-projection) and fused cross-attention as one launch (decode.hip: dec_attn_pair_kernel);
// pair_rows = [R][2] zeroed 128-byte counter lines the kernel leaves zeroed; 0 launched, < 0 not this kernel's shape
int launch_dec_attention_pair(const DecAttnArgs& self_args, const DecAttnArgs& cross_args, unsigned* pair_rows, unsigned* abort_word,
                              unsigned* host_abort, hipStream_t stream);

// multi-channel cross-attention with the query projection fused, one workgroup per (segment, head) (mc_cross_attn.hip)
struct McCrossArgs {
    const float* x_f32;         // [R][512] residual stream, row = row0 + seg*n_channels + channel
    const float* gain;          // [512]
    const float* ssq; int ssq_stride;
    const bf16_t* wq;           // [H*64][512]
    const bf16_t* k;            // [n_seg][H][T][64]
    const bf16_t* v;
    bf16_t* out;                // [R][H*64]
    int row0, n_seg, n_channels, H, T;
    float eps;
};
int init_mc_cross_kernels();
int launch_mc_cross_attention(const McCrossArgs& a, hipStream_t stream);

struct ArgmaxArgs {
    const float* logits;        // [R][V]
    float* h;                   // [R][d] residual stream to refill with the next embedding
    const bf16_t* embed;        // [V][d]
    const bf16_t* chan_embed;   // [K][d] or null
    DecodeShared* shared;
    int* finished;              // [R]
    float* ssq;                 // [SSQ_TILES][ssq_stride]
    int ssq_stride;
    int row0, R, V, d, n_channels, eos_id, pad_id;
    // slot mode (ymt3_transcribe_stream; both null otherwise): every row decodes at its own position row_pos[r] and
    // writes token p to tokens_out[row_out[r] + p]; a row stops (finished = 1, position frozen) after EOS or n_steps tokens
    int* row_pos;               // [R]
    const long long* row_out;   // [R]
    unsigned long long* stamp;  // measurement: as DecGemmArgs::stamp
    unsigned* ticket;           // or null: [rows / 32 + 1] sub-counters, one 128-byte line each (zero between launches): a two-level ticket for many rows
    unsigned* zero_sync;        // or null: counter lines (CHAIN_LINE words each) to leave zeroed for the next step's dec_step_kernel
    int zero_lines;
};
int launch_argmax_embed(const ArgmaxArgs& a, hipStream_t stream);
// tokens_out[r][from .. n_steps) = pad for rows [row0, row0 + R): the tail of a decode that stopped early
int launch_pad_tail(int32_t* tokens_out, int row0, int R, int n_steps, int from, int pad_id, hipStream_t stream);
// slot mode: (re)start rows [row0, row0 + n_channels) on a new segment: h = embed[pad] (+ channel), position 0,
// finished = 0, row_out = first_out + channel * n_steps
int launch_slot_start(const ArgmaxArgs& a, int row0, long long first_out, int n_steps, long long* row_out, hipStream_t stream);
// slot mode: PAD the unwritten tail [row_pos + 1, n_steps) of rows [row0, row0 + n_rows)
int launch_slot_retire(const ArgmaxArgs& a, int row0, int n_rows, int n_steps, int32_t* tokens_out, hipStream_t stream);
// all rows: h[r] = embed[pad] (+ chan_embed), finished = 0; a.shared[0..n_chains) reset
int launch_decode_init(const ArgmaxArgs& a, int n_chains, int n_steps, int step0, int32_t* tokens_out, const int32_t* forced,
                       float* logits_out, hipStream_t stream);

// ---------------------------------------------------------------- MoE decoder FFN (moe.hip)
struct MoeArgs {
    float* h;                   // [R][d_model] fp32 residual stream (read by router, updated by combine)
    const float* gain;          // [d_model]
    float* ssq; int ssq_stride; // carried sum(h^2) partials
    const bf16_t* router;       // [E][d_model]
    const bf16_t* wi;           // [E][d_ff][d_model]
    const bf16_t* wo;           // [E][d_model][d_ff]
    const uint8_t* wi_q8; const uint8_t* wo_q8;   // fp8 (OCP e4m3) forms, same shapes
    const float* wi_s; const float* wo_s;         // [E] dequantisation scales
    int fp8;
    bf16_t* xn;                 // [R][d_model] normed rows (bf16)
    int* sel; float* gate;      // [R][2] chosen experts and their gates; pair p = 2 * row + slot
    bf16_t* hidden;             // [2R][d_ff] by pair
    float* y;                   // [2R][d_model] gate-scaled expert outputs by pair
    int row0, R, E, top_k, d_model, d_ff;
    float eps;
    // debug hook ymt3_debug_moe_trace (null otherwise): the router also records its choices, [step][layer][row][2] int32, so a test can
    // teacher-force the ORACLE's routing with them and check every choice was a legitimate near-tie instead of excluding such steps
    int32_t* sel_trace; const DecodeShared* shared; int layer, n_layers, trace_rows, trace_steps;
};
int init_moe_kernels();
int launch_moe_stage(int stage, const MoeArgs& a, hipStream_t stream);

// The MoE layer's five skinny launches -- cross O-projection, router, expert FFN-in, expert FFN-out, the next QKV projection (or lm_head) with the
// combine folded in -- as ONE launch (moe_chain.hip); up to 64 rows, one channel, 8 experts, top-2; bf16 or fp8 expert weights.
struct MoeChainArgs {
    const bf16_t* wo_c; const void* wi; const void* wo; const bf16_t* w3;     // [512][512]; experts [8][2048][512] / [8][512][2048] (bf16 or e4m3); next wqkv / lm_head
    const bf16_t* attn; float* h; const float* part; float* ssq; int ssq_stride;   // as ChainArgs
    const float* gain_r; const bf16_t* router; bf16_t* xn; int* sel; float* gate;  // router: ln3, [8][512]; scratch [R][512], [2R], [2R]
    const float* wi_s; const float* wo_s;                                         // fp8: per-expert dequantisation scales
    bf16_t* hidden; float* y;                                                     // [2R][2048], [2R][512] by pair
    const float* gain3; int mode3, N3;                                            // stage 4: the next layer's ln1 or ln_f; DG_NORM_QKV_CACHE / DG_NORM_LOGITS
    float* h_out;                                                                 // QKV mode: the other residual buffer (column tile 0 stores h + y0 + y1 there)
    bf16_t* out_q; bf16_t* kcache; bf16_t* vcache; float* logits; int H, L;
    const DecodeShared* shared; const int* row_pos;
    int R, E, fp8;
    float eps;
    unsigned* sync; unsigned* host_abort;                                         // the chain's counter block (zeroed by the preceding cross-attention launch) + abort word
    unsigned long long* stamp;
    int32_t* sel_trace; int layer, n_layers, trace_rows, trace_steps;             // debug hook ymt3_debug_moe_trace, as MoeArgs
};
int init_moe_chain_kernels();
bool moe_chain_fits(int n_cus, bool fp8);
int launch_moe_chain(const MoeChainArgs& c, hipStream_t stream);       // 0 launched, < 0: not this kernel's shape
