/*
 * ymt3.h -- C ABI of the MI355X (gfx950) audio -> MIDI-token hot path.
 *
 * Drop-in boundary.  The reference checkout holds no source
 * (/root/reference = README.md:1-13 + LICENSE), so no reference file:line can be cited for
 * the interface each entry point replaces; the names `transcribe()` / `TaskManager` /
 * `inference_file()` come from BASELINE.json `north_star` and SURVEY.md section 9 (unverified
 * recollection of upstream `model/ymt3.py::inference_file(bsz, audio_segments)`).  The entry
 * points below are what a ctypes / cffi / pybind stub on that Python path would bind:
 *
 *   inference_file(bsz, audio_segments)  ->  ymt3_transcribe_segments()
 *   spectrogram module forward           ->  ymt3_logmel()
 *   encoder forward                      ->  ymt3_encode()
 *   decoder generate (greedy, KV cache)  ->  ymt3_decode_greedy()
 *
 * Conventions
 *   - every pointer named *_dev is DEVICE memory on the handle's GPU, owned by the caller;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); all calls are
 *     asynchronous with respect to the host and never synchronise the device.  Exceptions, all documented at their
 *     declarations: ymt3_transcribe_stream, ymt3_set_early_stop (opt-in), the ymt3_profile_decode / ymt3_debug_* measurement
 *     hooks, and -- on by default, ymt3_set_abort_recovery(h, 0) turns it off -- the one wait at the END of a decode call
 *     that ran the merged decode kernels of the 64-row regime (after its last step has been queued; nothing is
 *     launched behind it unless a kernel gave up);
 *   - the library owns weights, KV caches and scratch inside the handle; nothing is
 *     allocated after ymt3_create();
 *   - return value: 0 = ok, non-zero = error; the message is in ymt3_last_error()
 *     (thread local).  No exception ever crosses this boundary;
 *   - one handle per device per host thread.  No internal host threads.
 */
#ifndef YMT3_H
#define YMT3_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YMT3_ABI_VERSION 3

enum { YMT3_OK = 0, YMT3_ERR_ARG = 1, YMT3_ERR_BLOB = 2, YMT3_ERR_HIP = 3, YMT3_ERR_UNSUPPORTED = 4 };
enum { YMT3_ENC_T5 = 0, YMT3_ENC_PERCEIVER_TF = 1 };
enum { YMT3_FFN_DENSE = 0, YMT3_FFN_MOE = 1 };

/* Field order mirrors yourmt3_amd/config.py::CConfig. */
typedef struct ymt3_config {
    int32_t sample_rate, segment_samples, n_fft, hop, n_mels;
    float   f_min, f_max, log_floor;
    int32_t d_model, d_ff, n_heads, d_kv, n_enc_layers, n_dec_layers;
    int32_t vocab, rel_buckets, rel_max_distance;
    float   ln_eps;
    int32_t max_decode_len, n_channels, eos_id, pad_id;
    int32_t encoder_type, n_latents;
    int32_t dec_ffn, n_experts, moe_top_k, moe_fp8;
    int32_t ptf_d, ptf_blocks, ptf_dff;   /* Perceiver-TF encoder: token width (multiple of 64), blocks, FFN width; n_latents = latents per frame */
    int32_t max_batch;              /* segments per call the workspace is sized for */
} ymt3_config;

typedef struct ymt3_ctx* ymt3_handle;

int         ymt3_abi_version(void);
const char* ymt3_last_error(void);

/* Parse the weight blob (format: yourmt3_amd/weights.py), upload it to `device`, derive the
 * window / twiddle / mel / relative-position tables, allocate caches and scratch. */
int  ymt3_create(const ymt3_config* cfg, const void* blob_host, size_t blob_bytes, int device, ymt3_handle* out);
void ymt3_destroy(ymt3_handle h);

/* Bytes of device memory held by the handle (weights + caches + scratch). */
size_t ymt3_device_bytes(ymt3_handle h);

/* The step before the path: interleaved PCM on the device, (n_frames, n_channels)
 * int16 or f32 at any integer sample rate -> mono mix -> polyphase Kaiser-windowed-sinc resample to cfg.sample_rate
 * (the filter and alignment of scipy.signal.resample_poly) -> (n_segments, segment_samples) f32, zero padded: the
 * buffer ymt3_logmel / ymt3_transcribe_segments read.  ymt3_ingest_plan is host arithmetic only (how many samples
 * and segments n_frames become).  The first ymt3_ingest call for a new rate pair designs the filter and uploads
 * <= 2 MB synchronously (rate pairs whose reduced ratio exceeds 16384 are rejected); later calls allocate nothing and are asynchronous on `stream`. */
#define YMT3_PCM_S16 0
#define YMT3_PCM_F32 1
int ymt3_ingest_plan(ymt3_handle h, int64_t n_frames, int sample_rate_in, int64_t* n_samples_out, int* n_segments);
int ymt3_ingest(ymt3_handle h, const void* pcm_dev, int pcm_format, int64_t n_frames, int n_channels,
                int sample_rate_in, float* segments_dev, int n_segments, void* stream);

/* a1+a2: audio (B, segment_samples) f32 -> log-mel (B, n_frames, n_mels) f32. */
int ymt3_logmel(ymt3_handle h, const float* audio_dev, int B, float* mel_dev, void* stream);

/* a3+a4 (or a9): log-mel (B, n_frames, n_mels) f32 -> encoder output (B, n_frames, d_model) bf16. */
int ymt3_encode(ymt3_handle h, const float* mel_dev, int B, void* enc_dev, void* stream);

/* a6+a7+a8(+a10): encoder output bf16 -> greedy token ids (B, n_channels, n_steps) int32.
 * forced_dev (may be NULL): (B, n_channels, n_steps) int32 teacher-forcing ids fed back instead of
 * the argmax.  logits_dev (may be NULL): (B, n_channels, n_steps, vocab) f32 per-step logits. */
int ymt3_decode_greedy(ymt3_handle h, const void* enc_dev, int B, int n_steps, int32_t* tokens_dev,
                       const int32_t* forced_dev, float* logits_dev, void* stream);

/* Opt-in early stop (first step): with eos_id >= 0 and interval > 0, ymt3_decode_greedy /
 * ymt3_transcribe_segments check on the host every `interval` steps whether every row has emitted EOS and stop
 * launching once all have (the remainder of each row is PAD, exactly what the full-length run produces).  In this mode the
 * call synchronises the stream once per interval.  interval = 0 (default) restores the loop that queues every step at once. */
int ymt3_set_early_stop(ymt3_handle h, int interval);

/* The whole hot path: audio (B, segment_samples) f32 -> token ids (B, n_channels, n_steps) int32. */
int ymt3_transcribe_segments(ymt3_handle h, const float* audio_dev, int B, int n_steps,
                             int32_t* tokens_dev, void* stream);

/* Continuous batching: transcribe a queue of `n_segments` segments, audio (n_segments,
 * segment_samples) f32 -> ids (n_segments, n_channels, n_steps) int32, through `slots` decoder slots (<= 0 or
 * > max_batch: max_batch).  Every row decodes at its own position; every `interval` steps (0: 8) the host reads the
 * per-row stop flags, retires segments whose rows have all emitted EOS (or n_steps tokens; the tail is PAD, exactly
 * what the lock-step calls produce) and encodes the next pending segments into the freed slots.  The ids are bit-identical
 * to ymt3_transcribe_segments on the same segments.  Synchronises the stream (once per interval) and returns when the
 * queue is done. */
int ymt3_transcribe_stream(ymt3_handle h, const float* audio_dev, int n_segments, int n_steps, int32_t* tokens_dev,
                           int slots, int interval, void* stream);

/* Number of decoder steps the last decode call on this handle actually launched (ymt3_decode_greedy,
 * ymt3_transcribe_segments: n_steps unless ymt3_set_early_stop cut it short; ymt3_transcribe_stream: every step of every round). */
int ymt3_last_decode_steps(ymt3_handle h);

/* The merged decode kernels (up to 64 rows, one channel: a layer's two attentions as one launch, its four skinny GEMMs as one launch)
 * hand data between workgroups inside a launch and therefore need every workgroup of their grid resident at once; the handle enables
 * them only where the occupancy query says they fit.  If something else holds CUs while one runs (another process or handle on the
 * same GPU, a CU mask), a stage can wait in vain: every wait is bounded (1 s), then the kernel raises a sticky abort word and the
 * launch drains.  What follows is governed by `mode`:
 *   1 (default): ymt3_decode_greedy / ymt3_transcribe_segments / ymt3_transcribe_stream wait for their own work at the end of the call
 *      and look at the word.  Raised: the call is run AGAIN through the separate launches (fresh launches in the same process;
 *      the same arithmetic, bit-identical ids) and the handle stays on them.  The caller sees correct ids, later.
 *   0: calls stay fully asynchronous.  An aborted call's ids are all INT32_MIN (never plausible ids); the NEXT call on the handle
 *      notices, waits for the device, and switches to the separate launches before doing its own work.
 * ymt3_merged_fallbacks: how many times this handle has left the merged kernels that way (0 or 1; it never goes back).
 * (YMT3_ABORT_RECOVERY=0 in the environment at ymt3_create selects mode 0.) */
int ymt3_set_abort_recovery(ymt3_handle h, int mode);
int ymt3_merged_fallbacks(ymt3_handle h);

/* How many concurrent row ranges ("chains", each on a stream of the handle's own, joined into the caller's stream before the call returns
 * control of it) the last ymt3_decode_greedy / ymt3_transcribe_segments call decoded its batch as.  1 except for 168-256 rows of one channel
 * with the dense FFN, where two halves overlap (the attention kernels are bandwidth-bound there, the GEMMs between them latency-bound); the ids do
 * not depend on it.  YMT3_CHAINS=n in the environment at ymt3_create fixes the number (1..8). */
int ymt3_last_decode_chains(ymt3_handle h);

/* Measurement hook (bench.py `roofline`): decode eagerly (no graph) and bracket every kernel launch of
 * every `stride`-th step (positions stride/2, 3*stride/2, ...) with HIP events on `stream`; synchronises the stream before returning.
 * Classes: 0 qkv+cache GEMM, 1 self-attention, 2 self O-proj, 3 cross Q GEMM, 4 cross-attention,
 * 5 cross O-proj, 6 FFN wi, 7 FFN wo, 8 lm_head, 9 argmax+embed, 10 spans: ONE bracket around the stride-1
 * un-bracketed steps after each sampled step (true step time, used to calibrate out the stream time an
 * event pair itself costs).  Outputs are HOST arrays. */
#define YMT3_PROFILE_CLASSES 16
int ymt3_profile_decode(ymt3_handle h, const void* enc_dev, int B, int n_steps, int stride, int32_t* tokens_dev,
                        float* ms_by_class, int32_t* launches_by_class, void* stream);

/* Measurement hook: when the handle was created with YMT3_STAMP=1 in the environment, every decode-step kernel records a
 * 100 MHz wall-clock value per workgroup at entry and at the end of its last wave (last step executed wins).  This call
 * synchronises the device and reduces them: for kernel i of the step (launch order; cls[i] = class as in
 * ymt3_profile_decode, grid[i] = workgroups) stats[4i..4i+3] = earliest entry, latest entry, earliest exit, latest exit.
 * Arrays hold 64 kernels. */
int ymt3_debug_step_stamps(ymt3_handle h, int32_t* cls, int32_t* grid, uint64_t* stats, int* n_kernels);
/* The raw (entry, exit) pairs of kernel `kernel` of the step, one per workgroup in blockIdx order (capacity in workgroups). */
int ymt3_debug_kernel_stamps(ymt3_handle h, int kernel, uint64_t* stamps, int capacity_wgs);

/* Debug hook, NOT part of the product surface: refused (YMT3_ERR_UNSUPPORTED) unless the handle was created with
 * YMT3_DEBUG_HOOKS=1 in the environment.  The NEXT ymt3_decode_greedy / ymt3_transcribe_segments call -- that one call
 * only -- starts at cache position `step0` instead of 0; cache positions [0, step0) are zero-filled here, so the call
 * reads defined memory, but its TOKENS are meaningless.  The memory traffic of positions step0.. is exactly that of a
 * real decode, which lets a counter pass (rocprofv3 --pmc) cover late positions with a short process. */
int ymt3_debug_decode_start(ymt3_handle h, int step0);

/* Debug hook, gated like the one above: marks the handle as if one of its merged decode kernels (the attention pair / GEMM chain of the
 * 64-row regime, whose stages wait for each other inside one launch) had given up waiting during the NEXT decode call.  What must
 * follow -- and what the test of this hook checks -- is what a real abort triggers (ymt3_set_abort_recovery): in mode 1 that call
 * returns the correct ids through the separate launches and ymt3_merged_fallbacks reports 1; in mode 0 its ids are all INT32_MIN and
 * the call after it runs, correctly, on the separate launches.  YMT3_ERR_UNSUPPORTED if the handle does not run those kernels (fewer
 * than 256 CUs, both switched off, or already fallen back). */
int ymt3_debug_force_stage_abort(ymt3_handle h);

/* Debug hook, gated like the ones above (MoE decoder FFN only): from now on every lock-step decode call records the router's choices,
 * trace_dev[step][layer][row][2] int32 (the two chosen experts, best first; steps >= n_steps or rows >= n_rows are not recorded); NULL
 * stops recording.  A routing choice is discrete: at a near-tie of the 2nd / 3rd router logit the HIP path may legitimately pick another
 * expert than the CPU oracle.  With the trace a test feeds the HIP path's choices to the oracle, checks each was within the numerical
 * noise of the oracle's own top two, and compares logits / ids at EVERY step instead of excluding the near-tie steps. */
int ymt3_debug_moe_trace(ymt3_handle h, int32_t* trace_dev, int n_steps, int n_rows);

/* Unit-test hooks: C = A(bf16 MxK) * W^T(bf16 NxK), f32 out; runs the encoder GEMM kernel. */
int ymt3_test_gemm(ymt3_handle h, const void* a_dev, const void* w_dev, float* c_dev, int M, int N, int K, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* YMT3_H */
