// Per-kernel time of the decode GEMM / attention kernels in isolation: N back-to-back launches of ONE kernel in a
// hipGraph on fixed operands (caches warm, same code).  Compared with their in-situ durations this separates
// "the kernel is slow" from "the kernel is slow where it runs" (cold weights / I-cache).  Measurement tool.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include "../yourmt3_amd/csrc/kernels.h"
void ymt3_set_error(const char*, ...) {}
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)

template <typename F>
static int timeit(const char* name, F launch, hipStream_t st, int n = 500) {
    hipGraph_t g; hipGraphExec_t e;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < n; ++i) launch(i);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&e, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(e, st)); CK(hipStreamSynchronize(st));
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < 4; ++r) CK(hipGraphLaunch(e, st));
    CK(hipStreamSynchronize(st));
    printf("%-44s %.2f us per launch\n", name, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (4.0 * n));
    hipGraphExecDestroy(e); hipGraphDestroy(g);
    return 0;
}

int main() {
    const int R = 64, d = 512, dff = 2048, H = 8, L = 1024, T = 256, NL = 6;
    float *h, *ssq, *logits; bf16_t *w, *abf, *q, *kc, *vc, *ck; DecodeShared* sh;
    const size_t wbytes = (size_t)NL * (3 * d * d + dff * d) * 2;       // several layers' worth so "rotating" weights are cold-ish
    CK(hipMalloc((void**)&h, R * d * 4)); CK(hipMalloc((void**)&ssq, 32 * R * 4)); CK(hipMalloc((void**)&logits, R * 2048 * 4));
    CK(hipMalloc((void**)&w, wbytes)); CK(hipMalloc((void**)&abf, R * dff * 2)); CK(hipMalloc((void**)&q, R * d * 2));
    const size_t cache = (size_t)R * H * L * 64 * 2;
    CK(hipMalloc((void**)&kc, cache * NL)); CK(hipMalloc((void**)&vc, cache * NL)); CK(hipMalloc((void**)&ck, (size_t)2 * R * H * T * 64 * 2));
    CK(hipMalloc((void**)&sh, sizeof(DecodeShared)));
    CK(hipMemset(h, 0, R * d * 4)); CK(hipMemset(ssq, 0, 32 * R * 4)); CK(hipMemset(w, 0, wbytes)); CK(hipMemset(abf, 0, R * dff * 2));
    CK(hipMemset(q, 0, R * d * 2)); CK(hipMemset(kc, 0, cache * NL)); CK(hipMemset(vc, 0, cache * NL)); CK(hipMemset(ck, 0, (size_t)2 * R * H * T * 64 * 2));
    DecodeShared hs{}; hs.step = 511; hs.n_steps = 1024;
    CK(hipMemcpy(sh, &hs, sizeof(hs), hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));

    DecGemmArgs a{}; a.row0 = 0; a.R = R; a.eps = 1e-6f; a.H = H; a.L = L; a.shared = sh; a.ssq = ssq; a.ssq_stride = R;
    a.x_f32 = h; a.gain = h; a.a_bf16 = abf; a.out_bf16 = q; a.out_f32 = h; a.kcache = kc; a.vcache = vc;
    for (int rot = 0; rot < 2; ++rot) {
        const char* tag = rot ? " (weights rotate over 6 layers)" : " (same weights)";
        char nm[96];
        auto W = [&](int i, size_t per) { return w + (rot ? (size_t)(i % NL) * (wbytes / 2 / NL) : 0); };
        snprintf(nm, 96, "RESID K=512 N=512%s", tag);
        timeit(nm, [&](int i) { a.W = W(i, 0); a.N = d; a.K = d; launch_dec_gemm(DG_RESID, a, st); }, st);
        snprintf(nm, 96, "RESID K=2048 N=512%s", tag);
        timeit(nm, [&](int i) { a.W = W(i, 0); a.N = d; a.K = dff; launch_dec_gemm(DG_RESID, a, st); }, st);
        snprintf(nm, 96, "NORM_BF16 N=512%s", tag);
        timeit(nm, [&](int i) { a.W = W(i, 0); a.N = d; a.K = d; launch_dec_gemm(DG_NORM_BF16, a, st); }, st);
        snprintf(nm, 96, "NORM_QKV N=1536%s", tag);
        timeit(nm, [&](int i) { a.W = W(i, 0); a.N = 3 * d; a.K = d; launch_dec_gemm(DG_NORM_QKV_CACHE, a, st); }, st);
        snprintf(nm, 96, "NORM_RELU N=2048%s", tag);
        a.out_bf16 = abf;
        timeit(nm, [&](int i) { a.W = W(i, 0); a.N = dff; a.K = d; launch_dec_gemm(DG_NORM_BF16_RELU, a, st); }, st);
        a.out_bf16 = q;
    }
    DecAttnArgs t{}; t.q = q; t.out = abf; t.bias = h; t.shared = sh; t.row0 = 0; t.R = R; t.H = H; t.bias_stride = 0;
    for (int step : {0, 255, 511, 1023}) {
        hs.step = step; CK(hipMemcpy(sh, &hs, sizeof(hs), hipMemcpyHostToDevice));
        char nm[96]; snprintf(nm, 96, "self-attn t=%d (6 rotating layer caches)", step);
        t.slab_keys = L; t.rows_per_kv = 1; t.n_keys_const = 0;
        timeit(nm, [&](int i) { t.k = kc + (size_t)(i % NL) * (cache / 2); t.v = vc + (size_t)(i % NL) * (cache / 2); launch_dec_attention(true, t, st); }, st, 120);
    }
    for (int step : {255, 511, 1023}) {
        hs.step = step; CK(hipMemcpy(sh, &hs, sizeof(hs), hipMemcpyHostToDevice));
        char nm[96]; snprintf(nm, 96, "self-attn t=%d (SAME layer cache every launch)", step);
        t.slab_keys = L; t.rows_per_kv = 1; t.n_keys_const = 0;
        timeit(nm, [&](int) { t.k = kc; t.v = vc; launch_dec_attention(true, t, st); }, st, 120);
    }
    t.k = ck; t.v = ck + (size_t)R * H * T * 64; t.slab_keys = T; t.n_keys_const = T; t.bias = nullptr;
    timeit("cross-attn T=256 (same slabs)", [&](int) { launch_dec_attention(false, t, st); }, st, 200);
    // interaction between consecutive kernels: pairs and the whole 8-kernel layer sequence, rotating over 6 layers
    hs.step = 511; CK(hipMemcpy(sh, &hs, sizeof(hs), hipMemcpyHostToDevice));
    DecAttnArgs ts = t; ts.slab_keys = L; ts.rows_per_kv = 1; ts.n_keys_const = 0; ts.bias = h;
    DecAttnArgs tc = t;
    auto Wl = [&](int i, size_t off) { return w + (size_t)(i % NL) * (wbytes / 2 / NL) + off; };
    auto self_attn = [&](int i) { ts.k = kc + (size_t)(i % NL) * (cache / 2); ts.v = vc + (size_t)(i % NL) * (cache / 2); launch_dec_attention(true, ts, st); };
    auto resid512 = [&](int i, size_t off) { a.W = Wl(i, off); a.N = d; a.K = d; a.out_bf16 = q; launch_dec_gemm(DG_RESID, a, st); };
    auto norm512 = [&](int i, size_t off) { a.W = Wl(i, off); a.N = d; a.K = d; a.out_bf16 = q; launch_dec_gemm(DG_NORM_BF16, a, st); };
    auto qkv = [&](int i) { a.W = Wl(i, 0); a.N = 3 * d; a.K = d; a.out_bf16 = q; launch_dec_gemm(DG_NORM_QKV_CACHE, a, st); };
    auto wi = [&](int i) { a.W = Wl(i, (size_t)d * d * 5); a.N = dff; a.K = d; a.out_bf16 = abf; launch_dec_gemm(DG_NORM_BF16_RELU, a, st); a.out_bf16 = q; };
    auto wo2 = [&](int i) { a.W = Wl(i, (size_t)d * d * 5); a.N = d; a.K = dff; launch_dec_gemm(DG_RESID, a, st); };
    timeit("pair: self-attn(t=511) + RESID512", [&](int i) { self_attn(i); resid512(i, (size_t)d * d * 3); }, st, 120);
    timeit("pair: RESID512 + NORM512", [&](int i) { resid512(i, (size_t)d * d * 3); norm512(i, (size_t)d * d * 4); }, st, 240);
    timeit("pair: cross-attn + RESID512", [&](int i) { launch_dec_attention(false, tc, st); resid512(i, (size_t)d * d * 3); }, st, 240);
    timeit("layer: qkv,self,o,crossq,cross,oc,wi,wo2 (t=511)", [&](int i) {
        qkv(i); self_attn(i); resid512(i, (size_t)d * d * 3); norm512(i, (size_t)d * d * 4); launch_dec_attention(false, tc, st);
        resid512(i, (size_t)d * d * 3); wi(i); wo2(i); }, st, 60);
    timeit("layer without the two attention kernels", [&](int i) {
        qkv(i); resid512(i, (size_t)d * d * 3); norm512(i, (size_t)d * d * 4); resid512(i, (size_t)d * d * 3); wi(i); wo2(i); }, st, 60);
    return 0;
}
