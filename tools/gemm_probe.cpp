// Timing-only A/B builds of the encoder GEMM (yourmt3_amd/csrc/gemm.hip is compiled INTO this tool, optionally with
// -DYMT3_PROBE_NO_STORE / -DYMT3_PROBE_NO_DMA / -DYMT3_PROBE_NO_MFMA to price one part of the kernel).
// (Measurement tool; not part of the product path.)
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../include/ymt3.h"
#include "../yourmt3_amd/csrc/gemm.hip"
void ymt3_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)

static int run(const char* name, int epi, int M, int N, int K, hipStream_t st) {
    bf16_t *A, *W; void* C;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 4));
    std::vector<bf16_t> h((size_t)M * K);
    unsigned x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (bf16_t)(0x3c00 + ((x >> 16) & 0x3ff) + ((x >> 31) << 15)); }   // random small values
    CK(hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice));
    h.resize((size_t)N * K);
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (bf16_t)(0x3c00 + ((x >> 16) & 0x3ff) + ((x >> 31) << 15)); }
    CK(hipMemcpy(W, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice));
    CK(hipMemset(C, 0, (size_t)M * N * 4));
    GemmArgs g{A, W, C, nullptr, M, N, K, K, K, N, 256, 8, M / 256 > 0 ? M / 256 : 1};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) if (launch_gemm(epi, g, st)) { fprintf(stderr, "launch rejected\n"); return 1; }
    CK(hipStreamSynchronize(st));
    const int reps = 20;
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) launch_gemm(epi, g, st);
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / reps, tf = 2.0 * M * N * K / us / 1e6;
    printf("%-28s M=%5d N=%4d K=%4d : %7.2f us  %6.0f TFLOP/s\n", name, M, N, K, us, tf);
    hipFree(A); hipFree(W); hipFree(C);
    return 0;
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    if (init_gemm_kernels()) { fprintf(stderr, "init failed\n"); return 1; }
    run("wi   (bf16 relu out)", EPI_BF16_RELU, 16384, 2048, 512, st);
    run("qkv  (bf16 out)", EPI_BF16, 16384, 1536, 512, st);
    run("o    (fp32 resid)", EPI_RESID, 16384, 512, 512, st);
    run("wo   (fp32 resid)", EPI_RESID, 16384, 512, 2048, st);
    run("ckv  (head-major bf16)", EPI_KV_HEADMAJOR, 16384, 6144, 512, st);
    run("big square", EPI_BF16, 8192, 8192, 8192, st);
    // decode-side shapes of the many-row configs (832 rows = 64 segments x 13 channels; 256 rows): what the 128 x 128 encoder
    // kernel makes of them, to size a mid-tile decode GEMM against the 16-row-tile kernel's 17-21 us
    run("dec wi   832 rows", EPI_BF16_RELU, 832, 2048, 512, st);
    run("dec qkv  832 rows", EPI_BF16, 832, 1536, 512, st);
    run("dec wo   832 rows", EPI_RESID, 832, 512, 2048, st);
    run("dec o    832 rows", EPI_RESID, 832, 512, 512, st);
    run("dec wi   256 rows", EPI_BF16_RELU, 256, 2048, 512, st);
    run("dec wo   256 rows", EPI_RESID, 256, 512, 2048, st);
    return 0;
}
