// Standalone C host for the C ABI (no Python, no torch): loads a weight blob written by
// `python -m yourmt3_amd.export_blob`, transcribes synthetic segments, prints timing.
// Used for rocprofv3 --pmc passes (the profiler's counter mode crashes under the torch-bundled
// HSA runtime) and as the reference for a non-Python binding.
//   hipcc -O2 tools/ymt3_run.cpp -Iinclude -Lyourmt3_amd -lymt3_hip -Wl,-rpath,$PWD/yourmt3_amd -o tools/ymt3_run
//   tools/ymt3_run blob.bin [B=64] [L=1024] [passes=1] [step0=0] [config=1]      (config 2: the Perceiver-TF encoder of BASELINE configs[2])
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ymt3.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s blob.bin [B] [L] [passes]\n", argv[0]); return 2; }
    const int B = argc > 2 ? atoi(argv[2]) : 64, L = argc > 3 ? atoi(argv[3]) : 1024, passes = argc > 4 ? atoi(argv[4]) : 1, step0 = argc > 5 ? atoi(argv[5]) : 0, config = argc > 6 ? atoi(argv[6]) : 1;
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    fseek(f, 0, SEEK_END);
    const size_t n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<char> blob(n);
    if (fread(blob.data(), 1, n, f) != n) { fprintf(stderr, "short read\n"); return 1; }
    fclose(f);

    ymt3_config cfg{};                       // BASELINE configs[1] defaults (yourmt3_amd/config.py)
    cfg.sample_rate = 16000; cfg.segment_samples = 32767; cfg.n_fft = 2048; cfg.hop = 128; cfg.n_mels = 128;
    cfg.f_min = 50.f; cfg.f_max = 8000.f; cfg.log_floor = 1e-8f;
    cfg.d_model = 512; cfg.d_ff = 2048; cfg.n_heads = 8; cfg.d_kv = 64; cfg.n_enc_layers = 6; cfg.n_dec_layers = 6;
    cfg.vocab = 1536; cfg.rel_buckets = 32; cfg.rel_max_distance = 128; cfg.ln_eps = 1e-6f;
    cfg.max_decode_len = 1024; cfg.n_channels = 1; cfg.eos_id = -1; cfg.pad_id = 0;
    cfg.encoder_type = YMT3_ENC_T5; cfg.n_latents = 32; cfg.ptf_d = 128; cfg.ptf_blocks = 3; cfg.ptf_dff = 512; cfg.dec_ffn = YMT3_FFN_DENSE; cfg.n_experts = 8; cfg.moe_top_k = 2;
    if (config == 2) { cfg.encoder_type = YMT3_ENC_PERCEIVER_TF; cfg.n_enc_layers = 0; }
    cfg.max_batch = B;

    ymt3_handle h = nullptr;
    if (ymt3_create(&cfg, blob.data(), n, 0, &h) != YMT3_OK) { fprintf(stderr, "create: %s\n", ymt3_last_error()); return 1; }
    if (step0 && ymt3_debug_decode_start(h, step0) != YMT3_OK)   /* needs YMT3_DEBUG_HOOKS=1; applies to the first pass only */ { fprintf(stderr, "%s\n", ymt3_last_error()); return 1; }
    std::vector<float> audio((size_t)B * cfg.segment_samples);
    unsigned s = 12345u;
    for (size_t i = 0; i < audio.size(); ++i) {
        s = s * 1664525u + 1013904223u;
        audio[i] = 0.2f * ((s >> 8) / 8388608.0f - 1.0f) + 0.3f * sinf(0.1727876f * (float)(i % cfg.segment_samples));
    }
    float* a_dev; int32_t* t_dev;
    CK(hipMalloc((void**)&a_dev, audio.size() * 4));
    CK(hipMalloc((void**)&t_dev, (size_t)B * L * 4));
    CK(hipMemcpy(a_dev, audio.data(), audio.size() * 4, hipMemcpyHostToDevice));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    for (int p = 0; p < passes; ++p) {
        auto t0 = std::chrono::steady_clock::now();
        if (ymt3_transcribe_segments(h, a_dev, B, L, t_dev, st) != YMT3_OK) { fprintf(stderr, "run: %s\n", ymt3_last_error()); return 1; }
        CK(hipStreamSynchronize(st));
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("pass %d: %.1f ms, %.1f audio_s/wall_s\n", p, ms, B * 2.048 / (ms * 1e-3));
    }
    std::vector<int32_t> tok((size_t)B * L);
    CK(hipMemcpy(tok.data(), t_dev, tok.size() * 4, hipMemcpyDeviceToHost));
    long sum = 0;
    for (int32_t v : tok) sum += v;
    printf("token checksum %ld, first ids %d %d %d %d\n", sum, tok[0], tok[1], tok[2], tok[3]);
    ymt3_destroy(h);
    return 0;
}
