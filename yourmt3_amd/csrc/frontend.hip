// a1 + a2 of SURVEY.md section 8: frame + Hann window + real FFT + |.|^2 + HTK-mel + log, fused.
//
// One workgroup per frame.  The n_fft real samples are packed as N2 = n_fft/2 complex points,
// transformed by a radix-4 Stockham FFT held entirely in LDS (N2 = 4^S, one butterfly per thread
// per stage), untangled to the n_fft/2+1 real-FFT bins, squared, reduced through the sparse
// triangular mel filterbank (each bin feeds <= 2 filters; stored CSR per filter) and logged.
// HBM sees each audio sample ~once (the 16x frame overlap hits L2) and one fp32 write per mel
// value; arithmetic is fp32 throughout (tolerance vs. the oracle is stated in tests/test_gpu_frontend.py).
//
// Oracle: oracle/ymt3_oracle.py::logmel (TP: torch/functional.py:508-690 for the STFT half).
#include "common.h"
#include "kernels.h"

template <int S>
__global__ __launch_bounds__((1 << (2 * S)) / 4) void logmel_kernel(
    const float* __restrict__ audio, float* __restrict__ mel_out, const float* __restrict__ window,
    const float2* __restrict__ tw, const float2* __restrict__ untw, const int* __restrict__ mel_start,
    const int* __restrict__ mel_len, const int* __restrict__ mel_off, const float* __restrict__ mel_w,
    int n_samples, int n_frames, int hop, int n_mels, float log_floor) {
    constexpr int N2 = 1 << (2 * S);
    constexpr int NT = N2 / 4;
    constexpr int NFFT = 2 * N2;
    __shared__ float2 buf0[N2];
    __shared__ float2 buf1[N2 + 1];
    __shared__ float2 stw[N2];

    const int tid = threadIdx.x;
    const int frame = blockIdx.x % n_frames;
    const int b = blockIdx.x / n_frames;
    const float* x = audio + (size_t)b * n_samples;

#pragma unroll
    for (int q = 0; q < 4; ++q) stw[tid + q * NT] = tw[tid + q * NT];

    // windowed frame, reflect padded by NFFT/2 (torch.stft center=True, pad_mode="reflect")
    const int base = frame * hop - NFFT / 2;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = tid + q * NT;
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int n = 2 * j + e;
            int src = base + n;
            if (src < 0) src = -src;
            if (src >= n_samples) src = 2 * (n_samples - 1) - src;
            v[e] = x[src] * window[n];
        }
        buf0[j] = make_float2(v[0], v[1]);
    }
    __syncthreads();

    float2* in = buf0;
    float2* out = buf1;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int Ns = 1 << (2 * s);
        const int k = tid & (Ns - 1);
        float2 v0 = in[tid], v1 = in[tid + NT], v2 = in[tid + 2 * NT], v3 = in[tid + 3 * NT];
        if (s > 0) {
            const int m = k * (N2 / (4 * Ns));
            const float2 w1 = stw[m], w2 = stw[2 * m], w3 = stw[3 * m];
            v1 = make_float2(v1.x * w1.x - v1.y * w1.y, v1.x * w1.y + v1.y * w1.x);
            v2 = make_float2(v2.x * w2.x - v2.y * w2.y, v2.x * w2.y + v2.y * w2.x);
            v3 = make_float2(v3.x * w3.x - v3.y * w3.y, v3.x * w3.y + v3.y * w3.x);
        }
        const float2 a0 = make_float2(v0.x + v2.x, v0.y + v2.y);
        const float2 a1 = make_float2(v0.x - v2.x, v0.y - v2.y);
        const float2 a2 = make_float2(v1.x + v3.x, v1.y + v3.y);
        const float2 d = make_float2(v1.x - v3.x, v1.y - v3.y);
        const float2 a3 = make_float2(d.y, -d.x);                 // -i * (v1 - v3)
        const int j0 = ((tid - k) << 2) + k;
        out[j0] = make_float2(a0.x + a2.x, a0.y + a2.y);
        out[j0 + Ns] = make_float2(a1.x + a3.x, a1.y + a3.y);
        out[j0 + 2 * Ns] = make_float2(a0.x - a2.x, a0.y - a2.y);
        out[j0 + 3 * Ns] = make_float2(a1.x - a3.x, a1.y - a3.y);
        __syncthreads();
        float2* t = in; in = out; out = t;
    }

    // untangle the packed transform: X[k] = Xe + exp(-2 pi i k / NFFT) * Xo, k = 0..N2
    float* pw = reinterpret_cast<float*>(out);                    // N2 + 1 floats
    for (int k = tid; k <= N2; k += NT) {
        const float2 zk = in[k & (N2 - 1)];
        const float2 zr = in[(N2 - k) & (N2 - 1)];
        const float2 xe = make_float2(0.5f * (zk.x + zr.x), 0.5f * (zk.y - zr.y));
        const float2 dd = make_float2(zk.x - zr.x, zk.y + zr.y);  // Zk - conj(Zr)
        const float2 xo = make_float2(0.5f * dd.y, -0.5f * dd.x); // -i/2 * dd
        const float2 w = untw[k];
        const float re = xe.x + w.x * xo.x - w.y * xo.y;
        const float im = xe.y + w.x * xo.y + w.y * xo.x;
        pw[k] = re * re + im * im;
    }
    __syncthreads();

    // sparse mel: two threads per filter, alternate bins, combine with one shuffle
    float* dst = mel_out + ((size_t)b * n_frames + frame) * n_mels;
    for (int m = tid >> 1; m < n_mels; m += NT / 2) {
        const int st = mel_start[m], ln = mel_len[m], off = mel_off[m];
        float acc = 0.f;
        for (int i = tid & 1; i < ln; i += 2) acc += mel_w[off + i] * pw[st + i];
        acc += __shfl_xor(acc, 1, 64);
        if ((tid & 1) == 0) dst[m] = logf(fmaxf(acc, log_floor));
    }
}

int launch_logmel(const FrontendTables& t, const float* audio, float* mel, int B, hipStream_t stream) {
    const int grid = B * t.n_frames;
    if (grid == 0) return 0;
    if (t.n_fft == 2048) {
        logmel_kernel<5><<<grid, 256, 0, stream>>>(audio, mel, t.window, t.tw, t.untw, t.mel_start, t.mel_len,
                                                  t.mel_off, t.mel_w, t.n_samples, t.n_frames, t.hop, t.n_mels, t.log_floor);
    } else if (t.n_fft == 512) {
        logmel_kernel<4><<<grid, 64, 0, stream>>>(audio, mel, t.window, t.tw, t.untw, t.mel_start, t.mel_len,
                                                 t.mel_off, t.mel_w, t.n_samples, t.n_frames, t.hop, t.n_mels, t.log_floor);
    } else {
        return -1;
    }
    return 0;
}
