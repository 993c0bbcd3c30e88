// a1 + a2 of SURVEY.md section 8: frame + Hann window + real FFT + |.|^2 + HTK-mel + log, fused.
//
// One workgroup per 8 consecutive frames.  Their samples (n_fft + 7 hops: the 15/16 overlap between neighbouring
// frames) are staged in LDS once, and the per-thread constants -- window taps, the twiddles of every FFT stage, the
// untangle factors -- are loaded into registers once per workgroup (one frame per workgroup spent most of its time
// re-reading those 24 KB of tables from L2).  Per frame the n_fft real samples are packed as N2 = n_fft/2 complex
// points, transformed by a radix-4 Stockham FFT held entirely in LDS (N2 = 4^S, one butterfly per thread per stage),
// untangled to the n_fft/2+1 real-FFT bins, squared, reduced through the sparse triangular mel filterbank (each bin
// feeds <= 2 filters; stored CSR per filter) and logged.  HBM sees each audio sample ~once and one fp32 write per mel
// value; arithmetic is fp32 throughout (tolerance vs. the oracle: tests/test_gpu_parity.py).
//
// Oracle: oracle/ymt3_oracle.py::logmel (TP: torch/functional.py:508-690 for the STFT half).
#include "common.h"
#include "kernels.h"

constexpr int FPW = 8;          // consecutive frames per workgroup: tables and the 15/16 sample overlap are loaded once
constexpr int MAX_HOP = 256;
constexpr int MAX_MEL_W = 2304;  // >= 2 * (n_fft/2 + 1) triangle weights at n_fft = 2048

template <int S>
__global__ __launch_bounds__((1 << (2 * S)) / 4) void logmel_kernel(
    const float* __restrict__ audio, float* __restrict__ mel_out, const float* __restrict__ window,
    const float2* __restrict__ tw, const float2* __restrict__ untw, const int* __restrict__ mel_start,
    const int* __restrict__ mel_len, const int* __restrict__ mel_off, const float* __restrict__ mel_w,
    int n_samples, int n_frames, int hop, int n_mels, int n_mel_w, float log_floor) {
    constexpr int N2 = 1 << (2 * S);
    constexpr int NT = N2 / 4;
    constexpr int NFFT = 2 * N2;
    __shared__ __attribute__((aligned(16))) float samp[NFFT + (FPW - 1) * MAX_HOP];
    __shared__ float2 buf0[N2];
    __shared__ float2 buf1[N2 + 1];
    __shared__ float smw[MAX_MEL_W];                               // the CSR filterbank weights, once per workgroup

    const int tid = threadIdx.x;
    const int groups = (n_frames + FPW - 1) / FPW;
    const int f0 = (blockIdx.x % groups) * FPW;
    const int b = blockIdx.x / groups;
    const int nf = min(FPW, n_frames - f0);
    const float* x = audio + (size_t)b * n_samples;

    // per-thread constants, loaded once per workgroup: window taps, stage twiddles, untangle factors
    float win[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float2 w = reinterpret_cast<const float2*>(window)[tid + q * NT];
        win[q][0] = w.x; win[q][1] = w.y;
    }
    float2 tws[S > 1 ? S - 1 : 1][3];
#pragma unroll
    for (int s = 1; s < S; ++s) {
        const int Ns = 1 << (2 * s);
        const int m = (tid & (Ns - 1)) * (N2 / (4 * Ns));
        tws[s - 1][0] = tw[m]; tws[s - 1][1] = tw[2 * m]; tws[s - 1][2] = tw[3 * m];
    }
    float2 unt[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) unt[q] = untw[tid + q * NT];
    for (int i = tid; i < n_mel_w; i += NT) smw[i] = mel_w[i];
    // this thread's filter (two threads per filter; filters beyond the first NT/2 are handled in further rounds)
    const int my_m = tid >> 1;
    const int my_st = my_m < n_mels ? mel_start[my_m] : 0, my_ln = my_m < n_mels ? mel_len[my_m] : 0, my_off = my_m < n_mels ? mel_off[my_m] : 0;

    // samples of all nf frames, reflect padded by NFFT/2 (torch.stft center=True, pad_mode="reflect")
    const int base = f0 * hop - NFFT / 2;
    const int total = NFFT + (nf - 1) * hop;
    for (int i = tid; i < total; i += NT) {
        int src = base + i;
        if (src < 0) src = -src;
        if (src >= n_samples) src = 2 * (n_samples - 1) - src;
        samp[i] = x[src];
    }
    __syncthreads();

    for (int f = 0; f < nf; ++f) {
        const float* fs = samp + f * hop;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = tid + q * NT;
            const float2 v = *reinterpret_cast<const float2*>(fs + 2 * j);
            buf0[j] = make_float2(v.x * win[q][0], v.y * win[q][1]);
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
                const float2 w1 = tws[s - 1][0], w2 = tws[s - 1][1], w3 = tws[s - 1][2];
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
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = tid + q * NT;
            const float2 zk = in[k];
            const float2 zr = in[(N2 - k) & (N2 - 1)];
            const float2 xe = make_float2(0.5f * (zk.x + zr.x), 0.5f * (zk.y - zr.y));
            const float2 dd = make_float2(zk.x - zr.x, zk.y + zr.y);  // Zk - conj(Zr)
            const float2 xo = make_float2(0.5f * dd.y, -0.5f * dd.x); // -i/2 * dd
            const float re = xe.x + unt[q].x * xo.x - unt[q].y * xo.y;
            const float im = xe.y + unt[q].x * xo.y + unt[q].y * xo.x;
            pw[k] = re * re + im * im;
        }
        if (tid == 0) {                                               // k = N2 (Nyquist): Z[0] with the factor exp(-i pi) = -1
            const float2 z0 = in[0];
            const float re = z0.x - z0.y;
            pw[N2] = re * re;
        }
        __syncthreads();

        // sparse mel: two threads per filter, alternate bins, combine with one shuffle
        float* dst = mel_out + ((size_t)b * n_frames + f0 + f) * n_mels;
        for (int m = my_m; m < n_mels; m += NT / 2) {
            const bool first = m == my_m;
            const int st = first ? my_st : mel_start[m], ln = first ? my_ln : mel_len[m], off = first ? my_off : mel_off[m];
            float acc = 0.f;
            for (int i = tid & 1; i < ln; i += 2) acc += smw[off + i] * pw[st + i];
            acc += __shfl_xor(acc, 1, 64);
            if ((tid & 1) == 0) dst[m] = logf(fmaxf(acc, log_floor));
        }
        __syncthreads();                                              // buffers are rewritten by the next frame
    }
}

int launch_logmel(const FrontendTables& t, const float* audio, float* mel, int B, hipStream_t stream) {
    if (t.hop > MAX_HOP || t.n_mel_w > MAX_MEL_W) return -1;
    const int grid = B * ((t.n_frames + FPW - 1) / FPW);
    if (grid == 0) return 0;
    if (t.n_fft == 2048) {
        logmel_kernel<5><<<grid, 256, 0, stream>>>(audio, mel, t.window, t.tw, t.untw, t.mel_start, t.mel_len,
                                                  t.mel_off, t.mel_w, t.n_samples, t.n_frames, t.hop, t.n_mels, t.n_mel_w, t.log_floor);
    } else if (t.n_fft == 512) {
        logmel_kernel<4><<<grid, 64, 0, stream>>>(audio, mel, t.window, t.tw, t.untw, t.mel_start, t.mel_len,
                                                 t.mel_off, t.mel_w, t.n_samples, t.n_frames, t.hop, t.n_mels, t.n_mel_w, t.log_floor);
    } else {
        return -1;
    }
    return 0;
}
