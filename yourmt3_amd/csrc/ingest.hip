// Audio ingest, the step before the hot path: interleaved PCM (int16 or fp32, any
// channel count, any rate) -> mono mix -> polyphase FIR resample to the model rate -> zero-padded fixed-length
// segments, written straight into the (n_seg, segment_samples) layout ymt3_logmel reads.
//
//   y[n] = sum_j P[phase(n)][j] * x[k0(n) - j],   q = (n + r) * down,  phase = q % up,  k0 = q / up
//
// (the upfirdn form of TP: scipy/signal/_signaltools.py resample_poly; the taps P and the alignment r are derived on
// the host in runtime.hip).  One workgroup produces 256 consecutive output samples: the input span they need is
// mono-mixed once into LDS with coalesced reads of the interleaved frames (HBM sees every PCM byte once per
// workgroup span, ~1.1x overall), each thread then walks its own phase row of the tap table (16-byte loads, L2
// resident: <= 40 KB per rate pair).  HBM-bound streaming work; fp32 throughout.
// Oracle: oracle/ingest_oracle.py::ingest.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int OUT_PER_WG = 256;

template <bool S16>
__global__ __launch_bounds__(OUT_PER_WG) void ingest_kernel(IngestArgs a) {
    extern __shared__ float xw[];                                   // mono input window of this workgroup
    const int tid = threadIdx.x;
    const long long n0 = (long long)blockIdx.x * OUT_PER_WG;
    if (n0 >= a.n_out) {                                            // pure padding tail
        const long long n = n0 + tid;
        if (n < a.n_total) a.out[n] = 0.f;
        return;
    }
    long long n_last = n0 + OUT_PER_WG - 1;
    if (n_last >= a.n_out) n_last = a.n_out - 1;
    const long long k_hi = ((n_last + a.r) * a.down) / a.up;        // newest input frame any output here needs
    const long long k_lo = ((n0 + a.r) * a.down) / a.up - (a.J - 1);
    const int W = (int)(k_hi - k_lo + 1);                           // <= a.window (checked on the host)
    const float inv_c = 1.0f / (float)a.n_channels;
    for (int i = tid; i < W; i += OUT_PER_WG) {
        const long long k = k_lo + i;
        float v = 0.f;
        if (k >= 0 && k < a.n_in) {
            if (S16) {
                const short* p = static_cast<const short*>(a.pcm) + k * a.n_channels;
                float s = 0.f;
                for (int c = 0; c < a.n_channels; ++c) s += (float)p[c] * (1.0f / 32768.0f);
                v = s * inv_c;
            } else {
                const float* p = static_cast<const float*>(a.pcm) + k * a.n_channels;
                float s = 0.f;
                for (int c = 0; c < a.n_channels; ++c) s += p[c];
                v = s * inv_c;
            }
        }
        xw[i] = v;
    }
    __syncthreads();
    const long long n = n0 + tid;
    if (n >= a.n_total) return;
    float y = 0.f;
    if (n < a.n_out) {
        const long long q = (n + a.r) * a.down;
        const int phase = (int)(q % a.up);
        const int base = (int)(q / a.up - k_lo);                    // xw index of x[k0]
        const float4* row = reinterpret_cast<const float4*>(a.taps + (size_t)phase * a.Jp);
        for (int j4 = 0; j4 < a.Jp / 4; ++j4) {
            const float4 t = row[j4];
            const int i = base - 4 * j4;
            // taps beyond J are zero and the window below k_lo is never read: clamp the index, keep the product
            y += t.x * xw[max(i, 0)];
            y += t.y * xw[max(i - 1, 0)];
            y += t.z * xw[max(i - 2, 0)];
            y += t.w * xw[max(i - 3, 0)];
        }
    }
    a.out[n] = y;
}

}  // namespace

int launch_ingest(const IngestArgs& a, hipStream_t stream) {
    if (a.n_total <= 0) return 0;
    if (a.up <= 0 || a.down <= 0 || a.J <= 0 || a.Jp % 4 || a.Jp < a.J || a.n_channels <= 0 || a.window <= 0) return -1;
    const size_t lds = (size_t)a.window * sizeof(float);
    if (lds > 64 * 1024) return -2;
    const long long blocks = (a.n_total + OUT_PER_WG - 1) / OUT_PER_WG;
    if (blocks > 0x7fffffffLL) return -3;
    if (a.s16) ingest_kernel<true><<<(int)blocks, OUT_PER_WG, lds, stream>>>(a);
    else ingest_kernel<false><<<(int)blocks, OUT_PER_WG, lds, stream>>>(a);
    return 0;
}
