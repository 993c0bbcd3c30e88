// What would a stage boundary cost INSIDE one kernel for the decode step's skinny-GEMM chain (cross O -> FFN-in -> FFN-out -> QKV),
// against the dependent launch it would replace (gap 1.6-2.1 us + the activation round trip)?  The probe keeps the GEMM chain's real
// dependency shape at 64 rows: 4 row tiles x 64 column tiles = 256 co-resident workgroups of 512 threads; tile (mt, nt) of stage s
// needs the 16 complete rows of row tile mt from stage s - 1, i.e. all 64 tiles (mt, *), reads them (32 KB), reduces across its waves
// through LDS, writes its 16 x 8 outputs and signals.  Row tiles never wait for each other.
//   mode 0  one launch per stage in a captured graph (plain loads / stores): the baseline
//   mode 1  one kernel; agent-scope (sc1) stores and loads, one arrival counter per row tile (64 atomic arrivals)
//   mode 2  one kernel; one flag word per producer tile, the consumer's first wave polls the row tile's 64 flags with one load
//   mode 3  one kernel; no flags: every 16-byte chunk carries its stage tag, the consumer re-reads its rows until all tags match
//   mode 4  one kernel; grid-wide barrier on one counter (256 arrivals), for reference
// Every value read is checked (a stale line shows as a mismatch); every spin is bounded and a sticky abort flag drains the grid.
// (Measurement tool; not part of the product path.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ROWS = 64, COLS = 512, MT = 4, NTILES = 64, G = MT * NTILES, THREADS = 512;
constexpr unsigned MAX_POLLS = 1u << 16;

struct Sync { unsigned cnt[MT][32]; unsigned flag[MT][NTILES]; unsigned grid; unsigned abort; unsigned bad; unsigned polls; };

__device__ __forceinline__ f32x4 ld_agent(const f32x4* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void st_agent(f32x4* p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void wait4(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
}
__device__ __forceinline__ bool aborted(Sync* s) { return __hip_atomic_load(&s->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0; }
__device__ __forceinline__ void give_up(Sync* s) { __hip_atomic_store(&s->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// one stage of one tile; `in` = the previous stage's rows (tag `want`), `out` = this stage's (tag want + 1)
template <int MODE>
__device__ __forceinline__ void stage(Sync* s, const float* in, float* out, int st, float* red) {
    const int tid = threadIdx.x, mt = blockIdx.x & 3, nt = blockIdx.x >> 2;
    const float want = (float)st;
    // ---- wait for the row tile
    if (MODE == 1 && st > 0) {
        if (tid == 0) {
            unsigned polls = 0;
            while (__hip_atomic_load(&s->cnt[mt][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(NTILES * st)) {
                if (++polls > MAX_POLLS || aborted(s)) { give_up(s); break; }
            }
        }
        __syncthreads();
    }
    if (MODE == 2 && st > 0) {
        if (tid < 64) {
            unsigned polls = 0;
            while (true) {
                const unsigned f = __hip_atomic_load(&s->flag[mt][tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__ballot(f < (unsigned)st) == 0ull) break;
                if (++polls > MAX_POLLS || aborted(s)) { give_up(s); break; }
            }
        }
        __syncthreads();
    }
    if (MODE == 4 && st > 0) {
        if (tid == 0) {
            unsigned polls = 0;
            while (__hip_atomic_load(&s->grid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(G * st)) {
                if (++polls > MAX_POLLS || aborted(s)) { give_up(s); break; }
            }
        }
        __syncthreads();
    }
    // ---- read the 16 complete rows (32 KB: 4 x 16 B per thread), check them
    const f32x4* src = reinterpret_cast<const f32x4*>(in + (size_t)mt * 16 * COLS);
    f32x4 v0, v1, v2, v3;
    unsigned polls = 0;
    while (true) {
        if (MODE == 0) {
            v0 = src[tid]; v1 = src[tid + 512]; v2 = src[tid + 1024]; v3 = src[tid + 1536];
        } else {
            v0 = ld_agent(src + tid); v1 = ld_agent(src + tid + 512); v2 = ld_agent(src + tid + 1024); v3 = ld_agent(src + tid + 1536);
            wait4(v0, v1, v2, v3);
        }
        if (MODE != 3) break;
        const bool late = v0[3] != want || v1[3] != want || v2[3] != want || v3[3] != want;
        if (!__syncthreads_or(late)) break;
        if (++polls > MAX_POLLS / 16 || aborted(s)) { give_up(s); break; }
    }
    unsigned bad = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) bad += (v0[e] != want) + (v1[e] != want) + (v2[e] != want) + (v3[e] != want);
    if (bad) atomicAdd(&s->bad, bad);
    if (MODE == 3 && tid == 0 && polls) atomicAdd(&s->polls, polls);
    // ---- the cross-wave reduction a split-K tile does
    float acc = (v0[0] + v1[1]) + (v2[2] + v3[3]);
    red[tid] = acc;
    __syncthreads();
    float sum = 0.f;
    if (tid < 32) {
#pragma unroll
        for (int w = 0; w < 8; ++w) sum += red[w * 64 + tid];
    }
    // ---- 16 rows x 8 columns out (32 threads x 16 B), then signal
    if (tid < 32) {
        const float o = want + 1.f + (sum != sum ? 1.f : 0.f);      // (keeps the reduction alive)
        const f32x4 ov = {o, o, o, o};
        f32x4* dst = reinterpret_cast<f32x4*>(out + (size_t)(mt * 16 + (tid >> 1)) * COLS + nt * 8 + (tid & 1) * 4);
        if (MODE == 0) *dst = ov;
        else { st_agent(dst, ov); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    }
    if (MODE == 1 || MODE == 2 || MODE == 4) {
        __syncthreads();
        if (tid == 0) {
            if (MODE == 1) __hip_atomic_fetch_add(&s->cnt[mt][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (MODE == 2) __hip_atomic_store(&s->flag[mt][nt], (unsigned)(st + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (MODE == 4) __hip_atomic_fetch_add(&s->grid, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(THREADS) void k_chain(Sync* s, float* buf, int st0, int n_stages) {
    __shared__ float red[THREADS];
    for (int st = st0; st < st0 + n_stages; ++st) {
        stage<MODE>(s, buf + (size_t)((st + 1) & 1) * ROWS * COLS, buf + (size_t)(st & 1) * ROWS * COLS, st, red);
        if (MODE == 0 || MODE == 3) __syncthreads();               // red[] reuse
    }
}

template <int MODE>
static int run(const char* name, int n_stages, hipStream_t stream) {
    Sync* s; float* buf;
    CK(hipMalloc(&s, sizeof(Sync)));
    CK(hipMalloc(&buf, sizeof(float) * 2 * ROWS * COLS));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipGraph_t graph; hipGraphExec_t exec;
    CK(hipStreamBeginCapture(stream, hipStreamCaptureModeGlobal));
    if (MODE == 0) for (int st = 0; st < n_stages; ++st) k_chain<MODE><<<G, THREADS, 0, stream>>>(s, buf, st, 1);
    else k_chain<MODE><<<G, THREADS, 0, stream>>>(s, buf, 0, n_stages);
    CK(hipStreamEndCapture(stream, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    float best = 1e30f;
    unsigned bad = 0, ab = 0, polls = 0;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemsetAsync(s, 0, sizeof(Sync), stream));
        CK(hipMemsetAsync(buf, 0, sizeof(float) * 2 * ROWS * COLS, stream));        // stage 0 expects tag 0 in buffer 1
        CK(hipEventRecord(e0, stream));
        CK(hipGraphLaunch(exec, stream));
        CK(hipEventRecord(e1, stream));
        CK(hipStreamSynchronize(stream));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        Sync h; CK(hipMemcpy(&h, s, sizeof(Sync), hipMemcpyDeviceToHost));
        bad += h.bad; ab += h.abort; polls = h.polls;
    }
    printf("%-58s %4d stages: %6.2f us per stage   (mismatches %u, aborted %u, tag re-reads %u)\n", name, n_stages, best * 1000.f / n_stages, bad, ab, polls);
    CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
    CK(hipFree(s)); CK(hipFree(buf));
    return 0;
}

int main() {
    hipStream_t stream;
    CK(hipStreamCreate(&stream));
    const int n = 240;
    if (run<0>("mode 0: one launch per stage (graph)", n, stream)) return 1;
    if (run<1>("mode 1: one kernel, counter per row tile (64 arrivals)", n, stream)) return 1;
    if (run<2>("mode 2: one kernel, flag per producer tile", n, stream)) return 1;
    if (run<3>("mode 3: one kernel, stage tag in every 16-byte chunk", n, stream)) return 1;
    if (run<4>("mode 4: one kernel, grid barrier (256 arrivals)", n, stream)) return 1;
    return 0;
}
