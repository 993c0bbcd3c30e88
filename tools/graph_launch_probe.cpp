// Host cost of hipGraphLaunch for a 50-kernel graph, from 1, 2 and 4 host threads (one stream + graph each).
// Tells whether concurrent decode chains can be fed from several host threads.  Measurement tool.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void k_tiny(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(err_)); exit(1); } } while (0)
struct Chain { hipStream_t st; hipGraph_t g; hipGraphExec_t e; float* buf; };
int main() {
    const int NODES = 50, LAUNCHES = 1000;
    std::vector<Chain> ch(4);
    for (auto& c : ch) {
        CK(hipStreamCreateWithFlags(&c.st, hipStreamNonBlocking));
        CK(hipMalloc((void**)&c.buf, 256));
        CK(hipStreamBeginCapture(c.st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < NODES; ++i) k_tiny<<<64, 256, 0, c.st>>>(c.buf);
        CK(hipStreamEndCapture(c.st, &c.g));
        CK(hipGraphInstantiate(&c.e, c.g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(c.e, c.st)); CK(hipStreamSynchronize(c.st));
    }
    for (int nt : {1, 2, 4}) {
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        std::vector<double> host_us(nt);
        for (int t = 0; t < nt; ++t)
            th.emplace_back([&, t] {
                auto h0 = std::chrono::steady_clock::now();
                for (int i = 0; i < LAUNCHES; ++i) CK(hipGraphLaunch(ch[t].e, ch[t].st));
                host_us[t] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count() / LAUNCHES;
                CK(hipStreamSynchronize(ch[t].st));
            });
        for (auto& x : th) x.join();
        const double wall = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("%d thread(s): host %.1f us per hipGraphLaunch (thread 0), wall %.1f us per step of %d chain(s) = %.2f us per kernel node\n",
               nt, host_us[0], wall / LAUNCHES, nt, wall / LAUNCHES / (NODES * nt));
    }
    return 0;
}
