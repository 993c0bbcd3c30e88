// Which XCDs / CUs does a stream CU mask select on MI355X?  (measurement tool, not product code)
// For several masks: launch 2048 workgroups on a stream created with hipExtStreamCreateWithCUMask and histogram
// HW_REG_XCC_ID and the (SE, CU) fields of HW_REG_HW_ID per workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void census(unsigned* out) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    // keep the workgroup alive a little so that the grid spreads over every CU the mask allows
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 300) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}

static int run(const char* name, const std::vector<uint32_t>& mask) {
    hipStream_t st;
    CK(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
    const int G = 4096;
    unsigned* d;
    CK(hipMalloc((void**)&d, G * 8));
    census<<<G, 256, 0, st>>>(d);
    CK(hipStreamSynchronize(st));
    std::vector<unsigned> h(2 * G);
    CK(hipMemcpy(h.data(), d, G * 8, hipMemcpyDeviceToHost));
    std::map<unsigned, int> xcc;
    std::set<unsigned long long> cus;
    for (int i = 0; i < G; ++i) {
        const unsigned x = h[2 * i] & 0xf, hw = h[2 * i + 1];
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        xcc[x]++;
        cus.insert(((unsigned long long)x << 16) | (se << 8) | (sh << 4) | cu);
    }
    printf("%-28s distinct CUs %3zu  workgroups per XCC:", name, cus.size());
    for (auto& kv : xcc) printf(" %u:%d", kv.first, kv.second);
    printf("\n");
    hipFree(d);
    hipStreamDestroy(st);
    return 0;
}

int main() {
    std::vector<uint32_t> all(8, 0xffffffffu), lo(8, 0), hi(8, 0), even(8, 0x55555555u), mod8(8, 0), first32(8, 0), w0(8, 0), x01(8, 0);
    for (int i = 0; i < 4; ++i) { lo[i] = 0xffffffffu; hi[4 + i] = 0xffffffffu; }
    for (int i = 0; i < 8; ++i) mod8[i] = 0x0f0f0f0fu;          // bits with (i % 8) < 4
    first32[0] = 0xffffffffu;
    for (int i = 0; i < 8; ++i) w0[i] = 0x01010101u;            // bits with i % 8 == 0
    for (int i = 0; i < 8; ++i) x01[i] = 0x03030303u;           // bits with i % 8 in {0, 1}
    run("all 256 bits", all);
    run("bits 0..127", lo);
    run("bits 128..255", hi);
    run("even bits", even);
    run("bits with i%8 < 4", mod8);
    run("bits with i%8 == 0", w0);
    run("bits with i%8 in {0,1}", x01);
    run("bits 0..31", first32);
    return 0;
}
