// What v_permlane16_swap / v_permlane32_swap (gfx950) return when both operands are the same register: prints, per lane, the two result
// registers for the input x = lane id, next to what __shfl_xor(x, 16) / (x, 32) give.  (Measurement tool; not part of the product path.)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    const unsigned lane = threadIdx.x;
    auto r32 = __builtin_amdgcn_permlane32_swap(lane, lane, false, false);
    auto r16 = __builtin_amdgcn_permlane16_swap(lane, lane, false, false);
    out[lane * 6 + 0] = r32[0]; out[lane * 6 + 1] = r32[1];
    out[lane * 6 + 2] = r16[0]; out[lane * 6 + 3] = r16[1];
    out[lane * 6 + 4] = __shfl_xor(lane, 32, 64); out[lane * 6 + 5] = __shfl_xor(lane, 16, 64);
}
int main() {
    unsigned* d; unsigned h[64 * 6];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 1;
    k<<<1, 64>>>(d);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    printf("lane | swap32[0] swap32[1] | swap16[0] swap16[1] | xor32 xor16\n");
    int bad32 = 0, bad16 = 0;
    for (int l = 0; l < 64; ++l) {
        const unsigned* r = h + l * 6;
        if (l % 8 == 0) printf("%4d | %9u %9u | %9u %9u | %5u %5u\n", l, r[0], r[1], r[2], r[3], r[4], r[5]);
        const unsigned p32 = l < 32 ? r[1] : r[0], p16 = (l & 16) ? r[2] : r[3];
        bad32 += p32 != r[4]; bad16 += p16 != r[5];
    }
    printf("partner by (lane < 32 ? swap32[1] : swap32[0]): %d mismatches; by (lane & 16 ? swap16[0] : swap16[1]): %d mismatches\n", bad32, bad16);
    return 0;
}
