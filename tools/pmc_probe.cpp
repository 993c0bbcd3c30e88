// minimal program to check that rocprofv3 --pmc works at all on the box
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void copy_k(const float4* a, float4* b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
int main() {
    const size_t n = (size_t)64 << 20;  // 1 GiB of float4
    float4 *a, *b;
    hipMalloc((void**)&a, n * 16); hipMalloc((void**)&b, n * 16);
    hipMemset(a, 1, n * 16);
    for (int i = 0; i < 3; ++i) copy_k<<<2048, 256>>>(a, b, n);
    hipDeviceSynchronize();
    printf("ok\n");
    return 0;
}
