// Shared device helpers for the gfx950 kernels (wave = 64 lanes everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;   // raw bf16 bit pattern in memory
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define WAVE 64

__device__ __forceinline__ float bf2f(bf16_t b) { return __uint_as_float(((uint32_t)b) << 16); }

// fp32 -> bf16, round to nearest even (v_cvt_pk_bf16_f32 on gfx950; NaN stays NaN).
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 h = (__bf16)f;
    return __builtin_bit_cast(bf16_t, h);
}

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}

// A product and a sum that must stay two roundings (hipcc contracts a*b + c into an fma by default, and __fmul_rn / __fadd_rn are
// plain operators to it): the pragma strips the `contract` flag from these two operations only.
__device__ __forceinline__ float mul_sep(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float add_sep(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}

// Cross-lane moves inside a 16-lane row by DPP (no LDS crossbar round trip: a ds_bpermute costs ~100 cycles, and a chain of them sits
// on the critical path of every latency-bound decode kernel).
#define DPP_F(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xF, 0xF, true))
// the same restricted to the 16-lane rows in `rows` (bit r = row r); the other rows read 0.0f
#define DPP_ROWS(v, ctrl, rows) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), (rows), 0xF, false))
__device__ __forceinline__ float add_xor1(float v) { return v + DPP_F(v, 0xB1); }     // quad_perm [1,0,3,2]: + lane ^ 1
__device__ __forceinline__ float add_xor2(float v) { return v + DPP_F(v, 0x4E); }     // quad_perm [2,3,0,1]: + lane ^ 2
__device__ __forceinline__ float add_xor8(float v) { return v + DPP_F(v, 0x128); }    // row_ror:8: + lane ^ 8 within the row of 16
// v + lanes ^1, ^2, ^4: bit for bit `v += shfl_xor(v, 1); v += shfl_xor(v, 2); v += shfl_xor(v, 4)` -- the first two steps add the same
// lane pairs; after them the four lanes of a quad hold one value, so the third step may read ANY lane of the other quad
// (row_half_mirror: lane 7 - l) for what lane ^ 4 holds
__device__ __forceinline__ float sum8(float v) {
    v = add_xor1(v);
    v = add_xor2(v);
    v += DPP_F(v, 0x141);
    return v;
}

// The value of lane ^ 16 / lane ^ 32 at VALU speed: gfx950's v_permlane16_swap / v_permlane32_swap with both operands the same register
// return (tools/permlane_probe.cpp, checked on hardware against __shfl_xor) { [r0 r0 r2 r2], [r1 r1 r3 r3] } over the 16-lane rows and
// { [lo lo], [hi hi] } over the 32-lane halves.  One-dimensional workgroups (threadIdx.x & 63 is the lane), all lanes active.
__device__ __forceinline__ float lane_xor16(float v) {
    const unsigned x = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    return __builtin_bit_cast(float, (threadIdx.x & 16u) ? r[0] : r[1]);
}
__device__ __forceinline__ float lane_xor32(float v) {
    const unsigned x = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return __builtin_bit_cast(float, (threadIdx.x & 32u) ? r[0] : r[1]);
}
// lane ^ off for off = 8, 16, 32 (a compile-time constant after unrolling) without the LDS crossbar
__device__ __forceinline__ float lane_xor(float v, int off) {
    if (off == 8) return DPP_F(v, 0x128);
    if (off == 16) return lane_xor16(v);
    return lane_xor32(v);
}

// The 64-lane butterfly sum  v += lane^32; += lane^16; += lane^8; += lane^4; += lane^2; += lane^1  with every step as a register move
// (every lane receives exactly the lane the shuffle gave it: same operands, same order, same bits).  lane ^ 4 takes two DPP moves:
// row_shl:4 for the lanes whose bit 2 is clear (banks 0 and 2 read lane + 4), row_shr:4 for the others (banks 1 and 3 read lane - 4).
__device__ __forceinline__ float wave_sum(float v) {
    v += lane_xor32(v);
    v += lane_xor16(v);
    v = add_xor8(v);
    {
        int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x104, 0xF, 0x5, false);      // row_shl:4 -> banks 0, 2
        t = __builtin_amdgcn_update_dpp(t, __builtin_bit_cast(int, v), 0x114, 0xF, 0xA, false);          // row_shr:4 -> banks 1, 3
        v += __builtin_bit_cast(float, t);
    }
    v = add_xor2(v);
    v = add_xor1(v);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// unpack 8 bf16 (one 16-byte load) to fp32
__device__ __forceinline__ void unpack8(const uint4& u, float* f) {
    f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
    f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
    f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
    f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
}

// One wave-instruction of direct global -> LDS DMA: every lane's 16 bytes land at lds_dst + 16 * lane (1 KiB, linear).  Inline asm,
// so hipcc neither counts it in its vmcnt bookkeeping nor orders LDS reads behind it: the issuing wave waits with an explicit
// `s_waitcnt vmcnt(n)` (older DMAs retire before younger loads: vector-memory data returns in issue order).
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            ymt3_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return YMT3_ERR_HIP;                                                           \
        }                                                                                  \
    } while (0)

void ymt3_set_error(const char* fmt, ...);
