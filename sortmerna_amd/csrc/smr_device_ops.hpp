// smr_device_ops.hpp -- the few things the kernels ask of the gfx950 compiler beyond plain C++: dynamic LDS, packed 16-bit arithmetic
// (v_pk_add_i16 / v_pk_sub_i16 / v_pk_max_i16), v_perm_b32, v_rcp_f32.  Included as <smr_device_ops.hpp> (build.py passes -I csrc): the
// kernel emulator of the test suite puts its own file of this name in front (tests/emu/shim/smr_device_ops.hpp, scalar host code), so the
// product sources carry no second implementation.
#pragma once
#include <stdint.h>

#define SMR_DYN_LDS(type, name) extern __shared__ __align__(16) type name[]
// a pointer into global memory that was put together from integers: without the address space the compiler emits FLAT loads, which count
// against the LDS counter as well and make every LDS wait a wait for memory
#define SMR_GLOBAL_U32 const uint32_t __attribute__((address_space(1)))
#define SMR_SW_SELFCHECK_CASES 512u                       // random problems smr_create runs through both Smith-Waterman kernels

namespace smr {

typedef short pk16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk16 pk_from(uint32_t v) { return __builtin_bit_cast(pk16, v); }
__device__ __forceinline__ uint32_t pk_bits(pk16 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ pk16 pk_add(pk16 a, pk16 b) { return a + b; }
__device__ __forceinline__ pk16 pk_sub(pk16 a, pk16 b) { return a - b; }
__device__ __forceinline__ pk16 pk_max(pk16 a, pk16 b) { return __builtin_elementwise_max(a, b); }
// v_pk_sub_u16 ... clamp: per half max(a - b, 0) for non-negative a, b (the halves taken as unsigned)
typedef unsigned short pku16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk16 pk_subs_u(pk16 a, pk16 b) { return __builtin_bit_cast(pk16, __builtin_elementwise_sub_sat(__builtin_bit_cast(pku16, a), __builtin_bit_cast(pku16, b))); }
__device__ __forceinline__ uint32_t perm_b32(uint32_t s0, uint32_t s1, uint32_t sel) { return __builtin_amdgcn_perm(s0, s1, sel); }
// A value that is the same in every lane of the wave, said so to the compiler (v_readfirstlane_b32): it then lives in a scalar register, costs
// no vector register across a call and is worked on by the scalar unit.  The kernel emulator's version checks that the lanes do agree.
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned long long uni(unsigned long long v) { return (unsigned long long)uni((uint32_t)v) | ((unsigned long long)uni((uint32_t)(v >> 32)) << 32); }
__device__ __forceinline__ unsigned long uni(unsigned long v) { return (unsigned long)uni((unsigned long long)v); }
template <class T> __device__ __forceinline__ T uni_words(const T& v) {      // a struct of whole 32-bit words
  static_assert(sizeof(T) % 4 == 0, "uni_words: whole words only");
  uint32_t w[sizeof(T) / 4];
  __builtin_memcpy(w, &v, sizeof(T));
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 4; i++) w[i] = uni(w[i]);
  T o;
  __builtin_memcpy(&o, w, sizeof(T));
  return o;
}
// inclusive prefix sum over the 64 lanes in six v_add_u32_dpp: row_shr:1/2/4/8 inside the rows of 16 (a lane without a source adds 0), then
// row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3.  Written out because the compiler turns the same steps, given as
// x += update_dpp(0, x, ...), into a zeroed register + v_mov_b32_dpp + v_add_u32 each: 18 vector instructions instead of 6, three scans per
// chunk of the seed search.  (s_nop 1: two wait states between a VALU write and a DPP read of the same register.)
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t x) {
  asm volatile("s_nop 1\n\t"
               "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\t"
               "v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\t"
               "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\t"
               "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\t"
               "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
               "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
               : "+v"(x));
  return x;
}
// n / d for n = k * d, k < 2^16: one reciprocal and a multiply (the error of v_rcp_f32 is far below the 0.5 that is added)
__device__ __forceinline__ uint32_t div_multiple(uint32_t n, uint32_t d) { return (uint32_t)((float)n * __builtin_amdgcn_rcpf((float)d) + 0.5f); }

}  // namespace smr
