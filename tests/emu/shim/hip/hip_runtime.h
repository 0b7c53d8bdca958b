// TEST INFRASTRUCTURE -- not part of the product.
//
// A stand-in for <hip/hip_runtime.h> that lets g++ compile sortmerna_amd/csrc/smr_engine.hip (kernels included)
// for the HOST, so that the -m "not gpu" suite can run the *actual kernel source* on small cases and compare it
// with the oracle before a GPU is spent on it.  Execution model (tests/emu/emu_runtime.cpp):
//   * every work-item is a fiber; the 64 fibers of a wave run one after the other up to their next wave-level
//     operation (__shfl*, __ballot, __any, __all, DPP, __threadfence_block) or block barrier;
//   * a wave-level operation resolves when every unfinished lane of the wave has arrived at it -- and all of them
//     must have arrived at the SAME call site: a shuffle/ballot executed by a divergent subset of a wave is
//     reported as an error (on the hardware such lanes read 0 from the inactive ones, the bug class this catches);
//   * blocks are distributed over host threads; atomics are real atomics; fresh device memory is filled with 0xA5 and lies
//     between guard pages (an out-of-bounds access of a kernel faults at once).
// Nothing under sortmerna_amd/ includes this file; libsmr_emu.so is built under tests/emu/_build only.
#pragma once
#include <chrono>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <type_traits>

#define SMR_EMU 1
#define __global__
#define __device__
#define __host__
#define __constant__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define __align__(n) __attribute__((aligned(n)))

struct dim3 { uint32_t x, y, z; dim3(uint32_t a = 1, uint32_t b = 1, uint32_t c = 1) : x(a), y(b), z(c) {} };
struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; } __attribute__((aligned(16)));
struct int2 { int x, y; };
struct int4 { int x, y, z, w; } __attribute__((aligned(16)));
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }

// HIP's global min/max (mixed integer types allowed)
template <class A, class B> inline typename std::common_type<A, B>::type min(A a, B b) { typedef typename std::common_type<A, B>::type T; return (T)a < (T)b ? (T)a : (T)b; }
template <class A, class B> inline typename std::common_type<A, B>::type max(A a, B b) { typedef typename std::common_type<A, B>::type T; return (T)a > (T)b ? (T)a : (T)b; }

namespace emu {
struct Idx { uint32_t x, y, z; };
struct Ctx { Idx tid, bid, bdim, gdim; };
extern thread_local Ctx cur;                      // refreshed by the scheduler whenever a fiber is resumed
enum Op { OP_XCHG = 0, OP_FENCE = 1 };
// deposit v, wait for the wave; afterwards res[l] holds lane l's deposit and the return value is the participant mask
uint64_t wave_exchange(uint64_t v, const uint64_t** res, int site, const void* ret_addr);
void block_barrier();
void* dyn_lds();
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& fn);
unsigned long long clock();
template <class T> inline uint64_t to_bits(T v) { static_assert(sizeof(T) <= 8, "shuffle of a type wider than 64 bit"); uint64_t b = 0; memcpy(&b, &v, sizeof(T)); return b; }
template <class T> inline T from_bits(uint64_t b) { T v; memcpy(&v, &b, sizeof(T)); return v; }
}  // namespace emu

#define threadIdx (emu::cur.tid)
#define blockIdx (emu::cur.bid)
#define blockDim (emu::cur.bdim)
#define gridDim (emu::cur.gdim)
static const int warpSize = 64;

// ---- wave-level operations ------------------------------------------------------------------------------------------
// Every operation carries the number of its textual occurrence (__COUNTER__) as its site: lanes of one wave that wait in
// operations with different sites have diverged.  (The return address only serves the error report: the host compiler may
// clone a call, so addresses cannot identify a site.)
#define EMU_LANE ((int)(emu::cur.tid.x & 63))
#define EMU_RA __builtin_return_address(0)
namespace emu {
template <class T> __attribute__((noinline)) T shfl_abs(int site, T v, int src) {          // src = absolute lane; outside the wave or not participating -> 0
  const uint64_t* res; const uint64_t m = wave_exchange(to_bits(v), &res, site, EMU_RA);
  if (src < 0 || src > 63 || !((m >> src) & 1)) return from_bits<T>(0);
  return from_bits<T>(res[src]);
}
template <class T> inline __attribute__((always_inline)) T shfl(int site, T v, int src, int width = 64) {
  const int l = EMU_LANE; return shfl_abs(site, v, (l & ~(width - 1)) | (src & (width - 1)));
}
template <class T> inline __attribute__((always_inline)) T shfl_up(int site, T v, unsigned d, int width = 64) {
  const int l = EMU_LANE, s = l - (int)d; return shfl_abs(site, v, s >= (l & ~(width - 1)) ? s : l);
}
template <class T> inline __attribute__((always_inline)) T shfl_down(int site, T v, unsigned d, int width = 64) {
  const int l = EMU_LANE, s = l + (int)d; return shfl_abs(site, v, s <= (l | (width - 1)) ? s : l);
}
template <class T> inline __attribute__((always_inline)) T shfl_xor(int site, T v, int m, int width = 64) {
  const int l = EMU_LANE, s = l ^ m; return shfl_abs(site, v, (s & ~(width - 1)) == (l & ~(width - 1)) ? s : l);
}
__attribute__((noinline)) inline unsigned long long ballot(int site, int p) {
  const uint64_t* res; const uint64_t m = wave_exchange(p ? 1 : 0, &res, site, EMU_RA);
  unsigned long long b = 0; for (int l = 0; l < 64; l++) if (((m >> l) & 1) && res[l]) b |= 1ull << l; return b;
}
__attribute__((noinline)) inline int all(int site, int p) {
  const uint64_t* res; const uint64_t m = wave_exchange(p ? 1 : 0, &res, site, EMU_RA);
  for (int l = 0; l < 64; l++) if (((m >> l) & 1) && !res[l]) return 0; return 1;
}
// DPP (gfx9 encodings of dpp_ctrl): quad_perm 0x00-0xFF, row_shl:n 0x101-0x10F (lane l reads l+n of its row of 16), row_shr:n 0x111-0x11F,
// row_ror:n 0x121-0x12F, wave_shl:1 0x130, wave_rol:1 0x134, wave_shr:1 0x138, wave_ror:1 0x13C, row_mirror 0x140, row_half_mirror 0x141,
// row_bcast:15 0x142 (lane 15 of a row to every lane of the next row), row_bcast:31 0x143 (lane 31 to rows 2 and 3).  A lane whose row
// (row_mask) or bank of 4 (bank_mask) is masked off, or whose source lane does not exist / does not participate, keeps `old`
// (bound_ctrl = 1: a non-existent source writes 0 instead).
__attribute__((noinline)) inline int update_dpp(int site, int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  const uint64_t* res; const uint64_t m = wave_exchange((uint32_t)src, &res, site, EMU_RA);
  const int l = EMU_LANE, row = l >> 4, rl = l & 15;
  if (!((row_mask >> row) & 1) || !((bank_mask >> ((l >> 2) & 3)) & 1)) return old;
  int from = -1; bool exists = true;
  if (ctrl >= 0 && ctrl <= 0xFF) from = (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);
  else if (ctrl >= 0x101 && ctrl <= 0x10F) { const int s = rl + (ctrl & 15); exists = s <= 15; from = (row << 4) | (s & 15); }
  else if (ctrl >= 0x111 && ctrl <= 0x11F) { const int s = rl - (ctrl & 15); exists = s >= 0; from = (row << 4) | (s & 15); }
  else if (ctrl >= 0x121 && ctrl <= 0x12F) from = (row << 4) | ((rl - (ctrl & 15)) & 15);
  else if (ctrl == 0x130) { exists = l < 63; from = (l + 1) & 63; }
  else if (ctrl == 0x134) from = (l + 1) & 63;
  else if (ctrl == 0x138) { exists = l > 0; from = (l + 63) & 63; }
  else if (ctrl == 0x13C) from = (l + 63) & 63;
  else if (ctrl == 0x140) from = (row << 4) | (15 - rl);
  else if (ctrl == 0x141) from = (l & ~7) | (7 - (l & 7));
  else if (ctrl == 0x142) { exists = row > 0; from = ((row - 1) << 4) | 15; }
  else if (ctrl == 0x143) { exists = row >= 2; from = 31; }
  else { fprintf(stderr, "emu: unsupported DPP control 0x%x\n", ctrl); abort(); }
  if (!exists) return bound_ctrl ? 0 : old;
  if (!((m >> from) & 1)) return old;
  return (int)(uint32_t)res[from];
}
__attribute__((noinline)) inline int readlane(int site, int v, int l) {          // v_readlane_b32: every lane gets lane l's value
  const uint64_t* res; const uint64_t m = wave_exchange((uint32_t)v, &res, site, EMU_RA);
  if (!((m >> (l & 63)) & 1)) { fprintf(stderr, "emu: readlane of a lane that is not active\n"); abort(); }
  return (int)(uint32_t)res[l & 63];
}
__attribute__((noinline)) inline void wave_sync() { const uint64_t* res; wave_exchange(0, &res, 0, EMU_RA); }   // lockstep point (site 0 = not compared)
}  // namespace emu
#define __shfl(...) emu::shfl(__COUNTER__ + 1, __VA_ARGS__)
#define __shfl_up(...) emu::shfl_up(__COUNTER__ + 1, __VA_ARGS__)
#define __shfl_down(...) emu::shfl_down(__COUNTER__ + 1, __VA_ARGS__)
#define __shfl_xor(...) emu::shfl_xor(__COUNTER__ + 1, __VA_ARGS__)
#define __ballot(p) emu::ballot(__COUNTER__ + 1, (p))
#define __any(p) (emu::ballot(__COUNTER__ + 1, (p)) != 0)
#define __all(p) emu::all(__COUNTER__ + 1, (p))
#define __builtin_amdgcn_readlane(v, l) emu::readlane(__COUNTER__ + 1, (v), (l))
#define __builtin_amdgcn_update_dpp(...) emu::update_dpp(__COUNTER__ + 1, __VA_ARGS__)
static inline int emu_sbfe(int v, unsigned off, unsigned width) {          // v_bfe_i32
  off &= 31; width &= 31; if (width == 0) return 0;
  const uint32_t f = ((uint32_t)v >> off) & ((1u << width) - 1);
  return (int)(f << (32 - width)) >> (32 - width);
}
#define __builtin_amdgcn_sbfe emu_sbfe
#define __builtin_amdgcn_wave_barrier() emu::wave_sync()
inline void __threadfence_block() { emu::wave_sync(); }
inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline void __syncthreads() { emu::block_barrier(); }

// ---- scalar intrinsics ----------------------------------------------------------------------------------------------
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline unsigned __brev(unsigned v) { unsigned r = 0; for (int i = 0; i < 32; i++) r |= ((v >> i) & 1u) << (31 - i); return r; }
static inline unsigned long long __brevll(unsigned long long v) { unsigned long long r = 0; for (int i = 0; i < 64; i++) r |= ((v >> i) & 1ull) << (63 - i); return r; }
static inline unsigned long long clock64() { return emu::clock(); }

// ---- atomics (relaxed, like the hardware's) --------------------------------------------------------------------------
template <class T, class V> inline T atomicAdd(T* p, V v) { return __atomic_fetch_add(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class V> inline T atomicOr(T* p, V v) { return __atomic_fetch_or(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class V> inline T atomicAnd(T* p, V v) { return __atomic_fetch_and(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class V> inline T atomicMax(T* p, V v_) { const T v = (T)v_; T o = __atomic_load_n(p, __ATOMIC_RELAXED); while (o < v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} return o; }
template <class T, class V> inline T atomicMin(T* p, V v_) { const T v = (T)v_; T o = __atomic_load_n(p, __ATOMIC_RELAXED); while (o > v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} return o; }
template <class T, class U, class V> inline T atomicCAS(T* p, U cmp_, V v) { T cmp = (T)cmp_; __atomic_compare_exchange_n(p, &cmp, (T)v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED); return cmp; }
template <class T, class V> inline T atomicExch(T* p, V v) { return __atomic_exchange_n(p, (T)v, __ATOMIC_RELAXED); }

// ---- runtime API (synchronous; one device) ---------------------------------------------------------------------------
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
typedef struct emu_stream* hipStream_t;
typedef struct emu_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
struct hipDeviceProp_t { char name[256]; char gcnArchName[256]; size_t totalGlobalMem; int multiProcessorCount; int warpSize; size_t sharedMemPerBlock; };
hipError_t hipGetDeviceCount(int* n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int d);
hipError_t hipMalloc(void** p, size_t n);
template <class T> inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags = 0);
template <class T> inline hipError_t hipHostMalloc(T** p, size_t n, unsigned flags = 0) { return hipHostMalloc((void**)p, n, flags); }
hipError_t hipHostFree(void* p);
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st = nullptr);
hipError_t hipMemset(void* d, int v, size_t n);
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st = nullptr);
hipError_t hipStreamCreate(hipStream_t* s);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipDeviceSynchronize();
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = nullptr);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipGetLastError();
hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b);
const char* hipGetErrorString(hipError_t e);

#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
  emu::launch(dim3(grid), dim3(block), (size_t)(shmem), [=]() { (kern)(__VA_ARGS__); })
