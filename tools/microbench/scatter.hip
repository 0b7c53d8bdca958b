// scatter.hip -- what does a window's "there is a hit segment" mark cost?  wseg_put (smr_seed.hpp) stores the segment's place (4 bytes at a random
// slot of a 256 MB array) and sets the slot's bit in an 8 MB bitmap with a non-returning atomicOr.  32 M threads, one mark each, random slots:
//   0 the 4-byte store alone            1 the atomicOr alone               2 a 1-byte store into a 64 MB byte array alone
//   3 store + atomicOr (wseg_put now)   4 store + byte store               5 store + atomicOr, one lane in four active (sparser hits)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>
__global__ void __launch_bounds__(256) k_scatter(uint32_t* wseg, uint32_t* bits, uint8_t* bytes, uint32_t slots) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (MODE == 5 && (t & 3u)) return;
  const uint32_t slot = mix(t) & (slots - 1u);
  if (MODE == 0 || MODE >= 3) wseg[slot] = t;
  if (MODE == 1 || MODE == 3 || MODE == 5) atomicOr(&bits[slot >> 5], 1u << (slot & 31u));
  if (MODE == 2 || MODE == 4) bytes[slot] = 1;
}

template <int MODE> void run(uint32_t* wseg, uint32_t* bits, uint8_t* bytes, uint32_t slots, const char* what) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const uint32_t n = 32u << 20;
  hipLaunchKernelGGL(k_scatter<MODE>, dim3(n / 256), dim3(256), 0, 0, wseg, bits, bytes, slots);
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k_scatter<MODE>, dim3(n / 256), dim3(256), 0, 0, wseg, bits, bytes, slots);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  const double marks = MODE == 5 ? n / 4 : n;
  printf("%-64s %8.3f ms  %7.2f marks per ns\n", what, ms, marks / (ms * 1e6));
}

int main() {
  CK(hipSetDevice(0));
  const uint32_t slots = 64u << 20;
  uint32_t *wseg, *bits; uint8_t* bytes;
  CK(hipMalloc(&wseg, (size_t)slots * 4)); CK(hipMalloc(&bits, slots / 8)); CK(hipMalloc(&bytes, slots));
  CK(hipMemset(wseg, 0, (size_t)slots * 4)); CK(hipMemset(bits, 0, slots / 8)); CK(hipMemset(bytes, 0, slots));
  run<0>(wseg, bits, bytes, slots, "4-byte store, random slot of 256 MB");
  run<1>(wseg, bits, bytes, slots, "atomicOr (not returning) into an 8 MB bitmap");
  run<2>(wseg, bits, bytes, slots, "1-byte store into a 64 MB byte array");
  run<3>(wseg, bits, bytes, slots, "store + atomicOr (wseg_put)");
  run<4>(wseg, bits, bytes, slots, "store + byte store");
  run<5>(wseg, bits, bytes, slots, "store + atomicOr, one lane in four");
  return 0;
}
