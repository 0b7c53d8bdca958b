// atomics.hip -- how fast do returning global atomics on ONE cache line retire?  (Round 4: the three collect kernels -- 125 000 waves, one
// returning atomicAdd per wave on one counter, 200 MB of reads -- took 1.5 ms each; k_seed_pg / k_seed_finish allocated from 64 pool cursors that
// lay in four 128-byte lines.)  Every wave of the launch does `per` atomics with lane 0, in one of these shapes:
//   0 returning, one address                      1 returning, 64 addresses in 4 lines (8 bytes apart; wave -> address by block number)
//   2 returning, 64 addresses on 64 lines         3 NOT returning, one address
//   4 returning, one address, one atomic per 1024-thread block (the block's 16 waves folded through LDS first)
// Prints the launch time and atomics per microsecond.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ void k_atomics(unsigned long long* ctr, unsigned long long* out, uint32_t per) {
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + wv;
  unsigned long long acc = 0;
  if (MODE == 4) {
    __shared__ uint32_t s_n; __shared__ unsigned long long s_base;
    for (uint32_t i = 0; i < per; i++) {
      if (threadIdx.x == 0) s_n = 0;
      __syncthreads();
      if (lane == 0) atomicAdd(&s_n, 1u);
      __syncthreads();
      if (threadIdx.x == 0) s_base = atomicAdd(&ctr[0], (unsigned long long)s_n);
      __syncthreads();
      acc += s_base;
    }
  } else if (lane == 0) {
    for (uint32_t i = 0; i < per; i++) {
      if (MODE == 0) acc += atomicAdd(&ctr[0], 1ull);
      if (MODE == 1) acc += atomicAdd(&ctr[blockIdx.x & 63u], 1ull);
      if (MODE == 2) acc += atomicAdd(&ctr[(blockIdx.x & 63u) * 16u], 1ull);
      if (MODE == 3) atomicAdd(&ctr[0], 1ull);
    }
  }
  if (lane == 0 && acc == 0x123456789ull) out[wave & 1023u] = acc;       // (keeps the returned values alive)
}

template <int MODE> void run(unsigned long long* ctr, unsigned long long* out, uint32_t waves, uint32_t block, const char* what) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const uint32_t blocks = waves / (block / 64);
  CK(hipMemset(ctr, 0, 64 * 16 * 8));
  hipLaunchKernelGGL(k_atomics<MODE>, dim3(blocks), dim3(block), 0, 0, ctr, out, 1u);
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k_atomics<MODE>, dim3(blocks), dim3(block), 0, 0, ctr, out, 1u);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  const double n = MODE == 4 ? blocks : waves;
  printf("%-76s %7u waves  %8.3f ms  %8.1f global atomics per us\n", what, waves, ms, n / (ms * 1e3));
}

int main() {
  CK(hipSetDevice(0));
  unsigned long long *ctr, *out;
  CK(hipMalloc(&ctr, 64 * 16 * 8)); CK(hipMalloc(&out, 1024 * 8));
  for (uint32_t waves : {131072u, 1048576u}) {
    run<0>(ctr, out, waves, 64, "returning atomicAdd, one address, one per wave");
    run<1>(ctr, out, waves, 64, "returning, 64 addresses in 4 lines (pool cursors before)");
    run<2>(ctr, out, waves, 64, "returning, 64 addresses on 64 lines (pool cursors now)");
    run<3>(ctr, out, waves, 64, "not returning, one address");
    run<0>(ctr, out, waves, 1024, "returning, one address, one per wave, 1024-thread blocks");
    run<4>(ctr, out, waves, 1024, "returning, one address, ONE per 1024-thread block (block_append)");
  }
  return 0;
}
