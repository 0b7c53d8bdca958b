// TEST INFRASTRUCTURE: checks the emulator itself (tests/test_emu_selftest.py).
//   selftest ok        -> wave/block primitives give the values the hardware gives, prints "ok"
//   selftest diverge   -> a shuffle executed by half a wave must be reported and abort
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

__global__ void k_prims(uint32_t* out, unsigned long long* out64) {
  __shared__ uint32_t s_sum;
  __shared__ uint32_t s_part[4];
  const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t == 0) s_sum = 0;
  __syncthreads();
  uint32_t v = t + 1;
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);          // wave sum of t+1
  if (lane == 0) s_part[wave] = v;
  atomicAdd(&s_sum, 1u);
  __syncthreads();
  const unsigned long long odd = __ballot(lane & 1);
  const int up = __shfl_up((int)lane, 3), down = __shfl_down((int)lane, 5, 16), bc = __shfl((int)t, 7);
  const int dpp = __builtin_amdgcn_update_dpp(-1, (int)lane * 2, 0x138, 0xF, 0xF, false);
  uint32_t* o = out + (size_t)(blockIdx.x * blockDim.x + t) * 8;
  o[0] = s_part[wave]; o[1] = s_sum; o[2] = (uint32_t)up; o[3] = (uint32_t)down; o[4] = (uint32_t)bc; o[5] = (uint32_t)dpp;
  o[6] = (uint32_t)__any(t == 70); o[7] = (uint32_t)__all(t < 256);
  if (t == 0) out64[blockIdx.x] = odd;
  if (t >= 200) return;                                              // finished lanes do not take part any more
  const unsigned long long m2 = __ballot(1);
  if (t == 199) out64[gridDim.x + blockIdx.x] = m2;
}

__global__ void k_diverge(uint32_t* out) {
  const uint32_t lane = threadIdx.x & 63;
  uint32_t v = lane;
  if (lane < 32) v = __shfl(v, 40);      // lane 40 is not active here: the hardware returns 0, the emulator must complain
  else v = __ballot(1) != 0;
  out[threadIdx.x] = v;
}

int main(int argc, char** argv) {
  const bool diverge = argc > 1 && !strcmp(argv[1], "diverge");
  uint32_t* d; unsigned long long* d64;
  const int B = 5, T = 256;
  hipMalloc(&d, (size_t)B * T * 8 * 4); hipMalloc(&d64, 2 * B * 8);
  if (diverge) { hipLaunchKernelGGL(k_diverge, dim3(1), dim3(64), 0, 0, d); printf("not detected\n"); return 1; }
  hipLaunchKernelGGL(k_prims, dim3(B), dim3(T), 0, 0, d, d64);
  std::vector<uint32_t> h((size_t)B * T * 8); std::vector<unsigned long long> h64(2 * B);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(h64.data(), d64, h64.size() * 8, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int b = 0; b < B; b++) {
    for (int t = 0; t < T; t++) {
      const uint32_t* o = &h[((size_t)b * T + t) * 8];
      const int lane = t & 63, wave = t >> 6;
      uint32_t wsum = 0; for (int l = 0; l < 64; l++) wsum += wave * 64 + l + 1;
      const int up = lane >= 3 ? lane - 3 : lane, down = (lane & 15) + 5 <= 15 ? lane + 5 : lane, bc = wave * 64 + 7;
      const int dpp = lane == 0 ? -1 : (lane - 1) * 2;
      const uint32_t exp[8] = {wsum, (uint32_t)T, (uint32_t)up, (uint32_t)down, (uint32_t)bc, (uint32_t)dpp, (uint32_t)(wave == 1), 1u};
      for (int k = 0; k < 8; k++) if (o[k] != exp[k]) { if (bad++ < 10) printf("block %d thread %d field %d: %u != %u\n", b, t, k, o[k], exp[k]); }
    }
    if (h64[b] != 0xAAAAAAAAAAAAAAAAull) { bad++; printf("ballot %llx\n", h64[b]); }
    if (h64[B + b] != 0xFFull) { bad++; printf("ballot after exit %llx\n", h64[B + b]); }     // wave 3: lanes 192..199 remain
  }
  printf(bad ? "FAILED\n" : "ok\n");
  return bad != 0;
}
