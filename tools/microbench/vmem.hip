// vmem.hip -- how many vector-memory INSTRUCTIONS a CU can retire (round 4: k_seed_pg and k_chain did not get faster when VALU work, gathers'
// cache lines or dependent round trips were taken out of them; their time tracks their VMEM instruction count).
// Every wave issues `iters` x 8 independent loads from a buffer that stays in the L1 / L2, in one of these shapes:
//   0 dword, coalesced (lane i -> word base + i)         1 dword, all lanes the same word            2 dword gather (lane -> random word of 64 KB)
//   3 dwordx4, coalesced                                  4 dword from LDS (ds_read_b32), coalesced   5 scalar load (s_load_dword, uniform)
//   6 dword, coalesced, only 16 lanes active              7 / 9 gather inside 4 / 16 KB
// Prints cycles per load instruction per CU at full occupancy (8 waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256, 2) k_vmem(const uint32_t* __restrict__ buf, const uint32_t* __restrict__ idx, uint32_t iters, uint32_t* out) {
  __shared__ uint32_t lds[4096];
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  for (uint32_t i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = buf[i];
  __syncthreads();
  uint32_t acc = 0;
  const uint32_t base = (blockIdx.x * 4u + wv) * 64u & 8191u;
  const uint32_t gi = idx[(blockIdx.x * 256u + threadIdx.x) & 65535u];
  for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const uint32_t o = (it * 8u + (uint32_t)q) * 64u;
      if (MODE == 0) acc += buf[(base + o + lane) & 16383u];
      if (MODE == 1) acc += buf[(base + o) & 16383u];
      if (MODE == 2) acc += buf[(gi + o * 7u) & 16383u];
      if (MODE == 3) { const uint4 v = reinterpret_cast<const uint4*>(buf)[((base + o) / 4u + lane) & 4095u]; acc += v.x + v.y + v.z + v.w; }
      if (MODE == 4) acc += lds[(o + lane) & 4095u];
      if (MODE == 5) acc += __builtin_nontemporal_load(&buf[(blockIdx.x * 64u + o) & 16383u]);
      if (MODE == 6) { if (lane < 16) acc += buf[(base + o + lane) & 16383u]; }
      if (MODE == 7) acc += buf[(gi + o * 7u) & 1023u];                                    // gather inside 4 KB: L1-resident
      if (MODE == 9) acc += buf[(gi + o * 7u) & 4095u];                                    // gather inside 16 KB
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE> void run(const uint32_t* buf, const uint32_t* idx, uint32_t* out, const char* what) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const uint32_t iters = 2000, blocks = 256 * 8;                    // 8 blocks of 4 waves per CU = 32 waves per CU, one round
  hipLaunchKernelGGL(k_vmem<MODE>, dim3(blocks), dim3(256), 0, 0, buf, idx, 10u, out);
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k_vmem<MODE>, dim3(blocks), dim3(256), 0, 0, buf, idx, iters, out);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  const double instr_per_cu = (double)iters * 8 * 32;                // load instructions a CU retires
  printf("%-58s %8.3f ms  %6.2f ns per load instruction per CU (= %5.1f cycles at 2.1 GHz)\n", what, ms, ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.1);
}

int main() {
  CK(hipSetDevice(0));
  std::vector<uint32_t> h(16384), hi(65536);
  for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)i * 2654435761u;
  for (size_t i = 0; i < hi.size(); i++) hi[i] = (uint32_t)(i * 40503u + (i >> 3) * 977u) & 16383u;
  uint32_t *buf, *idx, *out;
  CK(hipMalloc(&buf, h.size() * 4)); CK(hipMalloc(&idx, hi.size() * 4)); CK(hipMalloc(&out, 64));
  CK(hipMemcpy(buf, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(idx, hi.data(), hi.size() * 4, hipMemcpyHostToDevice));
  run<0>(buf, idx, out, "global dword, coalesced");
  run<1>(buf, idx, out, "global dword, all lanes one word");
  run<2>(buf, idx, out, "global dword, gather over 64 KB");
  run<3>(buf, idx, out, "global dwordx4, coalesced");
  run<4>(buf, idx, out, "LDS dword (ds_read_b32), coalesced");
  run<5>(buf, idx, out, "global dword, uniform address (compiler's choice of s_load / vector)");
  run<6>(buf, idx, out, "global dword, coalesced, 16 of 64 lanes active");
  run<7>(buf, idx, out, "global dword, gather inside 4 KB (L1-resident)");
  run<9>(buf, idx, out, "global dword, gather inside 16 KB");
  return 0;
}
