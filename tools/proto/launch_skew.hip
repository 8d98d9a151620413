// Launch-skew probe (measurement helper, not product): how long does the dispatcher take to get the eight waves of each of 256
// workgroups onto the chip, as a function of the wave's register count, the workgroup's LDS size and its thread count?
//   hipcc --offload-arch=gfx950 -O3 -o launch_skew launch_skew.hip && ./launch_skew
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int NREG, int THREADS>
__global__ __launch_bounds__(THREADS) void probe_k(unsigned long long* out, int spin) {
  const unsigned long long t0 = __builtin_readcyclecounter();
  extern __shared__ float lds[];
  float r[NREG];
#pragma unroll
  for (int i = 0; i < NREG; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(r[i]) : "v"((float)threadIdx.x + i));
  if (spin) lds[threadIdx.x] = r[0];
  // stay resident for a while (so that all workgroups are on the chip together, like the real kernel)
  unsigned long long t = t0;
  while (t - t0 < 60000ull) t = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NREG; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s) : "v"(r[i]));
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    out[((size_t)blockIdx.x * (THREADS / 64) + wave) * 2] = t0;
    out[((size_t)blockIdx.x * (THREADS / 64) + wave) * 2 + 1] = (unsigned long long)(s != 12345.f);
  }
}

template <int NREG, int THREADS>
static void run(const char* name, int lds_bytes, int nwg) {
  const int waves = THREADS / 64;
  unsigned long long* d;
  hipMalloc(&d, sizeof(unsigned long long) * nwg * waves * 2);
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe_k<NREG, THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  std::vector<unsigned long long> h(nwg * waves * 2);
  std::vector<double> skews, totals;
  for (int rep = 0; rep < 6; ++rep) {
    hipLaunchKernelGGL((probe_k<NREG, THREADS>), dim3(nwg), dim3(THREADS), lds_bytes, 0, d, lds_bytes > 0);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> sk;
    unsigned long long lo = ~0ull, hi = 0;
    for (int b = 0; b < nwg; ++b) {
      unsigned long long a = ~0ull, z = 0;
      for (int w = 0; w < waves; ++w) { a = std::min(a, h[(b * waves + w) * 2]); z = std::max(z, h[(b * waves + w) * 2]); }
      sk.push_back((double)(z - a)); lo = std::min(lo, a); hi = std::max(hi, z);
    }
    std::sort(sk.begin(), sk.end());
    if (rep >= 2) { skews.push_back(sk[sk.size() / 2]); totals.push_back((double)(hi - lo)); }
  }
  std::sort(skews.begin(), skews.end()); std::sort(totals.begin(), totals.end());
  printf("%-44s wg=%d lds=%6d  median in-workgroup entry skew %8.0f cycles   first->last wave entry over the grid %8.0f cycles\n", name, nwg, lds_bytes,
         skews[skews.size() / 2], totals[totals.size() / 2]);
  hipFree(d);
}

int main() {
  run<8, 512>("512 threads,   8 regs", 0, 256);
  run<8, 512>("512 threads,   8 regs", 100 * 1024, 256);
  run<100, 512>("512 threads, 100 regs", 100 * 1024, 256);
  run<200, 512>("512 threads, 200 regs", 0, 256);
  run<200, 512>("512 threads, 200 regs", 100 * 1024, 256);
  run<200, 512>("512 threads, 200 regs", 140 * 1024, 256);
  run<200, 256>("256 threads, 200 regs", 100 * 1024, 256);
  run<200, 256>("256 threads, 200 regs (2 wg / CU)", 64 * 1024, 512);
  run<100, 1024>("1024 threads (16 waves), 100 regs", 100 * 1024, 256);
  return 0;
}
