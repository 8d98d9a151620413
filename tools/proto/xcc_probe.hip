// Which XCD does a workgroup run on, and do the XCDs' core-clock counters share an origin?  hipcc --offload-arch=gfx950 -o /tmp/xcc_probe tools/proto/xcc_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* id, unsigned long long* cyc, unsigned long long* wall) {
  id[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20);
  cyc[blockIdx.x] = __builtin_amdgcn_s_memtime();
  wall[blockIdx.x] = __builtin_readcyclecounter();
}
int main() {
  unsigned* id; unsigned long long *c, *w;
  hipMalloc(&id, 64 * 4); hipMalloc(&c, 64 * 8); hipMalloc(&w, 64 * 8);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k, dim3(16), dim3(1), 0, 0, id, c, w);
    unsigned h[16]; unsigned long long hc[16], hw[16];
    hipMemcpy(h, id, 64, hipMemcpyDeviceToHost); hipMemcpy(hc, c, 128, hipMemcpyDeviceToHost); hipMemcpy(hw, w, 128, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; ++i) printf("wg %2d xcc_reg 0x%x  s_memtime %llu  diff_to_wg0 %lld\n", i, h[i], hc[i], (long long)(hc[i] - hc[0]));
  }
  return 0;
}
