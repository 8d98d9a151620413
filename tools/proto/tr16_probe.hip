// Hardware probe (not product code): semantics of ds_read_b64_tr_b16 and the operand layout of
// v_mfma_f32_32x32x16_bf16 on gfx950, plus LDS cycles of three lane-address patterns.
//   hipcc --offload-arch=gfx950 -O3 tools/proto/tr16_probe.hip -o tools/proto/tr16_probe && tools/proto/tr16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define LDSP(p) ((s16x4 __attribute__((address_space(3)))*)(p))

__global__ void tr_sem(unsigned short* out, const int* byte_addr) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDSP(reinterpret_cast<char*>(lds) + byte_addr[threadIdx.x]));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}

__global__ void mfma_layout(float* C, const unsigned short* A, const unsigned short* B) {   // A[32][16], B[16][32] bf16 bits
  const int l = threadIdx.x, row = l & 31, h = l >> 5;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = __builtin_bit_cast(__bf16, A[row * 16 + 8 * h + j]);
    b[j] = __builtin_bit_cast(__bf16, B[(8 * h + j) * 32 + row]);
  }
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + row] = acc[r];
}

// LDS cycles of N tr reads per wave with a given lane-address pattern, 8 waves per workgroup
__global__ __launch_bounds__(512) void tr_time(long long* cyc, int pattern, int* sink) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[32768];
  for (int i = threadIdx.x; i < 32768; i += 512) lds[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x & 63, g = l >> 4, i = l & 15, w = threadIdx.x >> 6;
  int off;
  if (pattern == 0) off = 8 * l;                                             // linear 512 B
  else if (pattern == 1) off = (g & 1) * 512 + (g >> 1) * 256 + 8 * i;       // 32x32 mapping on the natural [16 key][16 feat] subtiles
  else if (pattern == 2) off = (g >> 1) * 512 + (g & 1) * 128 + 8 * i;       // permuted image: halves read 256 contiguous bytes
  else off = g * 1024 + 8 * i;                                                // 4 groups on the same banks
  char* base = reinterpret_cast<char*>(lds) + w * 4096 + off;
  int acc = 0;
  __syncthreads();
  const long long t0 = clock64();
#pragma unroll 1
  for (int it = 0; it < 256; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDSP(base + (u & 1) * 2048 + (u >> 1) * 0));
      acc += v[0] + v[3];
    }
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  if (acc == 123456789) sink[0] = acc;
}

int main() {
  unsigned short* d_out; int* d_addr;
  hipMalloc(&d_out, 64 * 4 * 2); hipMalloc(&d_addr, 64 * 4);
  int h_addr[64]; unsigned short h_out[256];
  for (int test = 0; test < 3; ++test) {
    for (int l = 0; l < 64; ++l) {
      if (test == 0) h_addr[l] = 8 * l;
      else if (test == 1) h_addr[l] = 8 * (63 - l);
      else h_addr[l] = (l >> 4) * 512 + ((l & 15) >> 2) * 64 + (l & 3) * 8;    // key rows 64 B apart
    }
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(tr_sem, dim3(1), dim3(64), 0, 0, d_out, d_addr);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    // hypothesis: result(lane l = 16 g + i, elem j) = element (i & 3) of the 8 bytes addressed by lane 16 g + 4 j + (i >> 2)
    int bad = 0;
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 4; ++j) {
        const int g = l >> 4, i = l & 15, src = 16 * g + 4 * j + (i >> 2);
        const int expect = h_addr[src] / 2 + (i & 3);
        if (h_out[l * 4 + j] != expect) { if (bad < 8) printf("  test %d lane %d elem %d: got %d expect %d\n", test, l, j, h_out[l * 4 + j], expect); ++bad; }
      }
    printf("tr16 semantics test %d: %s (%d mismatches)\n", test, bad ? "MISMATCH" : "ok", bad);
    if (test == 0) { printf("  lane0: %d %d %d %d  lane1: %d %d %d %d lane16: %d %d %d %d\n", h_out[0], h_out[1], h_out[2], h_out[3], h_out[4], h_out[5], h_out[6], h_out[7], h_out[64], h_out[65], h_out[66], h_out[67]); }
  }
  // MFMA layout
  std::vector<unsigned short> A(32 * 16), B(16 * 32); std::vector<float> Af(32 * 16), Bf(16 * 32), Cref(32 * 32, 0.f), C(32 * 32);
  srand(1);
  auto tobf = [](float v) { union { float f; unsigned u; } c; c.f = v; return (unsigned short)(c.u >> 16); };
  for (int i = 0; i < 512; ++i) { Af[i] = (float)(rand() % 17 - 8); Bf[i] = (float)(rand() % 13 - 6); A[i] = tobf(Af[i]); B[i] = tobf(Bf[i]); }
  for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) for (int k = 0; k < 16; ++k) Cref[m * 32 + n] += Af[m * 16 + k] * Bf[k * 32 + n];
  unsigned short *dA, *dB; float* dC;
  hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dC, 4096);
  hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(mfma_layout, dim3(1), dim3(64), 0, 0, dC, dA, dB);
  hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 1024; ++i) bad += C[i] != Cref[i];
  printf("mfma_f32_32x32x16_bf16 layout: %s (%d mismatches)\n", bad ? "MISMATCH" : "ok", bad);
  // timing
  long long* dcyc; int* dsink; hipMalloc(&dcyc, 8 * 256); hipMalloc(&dsink, 4);
  for (int p = 0; p < 4; ++p) {
    hipLaunchKernelGGL(tr_time, dim3(256), dim3(512), 0, 0, dcyc, p, dsink);
    hipLaunchKernelGGL(tr_time, dim3(256), dim3(512), 0, 0, dcyc, p, dsink);
    long long h[256]; hipMemcpy(h, dcyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 256; ++i) s += h[i];
    printf("tr read pattern %d: %.1f s_memtime ticks per wave-instruction per CU (8 waves x 2048 reads)\n", p, s / 256 / (8.0 * 2048));
  }
  return 0;
}
