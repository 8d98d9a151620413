// L2 -> CU delivery micro-benchmark for the weight stream of clica::fmlp::mlp_split_k.
//
// The whole-stack kernel has 256 workgroups (one per CU, 8 waves) that ALL stream the same fragment-order weight copy
// (6 B per weight, ~5.5 MB for the n = 10 stack) in the same order at the same time: per k-iteration a wave requests
// 3 pieces x NC column blocks x 1 KB (global_load_dwordx4, 16 B per lane), one k-iteration of requests in flight behind
// the 72 MFMAs of the current one.  DESIGN r3 claimed the eight L2s deliver only ~17-19 B/clk/CU to that pattern
// (~11.7 TB/s) against the guide's ~34.5 TB/s.  This program isolates the pattern:
//
//   order 0  same        every workgroup walks (layer, ki, piece, column block) in the kernel's order
//   order 1  rot-wg      k-iteration start rotated by blockIdx.x
//   order 2  rot-xcdl    k-iteration start rotated by the XCD-local workgroup index (blockIdx.x >> 3): the 32 CUs of an
//                        XCD are on different k-iterations (different L2 channels) at any moment
//   order 3  rot-xcd     rotated by the XCD id (blockIdx.x & 7) only -- control: the CUs of one XCD still move together
//   order 4  rot-cb      column-block assignment rotated by the XCD-local index (same k order)
//   order 5  rot-both    2 + 4
//   order 6  private     every workgroup streams its OWN 192 KB slice (no sharing; L2 resident: 32 x 192 KB = 6 MB ... 3 MB/XCD at 96 KB)
//   order 7  l1          every wave re-reads the same 12 KB (TCP-resident upper bound)
//   mfma  0  loads only (values xor-ed into a sink)      1  the kernel's 72 x v_mfma_f32_16x16x32_bf16 per k-iteration on the loaded fragments
//   path  0  global_load_dwordx4 into VGPRs              1  global_load_lds_dwordx4 (the DMA path of wgrad_split_k; loads only)
//   depth    k-iterations of requests in flight (1 = the kernel's ping-pong, 2 = twice that)
//
// Output: one line per variant with the median launch time, aggregate L2 -> CU rate, the average shader clock of the launch
// (s_memtime against the 100 MHz s_memrealtime) and bytes per clock per CU.
//   hipcc --offload-arch=gfx950 -O3 -o tools/l2_stream_bench tools/l2_stream_bench.hip && tools/l2_stream_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int WAVES = 8, CBW = 4, MAXL = 8;
struct LayerD { int ncb, kiters; long long off, ent; };     // 16-byte entries
struct Params {
  const u32x4* buf; int L; LayerD ly[MAXL];
  unsigned long long* stamps;     // per workgroup: {cycles, realtime ticks}
  u32x4* sink; long long private_entries;
  int repeat;                      // the layer list is walked this many times (--repeat: sustained-load clock)
};

template <int ORDER, int MFMA, int DEPTH, int NC>
__device__ __forceinline__ void layer_body(const Params& P, const LayerD ly, const int wave, const int lane, const int rot, const int crot,
                                           f32x4 (&acc)[3][CBW], const u32x4 (&xf)[3][3], u32x4& sink) {
  const u32x4* w0 = P.buf + ly.off;
  auto addr = [&](int p, int c, int ki) -> const u32x4* {
    if (ORDER == 6) {        // private slice, walked linearly
      const long long e = ((long long)((ki * 3 + p) * CBW + c) * WAVES + wave) * 64 + lane;
      return P.buf + (long long)blockIdx.x * P.private_entries + (e % P.private_entries);
    }
    if (ORDER == 7) return P.buf + ((p * CBW + c) * 64 + lane);
    int cb = wave + c * WAVES;
    if (ORDER == 4 || ORDER == 5) cb = (cb + crot) % ly.ncb;
    const int kk = (ORDER >= 1 && ORDER <= 3) || ORDER == 5 ? (ki + rot) % ly.kiters : ki;
    return w0 + p * ly.ent + ((long long)cb * ly.kiters + kk) * 64 + lane;
  };
  u32x4 w[DEPTH + 1][3][NC];
  auto fetch = [&](u32x4 (&d)[3][NC], int ki) {
    const int kk = ki < ly.kiters ? ki : ly.kiters - 1;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int c = 0; c < NC; ++c) d[p][c] = *addr(p, c, kk);
  };
  auto consume = [&](u32x4 (&d)[3][NC]) {
    if (MFMA) {
      constexpr int PW[6] = {2, 0, 1, 1, 0, 0}, PX[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
          for (int r = 0; r < 3; ++r)
            acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, d[PW[t]][c]), __builtin_bit_cast(bf16x8, xf[PX[t]][r]),
                                                                acc[r][c], 0, 0, 0);
    } else {
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int c = 0; c < NC; ++c) sink ^= d[p][c];
    }
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) fetch(w[d], d);
#pragma unroll 1
  for (int ki = 0; ki < ly.kiters; ki += DEPTH + 1) {
#pragma unroll
    for (int u = 0; u <= DEPTH; ++u) {
      if (ki + u < ly.kiters) {                // wave-uniform
        fetch(w[(u + DEPTH) % (DEPTH + 1)], ki + u + DEPTH);
        __builtin_amdgcn_sched_barrier(0);
        consume(w[u]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

template <int ORDER, int MFMA, int DEPTH>
__global__ __launch_bounds__(512) void stream_k(Params P) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xl = blockIdx.x >> 3;              // XCD-local index (workgroups are dealt round-robin over the eight XCDs)
  const int rot = ORDER == 1 ? (int)blockIdx.x : (ORDER == 2 || ORDER == 5) ? xl : ORDER == 3 ? (int)(blockIdx.x & 7) : 0;
  const int crot = (ORDER == 4 || ORDER == 5) ? xl : 0;
  const unsigned long long c0 = __builtin_readcyclecounter();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  u32x4 sink = {0u, 0u, 0u, 0u};
  f32x4 acc[3][CBW];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < CBW; ++c) acc[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  u32x4 xf[3][3];                               // stand-in activation fragments (registers; the real kernel reads them from LDS)
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int r = 0; r < 3; ++r) xf[p][r] = (u32x4){0x3F803F80u + lane, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u + r + p};

#pragma unroll 1
  for (int ll = 0; ll < P.L * P.repeat; ++ll) {
    const int l = ll % P.L;
    const LayerD ly = P.ly[l];
    int nc = (ly.ncb - wave + WAVES - 1) / WAVES;
    nc = nc < 0 ? 0 : (nc > CBW ? CBW : nc);
    switch (nc) {
      case 4: layer_body<ORDER, MFMA, DEPTH, 4>(P, ly, wave, lane, rot, crot, acc, xf, sink); break;
      case 3: layer_body<ORDER, MFMA, DEPTH, 3>(P, ly, wave, lane, rot, crot, acc, xf, sink); break;
      case 2: layer_body<ORDER, MFMA, DEPTH, 2>(P, ly, wave, lane, rot, crot, acc, xf, sink); break;
      case 1: layer_body<ORDER, MFMA, DEPTH, 1>(P, ly, wave, lane, rot, crot, acc, xf, sink); break;
      default: break;
    }
    __syncthreads();                             // the kernel's layer barrier
  }
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < CBW; ++c) { sink.x ^= __float_as_uint(acc[r][c][0] + acc[r][c][3]); sink.y ^= __float_as_uint(acc[r][c][1] + acc[r][c][2]); }
  if (sink.x == 0x12345u && sink.y == 0x777u) P.sink[threadIdx.x] = sink;      // never true: keeps the loads alive
  const unsigned long long c1 = __builtin_readcyclecounter();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { P.stamps[2 * blockIdx.x] = c1 - c0; P.stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

__device__ __forceinline__ void dma_1k(const char* lane_src, unsigned lds_off) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(lane_src), "s"(lds_off) : "memory", "m0");
#pragma clang diagnostic pop
}
template <int N> __device__ __forceinline__ void wait_vm() {   // s_waitcnt vmcnt(N) only (gfx9 encoding)
  __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}
// LDS-DMA path (loads only): the same addresses, 1 KB per wave instruction straight into LDS, DEPTH k-iterations in flight
template <int ORDER, int DEPTH>
__global__ __launch_bounds__(512) void stream_dma_k(Params P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xl = blockIdx.x >> 3;
  const int rot = ORDER == 1 ? (int)blockIdx.x : (ORDER == 2 || ORDER == 5) ? xl : ORDER == 3 ? (int)(blockIdx.x & 7) : 0;
  const int crot = (ORDER == 4 || ORDER == 5) ? xl : 0;
  const unsigned long long c0 = __builtin_readcyclecounter();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned lds_base = (unsigned)(uintptr_t)(lds) + (unsigned)(wave * 12 * 1024);   // one 12 KB landing zone per wave (the data is not read)
#pragma unroll 1
  for (int l = 0; l < P.L; ++l) {
    const LayerD ly = P.ly[l];
    int nc = (ly.ncb - wave + WAVES - 1) / WAVES;
    nc = nc < 0 ? 0 : (nc > CBW ? CBW : nc);
    if (nc == 0) continue;
    const u32x4* w0 = P.buf + ly.off;
    auto fetch = [&](int slot, int ki) {
      const int k0 = ki < ly.kiters ? ki : ly.kiters - 1;
      const int kk = rot ? (k0 + rot) % ly.kiters : k0;
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int c = 0; c < CBW; ++c)
          if (c < nc) {
            int cb = wave + c * WAVES;
            if (crot) cb = (cb + crot) % ly.ncb;
            const u32x4* src = w0 + p * ly.ent + ((long long)cb * ly.kiters + kk) * 64 + lane;
            dma_1k(reinterpret_cast<const char*>(src), lds_base + (unsigned)((p * CBW + c) * 1024)); (void)slot;
          }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) fetch(d, d);
#pragma unroll 1
    for (int ki = 0; ki < ly.kiters; ++ki) {
      fetch((ki + DEPTH) % (DEPTH + 1), ki + DEPTH);
      // wait until only the newest DEPTH k-iterations are outstanding
      switch (DEPTH * 3 * nc) {
        case 3: wait_vm<3>(); break;   case 6: wait_vm<6>(); break;   case 9: wait_vm<9>(); break;
        case 12: wait_vm<12>(); break; case 18: wait_vm<18>(); break; default: wait_vm<24>(); break;
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { P.stamps[2 * blockIdx.x] = c1 - c0; P.stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

struct Result { double us, tbs, ghz, bclkcu, mfma_frac; };

template <typename F>
static Result run(F launch, const Params& P, double bytes_per_wg, double mfma_cycles_per_simd, int wgs, int reps) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  std::vector<float> t;
  std::vector<unsigned long long> st(2 * wgs);
  double ghz = 0;
  for (int i = 0; i < 5; ++i) launch();
  for (int i = 0; i < reps; ++i) {
    CHECK(hipEventRecord(e0, 0)); launch(); CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms * 1e3f);
  }
  CHECK(hipMemcpy(st.data(), P.stamps, st.size() * 8, hipMemcpyDeviceToHost));
  double cyc = 0, rt = 0;
  for (int w = 0; w < wgs; ++w) { cyc += (double)st[2 * w]; rt += (double)st[2 * w + 1]; }
  ghz = cyc / (rt * 10.0);          // ticks of 10 ns
  std::sort(t.begin(), t.end());
  Result r;
  r.us = t[t.size() / 2];
  // in-kernel duration (mean over workgroups) for the rate: launch overhead of a ~100 us kernel is not what is being measured
  const double kern_us = rt / wgs * 0.01;
  r.tbs = bytes_per_wg * wgs / (kern_us * 1e-6) / 1e12;
  r.ghz = ghz;
  r.bclkcu = bytes_per_wg / (cyc / wgs);
  r.mfma_frac = mfma_cycles_per_simd / (cyc / wgs);
  r.us = kern_us;
  CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
  return r;
}

int main(int argc, char** argv) {
  int reps = 20, wgs = 256, repeat = 1;
  bool narrow = false, sustain_only = false;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--reps") && i + 1 < argc) reps = atoi(argv[++i]);
    if (!strcmp(argv[i], "--wgs") && i + 1 < argc) wgs = atoi(argv[++i]);
    if (!strcmp(argv[i], "--stack")) narrow = true;     // the whole n = 10 forward stack instead of 6 wide layers
    if (!strcmp(argv[i], "--repeat") && i + 1 < argc) { repeat = atoi(argv[++i]); sustain_only = true; }   // long kernels: sustained clock
  }
  // layers as (N, K): fragment-order copy per piece = ceil(N/16) * ceil(K/32) KB
  std::vector<std::pair<int, int>> layers;
  if (narrow) layers = {{100, 10}, {500, 100}, {500, 500}, {500, 500}, {500, 500}, {100, 500}, {10, 100}};
  else layers = {{500, 500}, {500, 500}, {500, 500}, {500, 500}, {500, 500}, {500, 500}};
  Params P; memset(&P, 0, sizeof(P));
  long long off = 0;
  double bytes_per_wg = 0, mfma_cyc = 0;
  P.L = (int)layers.size();
  for (int l = 0; l < P.L; ++l) {
    const int ncb = (layers[l].first + 15) / 16, kit = (layers[l].second + 31) / 32;
    P.ly[l] = {ncb, kit, off, (long long)ncb * kit * 64};
    off += 3 * P.ly[l].ent;
    bytes_per_wg += 3.0 * ncb * kit * 1024;
    // MFMA cycles per SIMD: 18 MFMAs of 16 cycles per (column block, k-iteration), two waves per SIMD
    int blocks = 0;
    for (int w = 0; w < WAVES; ++w) { int nc = (ncb - w + WAVES - 1) / WAVES; nc = nc < 0 ? 0 : nc > CBW ? CBW : nc; blocks += nc; }
    mfma_cyc += (double)blocks * kit * 18 * 16 / 4;
  }
  const long long private_entries = 96 * 1024 / 16;       // 96 KB per workgroup: 3 MB per XCD, L2 resident
  const long long total_entries = std::max(off, private_entries * wgs);
  u32x4* buf; CHECK(hipMalloc(&buf, total_entries * 16));
  {
    std::vector<unsigned> h(total_entries * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3F803F80u ^ (unsigned)((i * 2654435761u) & 0x007F007Fu);   // bf16 pairs near 1.0
    CHECK(hipMemcpy(buf, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  }
  P.buf = buf; P.private_entries = private_entries; P.repeat = repeat;
  bytes_per_wg *= repeat; mfma_cyc *= repeat;
  CHECK(hipMalloc(&P.stamps, 2 * wgs * 8));
  CHECK(hipMalloc(&P.sink, 512 * 16));
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  printf("# device %s, %d CUs, %d workgroups x 512 threads, stream %.2f MB per workgroup (%s), %d reps (median)\n", prop.gcnArchName,
         prop.multiProcessorCount, wgs, bytes_per_wg / 1e6, narrow ? "n = 10 forward stack" : "6 layers of 500 x 500", reps);
  printf("# guide: L2 aggregate ~34.5 TB/s = ~56 B/clk/CU at 2.4 GHz; DESIGN r3 claim for this pattern: 17-19 B/clk/CU (~11.7 TB/s)\n");
  printf("%-10s %-5s %-5s %-5s %9s %8s %7s %9s %9s\n", "order", "mfma", "path", "depth", "kern_us", "TB/s", "GHz", "B/clk/CU", "mfma_frac");
  static const char* names[8] = {"same", "rot-wg", "rot-xcdl", "rot-xcd", "rot-cb", "rot-both", "private", "l1"};
  auto report = [&](int order, int mfma, int path, int depth, const Result& r) {
    printf("%-10s %-5d %-5s %-5d %9.1f %8.2f %7.3f %9.2f %9.3f\n", names[order], mfma, path ? "dma" : "vgpr", depth, r.us, r.tbs, r.ghz, r.bclkcu,
           mfma ? r.mfma_frac : 0.0);
    fflush(stdout);
  };
#define RUN(ORDER, MF, DEPTH) do { auto f = [&]() { hipLaunchKernelGGL((stream_k<ORDER, MF, DEPTH>), dim3(wgs), dim3(512), 0, 0, P); }; \
    report(ORDER, MF, 0, DEPTH, run(f, P, bytes_per_wg, mfma_cyc, wgs, reps)); } while (0)
#define RUN_DMA(ORDER, DEPTH) do { const size_t shm = (size_t)WAVES * 12 * 1024; \
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_dma_k<ORDER, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm)); \
    auto f = [&]() { hipLaunchKernelGGL((stream_dma_k<ORDER, DEPTH>), dim3(wgs), dim3(512), shm, 0, P); }; \
    report(ORDER, 0, 1, DEPTH, run(f, P, bytes_per_wg, mfma_cyc, wgs, reps)); } while (0)
  if (sustain_only) {      // the matrix work alone (weights from L1) and with the L2 stream, kernels of several milliseconds
    printf("# --repeat %d: every kernel walks the layer list %d times (sustained load: does the shader clock hold?)\n", repeat, repeat);
    RUN(7, 1, 1); RUN(0, 1, 1); RUN(7, 1, 1); RUN(0, 1, 1);
    return 0;
  }
  // loads only: what the memory system delivers to the pattern
  RUN(0, 0, 1); RUN(1, 0, 1); RUN(2, 0, 1); RUN(3, 0, 1); RUN(4, 0, 1); RUN(5, 0, 1); RUN(6, 0, 1); RUN(7, 0, 1);
  RUN(0, 0, 2); RUN(2, 0, 2); RUN(5, 0, 2); RUN(6, 0, 2); RUN(7, 0, 2);
  // with the kernel's matrix work on the loaded fragments (no LDS panel reads, no epilogue): the k-loop's own ceiling
  RUN(0, 1, 1); RUN(1, 1, 1); RUN(2, 1, 1); RUN(4, 1, 1); RUN(5, 1, 1); RUN(6, 1, 1); RUN(7, 1, 1);
  // the DMA path of wgrad_split_k
  RUN_DMA(0, 1); RUN_DMA(2, 1); RUN_DMA(5, 1); RUN_DMA(0, 2); RUN_DMA(2, 2);
  return 0;
}
