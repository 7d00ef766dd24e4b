// Micro-benchmark (round 6): what does ONE CU sustain on the operand path of the 256x128 GEMM kernel, with no MFMA in the way?
//   mode 0  global_load_lds_dwordx4 (LDS-DMA), 6 x 1-KB pieces per wave and iteration into a 3-stage ring, vmcnt(12) (two tiles in flight)
//   mode 1  the same + the 16 ds_read_b128 per wave and iteration of the kernel's R phase
//   mode 2  global_load_dwordx4 into VGPRs + ds_write_b128 of the tile requested two iterations earlier (register-staged ring of 3)
//   mode 3  mode 2 + the 16 ds_read_b128
//   mode 4  global_load_dwordx4 into VGPRs only (consumed by an xor)
//   mode 5  only the 16 ds_read_b128
// 512 threads, 144 KB of LDS (one block per CU), `iters` iterations of one 48-KB "k-tile" per block; source = a private window of `win` bytes per
// block walked cyclically (48 KB: L1/L2-hot, 3 MB: L2 / Infinity-Cache, larger: HBM).  Prints cycles per iteration and B/clk/CU.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_dma_rate_probe.hip -o tools/bin/lds_dma_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int STAGE = 49152;

template <int MODE>
__global__ __launch_bounds__(512) void probe(const char* src, size_t win, int iters, unsigned long long* out, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* base = src + (size_t)blockIdx.x * win;
  const size_t ntile = win / STAGE;
  u32x4 acc = {0, 0, 0, 0};
  u32x4 regs[3][6];
  for (int s = 0; s < 3; ++s) for (int i = 0; i < 6; ++i) regs[s][i] = u32x4{0, 0, 0, 0};
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  auto issue = [&](int it, int st) {
    const char* tb = base + (size_t)(it % ntile) * STAGE;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const char* g = tb + (i * 8 + wave) * 1024 + lane * 16;
      if constexpr (MODE == 0 || MODE == 1) __builtin_amdgcn_global_load_lds((gptr_t*)g, (lptr_t*)(smem + st * STAGE + (i * 8 + wave) * 1024), 16, 0, 0);
      else if constexpr (MODE != 5) regs[st][i] = *reinterpret_cast<const u32x4*>(g);
    }
  };
  issue(0, 0); issue(1, 1);
  static_assert(true, "");
#pragma unroll 1
  for (int it0 = 0; it0 < iters; it0 += 3) {
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int it = it0 + s;
      constexpr int st = 0;
      issue(it + 2, (s + 2) % 3);
      if constexpr (MODE == 0 || MODE == 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      if constexpr (MODE == 2 || MODE == 3) {
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          asm volatile("" : "+v"(regs[s][i]));
          *reinterpret_cast<u32x4*>(smem + s * STAGE + (i * 8 + wave) * 1024 + lane * 16) = regs[s][i];
        }
      }
      if constexpr (MODE == 4) {
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 6; ++i) { asm volatile("" : "+v"(regs[s][i])); acc ^= regs[s][i]; }
      }
      if constexpr (MODE == 1 || MODE == 3 || MODE == 5) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(smem + s * STAGE + ((r * 8 + wave) & 47) * 1024 + (lane ^ (r & 7)) * 16);
          acc ^= v;
        }
      }
      __builtin_amdgcn_s_barrier();
      (void)st;
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (acc.x == 0x12345678u && acc.y == 77u) sink[0] = acc.z + acc.w;
}

template <int MODE>
void run(const char* name, const char* src, size_t win, int blocks, int iters, unsigned long long* dout, unsigned* sink) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * STAGE);
  std::vector<unsigned long long> h(blocks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double best_us = 1e30; double cyc = 0;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(512), 3 * STAGE, 0, src, win, iters, dout, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms * 1e3 < best_us) {
      best_us = ms * 1e3;
      hipMemcpy(h.data(), dout, blocks * 8, hipMemcpyDeviceToHost);
      std::sort(h.begin(), h.end()); cyc = (double)h[blocks / 2];
    }
  }
  const double per_it = cyc / iters;     // s_memtime / readcyclecounter ticks at 100 MHz on gfx950? print both views
  printf("%-34s win %8zu KB blocks %3d: %8.1f us, %7.1f ticks/iter (median block), %6.2f GB/s per CU, chip %6.2f TB/s\n", name, win / 1024, blocks, best_us,
         per_it, 49152.0 * iters / best_us / 1e3, 49152.0 * iters * blocks / best_us / 1e6);
}

int main(int argc, char** argv) {
  const int iters = 3 * 400;
  const int blocks = argc > 1 ? atoi(argv[1]) : 256;
  const size_t wins[3] = {49152, (size_t)49152 * 64, (size_t)49152 * 512};
  char* src; hipMalloc(&src, wins[2] * blocks + (1 << 20)); hipMemset(src, 1, wins[2] * blocks);
  unsigned long long* dout; hipMalloc(&dout, blocks * 8);
  unsigned* sink; hipMalloc(&sink, 64);
  for (size_t win : wins) {
    run<0>("0 LDS-DMA", src, win, blocks, iters, dout, sink);
    run<1>("1 LDS-DMA + 16 ds_read_b128", src, win, blocks, iters, dout, sink);
    run<2>("2 VGPR + ds_write_b128", src, win, blocks, iters, dout, sink);
    run<3>("3 VGPR + ds_write + 16 ds_read", src, win, blocks, iters, dout, sink);
    run<4>("4 VGPR only", src, win, blocks, iters, dout, sink);
    run<5>("5 16 ds_read_b128 only", src, win, blocks, iters, dout, sink);
  }
  return 0;
}
