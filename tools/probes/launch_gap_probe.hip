// What does the chip do BETWEEN two dependent kernels of one stream, un-traced?  Every workgroup stamps the 100 MHz
// s_memrealtime counter when its first wave starts and when it ends into its OWN slot (plain stores: atomics on one
// address serialise at ~12 ns each and 256 simultaneous ones measured as a 3.7 us "gap"); the host takes the earliest
// start and the latest end per launch.  gap(i) = first start of launch i+1 - last end of launch i: dispatch + end-of-kernel
// release + workgroup set-up, with no profiler in the way.  Swept over the footprint of the workgroups (threads, LDS
// bytes, VGPRs do not matter for a spin kernel), the grid, the kernel's length and whether it leaves written lines in L2
// (write-back stores) or not (write-through / no stores).
//   hipcc --offload-arch=gfx950 -O3 -o launch_gap_probe launch_gap_probe.hip && ./launch_gap_probe
// (HIP_FORCE_DEV_KERNARG=0/1 in the environment selects where the kernel arguments live.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

struct Stamp { unsigned long long start, end; };

// spin for `ticks` of the 100 MHz counter; mode 0 = no stores, 1 = write-back stores, 2 = non-temporal, 3 = sc1 (write-through), 4 = sc0 sc1
__global__ void spin_kernel(Stamp* st, int launch, int ticks, float* out, int floats_per_wg, int mode) {
  extern __shared__ float lds[];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  Stamp* mine = st + (size_t)launch * gridDim.x + blockIdx.x;
  if (floats_per_wg > 0 && mode != 0) {
    float* dst = out + (size_t)blockIdx.x * floats_per_wg;
    for (int i = threadIdx.x * 4; i < floats_per_wg; i += blockDim.x * 4) {
      typedef float f4 __attribute__((ext_vector_type(4)));
      const f4 v = {(float)i, 1.f, 2.f, 3.f};
      if (mode == 2) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(dst + i));
      else if (mode == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst + i), "v"(v) : "memory");
      else if (mode == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(dst + i), "v"(v) : "memory");
      else *reinterpret_cast<f4*>(dst + i) = v;
    }
  }
  if (threadIdx.x == 0) lds[0] = 1.f;
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) {}
  __syncthreads();
  if (threadIdx.x == 0) { mine->start = t0; mine->end = __builtin_amdgcn_s_memrealtime(); }
}

int main() {
  const int NL = 64;
  Stamp* st;
  float* out;
  const int MAXG = 4096;
  hipMalloc(&st, (size_t)NL * MAXG * sizeof(Stamp));
  const size_t out_floats = (size_t)64 << 20;       // 256 MB
  hipMalloc(&out, out_floats * sizeof(float));
  std::vector<Stamp> h((size_t)NL * MAXG);
  struct Cfg { const char* name; int grid, threads, lds, us, mb, mode; };
  const Cfg cfgs[] = {
      {"tiny: 1 wg x 64 thr, 2 us", 1, 64, 0, 2, 0, 0},
      {"256 wg x 256 thr, no LDS, 20 us", 256, 256, 0, 20, 0, 0},
      {"2048 wg x 256 thr, no LDS, 5 us each", 2048, 256, 0, 5, 0, 0},
      {"256 wg x 512 thr, 144 KiB LDS, 20 us (ring GEMM footprint)", 256, 512, 144 * 1024, 20, 0, 0},
      {"256 wg x 512 thr, 128 KiB LDS, 20 us (phased GEMM footprint)", 256, 512, 128 * 1024, 20, 0, 0},
      {"468 wg x 512 thr, 128 KiB LDS, 20 us each (two rounds)", 468, 512, 128 * 1024, 20, 0, 0},
      {"256 wg x 512 thr, 128 KiB LDS, 20 us, 60 MB write-back stores", 256, 512, 128 * 1024, 20, 60, 1},
      {"256 wg x 512 thr, 128 KiB LDS, 20 us, 60 MB non-temporal stores", 256, 512, 128 * 1024, 20, 60, 2},
      {"256 wg x 512 thr, 128 KiB LDS, 20 us, 120 MB write-back stores", 256, 512, 128 * 1024, 20, 120, 1},
      {"256 wg x 512 thr, 128 KiB LDS, 20 us, 60 MB sc1 (write-through) stores", 256, 512, 128 * 1024, 20, 60, 3},
      {"256 wg x 512 thr, 128 KiB LDS, 20 us, 60 MB sc0 sc1 stores", 256, 512, 128 * 1024, 20, 60, 4},
      {"256 wg x 512 thr, 128 KiB LDS, 20 us, 1 MB write-back stores", 256, 512, 128 * 1024, 20, 1, 1},
      {"256 wg x 256 thr, no LDS, 20 us, 1 MB sc1 stores", 256, 256, 0, 20, 1, 3},
      {"2459 wg x 256 thr, no LDS, 2 us each, 30 MB write-back stores (LayerNorm-like)", 2459, 256, 0, 2, 30, 1},
  };
  hipFuncSetAttribute((const void*)spin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  printf("%-84s %8s %8s %8s\n", "configuration (chain of 64 launches of the same kernel)", "gap med", "gap min", "gap max");
  for (const Cfg& c : cfgs) {
    for (int rep = 0; rep < 2; ++rep) {
      hipMemset(st, 0, (size_t)NL * MAXG * sizeof(Stamp));
      const int fpw = c.mb ? (int)((size_t)c.mb * 1024 * 1024 / 4 / c.grid) / 4 * 4 : 0;
      for (int i = 0; i < NL; ++i)
        hipLaunchKernelGGL(spin_kernel, dim3(c.grid), dim3(c.threads), c.lds > 0 ? c.lds : 16, 0, st, i, c.us * 100, out, fpw,
                           c.mode);
      hipDeviceSynchronize();
      hipMemcpy(h.data(), st, (size_t)NL * c.grid * sizeof(Stamp), hipMemcpyDeviceToHost);
      if (!rep) continue;
      std::vector<unsigned long long> first(NL, ~0ull), last(NL, 0);
      for (int i = 0; i < NL; ++i)
        for (int w = 0; w < c.grid; ++w) {
          const Stamp& x = h[(size_t)i * c.grid + w];
          first[i] = std::min(first[i], x.start);
          last[i] = std::max(last[i], x.end);
        }
      std::vector<double> g;
      for (int i = 8; i + 1 < NL; ++i) g.push_back(((double)first[i + 1] - (double)last[i]) / 100.0);
      std::sort(g.begin(), g.end());
      printf("%-84s %8.2f %8.2f %8.2f\n", c.name, g[g.size() / 2], g.front(), g.back());
    }
  }
  return 0;
}
