// What does a split-K PAIR exchange cost at the end of a 256x256 tile?  Two workgroups (512 threads, one per CU) hold the
// f32 partial sums of the same output tile (128 accumulator registers per lane each); each keeps one half of the tile and
// needs the partner's partials for it: every workgroup PUBLISHES 64 registers per lane (128 KiB, 16-byte stores, one
// 8 KiB line run per instruction), raises a flag, waits for the partner's flag and READS 128 KiB back.  All pairs do so at
// the same moment (they ran the same K loop), 117 pairs = 30 MB each way.  Per workgroup, 100 MHz s_memrealtime stamps:
//   t0 exchange starts   t1 own payload drained + flag raised   t2 partner's flag seen   t3 partner's payload in registers
// Modes (payload stores / flag / payload loads):
//   0  plain stores, release fence (agent), flag; relaxed sc1 poll, acquire fence, plain loads   (memory-model form)
//   1  sc1 write-through stores, s_waitcnt vmcnt(0), sc1 flag; sc1 poll, sc1 loads                (placement-independent)
//   2  plain stores, s_waitcnt vmcnt(0), sc1 flag; sc1 poll, plain loads -- correct ONLY when both workgroups share an XCD
//      (one L2); the probe counts wrong values, so the cross-XCD pairing shows what stale lines look like
//   3  sc1 stores, vmcnt(0), sc1 flag; plain loads          4  plain stores, vmcnt(0), sc1 flag; sc1 loads (skip this CU's L1)
// Pairings: same-XCD (b, b ^ 8) and cross-XCD (b, b ^ 1); the XCC id of every workgroup is recorded and compared.
//   hipcc --offload-arch=gfx950 -O3 -o splitk_exchange_probe splitk_exchange_probe.hip && ./splitk_exchange_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef float f4 __attribute__((ext_vector_type(4)));
struct Stamp { unsigned long long t0, t1, t2, t3; unsigned xcc, bad; };

__device__ __forceinline__ void store_sc1(f4* p, f4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }   // (nop: the data registers may be rewritten next)
__device__ __forceinline__ f4 load_sc1(const f4* p) {
  f4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ unsigned poll_sc1(const unsigned* p) {
  unsigned v;
  asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void flag_sc1(unsigned* p, unsigned v) { asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }

template <int NV>   // NV 16-byte vectors per lane each way (16 = half of a 128-register accumulator tile)
__global__ __launch_bounds__(512) void exch_kernel(f4* scratch, unsigned* flags, unsigned gen, Stamp* st, int mode, int cross,
                                                   float* sink) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const int partner = cross ? (b ^ 1) : (b ^ 8);
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 0xf;
  f4 v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = f4{(float)(b * 1000 + i), (float)tid, (float)gen, 1.0f};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  f4* mine = scratch + (size_t)b * NV * 512;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (mode == 1 || mode == 3) store_sc1(mine + i * 512 + tid, v[i]);
    else mine[i * 512 + tid] = v[i];
  }
  if (mode == 0) __threadfence();
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) flag_sc1(flags + b, gen);
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (tid == 0) {
    while (poll_sc1(flags + partner) != gen) __builtin_amdgcn_s_sleep(2);
  }
  __syncthreads();
  const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
  if (mode == 0) __atomic_thread_fence(__ATOMIC_ACQUIRE);
  const f4* theirs = scratch + (size_t)partner * NV * 512;
  f4 r[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (mode == 1 || mode == 4) r[i] = load_sc1(theirs + i * 512 + tid);
    else r[i] = theirs[i * 512 + tid];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < NV; ++i) asm volatile("" : "+v"(r[i]));      // (uses stay behind the wait)
  unsigned bad = 0;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    bad += (r[i][0] != (float)(partner * 1000 + i)) || (r[i][1] != (float)tid) || (r[i][2] != (float)gen);
    s += r[i][3];
  }
  const unsigned long long t3 = __builtin_amdgcn_s_memrealtime();
  __shared__ unsigned nbad;
  if (tid == 0) nbad = 0;
  __syncthreads();
  if (bad) atomicAdd(&nbad, bad);
  __syncthreads();
  if (s == 12345.f) sink[b] = s;
  if (tid == 0) {
    Stamp& o = st[b];
    o.t0 = t0; o.t1 = t1; o.t2 = t2; o.t3 = t3; o.xcc = xcc; o.bad = nbad;
  }
}

template <int NV>
static void sweep(f4* scratch, unsigned* flags, Stamp* st, float* sink, unsigned& gen) {
  const int G = 240, NL = 24;
  std::vector<Stamp> h(G);
  const char* mname[5] = {"plain + release fence / acquire + plain loads", "sc1 stores, vmcnt(0), sc1 flag / sc1 loads",
                          "plain stores, vmcnt(0), sc1 flag / plain loads", "sc1 stores, vmcnt(0), sc1 flag / plain loads",
                          "plain stores, vmcnt(0), sc1 flag / sc1 loads"};
  printf("## %d KiB per workgroup each way, %d workgroups x 512 threads (one per CU), %d launches per row; us\n", NV * 8, G, NL);
  printf("%-48s %-9s %9s %9s %9s %9s %9s %10s %8s\n", "mode", "pairing", "publish", "flag wait", "read", "total med", "total max",
         "same-XCD %", "bad");
  for (int cross = 0; cross < 2; ++cross)
    for (int mode = 0; mode < 5; ++mode) {
      std::vector<double> pub, wait, rd, tot, totmax;
      unsigned long long bad = 0, same = 0, pairs = 0;
      for (int l = 0; l < NL; ++l) {
        ++gen;
        hipLaunchKernelGGL(exch_kernel<NV>, dim3(G), dim3(512), 0, 0, scratch, flags, gen, st, mode, cross, sink);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), st, G * sizeof(Stamp), hipMemcpyDeviceToHost);
        if (l < 4) continue;
        double mx = 0;
        for (int b = 0; b < G; ++b) {
          const Stamp& s = h[b];
          pub.push_back((s.t1 - s.t0) * 0.01); wait.push_back((s.t2 - s.t1) * 0.01); rd.push_back((s.t3 - s.t2) * 0.01);
          tot.push_back((s.t3 - s.t0) * 0.01);
          mx = std::max(mx, (s.t3 - s.t0) * 0.01);
          bad += s.bad;
          same += s.xcc == h[cross ? (b ^ 1) : (b ^ 8)].xcc;
          ++pairs;
        }
        totmax.push_back(mx);
      }
      auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
      printf("%-48s %-9s %9.2f %9.2f %9.2f %9.2f %9.2f %10.1f %8llu\n", mname[mode], cross ? "b ^ 1" : "b ^ 8", med(pub), med(wait),
             med(rd), med(tot), med(totmax), 100.0 * same / pairs, bad);
    }
}

int main() {
  f4* scratch;
  unsigned* flags;
  Stamp* st;
  float* sink;
  hipMalloc(&scratch, (size_t)256 * 32 * 512 * sizeof(f4));
  hipMalloc(&flags, 4096);
  hipMemset(flags, 0, 4096);
  hipMalloc(&st, 256 * sizeof(Stamp));
  hipMalloc(&sink, 4096);
  unsigned gen = 0;
  sweep<16>(scratch, flags, st, sink, gen);      // half a tile each way (symmetric exchange)
  sweep<8>(scratch, flags, st, sink, gen);       // a quarter (four-way split, or fp16 partials)
  sweep<32>(scratch, flags, st, sink, gen);      // the whole tile one way (asymmetric: one workgroup finishes the tile)
  return 0;
}
