// How fast does a GEMM-epilogue-shaped store burst leave the chip, by lane -> address map?  (gfx950)
// Every workgroup (8 waves, one per CU, persistent over 256x256 tiles of a [M][N] fp16 matrix, N = 3072) writes its
// tile the way the register epilogues of csrc/gemm.hip do: a wave owns 128 rows x 64 columns (128 B per row) and
// issues 16 x 16-byte stores per lane.  Patterns (which 16 bytes a lane writes in store instruction s):
//   0 comb     : lane (r = lane & 15, f = lane >> 4) writes row 16 i + r, bytes f * 32 + h * 16   (today's epilogue: four
//                scattered 16-byte pieces per row and instruction)
//   1 run64    : ... bytes h * 64 + f * 16   (four lanes = one contiguous 64-byte run per row and instruction)
//   2 fullline : lane (r = lane >> 3, c = lane & 7) writes row 8 s + r, bytes c * 16   (8 rows x whole 128-byte lines)
//   3 fullline + non-temporal stores      4 comb + non-temporal
// Output: us per pass over the matrix and TB/s, all CUs storing in phase (nothing else running).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int PAT>
__global__ __launch_bounds__(512) void store_probe(uint16_t* __restrict__ C, int M, int N, int tiles_m, int tiles_n) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int ntile = tiles_m * tiles_n;
  for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * 256 + wr * 128, n0 = tn * 256 + wc * 64;
    u32x4 v = {(unsigned)tile, (unsigned)lane, 0x3c003c00u, 0x3c003c00u};
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      int row, byte;
      if (PAT == 0 || PAT == 4) { row = 16 * (s >> 1) + (lane & 15); byte = (lane >> 4) * 32 + (s & 1) * 16; }
      else if (PAT == 1) { row = 16 * (s >> 1) + (lane & 15); byte = (s & 1) * 64 + (lane >> 4) * 16; }
      else { row = 8 * s + (lane >> 3); byte = (lane & 7) * 16; }
      const int m = m0 + row;
      if (m < M) {
        u32x4* p = reinterpret_cast<u32x4*>(reinterpret_cast<char*>(C + (int64_t)m * N + n0) + byte);
        if (PAT >= 3) __builtin_nontemporal_store(v, p); else *p = v;
      }
      v.x += 1;
    }
  }
}

int main() {
  const int M = 9834, N = 3072;
  uint16_t* C;
  hipMalloc(&C, (size_t)M * N * 2);
  const int tm = (M + 255) / 256, tn = N / 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[5] = {"comb (today)", "run64", "fullline", "fullline nt", "comb nt"};
  for (int rep = 0; rep < 2; ++rep)
    for (int p = 0; p < 5; ++p) {
      const int reps = 50;
      for (int i = 0; i < reps + 5; ++i) {
        if (i == 5) hipEventRecord(e0, 0);
        switch (p) {
          case 0: hipLaunchKernelGGL(store_probe<0>, dim3(256), dim3(512), 0, 0, C, M, N, tm, tn); break;
          case 1: hipLaunchKernelGGL(store_probe<1>, dim3(256), dim3(512), 0, 0, C, M, N, tm, tn); break;
          case 2: hipLaunchKernelGGL(store_probe<2>, dim3(256), dim3(512), 0, 0, C, M, N, tm, tn); break;
          case 3: hipLaunchKernelGGL(store_probe<3>, dim3(256), dim3(512), 0, 0, C, M, N, tm, tn); break;
          default: hipLaunchKernelGGL(store_probe<4>, dim3(256), dim3(512), 0, 0, C, M, N, tm, tn); break;
        }
      }
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      const double us = ms * 1e3 / reps;
      printf("%-14s %7.1f us per 60.4 MB pass  %5.2f TB/s\n", names[p], us, (double)M * N * 2 / us / 1e6);
    }
  return 0;
}
