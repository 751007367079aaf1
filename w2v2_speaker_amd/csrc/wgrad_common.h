// wgrad_common.h -- argument block shared by the grouped weight-gradient kernels (wgrad.hip: 128x128, 256x128 ring and
// 256x256x32 ring; wgrad_phased.hip: 256x256x64 phased)
#pragma once
#include "common.h"
#include <stdlib.h>
#include <type_traits>

typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;
typedef __attribute__((address_space(3))) short4v lds_s4_t;

constexpr int WG_MAXP = 32;

struct WgProblem {
  const bf16_t* dY;
  const bf16_t* X;
  float* dW;
  float* dbias;
  int64_t ld_dy, ld_x, ld_dw;
  int n_out, n_in;
  int tile_begin, tiles_n;
};
struct WgArgs {
  WgProblem p[WG_MAXP];
  int n_problems, total_tiles, ktiles;
};


// wgrad_phased.hip: the 256x256x64 phased kernel; `dtype` = W2V2_BF16 / W2V2_F16, one workgroup per tile
// late = DMA pieces of a phase (0..2) issued between its MFMAs
void w2v2_launch_wgrad_phased(const WgArgs& a, int dtype, int tiles, int late, hipStream_t st);
