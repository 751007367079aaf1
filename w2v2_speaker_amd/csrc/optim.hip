// optim.hip -- fused Adam over one flat f32 parameter arena (torch.optim.Adam semantics,
// ref: config/optim/algo/adam.yaml:1-16, wired at src/main.py:323-335).  One launch updates every
// trainable parameter and refreshes the bf16 copy the MFMA GEMMs read.  HBM-bound:
// 4 f32 streams read (p, g, m, v) + 3 written (p, m, v) + 2 B/param bf16 copy = 30 B/param.
#include "common.cuh"

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   bf16_t* __restrict__ pb, int64_t n, float lr, float b1, float b2,
                                                   float eps, float step_size, float inv_sqrt_bc2, float gscale) {
  const int64_t nv = n >> 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    float* pa = &pp.x; const float* ga = &gg.x; float* ma = &mm.x; float* va = &vv.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gr = ga[e] * gscale;
      ma[e] = b1 * ma[e] + (1.0f - b1) * gr;
      va[e] = b2 * va[e] + (1.0f - b2) * gr * gr;
      const float denom = sqrtf(va[e]) * inv_sqrt_bc2 + eps;
      pa[e] -= step_size * ma[e] / denom;
    }
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
    if (pb != nullptr) {
      uint2 w;
      w.x = f32x2_to_bf16x2(pp.x, pp.y);
      w.y = f32x2_to_bf16x2(pp.z, pp.w);
      reinterpret_cast<uint2*>(pb)[i] = w;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (nv << 2) + threadIdx.x;
    const float gr = g[i] * gscale;
    const float mi = b1 * m[i] + (1.0f - b1) * gr;
    const float vi = b2 * v[i] + (1.0f - b2) * gr * gr;
    m[i] = mi; v[i] = vi;
    const float pn = p[i] - step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
    p[i] = pn;
    if (pb != nullptr) pb[i] = f32_to_bf16(pn);
  }
}

extern "C" int w2v2_adam_step(float* p, const float* g, float* m, float* v, void* pb, int64_t n, float lr,
                              float beta1, float beta2, float eps, float bias_corr1, float bias_corr2,
                              float grad_scale, void* stream) {
  W2V2_REQUIRE(p && g && m && v && n >= 0, "adam_step: bad arguments");
  W2V2_REQUIRE(bias_corr1 > 0.f && bias_corr2 > 0.f, "adam_step: bias corrections must be > 0");
  if (n == 0) return 0;
  int64_t nb = cdiv(n >> 2, 256);
  if (nb > 8192) nb = 8192;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), p, g, m, v, (bf16_t*)pb, n, lr,
                     beta1, beta2, eps, lr / bias_corr1, 1.0f / sqrtf(bias_corr2), grad_scale);
  W2V2_CHECK_LAUNCH("adam_step");
  return 0;
}
