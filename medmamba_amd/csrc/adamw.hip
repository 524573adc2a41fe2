// AdamW update of the training step (train.py:187-201, 285-287: torch.optim.AdamW) over a whole parameter list in ONE launch.
//
// Why: torch's fused AdamW walks the list with multi_tensor_apply in chunks of 65536 elements per workgroup — 300 workgroups in 11
// launches for the 19.5 M parameters of MedMamba-S, i.e. a quarter of the chip streaming 28 B per element: 0.47 ms per step
// (1.1 TB/s) at the serial tail of the step, where nothing overlaps it.  Here: 2048-element chunks (one per 256-thread workgroup,
// 9.5 k workgroups), parameters and moments addressed through device-resident pointer tables, the GRADIENT pointers — new tensors
// every step — travelling in the kernel arguments (up to 448 per launch: no host-to-device copy that a later step could overtake),
// 16-B accesses where the four pointers of a chunk allow it.  Same update rule and operation order as torch.optim.AdamW (decoupled weight decay, bias-corrected moments, amsgrad
// off), fp32 throughout:
//     p   -= lr * wd * p
//     m    = m + (1 - b1) * (g - m)
//     v    = b2 * v + (1 - b2) * g * g
//     p   -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps),     bc_i = 1 - b_i^step
#include "mm_common.h"
#include "medmamba_hip.h"

namespace {

constexpr int kAdamChunk = 2048;
constexpr int kAdamMaxT = 448;       // gradient pointers per launch (kernel arguments are limited to 4 KB)

struct AdamArgs {
  float* const* p;
  const float* g[kAdamMaxT];
  int t0;
  float* const* m;
  float* const* v;
  const int64_t* numel;
  const int32_t* chunk_tensor;
  const int32_t* chunk_index;
  float lr_wd, b1c, b2, b2c, step_size, inv_bc2_sqrt, eps;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamArgs& a) {
  p -= a.lr_wd * p;
  m = fmaf(a.b1c, g - m, m);
  v = a.b2 * v + a.b2c * g * g;
  const float denom = sqrtf(v) * a.inv_bc2_sqrt + a.eps;
  p -= a.step_size * m / denom;
}

__global__ __launch_bounds__(256) void adamw_kernel(AdamArgs a) {
  const int t = a.chunk_tensor[blockIdx.x];
  const int64_t n = a.numel[t];
  const int64_t off = (int64_t)a.chunk_index[blockIdx.x] * kAdamChunk;
  const int cnt = (int)min((int64_t)kAdamChunk, n - off);
  float* p = a.p[t] + off;
  const float* g = a.g[t - a.t0] + off;
  float* m = a.m[t] + off;
  float* v = a.v[t] + off;
  const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                     reinterpret_cast<uintptr_t>(v)) & 15) == 0;
  if (vec) {
    const int n4 = cnt >> 2;
    for (int i = threadIdx.x; i < n4; i += 256) {
      float4 pv = reinterpret_cast<float4*>(p)[i], mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
      const float4 gv = reinterpret_cast<const float4*>(g)[i];
      adam_one(pv.x, gv.x, mv.x, vv.x, a); adam_one(pv.y, gv.y, mv.y, vv.y, a);
      adam_one(pv.z, gv.z, mv.z, vv.z, a); adam_one(pv.w, gv.w, mv.w, vv.w, a);
      reinterpret_cast<float4*>(p)[i] = pv; reinterpret_cast<float4*>(m)[i] = mv; reinterpret_cast<float4*>(v)[i] = vv;
    }
    for (int i = (n4 << 2) + threadIdx.x; i < cnt; i += 256) adam_one(p[i], g[i], m[i], v[i], a);
  } else {
    for (int i = threadIdx.x; i < cnt; i += 256) adam_one(p[i], g[i], m[i], v[i], a);
  }
}

}  // namespace

extern "C" {

int mm_adamw_chunk(void) { return kAdamChunk; }
int mm_adamw_max_tensors(void) { return kAdamMaxT; }

int mm_adamw_step(float* const* params, const float* const* grads_host, int t0, int nt, float* const* exp_avg, float* const* exp_avg_sq,
                  const int64_t* numel, const int32_t* chunk_tensor, const int32_t* chunk_index, int nchunks, float lr, float beta1,
                  float beta2, float eps, float weight_decay, double step, void* stream) {
  if (!params || !grads_host || !exp_avg || !exp_avg_sq || !numel || !chunk_tensor || !chunk_index) return MM_ERR_NULL;
  if (nchunks <= 0 || step < 1.0 || nt <= 0 || nt > kAdamMaxT || t0 < 0) return MM_ERR_SHAPE;
  AdamArgs a;
  a.p = params; a.m = exp_avg; a.v = exp_avg_sq; a.numel = numel; a.chunk_tensor = chunk_tensor; a.chunk_index = chunk_index;
  a.t0 = t0;
  for (int i = 0; i < nt; ++i) {
    if (!grads_host[i]) return MM_ERR_NULL;
    a.g[i] = grads_host[i];
  }
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  a.lr_wd = lr * weight_decay;
  a.b1c = 1.0f - beta1;
  a.b2 = beta2;
  a.b2c = 1.0f - beta2;
  a.step_size = (float)((double)lr / bc1);
  a.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  a.eps = eps;
  hipLaunchKernelGGL(adamw_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

}  // extern "C"
