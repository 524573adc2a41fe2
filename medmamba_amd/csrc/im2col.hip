// im2col of the dense 3x3 convolutions' input (MedMamba.py:339, 342) for the deterministic weight gradient (the reference trains with
// cudnn.deterministic, train.py:28-29; ops.ConvBiasFn / csrc_host conv3x3_bwd): product code, part of libmedmamba_hip.so.
#include "mm_common.h"
#include "medmamba_hip.h"

// ---- im2col of a 3x3 / padding 1 / stride 1 convolution's input, all images in ONE launch (ATen's im2col launches once per image:
// 1792 launches per MedMamba-S step when the deterministic weight gradient below uses it) --------------------------------------
// cols[b][c*9 + r*3 + s][h*W + w] = x[b][c][h+r-1][w+s-1] (0 outside the image): the layout of torch.nn.functional.unfold(x, 3,
// padding=1), so that dW = sum_b dy[b] (K x HW) . cols[b]^T (HW x 9C) is the weight gradient in (K, C, 3, 3) order.  With
// group = gs > 1 the images of a group sit side by side: cols (batch/gs, 9C, gs*HW) — one GEMM then contracts over gs images and
// the caller sums batch/gs partial products instead of batch.
namespace {
__global__ __launch_bounds__(256) void im2col3x3_kernel(const float* __restrict__ x, float* __restrict__ cols, int C, int H, int W, int gs,
                                                        int planes) {
  const int HW = H * W;
  // grid.y is capped at 65535 by HIP: a workgroup row walks the planes b*C + c it owns (384 channels x 171 images already
  // exceed the cap — the deterministic weight gradient of trainer.set_seed runs this at any batch size)
  for (int plane = blockIdx.y; plane < planes; plane += gridDim.y) {
  const int b = plane / C, c = plane - b * C;
  const mm::rsrc_t rx = mm::make_rsrc(x + (int64_t)plane * HW, (int64_t)HW * 4);   // taps outside the image read 0 through the range check:
  // image j = b % gs of group g = b / gs: column block j of the group's (9C) x (gs*HW) matrix
  float* cp = cols + (((int64_t)(b / gs) * 9 * C + (int64_t)c * 9) * gs + (b % gs)) * HW;
  const int64_t tap = (int64_t)gs * HW;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
    const int h = i / W, w = i - h * W;
    float v[9];                                                                       // no branch, nine loads in flight
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s2 = 0; s2 < 3; ++s2) {
        const int hh = h + r - 1, ww = w + s2 - 1;
        v[r * 3 + s2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
            rx, (hh >= 0 && hh < H && ww >= 0 && ww < W) ? (hh * W + ww) * 4 : mm::kOOB, 0, 0));
      }
#pragma unroll
    for (int t = 0; t < 9; ++t) cp[t * tap + i] = v[t];
  }
  }
}
}  // namespace

extern "C" int mm_im2col3x3(const float* x, float* cols, int batch, int C, int H, int W, int group, void* stream) {
  if (!x || !cols) return MM_ERR_NULL;
  if (batch <= 0 || C <= 0 || H <= 0 || W <= 0 || (int64_t)batch * C > 0x7fffffffll) return MM_ERR_SHAPE;
  if (group < 1 || batch % group != 0) return MM_ERR_SHAPE;
  const int HW = H * W;
  int gx = (HW + 255) / 256;
  if (gx > 16) gx = 16;
  const int planes = batch * C;
  hipLaunchKernelGGL(im2col3x3_kernel, dim3(gx, planes < 65535 ? planes : 65535), dim3(256), 0, (hipStream_t)stream, x, cols, C, H, W,
                     group, planes);
  return (int)hipGetLastError();
}
