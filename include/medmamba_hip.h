/* medmamba_hip.h — C ABI of libmedmamba_hip.so (MI355X / gfx950 only).
 *
 * This is the drop-in boundary for MedMamba's SS2D hot path.  The reference reaches native code at
 * exactly one place:
 *     from mamba_ssm.ops.selective_scan_interface import selective_scan_fn     (MedMamba.py:12)
 *     out_y = self.selective_scan(xs, dts, As, Bs, Cs, Ds, z=None,
 *                                 delta_bias=dt_projs_bias, delta_softplus=True,
 *                                 return_last_state=False)                      (MedMamba.py:273-279)
 * whose native half in mamba_ssm 1.0.1 is `selective_scan_cuda.fwd / .bwd` (third-party, CUDA only,
 * not in the reference tree).  mm_scan_fwd / mm_scan_bwd replace those two entry points.
 * The mm_ss2d_* / mm_* glue entry points replace the chains of stock torch ops around that call
 * inside SS2D.forward / forward_corev0 / SS_Conv_SSM.forward (file:line cited at each).
 *
 * Conventions (all entry points):
 *   - plain pointers + sizes, no C++/torch types; every pointer is a DEVICE pointer owned by the caller
 *   - the library allocates nothing, never synchronises, and keeps no state — with ONE exception since ABI 20: after
 *     mm_blas_attach, mm_gemm_f32 keeps one rocBLAS handle (and that handle's device workspace) per host thread, device and
 *     stream for the life of the thread.  Kernels are enqueued on the hipStream_t passed as `stream` (NULL = default
 *     stream) and the call returns immediately
 *   - return 0 on success; < 0 = mm_status (bad argument / unsupported variant, nothing launched);
 *     > 0 = hipError_t of the failed launch.  Nothing throws across the ABI.
 *   - fp32 everywhere (the reference path is fp32: MedMamba.py:265-271, 280, 297)
 *   - strides are in ELEMENTS; the innermost (sequence) dimension always has stride 1
 */
#ifndef MEDMAMBA_HIP_H
#define MEDMAMBA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MM_ABI_VERSION 22

enum mm_status {
  MM_OK = 0,
  MM_ERR_NULL = -1,        /* a required pointer is NULL */
  MM_ERR_SHAPE = -2,       /* non-positive size, dim % G != 0, ... */
  MM_ERR_UNSUPPORTED = -3, /* variant not on the MedMamba path (N > 16, complex A, z, ...) */
  MM_ERR_ALIGN = -4,       /* a pointer is not 4-byte aligned */
  MM_ERR_WORKSPACE = -5,   /* workspace missing / too small for the requested operation */
  MM_ERR_BLAS = -6         /* mm_gemm_f32: no BLAS attached, a symbol is missing, or the library returned an error */
};

/* Operands of one selective_scan_fn call (MedMamba.py:273-279; semantics temp.py:57-139):
 *   delta' = delta_softplus ? softplus(delta + delta_bias[d]) : delta + delta_bias[d]
 *   x_t[n] = exp(delta'_t * A[d,n]) * x_{t-1}[n] + delta'_t * B[b,g(d),n,t] * u_t ,  x_{-1} = 0
 *   out_t  = sum_n C[b,g(d),n,t] * x_t[n] + D[d] * u_t ,        g(d) = d / (dim / G)
 * u, delta: (batch, dim, L) with strides (*_sb, *_sd, 1).  A: (dim, N) contiguous.
 * B, C: (batch, G, N, L) with strides (*_sb, *_sg, *_sn, 1) — the reference passes non-contiguous
 * views of x_dbl here (MedMamba.py:261, 267-268).  D, delta_bias: (dim,) or NULL.
 * out: (batch, dim, L) contiguous.
 *
 * x_chk (optional, forward output / backward input): state checkpoints
 * batch * ceil(L / mm_scan_chunk()) * dim * N floats, contiguous — the state after every chunk of
 * mm_scan_chunk() steps, laid out (batch, chunk, dim, N) since ABI 19 (a workspace between the two kernels: callers only
 * size it).  Required by mm_scan_bwd; pass NULL to mm_scan_fwd for inference.
 *
 * Backward (mm_scan_bwd): dout (batch, dim, L) contiguous in; du, ddelta (batch, dim, L) contiguous
 * out (fully written); dB, dC (batch, G, N, L) contiguous; dA (dim, N), dD (dim), ddelta_bias (dim).
 * Two output forms for the parameter / B / C gradients:
 *   - deterministic (since ABI 19; what medmamba_amd itself uses: dpar_sb / dBC_sc below non-zero): every output element has
 *     exactly one writer, plain stores, nothing to zero-fill, bitwise reproducible; the caller sums the per-batch-item
 *     (dA, dD, ddelta_bias) and per-workgroup (dB, dC) partials in an order of its choice;
 *   - accumulating (dpar_sb == dBC_sc == 0, the ABI <= 18 behaviour, kept for callers built against older headers): dA, dD,
 *     ddelta_bias — and dB, dC where several workgroups share a direction — are ADDED into with fp32 atomics, so the caller
 *     zero-fills them before the call and the result depends on the order in which the adds land.
 * dD / ddelta_bias may be NULL when D / delta_bias are NULL.
 *
 * The struct is SELF-DESCRIBING (since ABI 18): `struct_size` = sizeof(mm_scan_args) of the header the caller was built against.
 * The library accepts exactly the sizes at which a release of this header ended the struct — MM_SCAN_ARGS_SIZE_BASE (through
 * rev_mask), _DBC (through dC_sn), _STRIDED (through o_sd), _V18 (through dt_rank) and the current sizeof — and reads the fields a shorter caller
 * does not have as zero (= "contiguous", "no fused dt projection").  A LARGER struct (a newer header) is accepted when every
 * byte beyond this library's sizeof is zero, i.e. the caller uses none of the fields this library does not know.  Anything
 * else (0, a size in the middle of a field group) returns MM_ERR_SHAPE and nothing is read beyond struct_size bytes.
 */
typedef struct mm_scan_args {
  uint32_t struct_size;   /* sizeof(mm_scan_args) as the caller sees it; first member on purpose */
  int32_t batch, dim, L, N, G;
  int32_t delta_softplus;
  const float* u;
  const float* delta;
  const float* A;
  const float* B;
  const float* C;
  const float* D;
  const float* delta_bias;
  float* out;
  float* x_chk;
  int64_t u_sb, u_sd;
  int64_t delta_sb, delta_sd;
  int64_t B_sb, B_sg, B_sn;
  int64_t C_sb, C_sg, C_sn;
  /* backward only */
  const float* dout;
  float* du;
  float* ddelta;
  float* dA;
  float* dB;
  float* dC;
  float* dD;
  float* ddelta_bias;
  /* tuning override: 0 = library default; low byte = states-per-lane variant (1,2,4; forward: 32 = the workgroup-cooperative kernel);
   * bits 16-23 waves per workgroup; forward: bit 24 / 25 force the register-lean form on / off, bit 26 = every wave stages its own
   * B / C tile (default where it applies: one shared tile per 4-wave workgroup).  Results do not depend on any of these. */
  int32_t variant;
  /* Cross-scan without materialising it (replaces the stack/transpose/flip/cat of MedMamba.py:256-257 and the
   * flips of :282).  All zero = plain selective_scan_fn semantics.
   *   u_groups : 0 or G -> u holds one channel block per group, shape (batch, G*H, L).  Otherwise u (and dout in
   *              the backward) hold only u_groups blocks, shape (batch, u_groups*H, L), and group g reads block
   *              (u_map >> 4*g) & 15   (SS2D: 2 blocks = row-major and column-major image, 4 directions).
   *   rev_mask : bit g set -> group g runs over the sequence BACKWARDS: its time step t is memory position L-1-t in
   *              u, delta, B, C, dout and in every per-position output (out, du, ddelta, dB, dC); every tensor stays
   *              in position order and no flipped copy ever exists.
   * du is always written per group (batch, G*H, L): the caller sums the groups that share a block. */
  int32_t u_groups;
  uint32_t u_map;
  uint32_t rev_mask;
  /* backward: element strides of dB / dC (batch, group, state); all zero = contiguous (batch, G, N, L).  Lets the caller
   * receive dB/dC directly inside the gradient of x_dbl (MedMamba.py:261: B and C are row blocks of x_dbl). */
  int64_t dB_sb, dB_sg, dB_sn;
  int64_t dC_sb, dC_sg, dC_sn;
  /* backward: element strides of dout / du / ddelta rows; all zero = contiguous.  dout rows at dout + b*dout_sb + d*o_sd,
   * du and ddelta rows at base + b*dud_sb + d*o_sd (one channel stride for the three).  Lets them live in channel-major
   * planes (channel, batch, L) where the projections around the scan are single large GEMMs. */
  int64_t dout_sb, dud_sb, o_sd;
  /* forward, optional: the dt projection of SS2D (MedMamba.py:262: dts = einsum(dts, dt_projs_weight)) fused into the scan.
   * dt_w != NULL: `delta` is not read (may be NULL); instead delta[b,d,t] = sum_r dt_w[d*dt_rank + r] * dts[b, g(d), r, t] with
   * dt_w (dim, dt_rank) contiguous and dts (batch, G, dt_rank, L) with element strides (dts_sb, dts_sg, dts_sn, 1) — the first
   * dt_rank rows of x_dbl (MedMamba.py:261).  dt_rank <= mm_scan_dt_max(); needs L % 4 == 0, 16-B aligned rows and
   * delta_softplus; anything else returns MM_ERR_UNSUPPORTED and the caller materialises delta with a GEMM.  The backward
   * entry point ignores these fields (it needs delta as a tensor). */
  const float* dt_w;
  const float* dts;
  int64_t dts_sb, dts_sg, dts_sn;
  int32_t dt_rank;
  /* backward, optional (since ABI 19): outputs without atomics — deterministic, nothing to zero-fill.
   *   dpar_sb != 0: dA, dD, ddelta_bias point at per-batch-item PARTIAL buffers: the gradient contribution of batch item b is
   *     stored (plain stores, every element written exactly once) at dA + b*dpar_sb (dim*N floats), dD + b*dpar_sb (dim),
   *     ddelta_bias + b*dpar_sb (dim); the caller sums over b in an order of its choice (mm_ss2d_pack_bwd does it while
   *     un-packing).  One stride for the three: they are meant to be three offsets into one (batch, S) buffer.
   *   dBC_sc != 0: when a direction is shared by W > 1 workgroups (mm_scan_plan out[6]), workgroup w of a direction stores its
   *     partial dB / dC at dB + w*dBC_sc / dC + w*dBC_sc (same strides inside a plane) instead of adding with atomics; the
   *     caller sums the W planes.  With W == 1 the field is ignored and dB / dC are written in place (plain stores, no
   *     zero-fill needed either way).  All zero = the ABI-18 behaviour: atomicAdd into zero-filled dA, dD, ddelta_bias (and
   *     dB / dC when W > 1). */
  int64_t dpar_sb;
  int64_t dBC_sc;
} mm_scan_args;

/* struct sizes earlier layouts of this header ended at (see struct_size above) */
#define MM_SCAN_ARGS_SIZE_BASE ((uint32_t)offsetof(mm_scan_args, dB_sb))
#define MM_SCAN_ARGS_SIZE_DBC ((uint32_t)offsetof(mm_scan_args, dout_sb))
#define MM_SCAN_ARGS_SIZE_STRIDED ((uint32_t)offsetof(mm_scan_args, dt_w))
#define MM_SCAN_ARGS_SIZE_V18 ((uint32_t)offsetof(mm_scan_args, dpar_sb))

/* replaces selective_scan_cuda.fwd behind selective_scan_fn (MedMamba.py:273-279) */
int mm_scan_fwd(const mm_scan_args* args, void* stream);
/* replaces selective_scan_cuda.bwd behind SelectiveScanFn.backward (autograd of MedMamba.py:273-279) */
int mm_scan_bwd(const mm_scan_args* args, void* stream);
/* The launch plan mm_scan_fwd (backward == 0) / mm_scan_bwd (backward != 0) would take for these arguments, without launching
 * anything (sizes, strides, alignment of the pointers as given — NULL pointers count as aligned — and `variant` are read; no
 * memory is touched): out[0] = states per lane, out[1] = wavefronts per workgroup, out[2] = workgroups, out[3] = 1 if the
 * 16-byte vector path is taken, out[4] = 1 for the forward's register-lean kernel; backward only: out[5] = 1 if the call as
 * given uses no atomics at all (dpar_sb set, and dBC_sc set or one workgroup per direction) = bitwise reproducible,
 * out[6] = W = workgroups that share a direction (number of dB / dC partial planes the caller must provide for dBC_sc),
 * out[7] = channel tiles each workgroup walks in turn.  For callers that size partial buffers, tests and tuning. */
int mm_scan_plan(const mm_scan_args* args, int backward, int32_t out[8]);
/* checkpoint interval (steps) of x_chk */
int mm_scan_chunk(void);
/* largest dt_rank mm_scan_fwd fuses (mm_scan_args.dt_w) */
int mm_scan_dt_max(void);

/* Block glue of SS_Conv_SSM.forward (MedMamba.py:354-357 + channel_shuffle :308-320), one pass over HBM:
 *   out[b,p,2i] = left[b,i,p] + inp[b,p,2i] ;  out[b,p,2i+1] = ssm[b,p,i] + inp[b,p,2i+1]
 * left: conv-branch output NCHW (batch, C2, P); ssm: SS2D-branch output, NHWC (batch, P, C2) or — ssm_channel_first != 0 —
 * channel-first (batch, C2, P) like left; inp, out: block input / output NHWC (batch, P, 2*C2); contiguous fp32, P = H*W.
 * Two neighbours of the chain are folded in (both optional):
 *   left_relu != 0 : `left` is the PRE-activation of the conv branch's trailing nn.ReLU (MedMamba.py:347), applied here;
 *   ssm_scale (batch) or NULL: per-sample DropPath factor mask/keep_prob of self.drop_path (MedMamba.py:335, 353);
 *   left_bias (C2) or NULL: the bias of the conv branch's closing 1x1 convolution (MedMamba.py:345), added to `left` before the
 *   ReLU here instead of by a pass of its own (the 1x1 convolution is a bias-free GEMM then); the backward needs the same
 *   pointer for the ReLU mask (left_pre + left_bias > 0).
 * Backward: dleft (batch, C2, P) and dssm (same layout as ssm) from dout (batch, P, 2*C2); d(inp) = dout;
 *   left_pre = the same pre-activation (ReLU mask) or NULL.
 * ssm_sb / ssm_sd (dssm_*): batch / channel element strides of a channel-first ssm (plane (b,i) at ssm + b*sb + i*sd,
 *   unit stride along P); both 0 = contiguous (batch, C2, P).  Ignored for the NHWC form. */
int mm_shuffle_residual_fwd(const float* left, const float* ssm, int64_t ssm_sb, int64_t ssm_sd, const float* inp, float* out,
                            const float* ssm_scale, int left_relu, const float* left_bias, int batch, int P, int C2,
                            int ssm_channel_first, void* stream);
int mm_shuffle_residual_bwd(const float* dout, float* dleft, float* dssm, int64_t dssm_sb, int64_t dssm_sd,
                            const float* ssm_scale, const float* left_pre, const float* left_bias, int batch, int P, int C2,
                            int ssm_channel_first, void* stream);

/* ---- SS2D in channel-first planes (everything between in_proj and out_proj is (batch, channel, H*W)) ----------
 * Plane tensors are addressed as base + b*X_sb + d*X_sd (element strides, unit stride along H*W), so the same kernels
 * serve batch-major (batch, D, L) storage (sb = D*L, sd = L) and channel-major (D, batch, L) storage (sb = L,
 * sd = batch*L) — the latter turns every projection around the scan into one large GEMM over batch*L columns.
 * mm_dwconv_silu_cross_fwd: depthwise conv3x3 (pad 1) + bias + SiLU (MedMamba.py:153-162, 295) that writes the scan's
 *   two input orders directly (replaces the permute of :294 and the stack/transpose of :256):
 *   x planes (b,d) of H*W floats at x + b*x_sb + d*x_sd; w (D,1,3,3); bias (D) or NULL;
 *   u2 = 2*D planes per batch item (plane (b, j*D+d) at u2 + b*u2_sb + (j*D+d)*u2_sd):
 *   u2[b,0,d,h*W+w] = u2[b,1,d,w*H+h] = silu(conv(x)[b,d,h,w] + bias[d]).
 * mm_dwconv_silu_cross_bwd: du2 (same plane indexing) [+ du4 or NULL: the scan's per-direction input gradients, 4*D
 *   planes per batch item, plane (b, k*D+d) at du4 + b*du4_sb + (k*D+d)*du4_sd; directions 0,1 add to the row-major image,
 *   2,3 to the column-major one] -> dx planes and per-(plane, strip) partial sums
 *   ws[((b*D+d)*S + s)*10 + (0..8: dW[kh][kw], 9: dbias)], S = mm_dwconv_silu_cross_strips(H, W)  (the caller sums over b, s). */
int mm_dwconv_silu_cross_fwd(const float* x, int64_t x_sb, int64_t x_sd, const float* w, const float* bias, float* u2,
                             int64_t u2_sb, int64_t u2_sd, int batch, int D, int H, int W, void* stream);
/* Both kernels work on row strips of a plane (whole plane when it fits 48 KB of LDS, else 32 rows + halo), so there is no
 * plane-size limit in practice: mm_dwconv_silu_cross_supported is 0 only when a single strip row set cannot fit (W > ~8000).
 * mm_dwconv_silu_cross_strips(H, W) = number of strips per plane = rows of partial sums per plane in the backward's ws. */
int mm_dwconv_silu_cross_supported(int H, int W);
int mm_dwconv_silu_cross_strips(int H, int W);
int mm_dwconv_silu_cross_bwd(const float* du2, int64_t du2_sb, int64_t du2_sd, const float* du4, int64_t du4_sb, int64_t du4_sd,
                             const float* x, int64_t x_sb, int64_t x_sd,
                             const float* w, const float* bias, float* dx, int64_t dx_sb, int64_t dx_sd, float* ws, int batch,
                             int D, int H, int W, void* stream);
/* cross-merge (MedMamba.py:282-286 + the 4-way sum of :298), all tensors in position order:
 *   m[b,d,h*W+w] = out4[b,0,d,h*W+w] + out4[b,1,d,h*W+w] + out4[b,2,d,w*H+h] + out4[b,3,d,w*H+h]
 *   (directions: row-major forward / backward, column-major forward / backward).  out4 (batch,4,D,L) contiguous,
 *   m planes at m + b*m_sb + d*m_sd. */
int mm_cross_merge_fwd(const float* out4, float* m, int64_t m_sb, int64_t m_sd, int batch, int D, int H, int W, void* stream);
/* dst[b,d,w*H+h] = src[b,d,h*W+w]; planes at base + b*sb + d*sd (adjoint of the column-major half of the merge) */
int mm_plane_transpose(const float* src, int64_t src_sb, int64_t src_sd, float* dst, int64_t dst_sb, int64_t dst_sd, int batch,
                       int D, int H, int W, void* stream);
/* out_norm LayerNorm over the D channels (eps) + gate with SiLU(z) (MedMamba.py:300-301), channel-first:
 *   y[b,d,p] = ((m[b,d,p]-mu[b,p])*rstd[b,p]*gamma[d]+beta[d]) * silu(z[b,d,p]);  m,y (batch,D,L); z planes with batch
 *   stride z_sb; mu,rstd (batch,L) outputs.  Backward: dm, dz and per-workgroup
 *   partial sums ws[row*2*D + (0: dgamma, D: dbeta) + d] for row < mm_ln_gate_rows(batch, D, L) (the caller sums rows).
 *   Channel strides (the *_sd arguments): 15 * stride + batch * L < 2^29 elements, else MM_ERR_SHAPE (a step of up to 16 channel
 *   rows is addressed with 32-bit byte offsets; D * stride itself is not limited). */
int mm_ln_gate_fwd(const float* m, int64_t m_sb, int64_t m_sd, const float* z, int64_t z_sb, int64_t z_sd, const float* gamma,
                   const float* beta, float eps, float* y, int64_t y_sb, int64_t y_sd, float* mu, float* rstd, int batch, int D,
                   int L, void* stream);
int mm_ln_gate_bwd(const float* dy, int64_t dy_sb, int64_t dy_sd, const float* m, int64_t m_sb, int64_t m_sd, const float* z,
                   int64_t z_sb, int64_t z_sd, const float* gamma, const float* beta, const float* mu, const float* rstd,
                   float* dm, int64_t dm_sb, int64_t dm_sd, float* dz, int64_t dz_sb, int64_t dz_sd, float* ws, int batch,
                   int D, int L, void* stream);
int mm_ln_gate_rows(int batch, int D, int L);

/* Block prologue of SS_Conv_SSM.forward (MedMamba.py:350-352): inp (batch, P, 2*C2) NHWC ->
 *   left_nchw (batch, C2, P) = inp[..., :C2] transposed (conv-branch input, replaces chunk + permute + contiguous)
 *   rn (batch, P, C2) = LayerNorm_{C2}(inp[..., C2:]) * gamma + beta  (ln_1), statistics mu/rstd (batch*P).
 * left_affine (optional, inference): [scale C2 | shift C2] applied to the left half while it is transposed — the eval-mode
 *   BatchNorm2d that opens the conv branch (MedMamba.py:338), folded: left_nchw[b,i,p] = inp[b,p,i]*scale[i] + shift[i]
 *   (before the convolution's zero padding, so exact at the image border).  NULL = plain copy; the backward assumes NULL.
 * Backward writes BOTH halves of dinp (batch, P, 2*C2) — left from dleft_nchw, right from drn, plus dres (batch, P, 2*C2)
 * or NULL: the gradient that reaches the block input through the residual add of MedMamba.py:357 — and per-workgroup partial
 * sums ws[row*2*C2 + (0: dgamma, C2: dbeta) + c], row < mm_block_split_rows(batch, P, C2) (the caller sums rows). */
int mm_block_split_fwd(const float* inp, const float* gamma, const float* beta, float eps, const float* left_affine,
                       float* left_nchw, float* rn, float* mu, float* rstd, int batch, int P, int C2, void* stream);
int mm_block_split_bwd(const float* dleft_nchw, const float* drn, const float* dres, const float* inp, const float* gamma,
                       const float* mu, const float* rstd, float* dinp, float* ws, int batch, int P, int C2, void* stream);
int mm_block_split_rows(int batch, int P, int C2);

/* PatchMerging2D front half (MedMamba.py:93-116): 2x2 strided gather (reference channel order x0..x3 = (even row, even col),
 * (odd, even), (even, odd), (odd, odd); odd H / W cropped to the even part, :97-111) + LayerNorm over the 4*C gathered channels,
 * one pass:  x (batch, H, W, C) NHWC -> out (batch, H/2, W/2, 4*C) = LN(gather(x)) * gamma + beta, statistics mu / rstd
 * (batch * H/2 * W/2).  The Linear(4C -> 2C) of :117 stays a GEMM on `out`.
 * Backward: dy (same shape as out) -> dinp (batch, H, W, C) at the gathered positions (the caller zero-fills dinp when H or W
 * is odd: the cropped row / column receives no gradient) and per-wavefront partial sums
 * ws[row * 8*C + (0: dgamma | 4*C: dbeta) + c], row < mm_patch_merge_ln_rows(batch, H, W) (the caller sums the rows).
 * mm_patch_merge_ln_supported(C): C % 4 == 0 and 4*C <= 2048 (a row lives in one wavefront's registers); 16-B aligned pointers. */
int mm_patch_merge_ln_supported(int C);
int mm_patch_merge_ln_rows(int batch, int H, int W);
int mm_patch_merge_ln_fwd(const float* x, const float* gamma, const float* beta, float eps, float* out, float* mu, float* rstd,
                          int batch, int H, int W, int C, void* stream);
int mm_patch_merge_ln_bwd(const float* dy, const float* x, const float* gamma, const float* mu, const float* rstd, float* dinp,
                          float* ws, int batch, int H, int W, int C, void* stream);

/* PatchEmbed2D back half (MedMamba.py:70-76: `x = self.proj(x).permute(0, 2, 3, 1)`, then LayerNorm(C)) in one pass:
 * x (batch, C, HW) contiguous NCHW conv output -> out (batch, HW, C) NHWC rows = LN(x^T) * gamma + beta, statistics mu / rstd
 * (batch * HW).  Backward: dy (batch, HW, C) -> dx (batch, C, HW) and partial sums ws[row * 2*C + (0: dgamma | C: dbeta) + c],
 * row < mm_nchw_ln_rows_ws_rows(batch, HW) (the caller sums the rows).  mm_nchw_ln_rows_supported(C): C <= 512; an image (C * HW * 4 bytes)
 * must stay below 2 GB (MM_ERR_UNSUPPORTED otherwise: the kernels address it with 32-bit byte offsets). */
int mm_nchw_ln_rows_supported(int C);
int mm_nchw_ln_rows_ws_rows(int batch, int HW);
int mm_nchw_ln_rows_fwd(const float* x, const float* gamma, const float* beta, float eps, float* out, float* mu, float* rstd,
                        int batch, int C, int HW, void* stream);
int mm_nchw_ln_rows_bwd(const float* dy, const float* x, const float* gamma, const float* mu, const float* rstd, float* dx,
                        float* ws, int batch, int C, int HW, void* stream);

/* Training-mode BatchNorm2d of the conv branch (MedMamba.py:338, 340, 343) with the nn.ReLU that follows two of them (:341,
 * :344) folded in.  x, y, dy, dx: contiguous NCHW (batch, C, HW).  Semantics of torch.nn.BatchNorm2d in training mode: batch
 * statistics (biased variance) for the normalisation; running_mean / running_var (either may be NULL) updated in place with
 * `momentum` and the unbiased variance.  mean / rstd (C): saved batch statistics for the backward.
 * ws: scratch of 3 * C * mm_bn_splits(batch, C, HW) floats (partial statistics; need not be initialised).
 * relu != 0: y = max(0, bn(x)); the backward masks dy where bn(x) <= 0 (recomputed from x, y is not needed).
 * pre_bias (C) or NULL: x is a convolution's output WITHOUT its per-channel bias (MedMamba.py:339-340, 342-343: conv -> BatchNorm).
 *   BatchNorm(x + bias) and BatchNorm(x) normalise to the same values (the batch mean absorbs the constant); only the running mean
 *   differs, and it is updated with mean(x) + pre_bias — so the bias add over the conv output never has to run.  mean / rstd
 *   returned are those of x (what the backward, which is also given x, needs). */
int mm_bn_splits(int batch, int C, int HW);
int mm_bn_relu_fwd(const float* x, const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                   float* running_var, float* y, float* mean, float* rstd, float* ws, const float* pre_bias, int relu, int batch, int C,
                   int HW, void* stream);
/* The same forward with the batch statistics already available as `nparts` partials per channel, partials[(q*C + c)*3 + (0: count,
 * 1: mean, 2: M2 = sum of squared deviations from that mean)] — what mm_conv3x3_fwd emits for its output: one pass (apply) only. */
int mm_bn_relu_fwd_stats(const float* x, const float* partials, int nparts, const float* gamma, const float* beta, float eps,
                         float momentum, float* running_mean, float* running_var, float* y, float* mean, float* rstd, int relu,
                         int batch, int C, int HW, void* stream);
/* mm_bn_fused(batch, C, HW) = 1 when both directions run as ONE kernel with a whole channel in one workgroup's registers
 * (batch*HW <= 16384, C >= 64: the 14x14 / 7x7 stages; ws is not touched then).  Only that form can also return
 * dxsum (C) = the per-channel sum of dx — the bias gradient of the convolution that produced x (MedMamba.py:339, 342) —
 * pass NULL otherwise (MM_ERR_UNSUPPORTED if asked for with the two-kernel form). */
int mm_bn_fused(int batch, int C, int HW);
int mm_bn_relu_bwd(const float* dy, const float* x, const float* gamma, const float* beta, const float* mean, const float* rstd,
                   float* dx, float* dgamma, float* dbeta, float* ws, float* dxsum, int relu, int batch, int C, int HW, void* stream);

#ifdef MM_EXPERIMENTS
/* ---- EXPERIMENTS build only (lib/libmedmamba_hip_exp.so, `python -m medmamba_amd.build --experiments`): not part of the product
 * library.  Own dense 3x3 convolutions (correct, slower than MIOpen, kept as a measured experiment), and the
 * forward scan's timing-ablation bits: mm_scan_args.variant bits 8-15 (1 = no y store, 2 = no recurrence; results are WRONG when
 * set).  The product library ignores those bits. ---- */
/* Dense 3x3 convolution, padding 1, stride 1, of the conv branch (MedMamba.py:339, 342), forward, fp32 on the matrix cores:
 *   y[b,k,h,w] = bias[k] + sum_{c,r,s} w[k,c,r,s] * x'[b,c,h+r-1,w+s-1];  x, y contiguous NCHW, w (K, C, 3, 3), bias (K) or NULL.
 *   in_affine = [scale C | shift C] or NULL: x' = x*scale[c] + shift[c] (then ReLU if in_relu) inside the image, 0 in the padding —
 *   an eval-mode / already-normalised BatchNorm (+ReLU) in front of the convolution, applied while the input is staged.
 *   stats (optional): (mm_conv3x3_fwd_tiles(batch, H, W), K, 3) floats: per position tile and output channel (count, mean, M2) of
 *   the outputs (bias included) — the statistics pass of a training-mode BatchNorm that follows (mm_bn_relu_fwd_stats). */
int mm_conv3x3_fwd_tiles(int batch, int H, int W);
int mm_conv3x3_fwd(const float* x, const float* w, const float* bias, const float* in_affine, int in_relu, float* y, float* stats,
                   int batch, int C, int K, int H, int W, void* stream);

/* Second version of the same convolution (finer tiles with the two halves of the input channels accumulated by different waves
 * of a workgroup, raw-patch staging): `wt` is the weight PRE-TRANSPOSED to (C*9, K), wt[(c*9 + r*3 + s)*K + k] = w[k,c,r,s]
 * (K % 4 == 0, 16-B aligned).  Passing wt built from the spatially flipped, channel-swapped weight and dy as x gives the data
 * gradient.  stats: (mm_conv3x3_v2_tiles(batch, H, W), K, 3). */
int mm_conv3x3_v2_tiles(int batch, int H, int W);
int mm_conv3x3_v2_fwd(const float* x, const float* wt, const float* bias, const float* in_affine, int in_relu, float* y, float* stats,
                      int batch, int C, int K, int H, int W, void* stream);
#endif /* MM_EXPERIMENTS */

/* SS2D parameters (MedMamba.py:150-175) -> one buffer in kernel direction order, A = -exp(A_logs) (MedMamba.py:271):
 *   x_proj_w (4, C, D), dt_w (4, D, R), dt_b (4, D), A_logs (4*D, N), Ds (4*D) in the reference's direction order
 *   k = (row fwd, col fwd, row rev, col rev);  packed = [Wx 4*C*D | Wdt 4*D*R | A 4*D*N | D 4*D | bias 4*D] floats
 *   in kernel order g = (row fwd, row rev, col fwd, col rev); every segment starts on a multiple of 64 floats (256 B)
 *   and mm_ss2d_pack_size = total number of floats including that padding.
 * mm_ss2d_pack_bwd: dpacked (gradient in packed layout) -> grads (same segment layout, reference direction order,
 *   A segment = gradient w.r.t. A_logs = dA * A).
 *   nparts > 0: the A / D / bias segments of the gradient are NOT taken from dpacked but summed (fixed order, no atomics) over
 *   parts[q * S + (i - offset of the A segment)], q < nparts, S = mm_ss2d_pack_parts_size — the per-batch-item partial
 *   buffers mm_scan_bwd writes with mm_scan_args.dpar_sb = S, dA = parts, dD = parts + (D segment offset - A segment offset),
 *   ddelta_bias likewise.  nparts = 0 (parts may be NULL): everything from dpacked, as before ABI 19.
 *   Two more small reductions of the SS2D backward can ride on the same launch (both optional, NULL = skip):
 *   ln_ws (ln_rows, 2*D) = the partial rows mm_ln_gate_bwd wrote -> ln_out[2*D] = dgamma | dbeta of out_norm;
 *   dw_ws (dw_batch, D*dw_strips, 10) = the partial sums mm_dwconv_silu_cross_bwd wrote -> dw_out[D*9 | D] = the depthwise conv's
 *   weight gradient in (D,1,3,3) order followed by its bias gradient.  Fixed summation order (no atomics). */
int mm_ss2d_pack_size(int D, int C, int R, int N);
int mm_ss2d_pack_fwd(const float* x_proj_w, const float* dt_w, const float* dt_b, const float* A_logs, const float* Ds,
                     float* packed, int D, int C, int R, int N, void* stream);
int mm_ss2d_pack_bwd(const float* dpacked, const float* packed, const float* parts, float* grads, int D, int C, int R, int N,
                     int nparts, const float* ln_ws, int ln_rows, float* ln_out, const float* dw_ws, int dw_batch, int dw_strips,
                     float* dw_out, void* stream);
/* floats per part of `parts` = size of the [A | D | bias] tail of the packed layout (incl. its padding) */
int mm_ss2d_pack_parts_size(int D, int C, int R, int N);

/* Bias gradient of the conv branch's convolutions (MedMamba.py:338-346): out[c] = sum over batch and positions of the
 * contiguous NCHW tensor x (batch, C, HW).  When S = mm_channel_sum_nchw_split(batch, C) > 1 the batch is split over S
 * workgroups per channel and `out` receives S rows of C partial sums (S*C floats, plain stores — no atomics since ABI 19);
 * the caller adds the rows. */
int mm_channel_sum_nchw_split(int batch, int C);
int mm_channel_sum_nchw(const float* x, float* out, int batch, int C, int HW, void* stream);

/* dst[i] = sum_{j < nlead} src[j * lead_stride + i] for i < ninner (fp32, fixed summation order: 4 ... 32 interleaved parts of
 * the lead dimension — by the width of the tensor —, each summed in order, then joined in order) — the sum over the batch behind the batched weight-gradient
 * GEMMs of MedMamba.py:259, 262, 292, 302 in batch-major storage and behind per-workgroup partial rows (since ABI 21). */
int mm_sum_lead(const float* src, float* dst, int nlead, int64_t ninner, int64_t lead_stride, void* stream);
/* The same sum of a dense (nlead, nchunks, chunk) source into a destination whose chunks are `dst_chunk_stride` floats apart
 * (dst[c * dst_chunk_stride + o] = sum_j src[j][c][o]; the same bits as mm_sum_lead on a dense destination): the per-workgroup
 * partial dB / dC planes of mm_scan_bwd (mm_scan_args.dBC_sc) summed straight into the B / C rows of d(x_dbl), which sit between
 * the dt rows of the next direction (since ABI 22). */
int mm_sum_lead_chunks(const float* src, float* dst, int nlead, int64_t nchunks, int64_t chunk, int64_t dst_chunk_stride, void* stream);

/* fp32 GEMMs of the projections around the scan (MedMamba.py:259, 262, 292, 302; the conv branch's 1x1 conv :345) without the
 * host-side cost of a framework dispatch (measured: 7.6 us per call against 17-31 us through torch.bmm on this image).  The
 * library does not link a BLAS: mm_blas_attach(path) opens the rocBLAS the HOST PROCESS already uses (path of its librocblas.so;
 * solution indices are build-specific, so it must be that one) and resolves rocblas_create_handle / rocblas_set_stream /
 * rocblas_gemm_strided_batched_ex / rocblas_gemm_ex from it.  mm_gemm_f32 is then rocBLAS's column-major contract verbatim:
 *   C[i] = alpha * op(A[i]) * op(B[i]) + beta * C[i],  i < batch;  op = 'N' | 'T';  op(A) is m x k, op(B) k x n, C m x n;
 *   leading dimensions in elements, batch strides in elements (0 = the same matrix for every i);
 *   solution = a rocBLAS solution index recorded for exactly this problem (rocblas_gemm_algo_solution_index), 0 = the
 *   library's own choice.  One handle per host thread, device and stream (a handle's device workspace must not serve two
 *   GEMMs that run at the same time on different streams); atomics follow
 *   mm_blas_set_atomics (default: allowed, rocBLAS's own default).
 * Returns MM_ERR_BLAS when nothing is attached or rocBLAS reports an error (mm_blas_last_status() holds its status), and
 * MM_ERR_UNSUPPORTED (nothing launched) for the FIRST call on a stream that is currently being captured into a hipGraph: the
 * handle it would need allocates device memory. */
int mm_blas_attach(const char* librocblas_path);
int mm_blas_attached(void);
int mm_blas_set_atomics(int allowed);
int mm_blas_last_status(void);
int mm_gemm_f32(char opa, char opb, int m, int n, int k, float alpha, const float* A, int lda, int64_t stride_a, const float* B,
                int ldb, int64_t stride_b, float beta, float* C, int ldc, int64_t stride_c, int batch, int32_t solution,
                void* stream);

/* im2col of the input of a 3x3 / padding 1 / stride 1 convolution, every image in one launch: cols (batch, 9*C, H*W) with
 * cols[b][c*9 + r*3 + s][h*W + w] = x[b][c][h+r-1][w+s-1] (0 outside) — torch.nn.functional.unfold(x, 3, padding=1).  Used for the
 * reproducible weight gradient of the conv branch's dense convolutions (MedMamba.py:339, 342) under
 * torch.backends.cudnn.deterministic: dW = sum over b (fixed order) of dy[b] (K x HW) . cols[b]^T.
 * group = gs > 1 (batch % gs == 0): the gs images of a group side by side, cols (batch/gs, 9*C, gs*H*W) with image b in columns
 * [(b % gs)*H*W, ...) of group b / gs: one GEMM per group contracts over gs images (dy regrouped the same way). */
int mm_im2col3x3(const float* x, float* cols, int batch, int C, int H, int W, int group, void* stream);

/* AdamW update of the training step (train.py:187-201, 285-287: torch.optim.AdamW, amsgrad off) for a whole parameter list in one
 * launch per mm_adamw_max_tensors() tensors.  Tensor i has numel[i] fp32 elements at params[i] / exp_avg[i] / exp_avg_sq[i]: these
 * three pointer tables and numel live in DEVICE memory (built once).  The gradients are new tensors every step: grads_host is a HOST
 * array with the gradient pointers of tensors t0 .. t0+nt-1, copied into the kernel arguments (nothing the next step could
 * overwrite before the kernel has run).  The work is cut into chunks of mm_adamw_chunk() elements, one workgroup each: chunk c covers
 * elements [chunk_index[c] * chunk, ...) of tensor chunk_tensor[c] (absolute index, within [t0, t0+nt)); device int32 tables of
 * nchunks entries.  `step` = the step count AFTER this update (>= 1), the same for every tensor.  torch's rule and order:
 *   p -= lr*wd*p;  m += (1-b1)*(g-m);  v = b2*v + (1-b2)*g*g;  p -= lr/(1-b1^step) * m / (sqrt(v)/sqrt(1-b2^step) + eps). */
int mm_adamw_chunk(void);
int mm_adamw_max_tensors(void);
int mm_adamw_step(float* const* params, const float* const* grads_host, int t0, int nt, float* const* exp_avg, float* const* exp_avg_sq,
                  const int64_t* numel, const int32_t* chunk_tensor, const int32_t* chunk_index, int nchunks, float lr, float beta1,
                  float beta2, float eps, float weight_decay, double step, void* stream);

/* hipEventRecord(event, stream) for host code that holds no HIP headers (the C++ sequencing layer brackets the scan kernels with
 * the caller's timing events). */
int mm_event_record(void* event, void* stream);

int mm_abi_version(void);
const char* mm_status_string(int status);

#ifdef __cplusplus
}
#endif
#endif /* MEDMAMBA_HIP_H */
