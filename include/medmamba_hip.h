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
 *   - the library allocates nothing, keeps no state, never synchronises: kernels are enqueued on the
 *     hipStream_t passed as `stream` (NULL = default stream) and the call returns immediately
 *   - return 0 on success; < 0 = mm_status (bad argument / unsupported variant, nothing launched);
 *     > 0 = hipError_t of the failed launch.  Nothing throws across the ABI.
 *   - fp32 everywhere (the reference path is fp32: MedMamba.py:265-271, 280, 297)
 *   - strides are in ELEMENTS; the innermost (sequence) dimension always has stride 1
 */
#ifndef MEDMAMBA_HIP_H
#define MEDMAMBA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MM_ABI_VERSION 2

enum mm_status {
  MM_OK = 0,
  MM_ERR_NULL = -1,        /* a required pointer is NULL */
  MM_ERR_SHAPE = -2,       /* non-positive size, dim % G != 0, ... */
  MM_ERR_UNSUPPORTED = -3, /* variant not on the MedMamba path (N > 16, complex A, z, ...) */
  MM_ERR_ALIGN = -4,       /* a pointer is not 4-byte aligned */
  MM_ERR_WORKSPACE = -5    /* workspace missing / too small for the requested operation */
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
 * (batch, dim, ceil(L / mm_scan_chunk()), N) contiguous — the state after every chunk of
 * mm_scan_chunk() steps.  Required by mm_scan_bwd; pass NULL to mm_scan_fwd for inference.
 *
 * Backward (mm_scan_bwd): dout (batch, dim, L) contiguous in; du, ddelta (batch, dim, L) contiguous
 * out (fully written); dB, dC (batch, G, N, L) contiguous; dA (dim, N), dD (dim), ddelta_bias (dim):
 * ACCUMULATED into (atomicAdd across batch / channel tiles) — the caller zero-fills dA, dB, dC, dD,
 * ddelta_bias before the call.  dD / ddelta_bias may be NULL when D / delta_bias are NULL.
 */
typedef struct mm_scan_args {
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
  /* tuning override: 0 = library default; low byte = states-per-lane variant (1,2,4) */
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
} mm_scan_args;

/* replaces selective_scan_cuda.fwd behind selective_scan_fn (MedMamba.py:273-279) */
int mm_scan_fwd(const mm_scan_args* args, void* stream);
/* replaces selective_scan_cuda.bwd behind SelectiveScanFn.backward (autograd of MedMamba.py:273-279) */
int mm_scan_bwd(const mm_scan_args* args, void* stream);
/* checkpoint interval (steps) of x_chk */
int mm_scan_chunk(void);

int mm_abi_version(void);
const char* mm_status_string(int status);

#ifdef __cplusplus
}
#endif
#endif /* MEDMAMBA_HIP_H */
