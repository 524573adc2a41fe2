// extern "C" surface of libmedmamba_hip.so — argument validation + dispatch only (include/medmamba_hip.h).
#include <hip/hip_runtime.h>
#include <string.h>
#include "medmamba_hip.h"
#include "mm_common.h"

namespace mm {
int scan_fwd_launch(const mm_scan_args* a, hipStream_t stream, int32_t* plan_out = nullptr);
int scan_bwd_launch(const mm_scan_args* a, hipStream_t stream, int32_t* plan_out = nullptr);
}  // namespace mm

namespace {
inline bool al4(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 3) == 0; }

// The caller's struct (struct_size bytes of it) -> a full-size copy whose missing tail reads as zero (medmamba_hip.h).
int normalize(const mm_scan_args* in, mm_scan_args* out) {
  if (!in) return MM_ERR_NULL;
  const uint32_t sz = in->struct_size;
  if (sz != MM_SCAN_ARGS_SIZE_BASE && sz != MM_SCAN_ARGS_SIZE_DBC && sz != MM_SCAN_ARGS_SIZE_STRIDED && sz != MM_SCAN_ARGS_SIZE_V18 &&
      sz < sizeof(mm_scan_args))
    return MM_ERR_SHAPE;
  if (sz > 4096) return MM_ERR_SHAPE;
  const unsigned char* raw = reinterpret_cast<const unsigned char*>(in);
  for (uint32_t i = sizeof(mm_scan_args); i < sz; ++i)
    if (raw[i]) return MM_ERR_UNSUPPORTED;     // a newer caller uses a field this library does not know
  memset(out, 0, sizeof(*out));
  memcpy(out, in, sz < sizeof(*out) ? sz : sizeof(*out));
  out->struct_size = (uint32_t)sizeof(*out);
  return MM_OK;
}

int check_common(const mm_scan_args* a) {
  if (!a) return MM_ERR_NULL;
  if (a->batch <= 0 || a->dim <= 0 || a->L <= 0 || a->N <= 0 || a->G <= 0) return MM_ERR_SHAPE;
  if (a->dim % a->G != 0) return MM_ERR_SHAPE;
  if (a->N != mm::kNState) return MM_ERR_UNSUPPORTED;   // d_state = 16 on every MedMamba path (MedMamba.py:329,457)
  if (!a->u || (!a->delta && !a->dt_w) || !a->A || !a->B || !a->C) return MM_ERR_NULL;
  if (a->u_groups < 0) return MM_ERR_SHAPE;
  if (a->u_groups > 0 && a->u_groups < a->G) {
    if (a->G > 8) return MM_ERR_SHAPE;
    for (int g = 0; g < a->G; ++g)
      if ((int)((a->u_map >> (4 * g)) & 15) >= a->u_groups) return MM_ERR_SHAPE;
  }
  if (!al4(a->u) || (a->delta && !al4(a->delta)) || !al4(a->A) || !al4(a->B) || !al4(a->C)) return MM_ERR_ALIGN;
  return MM_OK;
}
}  // namespace

extern "C" {

int mm_abi_version(void) { return MM_ABI_VERSION; }
int mm_scan_chunk(void) { return mm::kChunk; }
int mm_scan_dt_max(void) { return 8; }

const char* mm_status_string(int s) {
  switch (s) {
    case MM_OK: return "ok";
    case MM_ERR_NULL: return "required pointer is NULL";
    case MM_ERR_SHAPE: return "bad shape";
    case MM_ERR_UNSUPPORTED: return "unsupported variant";
    case MM_ERR_ALIGN: return "misaligned pointer";
    case MM_ERR_WORKSPACE: return "workspace missing";
    case MM_ERR_BLAS: return "BLAS not attached or BLAS error";
    default: return s > 0 ? hipGetErrorString((hipError_t)s) : "unknown status";
  }
}

int mm_scan_fwd(const mm_scan_args* args, void* stream) {
  mm_scan_args full;
  int rc = normalize(args, &full);
  if (rc) return rc;
  const mm_scan_args* a = &full;
  rc = check_common(a);
  if (rc) return rc;
  if (!a->out) return MM_ERR_NULL;
  return mm::scan_fwd_launch(a, (hipStream_t)stream);
}

int mm_scan_bwd(const mm_scan_args* args, void* stream) {
  mm_scan_args full;
  int rc = normalize(args, &full);
  if (rc) return rc;
  const mm_scan_args* a = &full;
  rc = check_common(a);
  if (rc) return rc;
  if (!a->delta || !a->dout || !a->du || !a->ddelta || !a->dA || !a->dB || !a->dC) return MM_ERR_NULL;
  if (a->D && !a->dD) return MM_ERR_NULL;
  if (a->delta_bias && !a->ddelta_bias) return MM_ERR_NULL;
  if (!a->x_chk) return MM_ERR_WORKSPACE;
  return mm::scan_bwd_launch(a, (hipStream_t)stream);
}

int mm_scan_plan(const mm_scan_args* args, int backward, int32_t* out) {
  if (!out) return MM_ERR_NULL;
  mm_scan_args full;
  int rc = normalize(args, &full);
  if (rc) return rc;
  if (full.batch <= 0 || full.dim <= 0 || full.L <= 0 || full.G <= 0 || full.dim % full.G != 0) return MM_ERR_SHAPE;
  if (full.N != mm::kNState) return MM_ERR_UNSUPPORTED;
  for (int i = 0; i < 8; ++i) out[i] = 0;
  return backward ? mm::scan_bwd_launch(&full, nullptr, out) : mm::scan_fwd_launch(&full, nullptr, out);
}

}  // extern "C"
