// fp32 GEMM entry point of the C ABI: forwards to the rocBLAS of the host process (resolved at run time, nothing linked).
// See include/medmamba_hip.h (mm_blas_attach / mm_gemm_f32) for the contract.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <mutex>
#include <unordered_map>
#include "medmamba_hip.h"

namespace {

// the few rocBLAS types this file needs (values from rocblas-types.h; the C ABI of these entry points is stable across 4.x-5.x)
typedef void* rb_handle;
enum { RB_OP_N = 111, RB_OP_T = 112, RB_F32 = 151, RB_ALGO_STANDARD = 0, RB_ALGO_SOLUTION_INDEX = 1 };
enum { RB_ATOMICS_NOT_ALLOWED = 0, RB_ATOMICS_ALLOWED = 1 };
typedef int (*create_fn)(rb_handle*);
typedef int (*destroy_fn)(rb_handle);
typedef int (*set_stream_fn)(rb_handle, hipStream_t);
typedef int (*set_atomics_fn)(rb_handle, int);
typedef int (*gemm_sb_fn)(rb_handle, int, int, int, int, int, const void*, const void*, int, int, int64_t, const void*, int, int,
                          int64_t, const void*, const void*, int, int, int64_t, void*, int, int, int64_t, int, int, int, int32_t,
                          uint32_t);
typedef int (*gemm_fn)(rb_handle, int, int, int, int, int, const void*, const void*, int, int, const void*, int, int, const void*,
                       const void*, int, int, void*, int, int, int, int, int32_t, uint32_t);

struct Blas {
  void* dl = nullptr;
  create_fn create = nullptr;
  destroy_fn destroy = nullptr;
  set_stream_fn set_stream = nullptr;
  set_atomics_fn set_atomics = nullptr;
  gemm_sb_fn gemm_sb = nullptr;
  gemm_fn gemm = nullptr;
  int atomics = RB_ATOMICS_ALLOWED;
  int atomics_epoch = 0;
};
Blas g_blas;
std::mutex g_mu;
thread_local int t_last_status = 0;

// One handle per host thread, device AND stream: a rocBLAS handle owns one device workspace, and two GEMMs that use it (split-K
// reductions) must not run at the same time — which they do when a block's two branches issue GEMMs on two streams from one thread.
struct Handle { rb_handle h = nullptr; int epoch = -1; };
struct HandleKey {
  int dev; hipStream_t stream;
  bool operator==(const HandleKey& o) const { return dev == o.dev && stream == o.stream; }
};
struct HandleKeyHash {
  size_t operator()(const HandleKey& k) const { return std::hash<const void*>()((const void*)k.stream) * 31u + (size_t)k.dev; }
};
thread_local std::unordered_map<HandleKey, Handle, HandleKeyHash> t_handles;

}  // namespace

extern "C" {

int mm_blas_attach(const char* path) {
  if (!path) return MM_ERR_NULL;
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_blas.dl) return MM_OK;
  void* dl = dlopen(path, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);       // the copy the process already holds, if any
  if (!dl) dl = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!dl) return MM_ERR_BLAS;
  Blas b;
  b.dl = dl;
  b.create = (create_fn)dlsym(dl, "rocblas_create_handle");
  b.destroy = (destroy_fn)dlsym(dl, "rocblas_destroy_handle");
  b.set_stream = (set_stream_fn)dlsym(dl, "rocblas_set_stream");
  b.set_atomics = (set_atomics_fn)dlsym(dl, "rocblas_set_atomics_mode");
  b.gemm_sb = (gemm_sb_fn)dlsym(dl, "rocblas_gemm_strided_batched_ex");
  b.gemm = (gemm_fn)dlsym(dl, "rocblas_gemm_ex");
  if (!b.create || !b.set_stream || !b.gemm_sb || !b.gemm) return MM_ERR_BLAS;
  b.atomics = g_blas.atomics;
  g_blas = b;
  return MM_OK;
}

int mm_event_record(void* event, void* stream) {
  if (!event) return MM_ERR_NULL;
  return hipEventRecord((hipEvent_t)event, (hipStream_t)stream) == hipSuccess ? MM_OK : MM_ERR_UNSUPPORTED;
}

int mm_blas_attached(void) { return g_blas.dl != nullptr; }

int mm_blas_set_atomics(int allowed) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_blas.atomics = allowed ? RB_ATOMICS_ALLOWED : RB_ATOMICS_NOT_ALLOWED;
  ++g_blas.atomics_epoch;
  return MM_OK;
}

int mm_blas_last_status(void) { return t_last_status; }

int mm_gemm_f32(char opa, char opb, int m, int n, int k, float alpha, const float* A, int lda, int64_t stride_a, const float* B,
                int ldb, int64_t stride_b, float beta, float* C, int ldc, int64_t stride_c, int batch, int32_t solution,
                void* stream) {
  if (!g_blas.dl) return MM_ERR_BLAS;
  if (!A || !B || !C) return MM_ERR_NULL;
  if (m <= 0 || n <= 0 || k <= 0 || batch <= 0) return MM_ERR_SHAPE;
  if ((opa != 'N' && opa != 'T') || (opb != 'N' && opb != 'T')) return MM_ERR_UNSUPPORTED;
  if (lda < (opa == 'N' ? m : k) || ldb < (opb == 'N' ? k : n) || ldc < m) return MM_ERR_SHAPE;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return MM_ERR_BLAS;
  const HandleKey key{dev, (hipStream_t)stream};
  if (t_handles.find(key) == t_handles.end()) {
    // a new handle allocates its device workspace: not while `stream` is being captured into a hipGraph (the caller then issues
    // this GEMM through its own BLAS; streams that carried a GEMM before the capture keep their handle and are fine)
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)stream, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
    if (cs != hipStreamCaptureStatusNone) return MM_ERR_UNSUPPORTED;
  }
  Handle& hd = t_handles[key];
  if (!hd.h) {
    t_last_status = g_blas.create(&hd.h);
    if (t_last_status != 0) { hd.h = nullptr; t_handles.erase(key); return MM_ERR_BLAS; }
    t_last_status = g_blas.set_stream(hd.h, (hipStream_t)stream);
    if (t_last_status != 0) {
      // never keep a handle that is not bound to `stream`: later calls on this key would run on the default stream
      if (g_blas.destroy) (void)g_blas.destroy(hd.h);
      hd.h = nullptr;
      t_handles.erase(key);
      return MM_ERR_BLAS;
    }
  }
  if (hd.epoch != g_blas.atomics_epoch) {
    if (g_blas.set_atomics) (void)g_blas.set_atomics(hd.h, g_blas.atomics);
    hd.epoch = g_blas.atomics_epoch;
  }
  const int ta = opa == 'N' ? RB_OP_N : RB_OP_T, tb = opb == 'N' ? RB_OP_N : RB_OP_T;
  const int algo = solution ? RB_ALGO_SOLUTION_INDEX : RB_ALGO_STANDARD;
  if (batch == 1)
    t_last_status = g_blas.gemm(hd.h, ta, tb, m, n, k, &alpha, A, RB_F32, lda, B, RB_F32, ldb, &beta, C, RB_F32, ldc, C, RB_F32, ldc,
                                RB_F32, algo, solution, 0u);
  else
    t_last_status = g_blas.gemm_sb(hd.h, ta, tb, m, n, k, &alpha, A, RB_F32, lda, stride_a, B, RB_F32, ldb, stride_b, &beta, C, RB_F32,
                                   ldc, stride_c, C, RB_F32, ldc, stride_c, batch, RB_F32, algo, solution, 0u);
  return t_last_status == 0 ? MM_OK : MM_ERR_BLAS;
}

}  // extern "C"
