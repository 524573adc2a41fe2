#!/usr/bin/env python3
"""Host-side cost of one fp32 strided-batched GEMM call: torch.bmm(out=) vs rocblas_gemm_strided_batched_ex called directly
(ctypes on the librocblas.so that torch has loaded).  Enqueue time only, 2000 calls, nothing waits for the GPU."""
import ctypes, glob, os, sys, time
import torch
dev = torch.device("cuda:0")
M, N, K, B = 192, 196, 96, 64
a = torch.randn(B, M, K, device=dev); b = torch.randn(B, K, N, device=dev); c = torch.empty(B, M, N, device=dev)
for _ in range(10): torch.bmm(a, b, out=c)
torch.cuda.synchronize()
n = 2000
t0 = time.perf_counter()
for _ in range(n): torch.bmm(a, b, out=c)
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"torch.bmm(out=): {1e6 * (t1 - t0) / n:.1f} us per call (enqueue)")
w = torch.randn(M, K, device=dev)
t0 = time.perf_counter()
for _ in range(n): torch.matmul(w, b, out=c)
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"torch.matmul(w, b, out=) broadcast: {1e6 * (t1 - t0) / n:.1f} us per call")
lib = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librocblas.so"))
h = ctypes.c_void_p()
assert lib.rocblas_create_handle(ctypes.byref(h)) == 0
assert lib.rocblas_set_stream(h, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
f = lib.rocblas_gemm_strided_batched_ex
f.restype = ctypes.c_int
vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
f.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp, i32, i32, i64, vp, i32, i32, i64, vp, vp, i32, i32, i64, vp, i32, i32, i64, i32,
              i32, i32, i32, ctypes.c_uint32]
one, zero = ctypes.c_float(1.0), ctypes.c_float(0.0)
F32 = 151   # rocblas_datatype_f32_r
NONE = 111  # rocblas_operation_none
# row-major C = A B  ==  column-major C^T = B^T A^T: (N x M) = (N x K)(K x M)
def call():
    return f(h, NONE, NONE, N, M, K, ctypes.byref(one), b.data_ptr(), F32, N, K * N, a.data_ptr(), F32, K, M * K, ctypes.byref(zero),
             c.data_ptr(), F32, N, M * N, c.data_ptr(), F32, N, M * N, B, F32, 0, 0, 0)
assert call() == 0
torch.cuda.synchronize()
ref = torch.bmm(a, b)
print("max err", float((c - ref).abs().max()))
t0 = time.perf_counter()
for _ in range(n): call()
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"rocblas_gemm_strided_batched_ex via ctypes: {1e6 * (t1 - t0) / n:.1f} us per call (enqueue, includes ~3 us of ctypes)")
