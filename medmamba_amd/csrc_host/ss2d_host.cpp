// Host-side sequencing of the SS2D branch (MedMamba.py:288-305) in C++: in_proj -> depthwise conv + SiLU -> x / dt projections
// -> 4-direction selective scan -> cross-merge -> out_norm + gate -> out_proj, and its backward, each as ONE call from Python.
//
// Why this file exists: the kernels are launched through the C ABI (include/medmamba_hip.h) and the GEMMs through ATen either
// way, but issuing them from Python costs 0.2 ms (forward) + 0.4 ms (backward) of interpreter time per block on top of the
// launches themselves (tools/host_op_profile.py: SS2DCoreFn / InProjFn / OutProjFn and their backwards), on a training step that
// is within 10 % of being bound by the host's launch rate.  The sequence below is the one medmamba_amd/ops.py holds
// (InProjFn, SS2DCoreFn with the fused depthwise conv, OutProjFn) statement by statement — same allocations, same GEMM shapes
// and strides (hence the same recorded GEMM kernels), same kernels — so both routes give the same bits; ops.py stays as the
// route for everything this file does not take (biases on the projections, d_state != 16, hooks, inference with the fused dt
// projection) and as its test oracle on the GPU.
//
// No device code and no HIP headers here: the stream is handed over as an integer, device memory as at::Tensor.
#include <torch/extension.h>

#include <algorithm>
#include <string>
#include <unordered_map>
#include <vector>

#include "medmamba_hip.h"

namespace {

using at::Tensor;
const int64_t kDim0[1] = {0};
const at::IntArrayRef DIM0(kDim0, 1);

inline void check(int rc, const char* what) {
  TORCH_CHECK(rc == 0, what, " failed: status ", rc, " (", mm_status_string(rc), ")");
}

inline const float* fp(const Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }
inline float* fpm(const Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }

inline Tensor planes(int64_t B, int64_t D, int64_t L, const at::TensorOptions& o, bool cm) {
  return cm ? at::empty({D, B, L}, o).permute({1, 0, 2}) : at::empty({B, D, L}, o);
}

inline bool is_cm(const Tensor& t) {
  const int64_t B = t.size(0), L = t.size(2);
  return t.stride(2) == 1 && t.stride(1) == B * L && (B == 1 || t.stride(0) == L);
}

// (D, B*L) matrix over the storage of a channel-major (B, D, L) tensor (a copy is made for any other layout)
inline Tensor cm2d(const Tensor& t_) {
  Tensor t = t_;
  const int64_t B = t.size(0), D = t.size(1), L = t.size(2);
  if (!is_cm(t)) t = t.permute({1, 0, 2}).contiguous().permute({1, 0, 2});
  return t.permute({1, 0, 2}).reshape({D, B * L});
}

inline Tensor rows(const Tensor& t) { return t.stride(-1) == 1 ? t : t.contiguous(); }

// ---- GEMMs: rocBLAS directly (mm_gemm_f32, the recorded solution index) where TunableOp's table names a rocBLAS solution for
// exactly this problem — 6 instead of 17-21 us of host time per call through ATen's dispatcher + TunableOp's string-keyed lookup +
// library front end — ATen otherwise (hipBLASLt winners, unrecorded shapes, no table).  The key is the one medmamba_amd/blas.py
// builds (TunableOp's own): column-major problem of out^T = b^T a^T.
std::unordered_map<std::string, int32_t> g_gemm_table;      // filled once by blas.load_table (set_gemm_table); afterwards only a value
                                                             // may change, to kDeclined (a solution the library rejected)
constexpr int32_t kDeclined = INT32_MIN;
thread_local bool g_rocblas_only = false;      // set (per host thread) around the parameter half when it runs on its own stream (ops.py, MM_PARAM_STREAM)
std::unordered_map<std::string, int32_t> g_gemm_table_rb;   // rocBLAS-only winners for shapes whose overall winner is a hipBLASLt kernel

inline bool col_operand(const Tensor& t, char& op, int64_t& ld) {
  const int64_t s0 = t.stride(-2), s1 = t.stride(-1);
  if (s1 == 1 && s0 >= t.size(-1)) { op = 'n'; ld = s0; return true; }
  if (s0 == 1 && s1 >= t.size(-2)) { op = 't'; ld = s1; return true; }
  return false;
}

// out = a @ b; a (m, k) or (B, m, k), b (k, n) or (B, k, n), out (m, n) or (B, m, n): a 2-D operand of a 3-D product is shared
inline void gemm_out(Tensor& out, const Tensor& a, const Tensor& b, void* stream) {
  if (g_rocblas_only && out.stride(-1) == 1) {
    // the parameter half on its own stream (MM_PARAM_STREAM): rocBLAS kernels only — the rocBLAS-only record first, the main record if
    // its winner is a rocBLAS solution, rocBLAS's own pick (solution 0) otherwise
    char opa, opb;
    int64_t lda, ldb;
    if (col_operand(b, opa, lda) && col_operand(a, opb, ldb)) {
      const int64_t m = a.size(-2), k = a.size(-1), n = b.size(-1), ldc = out.stride(-2);
      const bool batched = out.dim() == 3;
      std::string key;
      key.reserve(64);
      key += batched ? 'B' : 'N'; key += opa; key += opb;
      key += '_'; key += std::to_string(n); key += '_'; key += std::to_string(m); key += '_'; key += std::to_string(k);
      if (batched) { key += "_B_"; key += std::to_string(out.size(0)); }
      key += "_ld_"; key += std::to_string(lda); key += '_'; key += std::to_string(ldb); key += '_'; key += std::to_string(ldc);
      int32_t sol = 0;
      auto it = g_gemm_table_rb.find(key);
      if (it != g_gemm_table_rb.end() && it->second != kDeclined) sol = it->second;
      else if ((it = g_gemm_table.find(key)) != g_gemm_table.end() && it->second != kDeclined) sol = it->second;
      const int64_t sa = b.dim() == 3 ? b.stride(0) : 0, sb = a.dim() == 3 ? a.stride(0) : 0, sc = batched ? out.stride(0) : 0;
      int rc = mm_gemm_f32((char)(opa - 32), (char)(opb - 32), (int)n, (int)m, (int)k, 1.0f, fp(b), (int)lda, sa, fp(a), (int)ldb, sb,
                           0.0f, fpm(out), (int)ldc, sc, batched ? (int)out.size(0) : 1, sol, stream);
      if (rc == MM_ERR_BLAS && sol != 0)
        rc = mm_gemm_f32((char)(opa - 32), (char)(opb - 32), (int)n, (int)m, (int)k, 1.0f, fp(b), (int)lda, sa, fp(a), (int)ldb, sb, 0.0f,
                         fpm(out), (int)ldc, sc, batched ? (int)out.size(0) : 1, 0, stream);
      if (rc == MM_OK) return;
    }
  }
  if (!g_gemm_table.empty() && !g_rocblas_only && out.stride(-1) == 1) {
    char opa, opb;
    int64_t lda, ldb;
    if (col_operand(b, opa, lda) && col_operand(a, opb, ldb)) {
      const int64_t m = a.size(-2), k = a.size(-1), n = b.size(-1), ldc = out.stride(-2);
      const bool batched = out.dim() == 3;
      std::string key;
      key.reserve(64);
      key += batched ? 'B' : 'N'; key += opa; key += opb;
      key += '_'; key += std::to_string(n); key += '_'; key += std::to_string(m); key += '_'; key += std::to_string(k);
      if (batched) { key += "_B_"; key += std::to_string(out.size(0)); }
      key += "_ld_"; key += std::to_string(lda); key += '_'; key += std::to_string(ldb); key += '_'; key += std::to_string(ldc);
      const auto it = g_gemm_table.find(key);
      if (it != g_gemm_table.end() && it->second != kDeclined) {
        const int64_t sa = b.dim() == 3 ? b.stride(0) : 0, sb = a.dim() == 3 ? a.stride(0) : 0, sc = batched ? out.stride(0) : 0;
        const int rc = mm_gemm_f32((char)(opa - 32), (char)(opb - 32), (int)n, (int)m, (int)k, 1.0f, fp(b), (int)lda, sa, fp(a), (int)ldb,
                                   sb, 0.0f, fpm(out), (int)ldc, sc, batched ? (int)out.size(0) : 1, it->second, stream);
        if (rc == MM_OK) return;
        // BLAS: the library rejected the recorded solution (an index of another architecture / build): forget the record, ATen from
        // here on (the value is overwritten, never erased: the autograd thread may be looking at the table too)
        if (rc == MM_ERR_BLAS) it->second = kDeclined;
        else if (rc != MM_ERR_UNSUPPORTED) check(rc, "mm_gemm_f32");  // UNSUPPORTED: a first call on a capturing stream -> ATen below
      }
    }
  }
  if (out.dim() == 3) {
    const int64_t nb = out.size(0);
    at::bmm_out(out, a.dim() == 3 ? a : a.unsqueeze(0).expand({nb, -1, -1}), b.dim() == 3 ? b : b.unsqueeze(0).expand({nb, -1, -1}));
  } else {
    at::mm_out(out, a, b);
  }
}

// t.sum(0) through mm_sum_lead for dense fp32 tensors (ops.sum_lead: the same call, the same conditions — both routes give the
// same bits); `out` (optional) must be dense with the shape of t[0]
inline Tensor sum_lead(const Tensor& t, void* stream, const Tensor& out_ = Tensor()) {
  const int64_t n = t.dim() >= 1 ? t.size(0) : 0;
  if (t.dim() >= 2 && n >= 2 && n <= 4096 && t.is_contiguous() && t.numel() > 0 && (!out_.defined() || (out_.is_contiguous() && out_.numel() * n == t.numel()))) {
    Tensor out = out_.defined() ? out_ : at::empty(t.sizes().slice(1), t.options());
    const int64_t ninner = t.numel() / n;
    check(mm_sum_lead(fp(t), fpm(out), (int)n, ninner, ninner, stream), "mm_sum_lead");
    return out;
  }
  if (out_.defined()) { Tensor o = out_; at::sum_out(o, t, DIM0); return o; }
  return t.sum(0);
}

inline Tensor gemm_new(const Tensor& a, const Tensor& b, void* stream) {      // batched product into a fresh (B, m, n) tensor
  Tensor out = at::empty({a.dim() == 3 ? a.size(0) : b.size(0), a.size(-2), b.size(-1)}, a.options());
  gemm_out(out, a, b, stream);
  return out;
}

void set_gemm_table_rb(const std::vector<std::pair<std::string, int64_t>>& entries) {
  g_gemm_table_rb.clear();
  for (const auto& e : entries) g_gemm_table_rb[e.first] = (int32_t)e.second;
}

// Are the four weight-gradient GEMMs of a channel-major block (ss2d_bwd_params, cm = true) all covered by an EXPLICIT rocBLAS solution?
// Only then may that half run on a stream of its own (ops.py, MM_PARAM_STREAM): rocBLAS's own pick (solution 0) may be a hipBLASLt kernel.
bool ss2d_params_covered(int64_t Bsz, int64_t L, int64_t dm, int64_t D, int64_t C, int64_t R) {
  const std::string Q = std::to_string(Bsz * L), sD = std::to_string(D), sdm = std::to_string(dm);
  const std::string keys[4] = {
      "Ntn_" + sD + "_" + sdm + "_" + Q + "_ld_" + Q + "_" + Q + "_" + sD,                                             // d(out_proj.weight)
      "Btn_" + std::to_string(R) + "_" + sD + "_" + Q + "_B_4_ld_" + Q + "_" + Q + "_" + std::to_string(R),             // d(dt_projs_weight)
      "Btn_" + sD + "_" + std::to_string(2 * C) + "_" + Q + "_B_2_ld_" + Q + "_" + Q + "_" + sD,                        // d(x_proj_weight)
      "Nnn_" + sdm + "_" + std::to_string(2 * D) + "_" + Q + "_ld_" + sdm + "_" + Q + "_" + sdm};                       // d(in_proj.weight)
  for (const auto& k : keys) {
    auto it = g_gemm_table_rb.find(k);
    if (it != g_gemm_table_rb.end() && it->second != kDeclined) continue;
    it = g_gemm_table.find(k);
    if (it != g_gemm_table.end() && it->second != kDeclined) continue;
    return false;
  }
  return true;
}

void set_gemm_table(const std::vector<std::pair<std::string, int64_t>>& entries) {
  g_gemm_table.clear();
  for (const auto& e : entries) g_gemm_table.emplace(e.first, (int32_t)e.second);
}

struct Seg { Tensor Wx, Wdt, A, Dp, bias; int64_t oA, oD, ob; };
inline Seg segments(const Tensor& P, int64_t D, int64_t C, int64_t R, int64_t N) {
  auto al = [](int64_t n) { return (n + 63) & ~int64_t(63); };        // segments start on 256-B boundaries (mm_ss2d_pack_fwd)
  const int64_t o1 = al(4 * C * D), o2 = al(o1 + 4 * D * R), o3 = al(o2 + 4 * D * N), o4 = al(o3 + 4 * D);
  Seg s;
  s.Wx = P.narrow(0, 0, 4 * C * D).view({4, C, D});
  s.Wdt = P.narrow(0, o1, 4 * D * R).view({4, D, R});
  s.A = P.narrow(0, o2, 4 * D * N).view({4 * D, N});
  s.Dp = P.narrow(0, o3, 4 * D);
  s.bias = P.narrow(0, o4, 4 * D);
  s.oA = o2; s.oD = o3; s.ob = o4;
  return s;
}

inline void fill_common(mm_scan_args& a, const Tensor& u, const Tensor& delta, const Tensor& A, const Tensor& Bm, const Tensor& Cm,
                        const Tensor& Dp, const Tensor& bias) {
  a.struct_size = sizeof(mm_scan_args);
  a.batch = (int)u.size(0); a.dim = (int)A.size(0); a.L = (int)u.size(2); a.N = (int)A.size(1); a.G = (int)Bm.size(1);
  a.delta_softplus = 1;
  a.u = fp(u); a.delta = fp(delta); a.A = fp(A); a.B = fp(Bm); a.C = fp(Cm); a.D = fp(Dp); a.delta_bias = fp(bias);
  a.u_sb = u.stride(0); a.u_sd = u.stride(1);
  if (delta.defined()) { a.delta_sb = delta.stride(0); a.delta_sd = delta.stride(1); }
  a.B_sb = Bm.stride(0); a.B_sg = Bm.stride(1); a.B_sn = Bm.stride(2);
  a.C_sb = Cm.stride(0); a.C_sg = Cm.stride(1); a.C_sn = Cm.stride(2);
  a.u_groups = 2; a.u_map = 0x1100u; a.rev_mask = 0b1010u;          // SS2D: 2 image orders, 4 directions (selective_scan_interface._CROSS_SHARED)
}

// uninitialised fp32 (lead, *dst.shape) whose planes have the same dimension order in memory as the view `dst` (dense)
inline Tensor like_strided(const Tensor& dst, int64_t lead) {
  const int64_t nd = dst.dim();
  std::vector<int64_t> order(nd);
  for (int64_t i = 0; i < nd; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int64_t x, int64_t y) { return dst.stride(x) > dst.stride(y); });
  std::vector<int64_t> shape{lead}, perm(nd + 1);
  for (int64_t i = 0; i < nd; ++i) shape.push_back(dst.size(order[i]));
  perm[0] = 0;
  for (int64_t pos = 0; pos < nd; ++pos) perm[order[pos] + 1] = pos + 1;
  return at::empty(shape, dst.options()).permute(perm);
}

// dst = planes.sum(0) for the dense planes of like_strided(dst, W): through mm_sum_lead_chunks when the view `dst` is a sequence of
// dense chunks a constant stride apart (selective_scan_interface._chunks_of: the same test, the same call), else ATen
inline void sum_planes(Tensor& dst, const Tensor& planes, void* stream) {
  std::vector<std::pair<int64_t, int64_t>> dims;                 // (stride, size), outermost first
  for (int64_t i = 0; i < dst.dim(); ++i)
    if (dst.size(i) > 1) dims.emplace_back(dst.stride(i), dst.size(i));
  std::sort(dims.begin(), dims.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
  int64_t chunk = 1;
  size_t i = dims.size();
  while (i > 0 && dims[i - 1].first == chunk) { chunk *= dims[i - 1].second; --i; }
  int64_t stride = chunk, n = 1;
  bool ok = true;
  if (i > 0) {
    stride = dims[i - 1].first;
    for (size_t j = i; j-- > 0;) {
      if (dims[j].first != stride * n) { ok = false; break; }
      n *= dims[j].second;
    }
    ok = ok && stride >= chunk;
  }
  const int64_t W = planes.size(0);
  if (ok && W >= 2 && W <= 4096) {
    check(mm_sum_lead_chunks(fp(planes), fpm(dst), (int)W, n, chunk, stride, stream), "mm_sum_lead_chunks");
  } else {
    at::sum_out(dst, planes, DIM0);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// forward.  Returns {out (B, d_model, L), xz, u2, x_dbl, delta, P, x_chk, m, mu, rstd, y}
// ---------------------------------------------------------------------------------------------------------------------
std::vector<Tensor> ss2d_fwd(const Tensor& x, const Tensor& in_w, const Tensor& conv_w, const c10::optional<Tensor>& conv_b_,
                             const Tensor& x_proj_w, const Tensor& dt_w, const Tensor& dt_b, const Tensor& A_logs, const Tensor& Ds,
                             const Tensor& ln_w, const Tensor& ln_b, const Tensor& out_w, int64_t H, int64_t W, double eps, bool cm,
                             bool need_grad, int64_t variant, int64_t stream_, int64_t ev0, int64_t ev1, py::object prescan,
                             bool fuse_dt) {
  void* stream = reinterpret_cast<void*>(stream_);
  const Tensor conv_b = conv_b_.has_value() ? *conv_b_ : Tensor();
  const int64_t Bsz = x.size(0), L = x.size(1), dm = x.size(2);
  const int64_t D = in_w.size(0) / 2, R = dt_w.size(2), N = A_logs.size(1), C = R + 2 * N, Q = Bsz * L;
  TORCH_CHECK(L == H * W && x.is_contiguous() && in_w.size(1) == dm && N == 16, "ss2d_fwd: unexpected shapes");
  const auto o = x.options();
  // in_proj (:291-292): the two channel-first halves x | z of one buffer
  Tensor xz;
  if (cm) {
    xz = at::empty({2 * D, Q}, o);
    gemm_out(xz, in_w, x.view({Q, dm}).t(), stream);
    xz = xz.view({2 * D, Bsz, L}).permute({1, 0, 2});
  } else {
    xz = at::empty({Bsz, 2 * D, L}, o);
    gemm_out(xz, in_w, x.transpose(1, 2), stream);
  }
  const Tensor x_cf = xz.narrow(1, 0, D), z_cf = xz.narrow(1, D, D);
  // depthwise conv3x3 + SiLU in both image orders (:294-295, :256)
  Tensor u2 = planes(Bsz, 2 * D, L, o, cm);
  check(mm_dwconv_silu_cross_fwd(fp(x_cf), x_cf.stride(0), x_cf.stride(1), fp(conv_w), fp(conv_b), fpm(u2), u2.stride(0), u2.stride(1),
                                 (int)Bsz, (int)D, (int)H, (int)W, stream), "mm_dwconv_silu_cross_fwd");
  // parameters in kernel direction order, A = -exp(A_logs) (:271)
  Tensor P = at::empty({mm_ss2d_pack_size((int)D, (int)C, (int)R, (int)N)}, o);
  check(mm_ss2d_pack_fwd(fp(x_proj_w), fp(dt_w), fp(dt_b), fp(A_logs), fp(Ds), fpm(P), (int)D, (int)C, (int)R, (int)N, stream),
        "mm_ss2d_pack_fwd");
  const Seg s = segments(P, D, C, R, N);
  Tensor x_dbl, xb, delta;
  if (cm) {
    const Tensor u2m = cm2d(u2);                                                          // (2D, Q) view
    x_dbl = at::empty({2, 2 * C, Q}, o);
    gemm_out(x_dbl, s.Wx.view({2, 2 * C, D}), u2m.view({2, D, Q}), stream);                 // :259
    x_dbl = x_dbl.view({4, C, Q});
    xb = x_dbl.view({4, C, Bsz, L}).permute({2, 0, 1, 3});                                // (B, 4, C, L) view
  } else {
    x_dbl = at::matmul(s.Wx.view({1, 2, 2 * C, D}), u2.view({Bsz, 2, D, L})).view({Bsz, 4, C, L});
    xb = x_dbl;
  }
  // inference (nothing to differentiate) with a small dt rank: the dt projection (:262) runs inside the scan's staging phase — no
  // (B, 4D, L) delta tensor is written or read, one GEMM less; training keeps the GEMM (the backward kernel consumes delta)
  const Tensor dts = xb.narrow(2, 0, R);
  fuse_dt = fuse_dt && !need_grad && R > 0 && R <= mm_scan_dt_max() && L % 4 == 0 && reinterpret_cast<uintptr_t>(dts.data_ptr()) % 16 == 0 &&
            dts.stride(3) == 1 && dts.stride(0) % 4 == 0 && dts.stride(1) % 4 == 0 && dts.stride(2) % 4 == 0;
  if (!fuse_dt && cm) {
    delta = at::empty({4, D, Q}, o);
    gemm_out(delta, s.Wdt, x_dbl.narrow(1, 0, R), stream);                                // :262
    delta = delta.view({4 * D, Bsz, L}).permute({1, 0, 2});
  } else if (!fuse_dt) {
    delta = at::matmul(s.Wdt.unsqueeze(0), x_dbl.narrow(2, 0, R)).view({Bsz, 4 * D, L});
  }
  if (!prescan.is_none()) prescan.attr("record")();       // the projections are queued; what follows is the latency-bound scan
  // the scan (:273-279) without materialising the cross-scan
  Tensor out4 = at::empty({Bsz, 4 * D, L}, o), x_chk;
  const int chunk = mm_scan_chunk();
  if (need_grad) x_chk = at::empty({Bsz, (L + chunk - 1) / chunk, 4 * D, N}, o);
  {
    mm_scan_args a = {};
    fill_common(a, u2, delta, s.A, xb.narrow(2, R, N), xb.narrow(2, R + N, N), s.Dp, s.bias);
    a.out = fpm(out4); a.x_chk = fpm(x_chk); a.variant = (int32_t)variant;
    if (fuse_dt) {
      a.dt_w = fp(s.Wdt); a.dts = fp(dts); a.dt_rank = (int32_t)R;
      a.dts_sb = dts.stride(0); a.dts_sg = dts.stride(1); a.dts_sn = dts.stride(2);
    }
    if (ev0) check(mm_event_record(reinterpret_cast<void*>(ev0), stream), "mm_event_record");
    check(mm_scan_fwd(&a, stream), "mm_scan_fwd");
    if (ev1) check(mm_event_record(reinterpret_cast<void*>(ev1), stream), "mm_event_record");
  }
  // cross-merge (:282-286, 298), out_norm + gate (:299-301)
  Tensor m = planes(Bsz, D, L, o, cm), y = planes(Bsz, D, L, o, cm);
  Tensor mu = at::empty({Bsz, L}, o), rstd = at::empty({Bsz, L}, o);
  check(mm_cross_merge_fwd(fp(out4), fpm(m), m.stride(0), m.stride(1), (int)Bsz, (int)D, (int)H, (int)W, stream), "mm_cross_merge_fwd");
  check(mm_ln_gate_fwd(fp(m), m.stride(0), m.stride(1), fp(z_cf), z_cf.stride(0), z_cf.stride(1), fp(ln_w), fp(ln_b), (float)eps, fpm(y),
                       y.stride(0), y.stride(1), fpm(mu), fpm(rstd), (int)Bsz, (int)D, (int)L, stream), "mm_ln_gate_fwd");
  // out_proj (:302)
  Tensor out;
  if (cm && Bsz > 1) {
    out = at::empty({out_w.size(0), Q}, o);
    gemm_out(out, out_w, cm2d(y), stream);
    out = out.view({-1, Bsz, L}).permute({1, 0, 2});
  } else {
    out = at::empty({Bsz, out_w.size(0), L}, o);
    gemm_out(out, out_w, y, stream);
  }
  if (!need_grad) return {out};
  return {out, xz, u2, x_dbl, delta, P, x_chk, m, mu, rstd, y};
}

// ---------------------------------------------------------------------------------------------------------------------
// backward, in two halves: what d(input) waits for, and the parameter gradients that nothing on the way to d(input) needs (the five
// weight-gradient GEMMs and the un-packing launch).  ss2d_bwd issues both on one stream.  Same kernels, same operands, same bits.
// ss2d_bwd_data returns {dx (B, L, d_model), dxz, ddelta, dx_dbl, parts, ws, wsc} — dx and what the parameter half reads.
// ---------------------------------------------------------------------------------------------------------------------
std::vector<Tensor> ss2d_bwd_data(const Tensor& dout_, const Tensor& x, const Tensor& in_w, const Tensor& conv_w,
                                  const c10::optional<Tensor>& conv_b_, const Tensor& ln_w, const Tensor& ln_b, const Tensor& out_w,
                                  const Tensor& xz, const Tensor& u2, const Tensor& x_dbl, const Tensor& delta, const Tensor& P,
                                  const Tensor& x_chk, const Tensor& m, const Tensor& mu, const Tensor& rstd, int64_t H, int64_t W, bool cm,
                                  int64_t variant, int64_t stream_, int64_t ev0, int64_t ev1) {
  void* stream = reinterpret_cast<void*>(stream_);
  const Tensor conv_b = conv_b_.has_value() ? *conv_b_ : Tensor();
  const int64_t Bsz = x.size(0), L = x.size(1), dm = x.size(2);
  const int64_t D = in_w.size(0) / 2, N = 16;
  const int64_t C = cm ? x_dbl.size(1) : x_dbl.size(2), R = C - 2 * N, Q = Bsz * L;
  const auto o = x.options();
  const Seg s = segments(P, D, C, R, N);
  const Tensor x_cf = xz.narrow(1, 0, D), z_cf = xz.narrow(1, D, D);
  // out_proj backward
  Tensor dy;
  if (cm && Bsz > 1 && is_cm(dout_)) {
    const Tensor g2 = cm2d(dout_);                                                        // (d_model, Q)
    dy = at::empty({D, Q}, o);
    gemm_out(dy, out_w.t(), g2, stream);
    dy = dy.view({D, Bsz, L}).permute({1, 0, 2});
  } else {
    const Tensor g = rows(dout_);
    dy = at::empty({Bsz, D, L}, o);
    gemm_out(dy, out_w.t(), g, stream);
  }
  // out_norm + gate backward, its plane transpose for the column-major directions
  Tensor dout2 = planes(Bsz, 2 * D, L, o, cm);            // channel block 0: dm, block 1: its plane transpose
  Tensor dxz = planes(Bsz, 2 * D, L, o, cm);              // d(x_cf) | d(z_cf), the layout in_proj produced them in
  Tensor dz = dxz.narrow(1, D, D);
  Tensor ws = at::empty({mm_ln_gate_rows((int)Bsz, (int)D, (int)L), 2 * D}, o);
  check(mm_ln_gate_bwd(fp(dy), dy.stride(0), dy.stride(1), fp(m), m.stride(0), m.stride(1), fp(z_cf), z_cf.stride(0), z_cf.stride(1),
                       fp(ln_w), fp(ln_b), fp(mu), fp(rstd), fpm(dout2), dout2.stride(0), dout2.stride(1), fpm(dz), dz.stride(0),
                       dz.stride(1), fpm(ws), (int)Bsz, (int)D, (int)L, stream), "mm_ln_gate_bwd");
  {
    Tensor d1 = dout2.narrow(1, D, D);
    check(mm_plane_transpose(fp(dout2), dout2.stride(0), dout2.stride(1), fpm(d1), d1.stride(0), d1.stride(1), (int)Bsz, (int)D, (int)H,
                             (int)W, stream), "mm_plane_transpose");
  }
  const int64_t S = mm_ss2d_pack_parts_size((int)D, (int)C, (int)R, (int)N);
  Tensor parts = at::empty({Bsz, S}, o);
  Tensor dx_dbl, xb, dxb;
  if (cm) {
    dx_dbl = at::empty({4, C, Q}, o);
    xb = x_dbl.view({4, C, Bsz, L}).permute({2, 0, 1, 3});
    dxb = dx_dbl.view({4, C, Bsz, L}).permute({2, 0, 1, 3});
  } else {
    dx_dbl = at::empty({Bsz, 4, C, L}, o);
    xb = x_dbl; dxb = dx_dbl;
  }
  // scan backward (selective_scan_interface._launch_bwd)
  Tensor du4 = planes(Bsz, 4 * D, L, o, cm), ddelta = planes(Bsz, 4 * D, L, o, cm);
  {
    const Tensor Bm = xb.narrow(2, R, N), Cm = xb.narrow(2, R + N, N);
    Tensor dBC = dxb.narrow(2, R, 2 * N);                  // rows [0, N): dB, rows [N, 2N): dC; fully overwritten
    mm_scan_args a = {};
    fill_common(a, u2, delta, s.A, Bm, Cm, s.Dp, s.bias);
    a.x_chk = fpm(x_chk); a.dout = fp(dout2);
    a.du = fpm(du4); a.ddelta = fpm(ddelta);
    a.dA = fpm(parts); a.dD = fpm(parts) + (s.oD - s.oA); a.ddelta_bias = fpm(parts) + (s.ob - s.oA);
    a.dpar_sb = parts.stride(0);
    a.dout_sb = dout2.stride(0); a.dud_sb = du4.stride(0); a.o_sd = du4.stride(1);
    TORCH_CHECK(dout2.stride(1) == du4.stride(1), "dout and du / ddelta must share the channel stride");
    a.variant = (int32_t)variant;
    auto point = [&](const Tensor& t, int64_t plane_stride) {
      a.dB = fpm(t); a.dC = fpm(t) + N * t.stride(2);
      a.dB_sb = a.dC_sb = t.stride(0); a.dB_sg = a.dC_sg = t.stride(1); a.dB_sn = a.dC_sn = t.stride(2);
      a.dBC_sc = plane_stride;
    };
    point(dBC, 0);
    int32_t plan[8];
    check(mm_scan_plan(&a, 1, plan), "mm_scan_plan");
    Tensor pl;
    if (plan[6] > 1) {        // a direction is shared by W workgroups: one partial plane each, summed below
      pl = like_strided(dBC, plan[6]);
      point(pl.select(0, 0), pl.stride(0));
    }
    if (ev0) check(mm_event_record(reinterpret_cast<void*>(ev0), stream), "mm_event_record");
    check(mm_scan_bwd(&a, stream), "mm_scan_bwd");
    if (ev1) check(mm_event_record(reinterpret_cast<void*>(ev1), stream), "mm_event_record");
    if (pl.defined()) sum_planes(dBC, pl, stream);
  }
  Tensor du2;
  if (cm) {
    const Tensor dd = ddelta.permute({1, 0, 2}).reshape({4, D, Q});                        // views of (4D, B, L) storage
    Tensor dxr = dx_dbl.narrow(1, 0, R);
    gemm_out(dxr, s.Wdt.transpose(1, 2), dd, stream);                                     // dt rows of d(x_dbl), in place
    const Tensor dx2 = dx_dbl.view({2, 2 * C, Q});
    Tensor du2m = at::empty({2, D, Q}, o);
    gemm_out(du2m, s.Wx.view({2, 2 * C, D}).transpose(1, 2), dx2, stream);                 // Wx^T d(x_dbl); pairs added later
    du2 = du2m.view({2 * D, Bsz, L}).permute({1, 0, 2});
  } else {
    const Tensor dd = ddelta.view({Bsz, 4, D, L});
    dx_dbl.narrow(2, 0, R).copy_(at::matmul(s.Wdt.transpose(-1, -2).unsqueeze(0), dd));    // dt rows of d(x_dbl)
    const Tensor Wx2 = s.Wx.view({2, 2 * C, D});
    const Tensor dxd2 = dx_dbl.view({Bsz, 2, 2 * C, L});
    const Tensor WxT = Wx2.transpose(1, 2).unsqueeze(0).expand({Bsz, -1, -1, -1}).reshape({Bsz * 2, D, 2 * C});
    du2 = at::bmm(WxT, dxd2.reshape({Bsz * 2, 2 * C, L}));                                 // Wx^T d(x_dbl); pairs added later
    du2 = du2.view({Bsz, 2 * D, L});
  }
  // d(u2) = projection part + the scan's two direction pairs, summed inside the depthwise conv's backward kernel
  Tensor dxc = dxz.narrow(1, 0, D);
  const int strips = mm_dwconv_silu_cross_strips((int)H, (int)W);
  Tensor wsc = at::empty({Bsz, D * strips, 10}, o);
  check(mm_dwconv_silu_cross_bwd(fp(du2), du2.stride(0), du2.stride(1), fp(du4), du4.stride(0), du4.stride(1), fp(x_cf), x_cf.stride(0),
                                 x_cf.stride(1), fp(conv_w), fp(conv_b), fpm(dxc), dxc.stride(0), dxc.stride(1), fpm(wsc), (int)Bsz, (int)D,
                                 (int)H, (int)W, stream), "mm_dwconv_silu_cross_bwd");
  // in_proj backward (data)
  Tensor dx;
  if (cm && is_cm(dxz)) {
    dx = at::empty({Q, dm}, o);
    gemm_out(dx, cm2d(dxz).t(), in_w, stream);
    dx = dx.view({Bsz, L, dm});
  } else {
    dx = at::empty({Bsz, L, dm}, o);
    gemm_out(dx, dxz.transpose(1, 2), in_w, stream);
  }
  return {dx, dxz, ddelta, dx_dbl, parts, ws, wsc};
}

// The parameter half.  Returns {d in_w, d conv_w, d conv_b, d x_proj_w, d dt_w, d dt_b, d A_logs, d Ds, d ln_w, d ln_b, d out_w}.
std::vector<Tensor> ss2d_bwd_params(const Tensor& dout_, const Tensor& x, const Tensor& in_w, const c10::optional<Tensor>& conv_b_,
                                    const Tensor& out_w, const Tensor& u2, const Tensor& x_dbl, const Tensor& P, const Tensor& y,
                                    const Tensor& dxz, const Tensor& ddelta, const Tensor& dx_dbl, const Tensor& parts, const Tensor& ws,
                                    const Tensor& wsc, int64_t H, int64_t W, bool cm, bool pack_fold, int64_t stream_) {
  void* stream = reinterpret_cast<void*>(stream_);
  const Tensor conv_b = conv_b_.has_value() ? *conv_b_ : Tensor();
  const int64_t Bsz = x.size(0), L = x.size(1), dm = x.size(2);
  const int64_t D = in_w.size(0) / 2, N = 16;
  const int64_t C = cm ? x_dbl.size(1) : x_dbl.size(2), R = C - 2 * N, Q = Bsz * L;
  const auto o = x.options();
  Tensor d_out_w;
  if (cm && Bsz > 1 && is_cm(dout_)) {
    d_out_w = at::empty({out_w.size(0), D}, o);
    gemm_out(d_out_w, cm2d(dout_), cm2d(y).t(), stream);
  } else {
    d_out_w = sum_lead(gemm_new(rows(dout_), rows(y).transpose(1, 2), stream), stream);
  }
  // packed parameter gradients: the GEMMs write the two weight segments, the scan kernel left per-batch-item partials of A / D / bias
  Tensor dP = at::empty_like(P);
  const Seg ds = segments(dP, D, C, R, N);
  if (cm) {
    const Tensor dd = ddelta.permute({1, 0, 2}).reshape({4, D, Q});                        // views of (4D, B, L) storage
    Tensor dWdt = ds.Wdt;
    gemm_out(dWdt, dd, x_dbl.narrow(1, 0, R).transpose(1, 2), stream);                     // (4, D, R)
    Tensor dWx = ds.Wx.view({2, 2 * C, D});
    gemm_out(dWx, dx_dbl.view({2, 2 * C, Q}), cm2d(u2).view({2, D, Q}).transpose(1, 2), stream);
  } else {
    Tensor dWdt = ds.Wdt;
    sum_lead(at::matmul(ddelta.view({Bsz, 4, D, L}), x_dbl.narrow(2, 0, R).transpose(-1, -2)), stream, dWdt);   // (4, D, R)
    Tensor dWx = ds.Wx.view({2, 2 * C, D});
    sum_lead(at::matmul(dx_dbl.view({Bsz, 2, 2 * C, L}), u2.view({Bsz, 2, D, L}).transpose(-1, -2)), stream, dWx);
  }
  const int strips = mm_dwconv_silu_cross_strips((int)H, (int)W);
  // one launch: packed gradients back to the module's layouts, ln_gate's partial rows, the depthwise conv's partial sums
  const int64_t npk = P.numel();
  Tensor G = at::empty({npk + 2 * D + 10 * D}, o);
  Tensor ln_out = G.narrow(0, npk, 2 * D), dw_out = G.narrow(0, npk + 2 * D, 10 * D);
  check(mm_ss2d_pack_bwd(fp(dP), fp(P), fp(parts), fpm(G), (int)D, (int)C, (int)R, (int)N, (int)Bsz, pack_fold ? fp(ws) : nullptr,
                         (int)ws.size(0), fpm(ln_out), pack_fold ? fp(wsc) : nullptr, (int)Bsz, strips, fpm(dw_out), stream),
        "mm_ss2d_pack_bwd");
  const Seg gs = segments(G.narrow(0, 0, npk), D, C, R, N);
  if (!pack_fold) {                                     // A/B switch (MM_PACK_FOLD=0): the reductions as separate ATen launches
    ln_out = sum_lead(ws, stream);
    const Tensor sc = wsc.view({Bsz, D, -1, 10}).sum(at::IntArrayRef({0, 2}));
    dw_out = at::cat({sc.narrow(1, 0, 9).reshape({-1}), sc.select(1, 9)});
  }
  Tensor dcw = dw_out.narrow(0, 0, 9 * D).view({D, 1, 3, 3});
  Tensor dcb = conv_b.defined() ? dw_out.narrow(0, 9 * D, D) : Tensor();
  Tensor d_in_w = at::empty_like(in_w);
  if (cm && is_cm(dxz)) gemm_out(d_in_w, cm2d(dxz), x.view({Q, dm}), stream);
  else sum_lead(gemm_new(dxz, x, stream), stream, d_in_w);
  return {d_in_w, dcw, dcb, gs.Wx, gs.Wdt, gs.bias.view({4, D}), gs.A, gs.Dp, ln_out.narrow(0, 0, D), ln_out.narrow(0, D, D), d_out_w};
}

// Both halves on one stream.  Returns {dx, d in_w, d conv_w, d conv_b, d x_proj_w, d dt_w, d dt_b, d A_logs, d Ds, d ln_w, d ln_b, d out_w}
std::vector<Tensor> ss2d_bwd(const Tensor& dout_, const Tensor& x, const Tensor& in_w, const Tensor& conv_w,
                             const c10::optional<Tensor>& conv_b_, const Tensor& ln_w, const Tensor& ln_b, const Tensor& out_w,
                             const Tensor& xz, const Tensor& u2, const Tensor& x_dbl, const Tensor& delta, const Tensor& P,
                             const Tensor& x_chk, const Tensor& m, const Tensor& mu, const Tensor& rstd, const Tensor& y, int64_t H,
                             int64_t W, bool cm, bool pack_fold, int64_t variant, int64_t stream_, int64_t ev0, int64_t ev1) {
  const std::vector<Tensor> d = ss2d_bwd_data(dout_, x, in_w, conv_w, conv_b_, ln_w, ln_b, out_w, xz, u2, x_dbl, delta, P, x_chk, m, mu,
                                              rstd, H, W, cm, variant, stream_, ev0, ev1);
  std::vector<Tensor> g = ss2d_bwd_params(dout_, x, in_w, conv_b_, out_w, u2, x_dbl, P, y, d[1], d[2], d[3], d[4], d[5], d[6], H, W, cm,
                                          pack_fold, stream_);
  g.insert(g.begin(), d[0]);
  return g;
}

// ---------------------------------------------------------------------------------------------------------------------
// The conv branch of a block (MedMamba.py:338-345 without the trailing ReLU and without the 1x1 conv's bias, both applied by
// mm_shuffle_residual_fwd): BN1 -> conv3x3 -> BN2 + ReLU -> conv3x3 -> BN3 + ReLU -> conv1x1, the sequence
// modules._conv_branch issues through BNReluFn / ConvBiasFn(add_bias=False) / PointwiseConvFn, statement by statement.
// The 3x3 convolutions stay ATen's (MIOpen); the conv biases act only through the BatchNorm running means (pre_bias).
// ---------------------------------------------------------------------------------------------------------------------
struct BnP { Tensor g, b, rm, rv; double eps, mom; };

inline std::pair<Tensor, Tensor> bn_fwd(const Tensor& x, const BnP& p, bool relu, const Tensor& pre_bias, void* stream) {
  const int64_t B = x.size(0), C = x.size(1), HW = x.numel() / (B * C);
  Tensor y = at::empty_like(x), stats = at::empty({2, C}, x.options());
  Tensor ws = at::empty({3 * C * mm_bn_splits((int)B, (int)C, (int)HW)}, x.options());
  check(mm_bn_relu_fwd(fp(x), fp(p.g), fp(p.b), (float)p.eps, (float)p.mom, fpm(p.rm), fpm(p.rv), fpm(y), fpm(stats),
                       fpm(stats) + C, fpm(ws), fp(pre_bias), relu ? 1 : 0, (int)B, (int)C, (int)HW, stream), "mm_bn_relu_fwd");
  return {y, stats};
}

inline Tensor bias_grad(const Tensor& dy, void* stream) {      // ops._bias_grad
  const int64_t B = dy.size(0), C = dy.size(1), HW = dy.numel() / (B * C);
  if (HW < 512) return dy.dim() == 4 ? dy.sum(at::IntArrayRef({0, 2, 3})) : dy.sum(at::IntArrayRef({0, 2}));
  const int split = mm_channel_sum_nchw_split((int)B, (int)C);
  Tensor out = at::empty({split, C}, dy.options());
  check(mm_channel_sum_nchw(fp(dy), fpm(out), (int)B, (int)C, (int)HW, stream), "mm_channel_sum_nchw");
  return split == 1 ? out.select(0, 0) : sum_lead(out, stream);
}

// returns {dx, dgamma, dbeta, d(pre_bias) or undefined}
inline std::vector<Tensor> bn_bwd(const Tensor& dy_, const Tensor& x, const Tensor& g, const Tensor& b, const Tensor& stats, bool relu,
                                  bool want_db, void* stream) {
  const int64_t B = x.size(0), C = x.size(1), HW = x.numel() / (B * C);
  const Tensor dy = dy_.contiguous();
  Tensor dx = at::empty_like(x);
  const bool fused = mm_bn_fused((int)B, (int)C, (int)HW) != 0;
  Tensor dgb = at::empty({3, C}, x.options());
  Tensor ws = fused ? dgb : at::empty({2 * C * mm_bn_splits((int)B, (int)C, (int)HW)}, x.options());
  check(mm_bn_relu_bwd(fp(dy), fp(x), fp(g), fp(b), fp(stats), fp(stats) + C, fpm(dx), fpm(dgb), fpm(dgb) + C, fpm(ws),
                       (want_db && fused) ? fpm(dgb) + 2 * C : nullptr, relu ? 1 : 0, (int)B, (int)C, (int)HW, stream), "mm_bn_relu_bwd");
  Tensor db;
  if (want_db) db = fused ? dgb.select(0, 2) : bias_grad(dx, stream);
  return {dx, dgb.select(0, 0), dgb.select(0, 1), db};
}

const int64_t kOne2[2] = {1, 1}, kZero2[2] = {0, 0};

// Backward of a dense 3x3 convolution (padding 1): {d input, d weight}.  Normally one at::convolution_backward (MIOpen: Winograd
// data gradient + an implicit-GEMM weight gradient that accumulates with atomics).  With torch.backends.cudnn.deterministic set
// (the reference's set_seed does, train.py:28-29) MIOpen answers the weight gradient with its per-image im2col + GEMM solver:
// 640 extra launches and +17.5 ms per MedMamba-S step.  The same arithmetic as ONE im2col and ONE batched GEMM summed over the
// batch in a fixed order (mm_im2col3x3: ATen's im2col launches once per image) is deterministic too and costs far less; the data gradient stays MIOpen's (Winograd, no atomics).
std::pair<Tensor, Tensor> conv3x3_bwd(const Tensor& dy, const Tensor& x, const Tensor& w, void* stream) {
  const at::IntArrayRef one(kOne2, 2), zero(kZero2, 2);
  if (!at::globalContext().deterministicCuDNN()) {
    auto r = at::convolution_backward(dy, x, w, c10::nullopt, one, one, one, false, zero, 1, {true, true, false});
    return {std::get<0>(r), std::get<1>(r)};
  }
  auto r = at::convolution_backward(dy, x, w, c10::nullopt, one, one, one, false, zero, 1, {true, false, false});
  const int64_t B = x.size(0), C = x.size(1), K = w.size(0), HW = x.size(2) * x.size(3);
  // gs images per GEMM: as many as still leave >= 512 output tiles for the chip (the partial products to sum shrink by gs:
  // 340 MB -> 21 MB per conv at the 7x7 stage); gs = 1 is F.unfold's layout
  const int64_t tiles = ((K + 63) / 64) * ((9 * C + 63) / 64);
  int64_t gs = 1;
  while (gs * 2 <= B && B % (gs * 2) == 0 && (B / (gs * 2)) * tiles >= 512) gs *= 2;
  Tensor cols = at::empty({B / gs, 9 * C, gs * HW}, x.options());
  check(mm_im2col3x3(fp(x), fpm(cols), (int)B, (int)C, (int)x.size(2), (int)x.size(3), (int)gs, stream), "mm_im2col3x3");
  const Tensor dyg = gs == 1 ? dy.reshape({B, K, HW})
                             : dy.reshape({B / gs, gs, K, HW}).transpose(1, 2).reshape({B / gs, K, gs * HW});   // one copy
  Tensor dw = sum_lead(at::bmm(dyg, cols.transpose(1, 2)), stream).view(w.sizes());
  return {std::get<0>(r), dw};
}

// returns {out (B, K, H, W), y1, s1, c1, y2, s2, c2, y3, s3}
std::vector<Tensor> conv_branch_fwd(const Tensor& x, const std::vector<Tensor>& bn1, const Tensor& w1, const Tensor& cb1,
                                    const std::vector<Tensor>& bn2, const Tensor& w2, const Tensor& cb2, const std::vector<Tensor>& bn3,
                                    const Tensor& w3, const std::vector<double>& eps_mom, int64_t stream_) {
  void* stream = reinterpret_cast<void*>(stream_);
  TORCH_CHECK(bn1.size() == 4 && bn2.size() == 4 && bn3.size() == 4 && eps_mom.size() == 6 && x.is_contiguous(), "conv_branch_fwd: arguments");
  const BnP p1{bn1[0], bn1[1], bn1[2], bn1[3], eps_mom[0], eps_mom[1]}, p2{bn2[0], bn2[1], bn2[2], bn2[3], eps_mom[2], eps_mom[3]},
      p3{bn3[0], bn3[1], bn3[2], bn3[3], eps_mom[4], eps_mom[5]};
  const at::IntArrayRef one(kOne2, 2);
  auto [y1, s1] = bn_fwd(x, p1, false, Tensor(), stream);
  Tensor c1 = at::conv2d(y1, w1, c10::nullopt, one, one, one, 1);
  auto [y2, s2] = bn_fwd(c1, p2, true, cb1, stream);
  Tensor c2 = at::conv2d(y2, w2, c10::nullopt, one, one, one, 1);
  auto [y3, s3] = bn_fwd(c2, p3, true, cb2, stream);
  const int64_t B = x.size(0), C = y3.size(1), H = x.size(2), W = x.size(3), K = w3.size(0);
  Tensor out = at::empty({B, K, H * W}, x.options());
  gemm_out(out, w3.view({K, C}), y3.view({B, C, H * W}), stream);
  return {out.view({B, K, H, W}), y1, s1, c1, y2, s2, c2, y3, s3};
}

// returns {dx, dg1, db1, dw1, dcb1, dg2, db2, dw2, dcb2, dg3, db3, dw3}
std::vector<Tensor> conv_branch_bwd(const Tensor& dout, const Tensor& x, const Tensor& g1, const Tensor& b1, const Tensor& w1,
                                    const Tensor& g2, const Tensor& b2, const Tensor& w2, const Tensor& g3, const Tensor& b3,
                                    const Tensor& w3, const Tensor& y1, const Tensor& s1, const Tensor& c1, const Tensor& y2,
                                    const Tensor& s2, const Tensor& c2, const Tensor& y3, const Tensor& s3, int64_t stream_) {
  void* stream = reinterpret_cast<void*>(stream_);
  const int64_t B = x.size(0), H = x.size(2), W = x.size(3), C = y3.size(1), K = w3.size(0);
  const Tensor dy = dout.contiguous().view({B, K, H * W});
  const Tensor w3v = w3.view({K, C}), y3v = y3.view({B, C, H * W});
  Tensor dy3 = at::empty({B, C, H * W}, x.options());
  gemm_out(dy3, w3v.t(), dy, stream);
  Tensor dw3 = sum_lead(gemm_new(dy, y3v.transpose(1, 2), stream), stream).view(w3.sizes());
  auto r3 = bn_bwd(dy3.view({B, C, H, W}), c2, g3, b3, s3, true, true, stream);
  auto cbw2 = conv3x3_bwd(r3[0], y2, w2, stream);
  auto r2 = bn_bwd(cbw2.first, c1, g2, b2, s2, true, true, stream);
  auto cbw1 = conv3x3_bwd(r2[0], y1, w1, stream);
  auto r1 = bn_bwd(cbw1.first, x, g1, b1, s1, false, false, stream);
  return {r1[0], r1[1], r1[2], cbw1.second, r2[3], r2[1], r2[2], cbw2.second, r3[3], r3[1], r3[2], dw3};
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, mod) {
  mod.def("conv_branch_fwd", &conv_branch_fwd);
  mod.def("conv_branch_bwd", &conv_branch_bwd);
  mod.doc() = "medmamba_amd: C++ sequencing of the SS2D branch over the C ABI of libmedmamba_hip.so";
  mod.def("ss2d_fwd", &ss2d_fwd);
  mod.def("ss2d_bwd", &ss2d_bwd);
  mod.def("set_rocblas_only", [](bool on) { g_rocblas_only = on; });
  mod.def("set_gemm_table_rb", &set_gemm_table_rb);
  mod.def("ss2d_params_covered", &ss2d_params_covered);
  mod.def("ss2d_bwd_data", &ss2d_bwd_data);
  mod.def("ss2d_bwd_params", &ss2d_bwd_params);
  mod.def("set_gemm_table", &set_gemm_table);
  mod.def("abi_version", []() { return mm_abi_version(); });
}
