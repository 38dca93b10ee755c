// pplie_autograd.cpp -- native autograd nodes for the row operators of libpplie.
//
// pypose/lietensor/operation.py:304-1113 defines every Lie operator as a Python torch.autograd.Function; so does
// pypose_amd/lietensor/operation.py, with one HIP kernel per forward / backward.  At small batch sizes (BASELINE configs[0]: 1024
// rows) the kernels take microseconds and a step of autograd through a PYTHON Function costs ~35 us per backward node (the engine's
// device thread has to take the GIL, build the Python context, call back) -- measured: 130 us for the two backward nodes of
// Exp().Log() against 55 us for two native nodes of the same size.  This extension is the same node in C++:
//
//     forward : out = kernel_fwd(inputs)          saved: the inputs / the output the backward kernel reads (table in operation.py)
//     backward: grads = kernel_bwd(saved..., grad_out)
//
// The kernels are reached through function pointers that Python resolves from the already loaded libpplie.so (no link-time
// dependency; the C ABI of every row operator is  fn(in0, in1, in2, out0, out1, n, stream), include/pplie.h).  Anything that is not
// the plain eager case -- functorch transforms, the optimizer's op tracers, dry traces, broadcasting, non-contiguous or host
// tensors -- stays on the Python Functions; a backward that itself has to be differentiated (create_graph=True) calls back into the
// Python rule registered for the operator, which is written in differentiable torch operations.
#include <torch/extension.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>      // (ROCm builds of torch call the device type "cuda")
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <unordered_map>
#include <vector>

namespace {
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;
typedef int (*rowfn_t)(const void*, const void*, const void*, void*, void*, int64_t, void*);

std::unordered_map<int64_t, py::object>& py_rules() {
  static std::unordered_map<int64_t, py::object>* m = new std::unordered_map<int64_t, py::object>();     // (never destroyed: no GIL at exit)
  return *m;
}

std::vector<at::Tensor> launch(int64_t fn_addr, const std::vector<at::Tensor>& ins, const std::vector<int64_t>& out_w) {
  const at::Tensor& x0 = ins[0];
  TORCH_CHECK(ins.size() >= 1 && ins.size() <= 3 && out_w.size() >= 1 && out_w.size() <= 2, "pplie: 1-3 inputs, 1-2 outputs");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(x0.device());
  std::vector<int64_t> lead(x0.sizes().begin(), x0.sizes().end() - 1);
  const int64_t n = x0.size(-1) > 0 ? x0.numel() / x0.size(-1) : 0;
  std::vector<at::Tensor> outs;
  for (int64_t w : out_w) {
    std::vector<int64_t> shp = lead;
    shp.push_back(w);
    outs.push_back(at::empty(shp, x0.options()));
  }
  if (n > 0) {
    const void* pi[3] = {nullptr, nullptr, nullptr};
    void* po[2] = {nullptr, nullptr};
    for (size_t k = 0; k < ins.size(); ++k) pi[k] = ins[k].data_ptr();
    for (size_t k = 0; k < outs.size(); ++k) po[k] = outs[k].data_ptr();
    const int code = reinterpret_cast<rowfn_t>(fn_addr)(pi[0], pi[1], pi[2], po[0], po[1], n,
                                                       (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(x0.device().index()).stream());
    TORCH_CHECK(code == 0, "pplie row operator failed with status ", code);
  }
  return outs;
}

bool plain(const at::Tensor& t, const at::Tensor& ref) {
  return t.defined() && t.is_cuda() && t.is_contiguous() && t.scalar_type() == ref.scalar_type() && t.device() == ref.device();
}

struct RowOp : public torch::autograd::Function<RowOp> {
  // saved: which tensors the backward kernel reads, in its argument order: k >= 0 = input k, -1 = the output
  static at::Tensor forward(AutogradContext* ctx, const at::Tensor& a, const c10::optional<at::Tensor>& b, int64_t fwd_fn, int64_t bwd_fn,
                            int64_t out_w, std::vector<int64_t> saved, std::vector<int64_t> bwd_out_w, int64_t rule, int64_t bwd_gb_fn) {
    std::vector<at::Tensor> ins{a};
    if (b.has_value()) ins.push_back(*b);
    at::Tensor out = launch(fwd_fn, ins, {out_w})[0];
    variable_list keep;
    for (int64_t k : saved) keep.push_back(k < 0 ? out : ins[(size_t)k]);
    ctx->save_for_backward(keep);
    ctx->saved_data["bwd_fn"] = bwd_fn;
    ctx->saved_data["bwd_gb_fn"] = bwd_gb_fn;
    ctx->saved_data["bwd_out_w"] = bwd_out_w;
    ctx->saved_data["rule"] = rule;
    ctx->saved_data["nin"] = (int64_t)ins.size();
    return out;
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    variable_list saved = ctx->get_saved_variables();
    const int64_t nin = ctx->saved_data["nin"].toInt();
    variable_list res(9);                                          // (one slot per forward argument; non-tensors stay undefined)
    at::Tensor g = grads[0];
    if (at::GradMode::is_enabled() || !g.has_storage()) {
      // create_graph=True: the backward has to be differentiable -- the Python rule of this operator (torch operations).  A
      // cotangent without storage is a batched tensor (torch.autograd.grad(is_grads_batched=True), what jacobian(vectorize=True)
      // uses): the Python launcher peels the batch dimension off.
      py::gil_scoped_acquire gil;
      py::object rule = py_rules().at(ctx->saved_data["rule"].toInt());
      py::list args;
      for (auto& t : saved) args.append(t);
      args.append(g);
      py::object r = rule(*py::tuple(args));
      if (py::isinstance<py::tuple>(r) || py::isinstance<py::list>(r)) {
        int64_t k = 0;
        for (auto item : r) {
          if (k < nin && !item.is_none()) res[(size_t)k] = item.cast<at::Tensor>();
          ++k;
        }
      } else {
        res[0] = r.cast<at::Tensor>();
      }
      return res;
    }
    std::vector<at::Tensor> ins(saved.begin(), saved.end());
    const at::Tensor& ref = ins[0];
    // A cotangent that is ONE row seen through stride-0 leading dimensions (what sum().backward() hands its producer): the
    // kernel's broadcast variant (pplie_<op>_bwd_gb, csrc/rowmap.h GB) reads that row once per workgroup -- materialising it
    // costs W x 4 bytes written and read per row, 15 % of the traffic of x.Exp().Log().sum().backward() at 10 M rows.
    const int64_t gb_fn = ctx->saved_data["bwd_gb_fn"].toInt();
    bool bcast = gb_fn != 0 && g.defined() && g.is_cuda() && g.dim() >= 1 && g.scalar_type() == ref.scalar_type() &&
                 g.device() == ref.device() && g.numel() > g.size(-1) && (g.size(-1) == 1 || g.stride(-1) == 1 || g.stride(-1) == 0);
    for (int64_t d = 0; bcast && d + 1 < g.dim(); ++d) bcast = g.size(d) == 1 || g.stride(d) == 0;
    if (bcast) {
      std::vector<int64_t> zero(g.dim(), 0);
      at::Tensor row = g.size(-1) > 1 && g.stride(-1) == 0 ? g.as_strided({g.size(-1)}, {0}).contiguous()     // (one scalar seen W times)
                                                           : g.as_strided({g.size(-1)}, {1});
      std::vector<at::Tensor> outs;
      {
        c10::hip::HIPGuardMasqueradingAsCUDA guard(ref.device());
        const int64_t n = ref.size(-1) > 0 ? ref.numel() / ref.size(-1) : 0;
        std::vector<int64_t> lead(ref.sizes().begin(), ref.sizes().end() - 1);
        for (int64_t w : ctx->saved_data["bwd_out_w"].toIntVector()) {
          std::vector<int64_t> shp = lead;
          shp.push_back(w);
          outs.push_back(at::empty(shp, ref.options()));
        }
        if (n > 0) {
          const void* pi[3] = {nullptr, nullptr, nullptr};
          void* po[2] = {nullptr, nullptr};
          for (size_t k = 0; k < ins.size(); ++k) pi[k] = ins[k].data_ptr();
          pi[ins.size()] = row.data_ptr();
          for (size_t k = 0; k < outs.size(); ++k) po[k] = outs[k].data_ptr();
          const int code = reinterpret_cast<rowfn_t>(gb_fn)(pi[0], pi[1], pi[2], po[0], po[1], n,
                                                           (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(ref.device().index()).stream());
          TORCH_CHECK(code == 0, "pplie row operator (broadcast cotangent) failed with status ", code);
        }
      }
      for (int64_t k = 0; k < nin && k < (int64_t)outs.size(); ++k) res[(size_t)k] = outs[(size_t)k];
      return res;
    }
    if (!plain(g, ref) || g.sizes().slice(0, g.dim() - 1) != ref.sizes().slice(0, ref.dim() - 1)) {
      std::vector<int64_t> shp(ref.sizes().begin(), ref.sizes().end() - 1);      // (an expanded / strided cotangent: e.g. out of sum())
      shp.push_back(g.size(-1));
      g = g.expand(shp).to(ref.options()).contiguous();
    }
    ins.push_back(g);
    std::vector<at::Tensor> outs = launch(ctx->saved_data["bwd_fn"].toInt(), ins, ctx->saved_data["bwd_out_w"].toIntVector());
    for (int64_t k = 0; k < nin && k < (int64_t)outs.size(); ++k) res[(size_t)k] = outs[(size_t)k];
    return res;
  }
};

at::Tensor row_op(const at::Tensor& a, const c10::optional<at::Tensor>& b, int64_t fwd_fn, int64_t bwd_fn, int64_t out_w,
                  std::vector<int64_t> saved, std::vector<int64_t> bwd_out_w, int64_t rule, int64_t bwd_gb_fn) {
  TORCH_CHECK(plain(a, a) && (!b.has_value() || (plain(*b, a) && b->sizes().slice(0, b->dim() - 1) == a.sizes().slice(0, a.dim() - 1))),
              "pplie native row operator: contiguous device tensors of one dtype and one leading shape");
  return RowOp::apply(a, b, fwd_fn, bwd_fn, out_w, saved, bwd_out_w, rule, bwd_gb_fn);
}

void set_rule(int64_t key, py::object fn) { py_rules()[key] = std::move(fn); }

// ---- prepared handles: everything one (operator, dtype) needs to be called, resolved ONCE -------------------------------------
// BASELINE configs[0] (1024 rows) is four launches of a few microseconds each; what it costs is the host path to them.  Through
// `row_op` above Python checked the operands (type, device, dtype, contiguity, widths: ~2 us), looked three kernel addresses up,
// and marshalled nine arguments, two of them Python lists, per call.  A handle holds all of that; its call checks the operands in
// C++ and either launches (no gradient wanted: the bare kernel; else the native autograd node) or returns None -- "not the plain
// eager case" (host tensors, broadcasting, another dtype ...) -- and the caller takes the general Python path.
struct RowHandle {
  int64_t fwd_fn = 0, bwd_fn = 0, bwd_gb_fn = 0, out_w = 0, rule = 0;
  int64_t in_w0 = 0, in_w1 = 0;                       // widths of the one or two operands (in_w1 = 0: unary)
  std::vector<int64_t> saved, bwd_out_w;
  at::ScalarType dtype = at::kFloat;

  py::object call(const at::Tensor& a, const c10::optional<at::Tensor>& b) const {
    const bool binary = in_w1 > 0;
    if (binary != b.has_value()) return py::none();
    // (a tensor without storage is a batched tensor of the legacy vmap -- torch.autograd.grad(is_grads_batched=True): Python peels it)
    if (!a.defined() || !a.has_storage() || !a.is_cuda() || a.scalar_type() != dtype || a.dim() < 1 || a.size(-1) != in_w0 || !a.is_contiguous())
      return py::none();
    if (binary) {
      const at::Tensor& bb = *b;
      if (!bb.defined() || !bb.has_storage() || !bb.is_cuda() || bb.scalar_type() != dtype || bb.device() != a.device() || bb.dim() != a.dim() ||
          bb.size(-1) != in_w1 || !bb.is_contiguous() || bb.sizes().slice(0, bb.dim() - 1) != a.sizes().slice(0, a.dim() - 1))
        return py::none();
    }
    const bool record = at::GradMode::is_enabled() && (a.requires_grad() || (binary && b->requires_grad()));
    if (!record) {
      std::vector<at::Tensor> ins{a};
      if (binary) ins.push_back(*b);
      at::Tensor out;
      {
        py::gil_scoped_release nogil;
        out = launch(fwd_fn, ins, {out_w})[0];
      }
      return py::cast(out);
    }
    return py::cast(RowOp::apply(a, b, fwd_fn, bwd_fn, out_w, saved, bwd_out_w, rule, bwd_gb_fn));
  }
};

// ---- the device-resident LM step (optim/fused.py DeviceLM): pplie_lm_se3inv_step behind ONE call -----------------------------------
// BASELINE configs[2] (LM on 10^6 independent SE3 problems) is two launches, ~32 us of GPU time per step; the host path to them was
// ~17 us of Python per step -- the current-stream lookup, a device-guard context manager and a ctypes call marshalling twelve
// arguments -- on top of the model's dry run (round 5: 40 us of host against 32 us of kernels, host-bound).  The handle keeps every
// pointer that does not change between steps; a step passes which state buffer is current and where the loss goes.
// `cfg_addr`: the address of the caller's pplie_lm_cfg (a ctypes.Structure it refills in place before a call).
typedef int (*lmstep_t)(void*, const void*, void*, void*, void*, void*, void*, const void*, int64_t, void*, void*, void*);
struct LmStepHandle {
  int64_t fn = 0, p_ptr = 0, x_ptr = 0, save = 0, partials = 0, st0 = 0, st1 = 0, sync = 0, cfg_addr = 0, n = 0;
  int device = 0;
  int launch(int cur, int64_t loss_ptr, int64_t last_ptr) const {
    c10::hip::HIPGuardMasqueradingAsCUDA guard(c10::Device(c10::kCUDA, (c10::DeviceIndex)device));
    void* stream = (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA((c10::DeviceIndex)device).stream();
    const int64_t st_in = cur ? st1 : st0, st_out = cur ? st0 : st1;
    return reinterpret_cast<lmstep_t>(fn)((void*)p_ptr, (const void*)x_ptr, (void*)save, (void*)partials, (void*)st_in, (void*)st_out,
                                          (void*)sync, (const void*)cfg_addr, n, (void*)loss_ptr, (void*)last_ptr, stream);
  }
};

// ---- IMU pre-integration (module/imu_preintegrator.py): pplie_imu_integrate / pplie_imu_integrate_bwd as ONE native node ----------
// Training through the pre-integrator is two kernels (64 + 108 us at 4096 x 1024); as a Python Function the pair cost ~190 us of
// host time per step (the backward runs on the engine's device thread behind the GIL) -- more than the kernels.  Same contract as
// _ImuIntegrate in Python, for the case it is used in training: gradients w.r.t. dt / gyro / acc (the initial state's and a double
// backward stay on the Python node).  C ABI: include/pplie.h.
typedef int (*imufwd_t)(const void*, const void*, const void*, const void*, const void*, const void*, const void*, const void*,
                        const double*, void*, void*, void*, void*, void*, void*, int64_t, int64_t, void*);
typedef int (*imubwd_t)(const void*, const void*, const void*, const void*, const void*, const void*, const void*, const double*,
                        const void*, const void*, const void*, void*, void*, void*, int64_t, int64_t, void*);

struct ImuOp : public torch::autograd::Function<ImuOp> {
  static variable_list forward(AutogradContext* ctx, const at::Tensor& dt, const at::Tensor& gyro, const at::Tensor& acc,
                               const c10::optional<at::Tensor>& rot, const at::Tensor& r0, const at::Tensor& v0, const at::Tensor& p0,
                               std::vector<double> gravity, int64_t fwd_fn, int64_t bwd_fn) {
    c10::hip::HIPGuardMasqueradingAsCUDA guard(dt.device());
    const int64_t B = dt.size(0), F = dt.size(1);
    at::Tensor orot = at::empty({B, F, 4}, dt.options()), ovel = at::empty({B, F, 3}, dt.options()), opos = at::empty({B, F, 3}, dt.options());
    const int code = reinterpret_cast<imufwd_t>(fwd_fn)(
        dt.data_ptr(), gyro.data_ptr(), acc.data_ptr(), rot.has_value() ? rot->data_ptr() : nullptr, r0.data_ptr(), v0.data_ptr(),
        p0.data_ptr(), nullptr, gravity.data(), orot.data_ptr(), ovel.data_ptr(), opos.data_ptr(), nullptr, nullptr, nullptr, B, F,
        (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dt.device().index()).stream());
    TORCH_CHECK(code == 0, "pplie_imu_integrate failed with status ", code);
    ctx->save_for_backward({dt, gyro, acc, rot.has_value() ? *rot : at::Tensor(), r0, orot, ovel});
    ctx->saved_data["bwd_fn"] = bwd_fn;
    ctx->saved_data["gravity"] = gravity;
    return {orot, ovel, opos};
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    TORCH_CHECK(!at::GradMode::is_enabled(), "pplie: the fused IMU backward is not differentiable a second time "
                                             "(IMUPreintegrator.fused_backward = False takes the composed route)");
    variable_list s = ctx->get_saved_variables();
    const at::Tensor &dt = s[0], &gyro = s[1], &acc = s[2], &rot = s[3], &r0 = s[4], &orot = s[5], &ovel = s[6];
    c10::hip::HIPGuardMasqueradingAsCUDA guard(dt.device());
    const int64_t B = dt.size(0), F = dt.size(1);
    at::Tensor g[3];
    const int64_t w[3] = {4, 3, 3};
    for (int k = 0; k < 3; ++k)
      if (grads[(size_t)k].defined()) g[k] = grads[(size_t)k].expand({B, F, w[k]}).to(dt.options()).contiguous();
    at::Tensor o_gyro = at::empty_like(gyro), o_acc = at::empty_like(acc), o_dt;
    if (ctx->needs_input_grad(0)) o_dt = at::empty_like(dt);
    std::vector<double> gravity = ctx->saved_data["gravity"].toDoubleVector();
    auto P = [](const at::Tensor& t) -> void* { return t.defined() ? t.data_ptr() : nullptr; };
    const int code = reinterpret_cast<imubwd_t>(ctx->saved_data["bwd_fn"].toInt())(
        dt.data_ptr(), gyro.data_ptr(), acc.data_ptr(), P(rot), orot.data_ptr(), ovel.data_ptr(), r0.data_ptr(), gravity.data(), P(g[0]),
        P(g[1]), P(g[2]), o_gyro.data_ptr(), o_acc.data_ptr(), P(o_dt), B, F,
        (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dt.device().index()).stream());
    TORCH_CHECK(code == 0, "pplie_imu_integrate_bwd failed with status ", code);
    variable_list res(10);
    res[0] = o_dt;
    res[1] = o_gyro;
    res[2] = o_acc;
    return res;
  }
};

variable_list imu_integrate(const at::Tensor& dt, const at::Tensor& gyro, const at::Tensor& acc, const c10::optional<at::Tensor>& rot,
                            const at::Tensor& r0, const at::Tensor& v0, const at::Tensor& p0, std::vector<double> gravity, int64_t fwd_fn,
                            int64_t bwd_fn) {
  TORCH_CHECK(dt.dim() == 3 && dt.size(2) == 1 && plain(dt, dt) && plain(gyro, dt) && plain(acc, dt) && plain(r0, dt) && plain(v0, dt) &&
                  plain(p0, dt) && (!rot.has_value() || plain(*rot, dt)) && gravity.size() == 3,
              "pplie native IMU node: contiguous device tensors of one dtype");
  const int64_t B = dt.size(0), F = dt.size(1);
  TORCH_CHECK(gyro.numel() == B * F * 3 && acc.numel() == B * F * 3 && r0.numel() == B * 4 && v0.numel() == B * 3 && p0.numel() == B * 3 &&
                  (!rot.has_value() || rot->numel() == B * F * 4),
              "pplie native IMU node: [B, F, 1 / 3 / 3 (/ 4)] inputs and [B, 4 / 3 / 3] initial states");
  return ImuOp::apply(dt, gyro, acc, rot, r0, v0, p0, gravity, fwd_fn, bwd_fn);
}

// ---- product scans (basics/scan.py): pplie_scan_<g> in place + pplie_scan_<g>_bwd as one native node ----------------------------
typedef int (*scanfwd_t)(void*, int64_t, int64_t, int64_t, int, void*);
typedef int (*scanbwd_t)(const void*, const void*, const void*, void*, int64_t, int64_t, int64_t, int, void*);

struct ScanOp : public torch::autograd::Function<ScanOp> {
  static at::Tensor forward(AutogradContext* ctx, at::Tensor x, int64_t fwd_fn, int64_t bwd_fn, int64_t nseq, int64_t L, int64_t inner,
                            bool left, int64_t rule, int64_t dim) {
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    ctx->mark_dirty({x});
    // left products: the backward transports cotangents through the scan's own FACTORS (kept: the scan overwrites them)
    at::Tensor x_in = left ? x.clone() : at::Tensor();
    const int code = reinterpret_cast<scanfwd_t>(fwd_fn)(x.data_ptr(), nseq, L, inner, left ? 1 : 0,
                                                         (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(x.device().index()).stream());
    TORCH_CHECK(code == 0, "pplie_scan failed with status ", code);
    ctx->save_for_backward({x, x_in});
    ctx->saved_data["bwd_fn"] = bwd_fn;
    ctx->saved_data["geom"] = std::vector<int64_t>{nseq, L, inner, left ? 1 : 0, rule, dim};
    return x;
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    variable_list s = ctx->get_saved_variables();
    const std::vector<int64_t> geom = ctx->saved_data["geom"].toIntVector();
    const bool left = geom[3] != 0;
    variable_list res(9);
    at::Tensor g = grads[0];
    if (at::GradMode::is_enabled() || !g.has_storage()) {            // create_graph=True / batched cotangents: the Python rule
      py::gil_scoped_acquire gil;
      py::object rule = py_rules().at(geom[4]);
      res[0] = rule(s[0], g, geom[5], left).cast<at::Tensor>();
      return res;
    }
    const at::Tensor& ref = s[0];
    c10::hip::HIPGuardMasqueradingAsCUDA guard(ref.device());
    if (!plain(g, ref) || g.sizes() != ref.sizes()) g = g.expand(ref.sizes()).to(ref.options()).contiguous();
    at::Tensor gx = at::empty_like(ref);
    const int code = reinterpret_cast<scanbwd_t>(ctx->saved_data["bwd_fn"].toInt())(
        left ? s[1].data_ptr() : nullptr, left ? nullptr : ref.data_ptr(), g.data_ptr(), gx.data_ptr(), geom[0], geom[1], geom[2],
        left ? 1 : 0, (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(ref.device().index()).stream());
    TORCH_CHECK(code == 0, "pplie_scan_bwd failed with status ", code);
    res[0] = gx;
    return res;
  }
};

at::Tensor scan_op(at::Tensor x, int64_t fwd_fn, int64_t bwd_fn, int64_t nseq, int64_t L, int64_t inner, bool left, int64_t rule,
                   int64_t dim) {
  TORCH_CHECK(plain(x, x) && x.numel() == nseq * L * x.size(-1), "pplie native scan node: a contiguous device tensor");
  return ScanOp::apply(x, fwd_fn, bwd_fn, nseq, L, inner, left, rule, dim);
}
}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "native autograd nodes for the row operators of libpplie (pypose_amd/csrc_torch/pplie_autograd.cpp)";
  m.def("row_op", &row_op, "forward of one row operator recorded as a native autograd node", py::arg("a"), py::arg("b"), py::arg("fwd_fn"),
        py::arg("bwd_fn"), py::arg("out_w"), py::arg("saved"), py::arg("bwd_out_w"), py::arg("rule"), py::arg("bwd_gb_fn") = 0);
  py::class_<RowHandle>(m, "RowHandle")
      .def(py::init([](int64_t fwd_fn, int64_t bwd_fn, int64_t bwd_gb_fn, int64_t in_w0, int64_t in_w1, int64_t out_w,
                       std::vector<int64_t> saved, std::vector<int64_t> bwd_out_w, int64_t rule, bool f64) {
             RowHandle h;
             h.fwd_fn = fwd_fn; h.bwd_fn = bwd_fn; h.bwd_gb_fn = bwd_gb_fn; h.in_w0 = in_w0; h.in_w1 = in_w1; h.out_w = out_w;
             h.saved = std::move(saved); h.bwd_out_w = std::move(bwd_out_w); h.rule = rule; h.dtype = f64 ? at::kDouble : at::kFloat;
             return h;
           }))
      .def("__call__", &RowHandle::call, py::arg("a"), py::arg("b") = py::none(),
           "launch (bare kernel, or native autograd node when a gradient is being recorded); None if the operands are not the plain eager case");
  py::class_<LmStepHandle>(m, "LmStepHandle")
      .def(py::init([](int64_t fn, int64_t p_ptr, int64_t x_ptr, int64_t save, int64_t partials, int64_t st0, int64_t st1, int64_t sync,
                       int64_t cfg_addr, int64_t n, int device) {
             LmStepHandle h;
             h.fn = fn; h.p_ptr = p_ptr; h.x_ptr = x_ptr; h.save = save; h.partials = partials; h.st0 = st0; h.st1 = st1; h.sync = sync;
             h.cfg_addr = cfg_addr; h.n = n; h.device = device;
             return h;
           }))
      .def("launch", &LmStepHandle::launch, py::arg("cur"), py::arg("loss_ptr"), py::arg("last_ptr"),
           "enqueue one device-resident LM step on the device's current stream; returns the C ABI's status");
  m.def("set_rule", &set_rule, "register the differentiable Python rule of an operator's backward (double backward)");
  m.def("scan_op", &scan_op, "pplie_scan_<group> in place, recorded as a native autograd node (backward: pplie_scan_<group>_bwd)");
  m.def("imu_integrate", &imu_integrate, "pplie_imu_integrate recorded as a native autograd node (backward: pplie_imu_integrate_bwd)");
}
