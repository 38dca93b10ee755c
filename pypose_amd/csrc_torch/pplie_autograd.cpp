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
                            int64_t out_w, std::vector<int64_t> saved, std::vector<int64_t> bwd_out_w, int64_t rule) {
    std::vector<at::Tensor> ins{a};
    if (b.has_value()) ins.push_back(*b);
    at::Tensor out = launch(fwd_fn, ins, {out_w})[0];
    variable_list keep;
    for (int64_t k : saved) keep.push_back(k < 0 ? out : ins[(size_t)k]);
    ctx->save_for_backward(keep);
    ctx->saved_data["bwd_fn"] = bwd_fn;
    ctx->saved_data["bwd_out_w"] = bwd_out_w;
    ctx->saved_data["rule"] = rule;
    ctx->saved_data["nin"] = (int64_t)ins.size();
    return out;
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    variable_list saved = ctx->get_saved_variables();
    const int64_t nin = ctx->saved_data["nin"].toInt();
    variable_list res(8);                                          // (one slot per forward argument; non-tensors stay undefined)
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
                  std::vector<int64_t> saved, std::vector<int64_t> bwd_out_w, int64_t rule) {
  TORCH_CHECK(plain(a, a) && (!b.has_value() || (plain(*b, a) && b->sizes().slice(0, b->dim() - 1) == a.sizes().slice(0, a.dim() - 1))),
              "pplie native row operator: contiguous device tensors of one dtype and one leading shape");
  return RowOp::apply(a, b, fwd_fn, bwd_fn, out_w, saved, bwd_out_w, rule);
}

void set_rule(int64_t key, py::object fn) { py_rules()[key] = std::move(fn); }
}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "native autograd nodes for the row operators of libpplie (pypose_amd/csrc_torch/pplie_autograd.cpp)";
  m.def("row_op", &row_op, "forward of one row operator recorded as a native autograd node");
  m.def("set_rule", &set_rule, "register the differentiable Python rule of an operator's backward (double backward)");
}
