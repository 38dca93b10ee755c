import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pypose_amd as pp
from pypose_amd.optim import fused as F
from tests.optim_models import InvNet
dev = "cuda:0"
torch.manual_seed(0)
B = 5000
init, inp = pp.randn_SE3(B, device=dev), pp.randn_SE3(B, device=dev)
net = InvNet(init.clone())
opt = pp.optim.LM(net, strategy=pp.optim.strategy.Constant(damping=1e-4))
for k in range(4):
    opt.step(inp)
    print("step", k, opt.linearization, type(opt.__dict__.get('_device_lm')).__name__, opt._structure_cache.get("fused"), opt._structure_cache.get("dry"))
d = opt._device_lm
pg = opt.param_groups[0]
params = [p for p in pg['params'] if p.requires_grad]
m = F.dry_program(opt, params, inp, None)
print("m", None if not m else (m[0], m[1] is d.P, m[2].data_ptr() == d.x_ptr, m[2].shape, m[2].dtype, m[2].device, m[2].is_contiguous()))
print("matches", d.matches(m) if m else None, "conditions", d._step_conditions(None, None, m))
print("parts", d.P.data_ptr(), d.p_ptr, opt.strategy is d.strategy, opt.fused, getattr(opt, 'structured', True), len(opt.param_groups), torch.is_inference_mode_enabled(), opt.weight)
out = F.checked_shortcut(opt, d, None, inp, None, None)
print("checked_shortcut ->", out)
