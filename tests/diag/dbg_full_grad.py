"""Per-tensor gradient error of the full MulResUnet3D (GPU fp32 and CPU-oracle fp32, both against CPU-oracle fp64)."""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from deep_prior_interpolation_amd import ops, utils as u
from deep_prior_interpolation_amd.architectures import get_net
from deep_prior_interpolation_amd.parameter import parse_arguments
from oracle import dpi_oracle as O
torch.set_num_threads(16)
kind = sys.argv[1] if len(sys.argv) > 1 else "mse"
a = parse_arguments(["--imgdir", "x", "--datadim", "3d", "--upsample", "linear"])
u.set_seed(0)
net = get_net(a, 1); u.init_weights(net, a.inittype, a.initgain)
init = {k: v.detach().clone() for k, v in net.state_dict().items()}
gen = torch.Generator().manual_seed(1)
sh = (32,32,32)
x = 0.1 * torch.randn((1, 64)+sh, generator=gen)
img = torch.randn((1, 1)+sh, generator=gen)
mask = (torch.rand((1, 1, 1)+sh[1:], generator=gen) > 0.5).float().expand((1, 1)+sh).contiguous()
cfg = {"ndim": 3, "filters": a.filters, "skip": a.skip, "upsample": "trilinear"}
def run_oracle(dtype):
    S = O.NetState(init, dtype=dtype)
    out = O.net_forward(S, x.to(dtype), cfg)
    loss = O.masked_loss(out, img.to(dtype), mask.to(dtype), kind); loss.backward()
    return S, out, loss
S32, o32, l32 = run_oracle(torch.float32)
S64, o64, l64 = run_oracle(torch.float64)
net = net.to('cuda')
out = net(x.cuda()); loss, met = ops.masked_loss(out, img.cuda(), mask.cuda(), kind); loss.backward()
def rel(a,b):
    a=a.double().cpu().numpy(); b=b.double().cpu().numpy(); return np.linalg.norm(a-b)/(np.linalg.norm(b)+1e-30)
print("out rel gpu-vs-f64", rel(out.detach(), o64.detach()), "cpu32-vs-f64", rel(o32.detach(), o64.detach()))
for k,p in net.named_parameters():
    if p.ndim>1:
        print("%-34s gpu %.2e cpu32 %.2e ratio %5.1f |g|=%.2e" % (k, rel(p.grad, S64.P[k].grad), rel(S32.P[k].grad, S64.P[k].grad),
              rel(p.grad, S64.P[k].grad)/max(rel(S32.P[k].grad, S64.P[k].grad),1e-12), float(S64.P[k].grad.norm())))
