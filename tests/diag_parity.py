"""One-off diagnostic: where does the HIP path differ from the fp64 oracle at the BASELINE graph size?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neurips21-self-supervised-bug-detection-and-repair_amd")]
import numpy as np, torch
from oracle import buglab_oracle as O
from tests import helpers as Hh
from buglab.data.collate import to_device

H, degree = int(sys.argv[1]), sys.argv[2]
cfg, _, mb = Hh.make_case(B=2, n=2000, E=10000, T=16, H=H, layers=8, vocab=15000, C=40, degree=degree, max_degree=512, seed=21)
params = O.init_params(cfg, seed=0)
tr64, tr32 = [], []
o64 = O.forward_loss({k: v.double() for k, v in params.items()}, mb, cfg, trace=tr64)
o32 = O.forward_loss(params, mb, cfg, trace=tr32)
module = Hh.build_module_like(cfg, params).eval()
with torch.no_grad():
    _, lp, gout, _ = module.compute_localization_logprobs(to_device(mb, "cuda")["graph_data"])
hip = gout.output_node_representations.cpu().double()
ref = o64["node_reprs"].detach()
d = (hip - ref).abs()
d32 = (o32["node_reprs"].detach().double() - ref).abs()
print("HIP vs fp64: max %.3e  mean %.3e ; fp32 oracle vs fp64: max %.3e mean %.3e" % (d.max(), d.mean(), d32.max(), d32.mean()))
n, c = np.unravel_index(int(d.argmax()), d.shape)
deg = np.diff(mb["graph_data"]["tgt_ptr"])
print("worst node", n, "channel", c, "in-degree", deg[n], "hip", float(hip[n, c]), "ref", float(ref[n, c]), "fp32 oracle", float(o32["node_reprs"][n, c]))
print("rows with diff > 1e-5:", int((d.max(1).values > 1e-5).sum()), "of", d.shape[0])
L64 = [t for t in tr64 if "arg" in t]; L32 = [t for t in tr32 if "arg" in t]
for li, (a, b) in enumerate(zip(L64, L32)):
    var = a["agg"].var(1, unbiased=False)
    print(f"layer {li}: fp32-vs-fp64 oracle out diff max {(a['out'] - b['out'].double()).abs().max():.3e}; min row variance of aggregate {float(var.min()):.3e}; argflips {(a['arg'] != b['arg']).sum().item()}")
