import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gnn_pressure_estimation_amd as G
from oracle import gatres_oracle as O

def relerr(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))

nb = int(os.environ.get("NB", "15"))
for bs in [int(v) for v in sys.argv[1:]] or (8, 9, 12, 16, 24, 32):
    torch.manual_seed(0)
    p = O.init_params(nb, 32, seed=3)
    def mk(fused):
        m = G.GATResMeanConv(num_blocks=nb, nc=32, fused=fused)
        sd = {}
        for k, v in p.items():
            sd[k] = v
            if k.endswith("lin_src.weight"): sd[k.replace("lin_src", "lin_dst")] = v
        m.load_state_dict(sd)
        return m.cuda()
    mf, mp = mk(True), mk(False)
    x, y, ei, mask = G.wdn_synth.make_batch(bs)
    dei = ei.cuda()
    g = torch.randn(x.shape[0], 1, device="cuda")
    def grad(m, g):
        m.zero_grad()
        dx = x.cuda().requires_grad_(True)
        o = m(dx, dei)
        o.backward(g)
        return dx.grad.clone(), torch.cat([q.grad.reshape(-1) for q in m.parameters()]).clone()
    xf, gf = grad(mf, g)
    xp, gp = grad(mp, g)
    xf2, gf2 = grad(mf, g)
    kinds = {"W": [], "att": [], "bias": [], "lin": []}
    off = 0
    for k, q in mf.named_parameters():
        n = q.numel()
        e = float((gf[off:off+n] - gp[off:off+n]).abs().max() / gp.abs().max())
        kind = "W" if "lin_src" in k else ("att" if "att" in k else ("bias" if "conv" in k else "lin"))
        kinds[kind].append(e)
        off += n
    print(f"bs={bs:3d} nb={nb}: g_x relerr {relerr(xf, xp):.2e} (rerun equal {torch.equal(xf, xf2)})  params relerr {relerr(gf, gp):.2e} "
          f"(rerun equal {torch.equal(gf, gf2)})  by kind: " + " ".join(f"{k}={max(v):.1e}" for k, v in kinds.items()))
