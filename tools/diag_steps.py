"""Diagnostic (GPU box): multi-step drift HIP vs fp32 oracle for one tensor."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import jamie_oracle as orc
from jamie_amd.engine import TrainEngine
from jamie_amd.model import edModelVar
B, dims, L, p = 512, (520, 260), 32, 0.6
torch.manual_seed(123); model = edModelVar(dims, L, dropout=p)
torch.manual_seed(123); P, Bf = orc.init_state(dims, L)
for v in P.values(): v.requires_grad_(True)
eng = TrainEngine(model, B); opt = orc.Adam(P.values(), 1e-3)
rng = np.random.default_rng(0)
Z = rng.standard_normal((B, 16))
X = [torch.from_numpy((Z @ rng.standard_normal((16, d)) + .1 * rng.standard_normal((B, d))).astype(np.float32)) for d in dims]
X = [(x - x.mean(0)) / x.std(0) for x in X]
names = list(P.keys())
lay = model.layout.reference_names()
for step in range(4):
    torch.manual_seed(1000 + step)
    noise = orc.draw_noise(dims, L, B, p)
    st = orc.train_step(P, Bf, opt, X, torch.eye(B), torch.zeros(B, B), noise, p, 0.3 + 0.1 * step, return_grads=True)
    dn = {'eps': [e.cuda() for e in noise['eps']],
          'enc_masks': [[m.to(torch.uint8).cuda() for m in pr] for pr in noise['enc_masks']],
          'dec_masks': [[m.to(torch.uint8).cuda() for m in pr] for pr in noise['dec_masks']]}
    for i in range(2): eng.ws[i]['x'].copy_(X[i])
    eng.set_kl_anneal(0.3 + 0.1 * step)
    eng.forward_backward(None, None, dn)
    gh = {k: (eng.g[m] if s is None else eng.g[m][s]).cpu().clone() for k, (m, s) in lay.items()}
    eng.optimizer_step()
    sd = model.state_dict()
    print(f'--- step {step}: grad_norm oracle {st["grad_norm"]:.6f} hip {torch.sqrt(eng.norm_partials.sum()).item():.6f}')
    for k in ['encoders.0.0.weight', 'encoders.0.4.weight', 'decoders.0.8.weight', 'encoders.0.1.weight', 'sigma']:
        g_o, g_h = st['grads'][k], gh[k]
        w_o, w_h = P[k].detach(), sd[k].cpu()
        dg = (g_h - g_o).abs(); dw = (w_h - w_o).abs()
        bad = dw > 2e-5
        msg = f'{k:24s} |g|max {g_o.abs().max():.3e} dg max {dg.max():.3e} relL2 {(g_h-g_o).norm()/g_o.norm():.2e} | dw max {dw.max():.3e} bad {int(bad.sum())}/{bad.numel()}'
        if bad.any() and g_o.dim() == 2:
            rows = bad.any(1).sum().item(); cols = bad.any(0).sum().item()
            gq = g_o.abs()[bad]
            msg += f' rows {rows} cols {cols} |g| at bad: med {gq.median():.2e} max {gq.max():.2e}; |g| overall med {g_o.abs().median():.2e}'
        print(msg)
