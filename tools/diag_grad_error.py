"""Diagnostic (GPU box): gradient error of the HIP step and of the fp32 CPU oracle, both measured against
the fp64 oracle, per parameter tensor.  Not a test; prints a table."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import jamie_oracle as orc
from jamie_amd.engine import TrainEngine
from jamie_amd.model import edModelVar

B, dims, L, p = int(sys.argv[1]), (int(sys.argv[2]), int(sys.argv[3])), int(sys.argv[4]), float(sys.argv[5])
torch.manual_seed(123)
model = edModelVar(dims, L, dropout=p)
torch.manual_seed(123)
P32, Bf32 = orc.init_state(dims, L)
P64 = {k: v.double().clone() for k, v in P32.items()}
Bf64 = {k: (v.double().clone() if v.dtype.is_floating_point else v.clone()) for k, v in Bf32.items()}
for d in (P32, P64):
    for v in d.values():
        v.requires_grad_(True)
eng = TrainEngine(model, B)
rng = np.random.default_rng(0)
Z = rng.standard_normal((B, 16))
X = [torch.from_numpy((Z @ rng.standard_normal((16, d)) + .1 * rng.standard_normal((B, d))).astype(np.float32)) for d in dims]
X = [(x - x.mean(0)) / x.std(0) for x in X]
torch.manual_seed(1000)
noise = orc.draw_noise(dims, L, B, p)
n64 = {'eps': [e.double() for e in noise['eps']],
       'enc_masks': [[None if m is None else m.double() for m in pr] for pr in noise['enc_masks']],
       'dec_masks': [[None if m is None else m.double() for m in pr] for pr in noise['dec_masks']]}
s32 = orc.train_step(P32, Bf32, None, X, torch.eye(B), torch.zeros(B, B), noise, p, 0.3, do_step=False, return_grads=True)
s64 = orc.train_step(P64, Bf64, None, [x.double() for x in X], torch.eye(B, dtype=torch.float64),
                     torch.zeros(B, B, dtype=torch.float64), n64, p, 0.3, do_step=False, return_grads=True)
dn = {'eps': [e.cuda() for e in noise['eps']],
      'enc_masks': [[None if m is None else m.to(torch.uint8).cuda() for m in pr] for pr in noise['enc_masks']],
      'dec_masks': [[None if m is None else m.to(torch.uint8).cuda() for m in pr] for pr in noise['dec_masks']]}
for i in range(2):
    eng.ws[i]['x'].copy_(X[i])
eng.set_kl_anneal(0.3)
eng.forward_backward(None, None, dn)
print('losses hip', eng.read_losses()[0], '\n       o32', s32['losses'], '\n       o64', s64['losses'])
names = model.layout.reference_names()
print(f'{"tensor":28s} {"|g|max":>10s} {"hip relL2":>10s} {"o32 relL2":>10s} {"hip maxabs":>11s} {"o32 maxabs":>11s}')
for ref, (mine, sl) in names.items():
    got = (eng.g[mine] if sl is None else eng.g[mine][sl]).cpu().double()
    g32, g64 = s32['grads'][ref].double(), s64['grads'][ref]
    nrm = g64.norm().item() + 1e-300
    print(f'{ref:28s} {g64.abs().max().item():10.3e} {(got - g64).norm().item() / nrm:10.3e} '
          f'{(g32 - g64).norm().item() / nrm:10.3e} {(got - g64).abs().max().item():11.3e} {(g32 - g64).abs().max().item():11.3e}')
tot = lambda d: float(torch.sqrt(sum((g.double() ** 2).sum() for g in d.values())))      # noqa: E731
print('||g||: hip', float(eng.grad.double().norm()), ' o32', tot(s32['grads']), ' o64', tot(s64['grads']))
