#!/usr/bin/env python
"""Diagnostic (GPU box): per-tensor relative L2 distance of the HIP bf16 step's gradients from (a) the oracle with the same
operand roundings and (b) the fp32 oracle, at a BASELINE size; and the engine's dW of a layer against the product of its own
stored bf16 operands.  python tools/diag_bf16_emulation.py [c2|c4|c5]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from oracle import jamie_oracle as orc  # noqa: E402
from test_hip_configs import _synth, _pair, _grad, _clone_state  # noqa: E402
from test_hip_step import _noise_to_dev  # noqa: E402
import jamie_amd  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else 'c2'
B, dims, L, p = {'c2': (512, (2000, 1000), 32, 0.6), 'c4': (512, (2000, 1000, 500), 64, 0.6),
                 'c5': (512, (5000, 2000), 64, 0.6), 'small': (256, (304, 184), 16, 0.6)}[cfg]
model, eng, P, Bf = _pair(jamie_amd, dims, L, B, 'bf16')
X = _synth(B, dims)
torch.manual_seed(42)
noise = orc.draw_noise(dims, L, B, p)
corr = torch.eye(B) if len(dims) == 2 else None
P_e, Bf_e = _clone_state(P, Bf)
st = orc.train_step(P, Bf, None, X, corr, None, noise, p, 0.5, do_step=False, return_grads=True)
prec, gbf = eng.operand_precision()
st_e = orc.train_step(P_e, Bf_e, None, X, corr, None, noise, p, 0.5, do_step=False, return_grads=True, emulate=prec)
eng.set_batch([x.cuda() for x in X])
eng.set_kl_anneal(0.5)
eng.forward_backward(None, None, _noise_to_dev(noise, p))
print('losses hip', eng.read_losses()[0], '\n   emulated', st_e['losses'], '\n       fp32', st['losses'])


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(1e-30, np.linalg.norm(b)))


print(f'{"tensor":28s} {"hip-vs-emulated":>16s} {"hip-vs-fp32":>12s} {"emulated-vs-fp32":>17s} {"|g|":>10s}')
for ref in P:
    if orc.is_dead_bias(ref):
        continue
    want = st_e['grads'][ref]
    if gbf and want.dim() == 2:
        want = orc.bf16_round(want)
    got = _grad(eng, model, ref)
    print(f'{ref:28s} {rel(got, want.numpy()):16.3e} {rel(got, st["grads"][ref].numpy()):12.3e} '
          f'{rel(st_e["grads"][ref].numpy(), st["grads"][ref].numpy()):17.3e} {float(st["grads"][ref].norm()):10.3e}')
# the engine's own stored operands
for i in range(len(dims)):
    w = eng.ws[i]
    for lin, dy, a in (('dec0', 'de1', 'comb'), ('head', 'dml', 'a2'), ('dec1', 'de2', 'e1')):
        chk = w[dy + '_bf'].float().t() @ w[a + '_bf'].float()
        got = eng.grad_view(f'm{i}.{lin}.W')
        chk = chk[:got.shape[0], :got.shape[1]]
        print(f'm{i}.{lin}.W vs its own stored bf16 operands: {rel(got.cpu().numpy(), chk.to(torch.bfloat16).float().cpu().numpy() if gbf else chk.cpu().numpy()):.3e}')
    print(f'm{i} comb_bf vs bf(comb): {rel(w["comb_bf"].float().cpu().numpy(), w["comb"].to(torch.bfloat16).float().cpu().numpy()):.3e}',
          f' comb vs emulated: {rel(w["comb"].cpu().numpy(), st_e["combined"][i].numpy()):.3e}')
