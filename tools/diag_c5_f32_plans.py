"""GPU diagnostic: one fp32 step at config 5's dimensions against the oracle under two split-K plans (engine.tune(f32_rows=...)):
relative L2 of every gradient tensor and of the post-step weights.  Usage: python tools/diag_c5_f32_plans.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import jamie_amd as jam
from oracle import jamie_oracle as orc
import test_hip_configs as T
B, dims, L, p = 512, (5000, 2000), 64, 0.6
for plan in ('1,1;2,1', ''):
    from jamie_amd import engine as _e
    _e.tune(f32_rows=plan or None)
    model, eng, P, Bf = T._pair(jam, dims, L, B, 'f32')
    opt = orc.Adam(P.values(), 1e-3)
    X = T._synth(B, dims, seed=5)
    torch.manual_seed(43)
    noise = orc.draw_noise(dims, L, B, p)
    st = orc.train_step(P, Bf, opt, X, torch.eye(B), torch.zeros(B, B), noise, p, 0.5, return_grads=True)
    eng.set_batch([x.cuda() for x in X]); eng.set_kl_anneal(0.5)
    eng.forward_backward(None, None, T._noise_to_dev(noise, p))
    print('plan', plan or 'model', {k: v for k, v in eng.ws[0]['sk'].items()}, flush=True)
    for ref in P:
        if orc.is_dead_bias(ref): continue
        g, w = T._grad(eng, model, ref), st['grads'][ref].numpy()
        g = g.cpu().numpy() if torch.is_tensor(g) else g
        print(f'   grad {ref:28s} relL2 {np.linalg.norm(g - w) / np.linalg.norm(w):.3e}')
    eng.optimizer_step()
    sd = model.state_dict()
    for k, v in P.items():
        if orc.is_dead_bias(k): continue
        a, b = sd[k].cpu().numpy(), v.detach().numpy()
        d = np.abs(a - b)
        print(f'   weight {k:26s} relL2 {np.linalg.norm(a - b) / np.linalg.norm(b):.3e}  frac(|d|>1e-4) {float((d > 1e-4).mean()):.2e}  max {d.max():.2e}')
    del model, eng
    torch.cuda.empty_cache()
