"""Correspondence stage of JAMIE on the MI355X: `Prime_Dual` (reference jamie/jamie.py:314-414), SURVEY.md §8(f)
rank 3 -- the largest end-to-end cost of the reference (45 009 s of 52 557 s on scGLUE, time-and-memory.ipynb).

One iteration of the reference is seven dense [N,N] products and ~25 element-wise dispatches.  Here it is four
launches of the exact-fp32 MFMA GEMM (`jamie_gemm_f32`) and four small kernels (`jamie_pd_step`, `jamie_pd_alpha`):

    T1  = F^T (F Ky)          TN      G1 = (F Ky) T1          NN       gradient, moments, projected step, new F,
    FKy = F Ky   (new F)      NN      G2 = Kx (F Ky)          NN       S / Mu / Lambda in jamie_pd_step

The scaling factor a = tr(Kx (F Ky) F^T) / tr(Kx Kx) (jamie.py:397-402, three more products in the reference) is
sum(G2 o F) / tr(Kx Kx): G2 of the NEW F is exactly what the next iteration's gradient needs.  Nothing is read back
per iteration; `a` lives in a device scalar.
"""
import math
import warnings

import numpy as np
import torch

from . import _native as nv


class PrimeDual:
    """State of one Prime_Dual solve on the device.  Kx [m,m], Ky [n,n]: distance matrices (numpy or tensors)."""

    def __init__(self, Kx, Ky, dx, dy, rho=10, epsilon=1e-3, delay=0, device='cuda', gemm_cfg=None):
        nv.require_gpu()
        dev = torch.device(device)
        Kx = np.asarray(Kx) if not torch.is_tensor(Kx) else Kx
        Ky = np.asarray(Ky) if not torch.is_tensor(Ky) else Ky
        self.m, self.n = int(Kx.shape[0]), int(Ky.shape[0])
        N = int(max(self.m, self.n))                                             # jamie.py:330-334
        f32 = dict(device=dev, dtype=torch.float32)
        self.Kx = (torch.as_tensor(Kx / N) if not torch.is_tensor(Kx) else Kx / N).to(**f32).contiguous()
        self.Ky = (torch.as_tensor(Ky / N) if not torch.is_tensor(Ky) else Ky / N).to(**f32).contiguous()
        self.rho, self.epsilon, self.delay = float(rho), float(epsilon), int(delay)
        m, n = self.m, self.n
        self.alpha = torch.full((1,), math.sqrt(dy / dx), **f32)                 # jamie.py:335
        # tr(Kx Kx) = sum_ij Kx_ij Kx_ji: constant over the iterations (the reference recomputes it each time)
        tr = 0.0
        for lo in range(0, m, 4096):
            blk = self.Kx[lo:lo + 4096].double()
            tr += float((blk * self.Kx[:, lo:lo + 4096].t().double()).sum())
        self.inv_trkk = 1.0 / tr if tr != 0 else float('inf')
        z = lambda *s: torch.zeros(*s, **f32)   # noqa: E731
        self.F, self.m1, self.m2 = z(m, n), z(m, n), z(m, n)                     # jamie.py:339, 350-351
        self.FKy, self.G1, self.G2, self.T1 = z(m, n), z(m, n), z(m, n), z(n, n)
        self.Mu, self.Lambda, self.S = z(m), z(n), z(n)                          # jamie.py:342-344
        self.rowsum, self.colsum = z(m), z(n)
        ra, ca = nv.pd_workspace(m, n)
        self.rowpart, self.colpart = z(ra), z(ca)
        self.partials = z(2048)
        self.iteration = 0
        st = nv.PdState()
        for k in ('F', 'G1', 'G2', 'm1', 'm2', 'Mu', 'Lambda', 'S', 'rowsum', 'colsum', 'alpha', 'rowpart', 'colpart'):
            setattr(st, k, nv.ptr(getattr(self, k)))
        st.m, st.n, st.rho, st.epsilon = m, n, self.rho, self.epsilon
        self._state = st
        # large squares: the fp32 products on the bf16 matrix pipe (gemm_f32.hip configurations 21 / 20: every element cut into three
        # bf16 pieces, six MFMAs per product, fp32-level error): 241 against 139 TFLOP/s at N = 8192, 172 against 107 at 2048
        # (tools/bench_prime_dual.py, profiles/r05_bench_prime_dual_bf16x3.log; gemm_cfg=17: the fp32 pipe's 128x128x32 tile);
        # small problems: the 64x64 default
        mn = min(m, n)
        self.gemm_cfg = (21 if mn >= 3072 else 20 if mn >= 1024 else -1) if gemm_cfg is None else int(gemm_cfg)
        P = nv.gemm_problem
        self._t1 = [P(self.F, self.FKy, self.T1, n, n, m, n, n, n)]             # T1 [n,n] = F^T FKy        (TN)
        self._g1 = [P(self.FKy, self.T1, self.G1, m, n, n, n, n, n)]            # G1 [m,n] = FKy T1         (NN)
        self._fky = [P(self.F, self.Ky, self.FKy, m, n, n, n, n, n)]            # FKy [m,n] = F Ky          (NN)
        self._g2 = [P(self.Kx, self.FKy, self.G2, m, n, m, m, n, n)]            # G2 [m,n] = Kx FKy         (NN)

    def flop_per_iteration(self):
        m, n = self.m, self.n
        return 2.0 * (n * n * m + m * n * n + m * n * n + m * n * m)

    def step(self):
        """One iteration of jamie.py:354-402."""
        self.iteration += 1
        c = self.gemm_cfg
        if self.iteration > 1:                        # F = 0 in the first iteration: T1 = G1 = 0 already
            nv.gemm(self._t1, nv.TN, c)
            nv.gemm(self._g1, nv.NN, c)
        nv.pd_step(self._state, self.iteration)
        nv.gemm(self._fky, nv.NN, c)
        nv.gemm(self._g2, nv.NN, c)
        if self.iteration >= self.delay:                                         # jamie.py:397-402
            nv.pd_alpha(self.G2, self.F, self.partials, self.inv_trkk, self.alpha)

    def error(self):
        """||a Kx - (F Ky) F^T|| (the reference's progress line, jamie.py:405-408).  Synchronises."""
        R = torch.empty(self.m, self.m, device=self.F.device, dtype=torch.float32)
        nv.gemm([nv.gemm_problem(self.FKy, self.F, R, self.m, self.m, self.n, self.n, self.n, self.m)], nv.NT)
        a = float(self.alpha.item())
        return float(torch.linalg.norm(a * self.Kx - R)), a

    def run(self, epoch_pd, log_pd=None, verbose=False):
        while self.iteration < epoch_pd:
            self.step()
            if verbose and log_pd and self.iteration % log_pd == 0:
                err, a = self.error()
                print('epoch:[{:d}/{:d}] err:{:.4f} alpha:{:.4f}'.format(self.iteration, epoch_pd, err, a))
        return self.F


def prime_dual(dist, dx, dy, epoch_pd=2000, rho=10, epsilon=1e-3, delay=0, log_pd=500, verbose=True, device='cuda'):
    """Drop-in for `JAMIE.Prime_Dual(dist, dx, dy, verbose)` (jamie.py:314-414): returns F as a float32 numpy array."""
    Kx, Ky = dist
    if tuple(np.shape(Kx)) == (1, 1) and tuple(np.shape(Ky)) == (1, 1):          # jamie.py:326-328
        warnings.warn('1x1 distance matrix, escaping...')
        return np.ones((1, 1), np.float32)
    pd = PrimeDual(Kx, Ky, dx, dy, rho=rho, epsilon=epsilon, delay=delay, device=device)
    return pd.run(epoch_pd, log_pd, verbose).cpu().numpy()
