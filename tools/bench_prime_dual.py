"""GPU benchmark of the correspondence stage (Prime_Dual): iterations/s and TFLOP/s of the
four fp32 MFMA products per iteration at N x N distance matrices, per GEMM tile configuration, next to the CPU oracle
(the reference's seven-product formulation) on a bounded number of iterations.
  python tools/bench_prime_dual.py [N ...]      env: CFGS=4,1  ITERS=20  CPU=1"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jamie_amd import _native as nv
from jamie_amd.correspondence import PrimeDual
nv.require_gpu()
sizes = [int(a) for a in sys.argv[1:]] or [2048, 4096, 8192]
cfgs = [int(c) for c in os.environ.get('CFGS', '4,1,2').split(',')]
iters = int(os.environ.get('ITERS', '20'))
for N in sizes:
    g = torch.Generator(device='cuda').manual_seed(0)
    X = torch.randn(N, 32, generator=g, device='cuda')
    Y = torch.randn(N, 24, generator=g, device='cuda')
    Kx, Ky = torch.cdist(X, X), torch.cdist(Y, Y)
    for cfg in cfgs:
        pd = PrimeDual(Kx, Ky, 32, 24, device='cuda', gemm_cfg=cfg)
        for _ in range(3): pd.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters): pd.step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        print(f'N {N:6d} cfg {cfg}: {dt*1e3:9.3f} ms/iteration  {pd.flop_per_iteration()/dt/1e12:7.1f} TFLOP/s (4 products)  '
              f'2000 iterations = {2000*dt:8.1f} s', flush=True)
        del pd
    if os.environ.get('CPU', '1') == '1' and N <= 4096:
        from oracle import jamie_oracle as orc
        torch.set_num_threads(min(32, os.cpu_count() or 8))
        k = 3 if N >= 4096 else 6
        Kxc, Kyc = Kx.cpu().numpy(), Ky.cpu().numpy()
        t0 = time.perf_counter()
        orc.prime_dual(Kxc, Kyc, 32, 24, k)
        dtc = (time.perf_counter() - t0) / k
        print(f'N {N:6d} CPU oracle ({torch.get_num_threads()} threads): {dtc*1e3:9.1f} ms/iteration -> GPU/CPU x{dtc/dt:.0f}', flush=True)
