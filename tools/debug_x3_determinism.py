"""GPU: is configuration 20 (bf16x3) bit-reproducible, alone and inside grouped launches?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jamie_amd import _native as nv
nv.require_gpu()
B = 512
torch.manual_seed(0)
def mk(layout, M, N, K):
    if layout == nv.NT: return torch.randn(M, K, device='cuda'), torch.randn(N, K, device='cuda'), K, K
    if layout == nv.NN: return torch.randn(M, K, device='cuda'), torch.randn(K, N, device='cuda'), K, N
    return torch.randn(K, M, device='cuda'), torch.randn(K, N, device='cuda'), M, N
for layout, name, shapes in ((nv.TN, 'TN', [(4000, 2000, B), (2000, 1000, B), (2000, 4000, B), (1000, 2000, B)]),
                             (nv.NT, 'NT', [(B, 4000, 2000), (B, 2000, 1000)]), (nv.NN, 'NN', [(B, 2000, 4000), (B, 1000, 2000)])):
    ops = [mk(layout, *s) for s in shapes]
    def run(idx, cfg, sk=1):
        outs = [torch.full((sk, shapes[i][0], shapes[i][1]), float('nan'), device='cuda') for i in idx]
        nv.gemm([nv.gemm_problem(ops[i][0], ops[i][1], o, *shapes[i], ops[i][2], ops[i][3], shapes[i][1], splitk=sk, slab_stride=shapes[i][0] * shapes[i][1])
                 for i, o in zip(idx, outs)], layout, cfg)
        torch.cuda.synchronize()
        return outs
    for sk in (1, 2):
        ref = run(list(range(len(shapes))), 20, sk)
        bad = 0
        for rep in range(10):
            again = run(list(range(len(shapes))), 20, sk)
            bad += sum(int((a != b).sum()) for a, b in zip(ref, again))
        solo = [run([i], 20, sk)[0] for i in range(len(shapes))]
        bad_solo = sum(int((a != b).sum()) for a, b in zip(ref, solo))
        pairs = run([0, 1], 20, sk)
        bad_pair = sum(int((a != b).sum()) for a, b in zip(ref[:2], pairs))
        f32 = run(list(range(len(shapes))), 17, sk)
        err = max(float((a.sum(0) - b.sum(0)).abs().max()) for a, b in zip(ref, f32))
        print(f'{name} sk {sk}: repeat mismatches {bad}, grouped vs solo {bad_solo}, vs pair {bad_pair}, max |cfg20 - cfg17| {err:.3e}', flush=True)
