import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
sys.argv = ['x']
os.environ['CFGS'] = '0'
import importlib.util, torch
spec = importlib.util.spec_from_file_location('psk', os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tools/bench_gemm_f32_psk.py'))
src = open(spec.origin).read().split('cases = [')[0]
exec(src)
B, d = 512, (2000, 1000)
tests = [('NT d->2d', nv.NT, [(B, 2 * x, x) for x in d], {1: [(1, 1)], 2: [(3, 2)], 4: [(3, 2), (4, 3)]}),
         ('NT 2d->d', nv.NT, [(B, x, 2 * x) for x in d], {1: [(4, 2)], 4: [(6, 4), (6, 3), (7, 4), (7, 3)], 2: [(6, 3), (7, 4)]}),
         ('NN dy[2d]W', nv.NN, [(B, x, 2 * x) for x in d], {1: [(4, 2)], 4: [(6, 4), (6, 3), (7, 4)], 2: [(6, 4), (6, 3)]}),
         ('NN dy[d]W', nv.NN, [(B, 2 * x, x) for x in d], {1: [(1, 1)], 4: [(3, 2)], 2: [(3, 2)]})]
for name, layout, shapes, plan in tests:
    for cfg, skl in plan.items():
        for sks in skl:
            us, tf = run(layout, shapes, sks, cfg)
            print(f'{name:12s} cfg {cfg} sk {sks}: {us:8.1f} us {tf:7.1f} TFLOP/s', flush=True)
