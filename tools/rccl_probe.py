#!/usr/bin/env python
"""Which algorithm / protocol / channel count does RCCL pick for the gradient exchange of this step, and at what bus
bandwidth?  (SURVEY.md §5 last row: check before hand-rolling P2P kernels.)  For whoever has an 8-GPU lease:

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29555 \
      tools/rccl_probe.py [--mb 40 --count 4 --dtype bf16|f32 --iters 20]

Defaults are config 2's exchange: 4 messages of ~40 MB (bf16: 80 MB per step in all; f32: use --mb 40 --count 4 for the
161 MB).  Rank 0 prints one JSON line: per-message time, algorithmic and bus bandwidth (2 (n-1)/n x bytes / time) of the
all-reduce, the same for the sharded optimiser's reduce-scatter and all-gather ((n-1)/n x bytes / time), the latency of a
256-byte all-reduce (bench.py's dp_model assumes 300 GB/s and 20 us: put the measured figures next to its prediction), and the
lines of RCCL's own INFO log (subsystems INIT, GRAPH, TUNING, COLL; one file per rank under --log-dir) that name the
algorithm (Ring / Tree / ...), the protocol (LL / LL128 / Simple), the channels and the transport (P2P over xGMI)."""
import argparse
import glob
import json
import os
import re
import tempfile
import time


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mb', type=float, default=40.0)
    ap.add_argument('--count', type=int, default=4)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--log-dir', default=None)
    args = ap.parse_args()
    log_dir = args.log_dir or tempfile.mkdtemp(prefix='rccl_probe_')
    os.makedirs(log_dir, exist_ok=True)
    os.environ.setdefault('NCCL_DEBUG', 'INFO')
    os.environ.setdefault('NCCL_DEBUG_SUBSYS', 'INIT,GRAPH,TUNING,COLL')
    os.environ.setdefault('NCCL_DEBUG_FILE', os.path.join(log_dir, 'rccl_%h_%p.log'))
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    local = int(os.environ.get('LOCAL_RANK', rank))
    torch.cuda.set_device(local)
    dist.init_process_group('nccl')
    dt = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    n = int(args.mb * 1e6 / (2 if args.dtype == 'bf16' else 4))
    bufs = [torch.randn(n, device='cuda').to(dt) for _ in range(args.count)]
    for b in bufs:                                  # warm-up: communicator, channels, buffers
        dist.all_reduce(b)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        works = [dist.all_reduce(b, async_op=True) for b in bufs]
        for w in works:
            w.wait()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / args.iters
    t = torch.tensor([el], device='cuda', dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el = float(t.item())
    dist.barrier()
    # the sharded optimiser's exchange (distributed.ShardedGradExchange): every message is reduce-scattered into a 1/world piece,
    # and a 1/world piece of the updated weights is all-gathered into a message-sized buffer
    m = n // world * world
    pieces = [torch.empty(m // world, device='cuda', dtype=dt) for _ in bufs]
    fulls = [torch.empty(m, device='cuda', dtype=dt) for _ in bufs]

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
        for _ in range(args.iters):
            for w in fn():
                w.wait()
        torch.cuda.synchronize()
        tt = torch.tensor([(time.perf_counter() - t1) / args.iters], device='cuda', dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())
    el_rs = timed(lambda: [dist.reduce_scatter_tensor(pc, b[:m], async_op=True) for pc, b in zip(pieces, bufs)])
    el_ag = timed(lambda: [dist.all_gather_into_tensor(f, pc, async_op=True) for f, pc in zip(fulls, pieces)])
    small = torch.zeros(64, device='cuda')
    el_small = timed(lambda: [dist.all_reduce(small, async_op=True)])
    if rank == 0:
        nbytes = n * bufs[0].element_size() * args.count
        text = ''
        for f in sorted(glob.glob(os.path.join(log_dir, 'rccl_*.log'))):
            text += open(f, errors='replace').read()
        pick = [ln.split('NCCL INFO')[-1].strip()[:200] for ln in text.splitlines()
                if re.search(r'Algo|algorithm|proto|Ring|Tree|channels|via P2P|xGMI|XGMI', ln)]
        seen, uniq = set(), []
        for ln in pick:
            key = re.sub(r'\d+', '#', ln)
            if key not in seen:
                seen.add(key)
                uniq.append(ln)
        print(json.dumps({'world': world, 'dtype': args.dtype, 'messages': args.count, 'bytes_per_step': nbytes,
                          'ms_per_step': 1e3 * el, 'alg_GBps': nbytes / el / 1e9,
                          'bus_GBps': 2 * (world - 1) / world * nbytes / el / 1e9,
                          'reduce_scatter': {'ms_per_step': 1e3 * el_rs, 'bus_GBps': (world - 1) / world * nbytes / el_rs / 1e9},
                          'all_gather': {'ms_per_step': 1e3 * el_ag, 'bus_GBps': (world - 1) / world * nbytes / el_ag / 1e9},
                          'all_reduce_256_bytes_us': 1e6 * el_small,
                          'rccl_version': '.'.join(map(str, torch.cuda.nccl.version())), 'log_dir': log_dir,
                          'rccl_log_lines': uniq[:40]}))
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
