"""The N>1 path on CPU: 2 processes, gloo, flat-gradient all-reduce around the model's host plan (numpy op mirror)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, PKG


def _worker(rank, world, port, out_dir, buckets=1, arc=(3, 3)):
    for p in (ROOT, PKG, os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from fake_backend import use_oracle_ops
    from gast_hip.dist import FlatGradAllReduce, shard_batch
    from model.gast_net import SpatioTemporalModelOptimized1f
    from oracle.gast_oracle import adj_from_parents
    from tests_helpers import PARENTS
    torch.manual_seed(0)                                   # identical replicas
    adj = torch.from_numpy(adj_from_parents(PARENTS[17]))
    m = use_oracle_ops(SpatioTemporalModelOptimized1f(adj, 17, 2, 17, filter_widths=list(arc), channels=16, dropout=0.0))
    sync = FlatGradAllReduce(m.parameters(), model=m, buckets=buckets)     # backward accumulates straight into the flat buffer
    g = torch.Generator().manual_seed(7)
    X = torch.rand(6, int(np.prod(arc)), 17, 2, generator=g) * 2 - 1       # the global batch; sharded on dim 0
    Y = torch.randn(6, 1, 17, 3, generator=g) * 0.3
    idx = shard_batch(6, rank, world)
    m.train()
    sync.zero_()
    loss = torch.mean(torch.norm(m(X[idx]) - Y[idx], dim=-1))
    loss.backward()
    local = sync.flat.clone()
    sync.sync()
    torch.save({'local': local, 'synced': sync.flat.clone(), 'ranges': sync.ranges, 'views_ok': all(p.grad.data_ptr() >= sync.flat.data_ptr() for p in m.parameters())},
               os.path.join(out_dir, 'rank%d.pt' % rank))
    dist.destroy_process_group()


def test_flat_grad_allreduce_world2(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, 'rank0.pt'))
    r1 = torch.load(os.path.join(tmp_path, 'rank1.pt'))
    assert r0['views_ok'] and r1['views_ok']
    expect = (r0['local'] + r1['local']) / 2
    assert torch.allclose(r0['synced'], expect, atol=1e-7) and torch.equal(r0['synced'], r1['synced'])
    assert (r0['local'] - r1['local']).abs().max() > 1e-6    # the ranks really saw different shards


def test_bucketed_allreduce_equals_monolithic_world2(tmp_path):
    """Three buckets (layers_graph_conv.2 | .1 | the rest), each all-reduced as soon as the engine reports its stage done, give the
    same averaged gradients as the single all-reduce after backward."""
    port = 31500 + (os.getpid() % 2000)
    d1, d3 = tmp_path / 'mono', tmp_path / 'bucketed'
    d1.mkdir(); d3.mkdir()
    mp.spawn(_worker, args=(2, port, str(d1), 1, (3, 3, 3)), nprocs=2, join=True)
    mp.spawn(_worker, args=(2, port + 1, str(d3), 3, (3, 3, 3)), nprocs=2, join=True)
    a0, b0 = torch.load(os.path.join(d1, 'rank0.pt')), torch.load(os.path.join(d3, 'rank0.pt'))
    b1 = torch.load(os.path.join(d3, 'rank1.pt'))
    assert len(b0['ranges']) == 3 and b0['ranges'][-1][0] == 0 and b0['ranges'][0][1] == a0['synced'].numel()
    assert sorted(b0['ranges']) == [(b0['ranges'][2][0], b0['ranges'][2][1]), b0['ranges'][1], b0['ranges'][0]]
    assert sum(e - s for s, e in b0['ranges']) == a0['synced'].numel()          # the buckets tile the buffer
    # (in bucketed mode the exchange has already happened when backward() returns: only the final buffers are comparable)
    assert torch.allclose(b0['synced'], a0['synced'], atol=1e-7, rtol=1e-6) and torch.equal(b0['synced'], b1['synced'])


@pytest.mark.parametrize('overlap', [False, True], ids=['monolithic', 'bucketed'])
def test_bench_launcher_dry_run_world2(overlap):
    """`python bench.py --gpus 2` exactly as a user (or the driver's fallback) would start it, minus the GPU: bench.py re-launches
    itself through torch.distributed.run on a free 127.0.0.1 port, every rank builds its shard from its own seed, runs the step with
    the (monolithic or bucketed) flat-gradient exchange over gloo, the barrier + max-over-ranks timing, and rank 0 prints ONE JSON
    line.  `--dry-run-cpu` swaps the HIP op set for the numpy mirror and marks the line as not-a-measurement."""
    import json
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run-cpu', '--batch', '4', '--channels', '16',
           '--steps', '2', '--warmup', '1'] + (['--overlap'] if overlap else [])
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]                      # one line, from rank 0 only
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 2 and d['warmup'] == 1 and d['scaling'] == 'weak'
    assert d['dry_run'] is True and d['value'] is None and 'DRY RUN' in d['data']
    assert d['cpu_baseline'] is None and 'N = 1' in d['cpu_baseline_note']
    cfg = d['config']
    assert cfg['rccl_world_size'] == 2 and cfg['global_batch'] == 8 and cfg['parallelism'] == 'dp2'
    assert cfg['per_rank_input_seeds'] == [1234, 1235]
    gx = cfg['gradient_exchange']
    assert gx['buckets'] == (3 if overlap else 1) and gx['overlap_with_backward'] is overlap
    assert sum(gx['bucket_bytes']) == gx['bytes_per_step'] > 0
    assert len(d['per_rank_ms_per_step']) == 2 and d['ms_per_step'] == max(d['per_rank_ms_per_step'])
    assert np.isfinite(cfg['loss_last'])
