"""CPU, world_size 2 (gloo): the N>1 host logic -- contiguous particle shards, shard-independent Philox
offsets, the final gather of the (P,H,d) means and the GPMP2 trust-region all-reduce arithmetic.
No compute calls (there is no GPU here): kernels are covered by the -m gpu tests."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from motion_planning_baselines_amd import parallel
    P_total, H, d = 10, 8, 4
    lo, hi = parallel.shard_range(P_total, rank, world)
    full = torch.arange(P_total * H * d, dtype=torch.float32).reshape(P_total, H, d)
    local = full[lo:hi].clone() * 2          # stand-in for "optimised" local means
    gathered = parallel.gather_means(local, P_total)          # equal shards: one all_gather_into_tensor
    # a ragged split (11 particles over 2 ranks: 6 + 5) pads to the largest shard
    rlo, rhi = parallel.shard_range(11, rank, world)
    rfull = torch.arange(11 * H * d, dtype=torch.float32).reshape(11, H, d)
    ragged = parallel.gather_means(rfull[rlo:rhi].clone(), 11)
    # reference eps order (S,d,P,H): slicing the particle axis is non-contiguous
    eps = torch.arange(3 * d * P_total * H, dtype=torch.float32).reshape(3, d, P_total, H)
    loc_eps = parallel.shard_eps(eps, rank, world)
    # trust-region damping: global mean from local sums
    dsum = torch.full((H * d,), float(rank + 1), dtype=torch.float64) * (hi - lo)
    mean = parallel.global_diag_mean(dsum, hi - lo)
    if rank == 0:
        torch.save(dict(gathered=gathered, ragged_ok=bool(torch.equal(ragged, rfull)), ok_eps=bool(torch.equal(loc_eps, eps[:, :, lo:hi].contiguous())),
                        contiguous=loc_eps.is_contiguous(), mean=mean, lo=lo, hi=hi), out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_gather(tmp_path):
    out = str(tmp_path / 'r0.pt')
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    r = torch.load(out)
    full = torch.arange(10 * 8 * 4, dtype=torch.float32).reshape(10, 8, 4) * 2
    assert torch.equal(r['gathered'], full) and r['ragged_ok']
    assert r['ok_eps'] and r['contiguous']
    assert (r['lo'], r['hi']) == (0, 5)
    # ranks hold 5 particles each with per-particle diag 1 and 2 -> global mean 1.5
    assert torch.allclose(r['mean'], torch.full((32,), 1.5, dtype=torch.float64))


def test_shard_range_covers_everything():
    from motion_planning_baselines_amd import parallel
    for P in (1, 7, 128, 32768):
        for w in (1, 2, 3, 8):
            spans = [parallel.shard_range(P, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == P
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
