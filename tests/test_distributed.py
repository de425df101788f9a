"""world_size-2 and -3 gloo tests (CPU) of the frame-parallel path: the ring exchange itself, and a two-rank emulation of the
bench loop in which the oracle stands in for the GPU frame pass — the frames each rank produces must equal the frames a
single process produces when it follows the same reference schedule."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

W, H, STEPS = 72, 72, 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene(idx):
    from x265_amd.synth import make_scene
    return make_scene(W, H, depth=8, seed=100 + idx, tile=24, vmax=3)


def _recon(src, ref):
    from frame_oracle import oracle_frame_pass
    r = oracle_frame_pass(src, ref, depth=8, qp=30, merange=8)
    return np.ascontiguousarray(r["recon"][96:96 + H, 96:96 + W])


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from x265_amd.exchange import ReferenceRing, frame_index, ring_shift
    # 1. the raw ring shift
    a, b = torch.full((5, 7), rank, dtype=torch.uint8), torch.zeros((5, 7), dtype=torch.uint8)
    ring_shift(a, b, rank, world)
    assert int(b[0, 0]) == (rank - 1) % world
    # 2. the bench loop with the oracle as the frame pass
    first = torch.from_numpy(_scene(1000 + rank)["ref"].copy())
    ring = ReferenceRing(first, torch.empty_like(first), rank, world)
    ref = ring.current
    produced = []
    for s in range(STEPS):
        sc = _scene(frame_index(s, rank, world))
        rec = torch.from_numpy(_recon(sc["src"], ref.numpy()))
        produced.append(rec.numpy().copy())
        ref = ring.exchange(rec)
    np.save(os.path.join(out, "rank%d.npy" % rank), np.stack(produced))
    # 3. bench.py's schedule with F = 2 frame passes in flight per rank (FrameChains: hand-over started first, chain 1 launched while it is
    #    in flight, chain 0 after the incoming reference arrived)
    from x265_amd.exchange import FrameChains
    F = 2
    firsts = [torch.from_numpy(_scene(2000 + 10 * rank + j)["ref"].copy()) for j in range(F)]
    ring2 = ReferenceRing(firsts[0], torch.empty_like(firsts[0]), rank, world)
    chains = FrameChains(ring2, F, [ring2.current] + firsts[1:])
    order, out2 = [], []

    def launch(j, k, ref):
        order.append(j)
        sc = _scene((k * world + rank) * F + j)
        return torch.from_numpy(_recon(sc["src"], ref.numpy()))
    for s in range(STEPS + 1):
        out2.append(np.stack([r.numpy().copy() for r in chains.step(launch)]))
    assert order == [1, 0] * (STEPS + 1)                     # chain 0 (the one that needs the transfer) is always launched last
    np.save(os.path.join(out, "chains%d.npy" % rank), np.stack(out2))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ring_exchange_two_ranks_matches_sequential_schedule(tmp_path, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = [np.load(os.path.join(str(tmp_path), "rank%d.npy" % r)) for r in range(world)]
    # single-process emulation of the same schedule
    refs = [_scene(1000 + r)["ref"] for r in range(world)]
    for s in range(STEPS):
        recs = []
        for r in range(world):
            sc = _scene(s * world + r)
            recs.append(_recon(sc["src"], refs[r]))
            assert np.array_equal(got[r][s], recs[r]), (s, r)
        refs = [recs[(r - 1) % world] for r in range(world)]


@pytest.mark.parametrize("world", [2, 3])
def test_frame_chains_two_ranks_match_the_reference_relation(tmp_path, world):
    """F = 2 chains per rank, 2 (and 3: an odd ring) ranks: every reconstruction equals what the dependency rule gives — chain j references
    chain j-1 of the same rank one step earlier, chain 0 references the last chain of the previous rank one step earlier."""
    F = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = [np.load(os.path.join(str(tmp_path), "chains%d.npy" % r)) for r in range(world)]       # [step][chain]
    prev = [[_scene(2000 + 10 * r + j)["ref"] for j in range(F)] for r in range(world)]           # references of step 0
    for s in range(STEPS + 1):
        recs = [[None] * F for _ in range(world)]
        for r in range(world):
            for j in range(F):
                sc = _scene((s * world + r) * F + j)
                recs[r][j] = _recon(sc["src"], prev[r][j])
                assert np.array_equal(got[r][s][j], recs[r][j]), (s, r, j)
        prev = [[recs[(r - 1) % world][F - 1]] + recs[r][:F - 1] for r in range(world)]


def test_single_rank_ring_is_identity():
    from x265_amd.exchange import ReferenceRing, reference_owner
    a, b = torch.zeros(4), torch.ones(4)
    ring = ReferenceRing(a, b, 0, 1)
    rec = torch.full((4,), 7.0)
    assert ring.exchange(rec) is rec and reference_owner(0, 1) == 0 and reference_owner(0, 8) == 7
