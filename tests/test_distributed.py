"""world_size-2 gloo tests (CPU) of the frame-parallel path: the ring exchange itself, and a two-rank emulation of the
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
    dist.barrier()
    dist.destroy_process_group()


def test_ring_exchange_two_ranks_matches_sequential_schedule(tmp_path):
    world = 2
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


def test_single_rank_ring_is_identity():
    from x265_amd.exchange import ReferenceRing, reference_owner
    a, b = torch.zeros(4), torch.ones(4)
    ring = ReferenceRing(a, b, 0, 1)
    rec = torch.full((4,), 7.0)
    assert ring.exchange(rec) is rec and reference_owner(0, 1) == 0 and reference_owner(0, 8) == 7
