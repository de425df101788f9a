"""Reconstructed-reference exchange between frame-parallel ranks.

x265 runs frames concurrently on frame threads and the only cross-frame edge is the reconstructed reference
(reference: encoder/frameencoder.cpp:848-861 waits on Frame::m_reconRowFlag, set by framefilter.cpp:664).  Here rank g owns
frames g, g+N, g+2N ... of the sequence; the frame a rank encodes at step s references the frame rank g-1 finished at
step s-1 (rank 0 takes rank N-1's), so after every step each rank pushes its border-extended reconstruction to rank g+1:
one point-to-point copy per rank per step (RCCL send/recv over xGMI on GPUs, gloo in the CPU tests) — no reduction."""
import torch.distributed as dist


def frame_index(step, rank, world):
    """Global index of the frame rank `rank` encodes at step `step`."""
    return step * world + rank


def reference_owner(rank, world):
    """Rank whose previous-step reconstruction is this rank's reference."""
    return (rank - 1) % world


def ring_shift(send, recv, rank, world):
    """Send `send` to rank+1 and receive rank-1's tensor into `recv` (same shape/dtype). Blocking on completion."""
    if world == 1:
        recv.copy_(send)
        return
    staged = send.is_cuda and dist.get_backend() == "gloo"      # debugging aid: gloo moves host memory only
    s, r = (send.cpu(), recv.cpu()) if staged else (send, recv)
    ops = [dist.P2POp(dist.isend, s, (rank + 1) % world), dist.P2POp(dist.irecv, r, (rank - 1) % world)]
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    if staged:
        recv.copy_(r)


class ReferenceRing:
    """Double-buffered reference planes of one rank: `current` is what the next frame pass searches; `exchange(recon)`
    publishes this rank's reconstruction and installs the incoming one as the new current reference."""

    def __init__(self, first_reference, spare, rank, world):
        self.bufs = [first_reference, spare]
        self.cur = 0
        self.rank, self.world = rank, world

    @property
    def current(self):
        return self.bufs[self.cur]

    def exchange(self, recon):
        if self.world == 1:
            # single rank: the frame just reconstructed IS the next reference; no copy, just swap roles
            return recon
        inbox = self.bufs[self.cur ^ 1]
        ring_shift(recon, inbox, self.rank, self.world)
        self.cur ^= 1
        return inbox
