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
    """Double-buffered reference planes of one rank: `current` is what the next frame pass searches.

    `exchange(recon)` publishes this rank's reconstruction and installs the incoming one as the new current reference.
    `begin(recon)` / `finish()` split that so the transfer overlaps the frame passes that do not need it (bench.py launches the
    chains whose reference is local between the two calls; only chain 0 waits for the incoming picture)."""

    def __init__(self, first_reference, spare, rank, world):
        self.bufs = [first_reference, spare]
        self.cur = 0
        self.rank, self.world = rank, world
        self._pending = None

    @property
    def current(self):
        return self.bufs[self.cur]

    def begin(self, recon):
        assert self._pending is None
        if self.world == 1:
            self._pending = (recon, None, None)       # single rank: the frame just reconstructed IS the next reference
            return
        inbox = self.bufs[self.cur ^ 1]
        staged = recon.is_cuda and dist.get_backend() == "gloo"      # debugging aid: gloo moves host memory only
        s, r = (recon.cpu(), inbox.cpu()) if staged else (recon, inbox)
        ops = [dist.P2POp(dist.isend, s, (self.rank + 1) % self.world), dist.P2POp(dist.irecv, r, (self.rank - 1) % self.world)]
        self._pending = (inbox, dist.batch_isend_irecv(ops), r if staged else None)

    def finish(self):
        ref, works, staged = self._pending
        self._pending = None
        if works is None:
            return ref
        for w in works:
            w.wait()
        if staged is not None:
            ref.copy_(staged)
        self.cur ^= 1
        return ref

    def exchange(self, recon):
        self.begin(recon)
        return self.finish()


class FrameChains:
    """The frame schedule of one rank with F frame passes in flight (bench.py; x265 runs several frame encoders per device the same way).

    Chain j of rank g encodes frame (step * N + g) * F + j.  Its reference is the reconstruction chain j-1 produced one step earlier; chain 0
    takes the last chain of rank g-1 through the ring.  `step(launch)` drives one step: the hand-over of the previous step's last
    reconstruction is STARTED first, chains 1..F-1 (whose references are local) are launched while it is in flight, and only chain 0 waits
    for the incoming picture.  `launch(j, k, ref)` runs chain j of step k against `ref` and returns its reconstruction; `before_exchange(k)`
    / `after_exchange(k)` are optional hooks around the transfer (bench.py orders its HIP streams there)."""

    def __init__(self, ring, frames_in_flight, first_refs):
        assert frames_in_flight >= 1 and len(first_refs) == frames_in_flight
        self.ring, self.F = ring, frames_in_flight
        self.refs = list(first_refs)
        self.last = None               # reconstruction of the last chain of the previous step, still to be handed to the next rank
        self.k = 0

    def step(self, launch, before_exchange=None, after_exchange=None):
        k, F = self.k, self.F
        recs = [None] * F
        if self.last is not None:
            if before_exchange:
                before_exchange(k)
            self.ring.begin(self.last)
        for j in range(1, F):
            recs[j] = launch(j, k, self.refs[j])
        if self.last is not None:
            self.refs[0] = self.ring.finish()
            if after_exchange:
                after_exchange(k)
        recs[0] = launch(0, k, self.refs[0])
        self.refs = [self.refs[0]] + recs[:F - 1]
        self.last = recs[F - 1]
        self.k = k + 1
        return recs
