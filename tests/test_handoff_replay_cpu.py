"""Host replay of the ONE-barrier LDS hand-off of the two-wave-group kernels: conv_halo.hip (3x3 forward / data gradient,
one tap per step), conv_wgrad_rows.hip and conv_wgrad_pw.hip.  Round 3 removed the second workgroup barrier per step from all three; until
now the argument that the remaining barrier (b1) carries every hand-off was a source comment plus a 200-repeat stress run.

The replay executes both groups' PROGRAM ORDER (what each wave issues, reads, waits for and where it arrives at b1) for a whole block
lifetime and derives, per operation, the barrier EPOCH it falls in (epoch e = between passing b1(e-1) and arriving at b1(e)).  A wave
passes b1(e) only after every wave has arrived there, so an operation COMPLETED in epoch e1 happens-before any operation of epoch e2 > e1 of
ANY wave, while two operations of the same epoch on different waves are unordered.  LDS reads are complete at the wave's next lgkmcnt(0)
(always before its next barrier arrival in these kernels); an LDS-DMA write may land anywhere between its issue and the counted
s_waitcnt vmcnt(N) that retires it (in-order retirement: everything but the N youngest operations of THAT wave is done).  Rules checked:

  RAW  every DMA piece of step u (both groups' pieces) is retired in an epoch < the epoch of every read of step u;
  WAR  a DMA into a ring slot / patch buffer is ISSUED in an epoch > the epoch of every read of the slot's previous occupant.

The same engine is run with the kernels' real constants (ring depth, prefetch distance, DMA instructions per wave and step, patch pieces,
steps per chunk) and, as a self-check, with perturbed constants that MUST fail."""
import pytest


class Wave(object):
    """One wave group's instruction stream (all waves of a group run the same program; each wave moves its own pieces of every step)."""

    def __init__(self, name):
        self.name = name
        self.epoch = 0
        self.queue = []          # outstanding vector-memory operations, oldest first: [kind, step, issue_epoch]
        self.writes = []         # (kind, step, issue_epoch, retire_epoch)
        self.reads = []          # (kind, step, epoch)

    def dma(self, kind, step, n=1):
        for _ in range(n):
            self.queue.append((kind, step, self.epoch))

    def wait(self, allowed):
        """s_waitcnt vmcnt(allowed): everything but the `allowed` youngest operations is retired NOW (epoch of the upcoming barrier)."""
        while len(self.queue) > allowed:
            kind, step, e_issue = self.queue.pop(0)
            self.writes.append((kind, step, e_issue, self.epoch))

    def read(self, kind, step):
        self.reads.append((kind, step, self.epoch))

    def barrier(self):
        self.epoch += 1

    def drain(self):
        self.wait(0)


def check(groups, occupant_distance):
    """occupant_distance[kind] = steps between two occupants of the same LDS region (ring depth; patch: 2 chunks)."""
    errors = []
    for kind, dist in occupant_distance.items():
        reads = {}
        for g in groups:
            for k, step, e in g.reads:
                if k == kind:
                    reads.setdefault(step, []).append((g.name, e))
        loaded = set()
        for g in groups:
            for k, step, e_issue, e_ret in g.writes:
                if k != kind:
                    continue
                loaded.add(step)
                for rname, e_read in reads.get(step, []):
                    if not e_ret < e_read:
                        errors.append("RAW %s step %d: %s's piece retires in epoch %d, %s reads it in epoch %d" % (kind, step, g.name, e_ret, rname, e_read))
                for rname, e_read in reads.get(step - dist, []):
                    if not e_read < e_issue:
                        errors.append("WAR %s step %d: %s issues in epoch %d, %s still reads step %d in epoch %d" % (kind, step, g.name, e_issue, rname, step - dist, e_read))
        for step in reads:
            if step not in loaded:
                errors.append("%s step %d is read but no wave ever loaded it" % (kind, step))
    return errors


# --------------------------------------------------------------------------------------------------------------------------------------
# conv_wgrad_rows.hip (DEPTH 6, P 4, 2 DMA per wave and K-step) and conv_wgrad_pw.hip (DEPTH 4, P 2, 4 DMA): main loop at
# conv_wgrad_rows.hip:303-369 / conv_wgrad_pw.hip:202-245.
#   prologue: DMA(0..P-1); vmcnt(0); barrier
#   A, cycle v:  DMA(v+P); read(v); vmcnt((P-1)*DPW); b1; MFMA(v)
#   B:           read(0); DMA(P);  cycle v:  MFMA(v); vmcnt((P-1)*DPW); b1; DMA(v+1+P); read(v+1)
def replay_wgrad(depth, ahead, dpw, steps, wait_steps=None):
    wait_steps = ahead - 1 if wait_steps is None else wait_steps
    A, B = Wave("A"), Wave("B")
    for w in (A, B):
        for s in range(ahead):
            w.dma("ring", s, dpw)
        w.drain()
        w.barrier()                                  # the prologue's barrier is epoch boundary 0 -> 1
    B.read("ring", 0)
    B.dma("ring", ahead, dpw)
    for v in range(steps):
        A.dma("ring", v + ahead, dpw)
        A.read("ring", v)
        A.wait(wait_steps * dpw)
        A.barrier()
        B.wait(wait_steps * dpw)
        B.barrier()
        B.dma("ring", v + 1 + ahead, dpw)
        B.read("ring", v + 1)
    A.drain(); B.drain()
    return check([A, B], {"ring": depth})


@pytest.mark.parametrize("name,depth,ahead,dpw", [("conv_wgrad_rows", 6, 4, 2), ("conv_wgrad_pw", 4, 2, 4)])
def test_weight_gradient_kernels_one_barrier_handoff(name, depth, ahead, dpw):
    assert replay_wgrad(depth, ahead, dpw, 40) == [], name
    # self-check of the engine: a ring as deep as the prefetch distance reuses a slot its readers may still hold (WAR), and a counted
    # wait that lets one more K-step stay in flight hands an unfinished step to the readers (RAW)
    assert any(e.startswith("WAR") for e in replay_wgrad(ahead, ahead, dpw, 40))
    assert any(e.startswith("RAW") for e in replay_wgrad(depth, ahead, dpw, 40, wait_steps=ahead))


# --------------------------------------------------------------------------------------------------------------------------------------
# conv_halo.hip, one tap per step (TPS 1: NSW 4, D 3, 9 steps per 64-channel chunk), main loop at conv_halo.hip:760-900.
#   weight ring: NSW stages, tile of step c + D issued in cycle c; WL DMA instructions per wave and step
#   patch: two buffers; chunk q + 1's PL pieces are issued ONE per step in steps 0 .. PL-1 of chunk q, after that step's weight pieces
#   bits (data gradient with ConvArgs::mask_bits): group A DMAs the item's 4 KiB of ReLU bits in step 1 of the item's first chunk; both
#        groups read them in the item's epilogue (A: in the memory phase of the next item's step 0, B: after b1 of the item's last step)
#   prologue: weight tiles 0 .. D-1 and the first patch; vmcnt(0); barrier
#   A, cycle c:  read(c) [+ epilogue reads of the previous item at its step 0]; W(c+D); patch piece; bits; counted wait; b1; MFMA(c)
#   B:           read(0);  cycle c:  MFMA(c); counted wait; b1; [epilogue reads at the item's last step]; read(c+1); W(c+D); patch piece
def replay_halo(nsw, wl, pl, chunks_per_item, items, bits, spc=9, drop_patch_wait=False, patch_buffers=2):
    D = nsw - 1
    A, B = Wave("A"), Wave("B")
    total = items * chunks_per_item * spc
    for w in (A, B):
        for s in range(D):
            w.dma("w", s, wl)
        w.dma("patch", 0, pl)
        w.drain()
        w.barrier()

    def np_a(s):
        return sum(1 for k in range(pl) if s - (D - 1) <= k <= s)

    def np_b(s):
        return sum(1 for k in range(pl) if s - (D - 1) <= k <= s - 1)

    def reads(w, c):
        w.read("w", c)
        w.read("patch", c // spc)

    reads(B, 0)
    for c in range(total):
        chunk, step = divmod(c, spc)
        item, cc = divmod(chunk, chunks_per_item)
        more_w = c + D < total
        live = chunk + 1 < items * chunks_per_item            # a next chunk exists: its patch is loaded during this one
        # ---- group A
        reads(A, c)
        if bits and step == 0 and cc == 0 and item > 0:
            A.read("bits", item - 1)                           # epilogue of the previous item
        if more_w:
            A.dma("w", c + D, wl)
        if live and step < pl:
            A.dma("patch", chunk + 1, 1)
        if bits and step == 1 and cc == 0:
            A.dma("bits", item, 1)
        if not more_w:
            A.wait(0)
        else:
            A.wait((D - 1) * wl + (np_a(step) if (live and not drop_patch_wait) else (pl if drop_patch_wait and live else 0)))
        A.barrier()
        # ---- group B
        if c > 0 and c + D - 1 >= total:
            B.wait(0)
        else:
            B.wait((D - 2) * wl + (np_b(step) if (live and not drop_patch_wait) else (pl if drop_patch_wait and live else 0)))
        B.barrier()
        if bits and step == spc - 1 and cc == chunks_per_item - 1:
            B.read("bits", item)                               # epilogue of this item
        if c + 1 < total:
            reads(B, c + 1)
        if more_w:
            B.dma("w", c + D, wl)
        if live and step < pl:
            B.dma("patch", chunk + 1, 1)
    if bits:
        A.read("bits", items - 1)
    A.drain(); B.drain()
    dist = {"w": nsw, "patch": patch_buffers}
    if bits:
        dist["bits"] = 1
    return check([A, B], dist)


@pytest.mark.parametrize("wl,pl", [(2, 6), (1, 6)])                     # 128-wide tiles (WL = 2) / 64-wide (WL = 1); 8x32 and 16x16 patches: 6 pieces per wave
@pytest.mark.parametrize("chunks_per_item", [1, 2, 4, 8])                # C = 64 .. 512
@pytest.mark.parametrize("bits", [False, True])
def test_halo_kernel_one_barrier_handoff(wl, pl, chunks_per_item, bits):
    assert replay_halo(4, wl, pl, chunks_per_item, 3, bits) == []


def test_halo_replay_engine_detects_broken_protocols():
    # a three-stage ring with the same prefetch distance D = 3 rewrites the stage the slower group still reads
    errs = replay_halo(4, 2, 6, 2, 2, False)
    assert errs == []
    A_like = replay_wgrad(3, 3, 2, 30)
    assert any(e.startswith("WAR") for e in A_like)
    # patch pieces left out of the counted waits' bookkeeping (all PL allowed to stay in flight at every wait) reach their readers unfinished
    assert any(e.startswith("RAW patch") for e in replay_halo(4, 2, 6, 2, 2, False, drop_patch_wait=True))
    # ONE patch buffer: chunk q + 1's pieces would overwrite the patch chunk q is still being read from
    assert any(e.startswith("WAR patch") for e in replay_halo(4, 2, 6, 2, 2, False, patch_buffers=1))
