"""csrc/dense_fused.hip, loader waves: a model check of the issue / counted-wait schedule (the counted `s_waitcnt vmcnt(N)` of the
fused kernels' loaders was covered by the GPU parity tests alone).  The schedule is restated here from `LoaderPlan` and the loader loop --
which pieces a loader requests in the iteration of granule G, in which order, and the N it then waits with -- and replayed against the two
facts the hardware gives: a wave's vector-memory operations retire IN ORDER, and `vmcnt(N)` returns when all but the N youngest have.  At
every barrier the test checks what the compute waves are about to read: the DMA'd weight pieces of granule G+1, and, when G+1 opens a
chunk, every row piece of that chunk.  It also checks that nothing is written into a staging buffer or weight slot that a compute wave may
still be reading.  All four loaders, the tail (EXT 5) and both growth pairs (EXT 2, 3), many steps."""
import pytest

NLOAD = 4


class Plan:
    def __init__(self, ext, nb0, nb1, mode, lw):
        self.ext, self.nb0, self.nb1, self.mode, self.lw = ext, nb0, nb1, mode, lw
        self.nbt, self.extg, self.ngr = nb0 + nb1, 3 * ext, 3 * (ext + 1)
        self.rt_e, self.rt_r = (2 if mode == 1 else 3), 2
        self.nsb, self.nws = (3, 2) if mode == 0 else (2, 3)
        self.wl, self.sl = self.nws - 1, self.nsb - 1

    def nrows(self, kx):
        return 2 if kx == 0 else (1 if (kx == 1 and self.lw < 3) else 0)

    def rows(self, kx):
        return [self.lw, self.lw + 4] if kx == 0 else ([8 + self.lw] if (kx == 1 and self.lw < 3) else [])

    def nst_at(self, i):
        i %= self.ngr
        return 3 * self.nrows(i % 3) if i < self.extg else 0

    def nwdma(self, iw):
        ext = iw < self.extg
        nw, rt = (self.nbt * 3, self.rt_e) if ext else (self.nb1 * 3, self.rt_r)
        return sum(1 for t in range((nw + 3) // 4) if self.lw + 4 * t < nw and t >= rt)

    def wait_n(self, i):
        if self.mode == 0:
            return self.nst_at(i) + self.nst_at(i - 1) + self.nst_at(i - 2)
        third = i < self.extg and i % 3 == 2
        return self.nwdma((i + self.wl) % self.ngr) + self.nst_at(i) + (0 if third else self.nst_at(i - 1))


def simulate(ext, nb0, nb1, mode, lw, nsteps):
    """-> (events, plan): events = per global granule G the list of ops issued in its iteration (in order) and the wait N that follows."""
    p = Plan(ext, nb0, nb1, mode, lw)
    issued = []                                   # ops in issue order: ("w", granule) | ("r", chunk number, row)
    done = 0                                      # ops [0, done) have completed (in-order retirement)
    # prologue: weights of the first WL granules (DMA pieces only count), the first SL chunks' rows; then vmcnt(0) before the first barrier
    for g in range(p.wl):
        issued += [("w", g)] * p.nwdma(g % p.ngr)
    for c0 in range(p.sl):
        for kx in (0, 1):
            for row in p.rows(kx):
                issued += [("r", c0, row)] * 3
    done = len(issued)
    G, nch = 0, 0
    checks = []
    for s in range(nsteps):
        for i in range(p.ngr):
            # --- barrier that opens granule G: what must have landed (as seen by THIS loader's own pieces)
            need = [op for op in issued if (op[0] == "w" and op[1] == G)]
            if i < p.extg and i % 3 == 0:
                need += [op for op in issued if op[0] == "r" and op[1] == nch]
            for op in need:
                idx = max(k for k, o in enumerate(issued) if o == op)
                assert idx < done, (ext, mode, lw, "granule", G, "not landed", op)
            # --- the iteration: weights of granule G + WL into slot (G + WL) % NWS -- last read in granule G - 1 when NWS = WL + 1 + ... check below
            assert (G + p.wl) % p.nws != G % p.nws or p.wl == 0                 # never the slot being read
            issued += [("w", G + p.wl)] * p.nwdma((i + p.wl) % p.ngr)
            if i < p.extg:
                c, kx = divmod(i, 3)
                target = nch + p.sl                                              # chunk number staged now -> buffer target % NSB
                assert target % p.nsb != nch % p.nsb                             # not the buffer being read ...
                assert all((nch + d) % p.nsb != target % p.nsb for d in range(1, p.sl))   # ... nor one already staged and not yet read
                for row in p.rows(kx):
                    issued += [("r", target, row)] * 3
            n = p.wait_n(i)
            assert 0 <= n < 64
            done = max(done, len(issued) - n)
            checks.append((G, n))
            if i < p.extg and i % 3 == 2:
                nch += 1
            G += 1
    return checks, p


@pytest.mark.parametrize("cfg", [(5, 2, 4, 1), (2, 2, 2, 0), (3, 2, 2, 0)])
def test_every_piece_has_landed_when_its_barrier_opens(cfg):
    ext, nb0, nb1, mode = cfg
    rows_seen = {}
    for lw in range(NLOAD):
        checks, p = simulate(ext, nb0, nb1, mode, lw, nsteps=7)
        assert len(checks) == 7 * p.ngr
        for kx in range(3):
            for r in p.rows(kx):
                rows_seen[r] = rows_seen.get(r, 0) + 1
    assert sorted(rows_seen) == list(range(11)) and all(v == 1 for v in rows_seen.values())      # the four loaders cover the 11 staged rows once


def test_weight_pieces_are_covered_once_by_the_four_loaders():
    for ext, nb0, nb1, mode in ((5, 2, 4, 1), (2, 2, 2, 0), (3, 2, 2, 0)):
        plans = [Plan(ext, nb0, nb1, mode, lw) for lw in range(NLOAD)]
        for iw in range(plans[0].ngr):
            nw = plans[0].nbt * 3 if iw < plans[0].extg else nb1 * 3
            pieces = sorted(lw + 4 * t for lw in range(NLOAD) for t in range((nw + 3) // 4) if lw + 4 * t < nw)
            assert pieces == list(range(nw)), (ext, iw)
            if mode == 0:
                assert all(p.nwdma(iw) == 0 for p in plans)                      # growth pairs: every weight piece is register resident


def test_model_catches_a_wrong_wait():
    """Self-test: leaving one more piece in flight than the schedule allows must trip the landed-check."""
    real = Plan.wait_n
    try:
        Plan.wait_n = lambda self, i: real(self, i) + 1
        with pytest.raises(AssertionError):
            for lw in range(NLOAD):
                simulate(5, 2, 4, 1, lw, nsteps=4)
    finally:
        Plan.wait_n = real


# ---- conv1_stream_kernel (csrc/dense_fused.hip): units k = (step k // 2, chunk k % 2) through four staging buffers ----------------------------

def _conv1_loader_replay(lw, nunits):
    """The loader loop of conv1_stream_kernel restated: prologue units 0..2; per barrier k: [k == 0: wait all but 2 NP]; barrier; issue unit
    k + 3 into buffer (k + 3) % 4; wait all but 2 NP.  Returns, per barrier k, the units of this loader's pieces that have landed when the
    barrier opens, and the unit whose buffer the iteration then overwrites."""
    np_ = 3 * (3 if lw < 2 else 2)
    issued, done = [], 0

    def wait_all_but(n):
        nonlocal done
        done = max(done, len(issued) - n)

    for k in (0, 1, 2):
        issued += [k] * np_
    landed_at, overwritten_at = [], []
    for k in range(nunits):
        if k == 0:
            wait_all_but(2 * np_)
        landed_at.append(list(issued[:done]))
        overwritten_at.append(k + 3 - 4)                    # the unit that lived in buffer (k + 3) % 4 before unit k + 3
        issued += [k + 3] * np_
        wait_all_but(2 * np_)
    return landed_at, overwritten_at


@pytest.mark.parametrize("lw", range(NLOAD))
def test_conv1_stream_loader_schedule(lw):
    """Barrier k lets the compute waves read unit k (buffer k % 4): every piece of it has landed; and the unit an iteration overwrites is one
    the compute waves finished before that barrier (unit k - 1), never the one they are about to read or a later one."""
    landed_at, overwritten_at = _conv1_loader_replay(lw, 40)
    for k, (landed, over) in enumerate(zip(landed_at, overwritten_at)):
        assert landed.count(k) == 3 * (3 if lw < 2 else 2), (lw, k)     # every piece of unit k has landed when barrier k opens
        assert over == k - 1, (lw, k, over)                 # its buffer is free: the compute waves passed barrier k only after finishing unit k - 1
        assert (k + 3) % 4 == (k - 1) % 4
