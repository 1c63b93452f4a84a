"""Host logic of the dataflow Cholesky (csrc/cholflow.hip): the task list every form of the factorisation runs from.
No device needed: the list is built on the host.  The test replays the list in ticket order next to a model of the chain
kernel's workgroups and checks what the kernels rely on (cholflow.hip, "Order and progress"):

  * a ticket only ever waits for tiles of EARLIER tickets or for the chain kernel's workgroups, which in turn only wait for
    earlier tickets (no deadlock whatever the residency);
  * every tile of the lower triangle (and of the right-hand-side rows) gets every block-column update exactly once -- from
    the tile kernel or from the chain workgroup the form assigns it to -- and is finished exactly once, after all of them.
"""
import ctypes as C

import numpy as np
import pytest

from bundle_adjustment_amd import engine

FIN = 1 << 20


PART, NPART_SHIFT, BUF_SHIFT = 1 << 21, 22, 12


def tasks(nb, rows, w, chain, second=0, split_m=1, split_from=1 << 30):
    lib = engine.load_library()
    lib.jaicov_debug_flow_tasks2.argtypes = [C.c_int] * 7 + [C.c_void_p, C.c_int]
    n = lib.jaicov_debug_flow_tasks2(nb, rows, w, chain, second, split_m, split_from, None, 0)
    assert n > 0
    out = np.zeros((n, 4), np.int32)
    assert lib.jaicov_debug_flow_tasks2(nb, rows, w, chain, second, split_m, split_from, out.ctypes.data, n) == n
    return out


class Replay:
    def __init__(self, nb, rows, chain, second):
        self.nb, self.rows, self.chain, self.second = nb, rows, chain, second
        self.done = np.zeros((rows, nb), bool)       # L[i][k] final (done[k][k]: the inverse too)
        self.stored = np.zeros((rows, nb), bool)     # the tile is in memory ...
        self.applied = np.zeros((rows, nb), int)     # ... with that many block columns subtracted
        self.updates = np.zeros((rows, nb), int)
        self.factored = np.zeros(nb, bool)
        self.c0 = 0          # workgroup 0: next block column
        self.c2 = 0          # workgroup 2: next block column
        self.half2 = False   # workgroup 2 has finished tile (c2+2, c2), its update of (c2+2, c2+1) is still to come

    def has(self, i, j, k):
        return self.stored[i, j] and self.applied[i, j] == k

    def chain_kernel(self):
        """potrf_chain_kernel's workgroups, as far as what is in memory lets them go."""
        nb = self.nb
        moved = True
        while moved:
            moved = False
            c = self.c0
            if c < nb and not self.factored[c]:      # workgroup 0: potrf(c); tile (0, 0) from memory, the others are in its LDS
                if c > 0 or self.has(0, 0, 0):
                    self.factored[c] = moved = True
            if c < nb and self.factored[c]:
                if c + 1 >= nb:
                    self.c0 = nb
                elif self.has(c + 1, c, c) and self.has(c + 1, c + 1, c):
                    self.done[c + 1, c] = True       # L[c+1][c] by forward substitution
                    self.updates[c + 1, c + 1] += 1  # ... subtracted from the next diagonal tile, which stays in LDS
                    self.c0 = c + 1
                    moved = True
            for k in range(nb):                      # workgroup 1: the inverses
                if self.factored[k] and not self.done[k, k]:
                    self.done[k, k] = moved = True
            if self.second >= 1:
                c = self.c2
                if c + 2 < nb and self.factored[c]:
                    if not self.half2 and self.has(c + 2, c, c):
                        self.done[c + 2, c] = True
                        self.half2 = moved = True
                    if self.half2:
                        if self.second < 2:
                            self.c2, self.half2, moved = c + 1, False, True
                        elif self.has(c + 2, c + 1, c) and self.done[c + 1, c]:
                            self.updates[c + 2, c + 1] += 1
                            self.applied[c + 2, c + 1] = c + 1
                            self.c2, self.half2, moved = c + 1, False, True

    def run(self, t):
        psum = {}                                    # partial-sum buffer -> (tile, block columns it holds)
        for (i, j, z, wd) in t:
            k0, buf = int(z) & ((1 << BUF_SHIFT) - 1), int(z) >> BUF_SHIFT
            k1, fin = int(wd) & (FIN - 1), bool(int(wd) & FIN)
            part, npart = bool(int(wd) & PART), (int(wd) >> NPART_SHIFT) & 7
            assert 0 <= j <= i < self.rows and j < self.nb and 0 <= k0 <= k1 <= j
            if self.chain:
                self.chain_kernel()
            for k in range(k0, k1):                  # operands: final by now
                assert self.done[i, k] and self.done[j, k], ("ticket would wait for a later ticket", i, j, k)
            if part:                                 # a partial sum of a split range: its own buffer, written once
                assert not fin and npart == 0 and buf not in psum and k1 > k0
                psum[buf] = ((i, j), k0, k1)
                continue
            if npart > 0:                            # the tile's own task: first visit, the last piece, + the partial sums of EARLIER tickets
                assert not self.stored[i, j]
                at = 0
                for q in range(npart):
                    assert buf + q in psum, "the partial sum comes from an earlier ticket"
                    tile, a, b = psum.pop(buf + q)
                    assert tile == (i, j) and a == at, "the pieces tile the range"
                    self.updates[i, j] += b - a
                    at = b
                assert at == k0
            elif k0 > 0:
                assert self.has(i, j, k0), "a later visit continues where the earlier one stopped"
            else:
                assert not self.stored[i, j]
            self.updates[i, j] += k1 - k0
            self.stored[i, j] = True
            self.applied[i, j] = k1
            if fin:
                assert k1 == j, "a tile is finished after ALL its updates"
                if i == j:
                    assert not self.chain
                    self.done[j, j] = True           # the diagonal kernel / inline
                else:
                    if self.chain:
                        self.chain_kernel()
                    assert self.done[j, j], ("the inverse this ticket waits for cannot be there yet", i, j)
                    self.done[i, j] = True
        if self.chain:
            self.chain_kernel()
        assert not psum, "every partial sum is added exactly once"


@pytest.mark.parametrize("nb", [1, 2, 3, 5, 29, 118])
@pytest.mark.parametrize("form", [(0, 0), (1, 0), (1, 1), (1, 2)])
@pytest.mark.parametrize("w", [1, 4])
def test_task_list_is_complete_and_deadlock_free(nb, form, w):
    chain, second = form
    rows = nb + 1                                    # + the right-hand-side row block
    r = Replay(nb, rows, chain, second)
    r.run(tasks(nb, rows, w, chain, second))
    low = np.tril(np.ones((rows, nb), bool))
    assert r.done[low].all(), "every tile of the lower triangle is finished"
    want = np.tile(np.arange(nb), (rows, 1))
    assert (r.updates[low] == want[low]).all(), "every tile gets each block-column update exactly once"


@pytest.mark.parametrize("nb", [20, 47, 118, 142])
@pytest.mark.parametrize("form", [(0, 0), (1, 0), (1, 2)])
@pytest.mark.parametrize("split", [(2, 0), (2, None), (3, 30), (8, 0)])
def test_split_update_ranges_keep_the_list_complete_and_deadlock_free(nb, form, split):
    """Round 5: the tasks of the late block columns hand part of their update range to partial-sum tasks (cholflow.hip, FLOW_PART).  Same
    guarantees: operands only from earlier tickets, every update exactly once, every partial sum added exactly once by its tile's own task."""
    chain, second = form
    m, frm = split
    if frm is None:                                  # the solver's own rule
        lib = engine.load_library()
        a, b = C.c_int(0), C.c_int(0)
        lib.jaicov_debug_flow_split(nb, C.byref(a), C.byref(b))
        m, frm = a.value, b.value
        assert (m, frm) == ((2, nb // 2) if nb >= 80 else (1, 1 << 30))
    rows = nb + 1
    t = tasks(nb, rows, 1, chain, second, m, frm)
    plain = tasks(nb, rows, 1, chain, second)
    n_part = int(((t[:, 3] & PART) != 0).sum())
    assert len(t) == len(plain) + n_part and (n_part > 0) == (m > 1 and nb - 1 >= max(8 * m, frm))
    r = Replay(nb, rows, chain, second)
    r.run(t)
    low = np.tril(np.ones((rows, nb), bool))
    assert r.done[low].all()
    want = np.tile(np.arange(nb), (rows, 1))
    assert (r.updates[low] == want[low]).all()
