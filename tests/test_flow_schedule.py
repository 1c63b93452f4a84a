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


def tasks(nb, rows, w, chain, second=0):
    lib = engine.load_library()
    lib.jaicov_debug_flow_tasks.argtypes = [C.c_int] * 5 + [C.c_void_p, C.c_int]
    n = lib.jaicov_debug_flow_tasks(nb, rows, w, chain, second, None, 0)
    assert n > 0
    out = np.zeros((n, 4), np.int32)
    assert lib.jaicov_debug_flow_tasks(nb, rows, w, chain, second, out.ctypes.data, n) == n
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
        for (i, j, k0, wd) in t:
            k1, fin = int(wd) & (FIN - 1), bool(int(wd) & FIN)
            assert 0 <= j <= i < self.rows and j < self.nb and 0 <= k0 <= k1 <= j
            if self.chain:
                self.chain_kernel()
            if k0 > 0:
                assert self.has(i, j, k0), "a later visit continues where the earlier one stopped"
            else:
                assert not self.stored[i, j]
            for k in range(k0, k1):                  # operands: final by now
                assert self.done[i, k] and self.done[j, k], ("ticket would wait for a later ticket", i, j, k)
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
