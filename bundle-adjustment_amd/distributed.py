"""Multi-GPU plumbing: shard observation groups by image, one all-reduce of the packed normal equations per pass.

SURVEY.md 8(e): W is block-diagonal by image, so N = sum_rank N_rank and n = sum_rank n_rank.  Every rank holds the
whole (small) structure and all parameter values, accumulates the image range it owns, the packed buffer
[N (U(U+1)/2) | n (U)] is summed in place with ``torch.distributed.all_reduce`` (backend "nccl" == RCCL over xGMI on
MI355X; "gloo" in the CPU rehearsal), and every rank then applies datum/damping/preconditioner and solves the
identical system.  Scale bars and directly observed groups are contributed by rank 0 only.
"""
from __future__ import annotations

import numpy as np


def image_costs(fp) -> np.ndarray:
    """Relative assembly cost per image: m_g^2 * k_g for a dense image block, n_obs * k^2 otherwise."""
    I = fp.n_images
    counts = np.bincount(fp.ip_image, minlength=I).astype(np.float64)
    k = 12.0 + np.diff(fp.cam_dist_begin)[fp.image_camera]
    cost = counts * k * k
    for b in range(fp.n_image_blocks):
        lo, hi = int(fp.blk_ip_begin[b]), int(fp.blk_ip_begin[b + 1])
        if hi > lo:
            img = int(fp.ip_image[lo])
            m = 2.0 * (hi - lo)
            cost[img] = m * m * (3 * (hi - lo) + k[img])
    return cost


def partition_images(fp, world: int):
    """Contiguous image ranges [lo, hi) per rank, balanced by ``image_costs`` (prefix-sum split)."""
    cost = image_costs(fp)
    csum = np.concatenate([[0.0], np.cumsum(cost)])
    total = csum[-1]
    bounds = [0]
    for r in range(1, world):
        target = total * r / world
        b = int(np.searchsorted(csum, target, side="left"))
        b = min(max(b, bounds[-1]), fp.n_images)
        bounds.append(b)
    bounds.append(fp.n_images)
    return [(bounds[r], bounds[r + 1]) for r in range(world)]


class DeviceArray:
    """Raw device pointer as a torch-visible array (``__cuda_array_interface__``, no copy)."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def allreduce_engine_buffer(eng, dist, device):
    """Sums the engine's packed partial normal equations over all ranks, in place on the device.

    On a GPU the collective is enqueued in the order of the engine's own stream (torch sees it as an ExternalStream: RCCL
    waits for the pack kernel through stream events, the engine's finalize waits for RCCL the same way), so the host never
    blocks and the factorisation's launches are queued while assembly and collective still run.  gloo (CPU rehearsal)
    needs host tensors and takes the synchronous route."""
    import os
    import torch
    if (device is not None and torch.device(device).type == "cuda" and dist.get_backend() == "nccl"
            and not os.environ.get("JAICOV_SYNC_COLLECTIVE")):      # the switch forces the host-synchronised route
        ptr, cnt, stream = eng.reduce_buffer_async()
        ext = torch.cuda.ExternalStream(stream, device=device)
        with torch.cuda.stream(ext):
            buf = torch.as_tensor(DeviceArray(ptr, cnt), device=device)
            dist.all_reduce(buf)
        return ext
    ptr, cnt = eng.reduce_buffer()
    buf = torch.as_tensor(DeviceArray(ptr, cnt), device=device)
    dist.all_reduce(buf)
    torch.cuda.synchronize(device)
    return None


def _check_same_buffer_on_all_ranks(eng, dist, device):
    """Every rank must hold the same system (same reduced order, hence the same reduce-buffer length) before the
    collective: ranks that disagree would hang in RCCL or sum unrelated entries.  Checked once per engine and order."""
    import torch
    order = eng.reduced_order()
    if getattr(eng, "_order_agreed", None) == order or dist.get_world_size() == 1:
        return
    dev = device if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([order, -order], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    hi, lo = int(t[0]), -int(t[1])
    if hi != lo:
        raise RuntimeError(f"ranks disagree on the order of the assembled system ({lo} .. {hi}): the engines were created "
                           "from different problem descriptions or inversion modes")
    eng._order_agreed = order


def sharded_step(eng, dist, device, sigma2, lam=0.0, invert=False):
    """One pass of the loop body on a sharded engine: accumulate -> all-reduce -> finalize -> solve.

    When the engine pre-eliminates the exterior-orientation blocks, every rank's dx carries only its own images' EO
    entries (the rest is zero): they are summed over the ranks (6 doubles per image)."""
    import torch
    eng.prepare_inverse(invert)
    eng.accumulate(sigma2, lam)
    _check_same_buffer_on_all_ranks(eng, dist, device)
    ext = allreduce_engine_buffer(eng, dist, device)
    if int(invert) == 3 and getattr(eng, "expansion_exchange", False) and eng.reduced_order() < eng.U:
        # MatrixInversion.FULL expanded from the reduced inverse (BA:268-271): every rank holds the F bands and L_E^-1 of its own
        # images only; one more all-reduce (zeros elsewhere) gives every rank all of them (371 MB at config 4).  Like the first
        # collective it is enqueued in the order of the engine's own stream when the backend is RCCL (round 5: it used to run on torch's
        # stream behind a device-wide synchronisation): finalize and the solve follow it on that stream, the host does not wait.
        ptr, cnt = eng.expansion_buffer()
        t = torch.as_tensor(DeviceArray(ptr, cnt), device=device)
        if ext is not None:
            with torch.cuda.stream(ext):
                dist.all_reduce(t)
        else:
            dist.all_reduce(t)
            if torch.device(device).type == "cuda":
                torch.cuda.synchronize(device)
    eng.finalize(sigma2, lam)
    dx = eng.solve(invert)
    e0 = eng.reduced_order()
    import os
    if e0 < eng.U and (dist.get_world_size() > 1 or os.environ.get("JAICOV_FORCE_EO_EXCHANGE")):   # the switch: rehearsal on one rank
        if device is not None and torch.device(device).type == "cuda" and dist.get_backend() == "nccl":
            # the EO steps stay where the back-substitution left them: all-reduce the engine's device array in place, one copy back
            ptr, cnt = eng.eo_step_buffer()
            t = torch.as_tensor(DeviceArray(ptr, cnt), device=device)
            dist.all_reduce(t)
            dx[e0:] = t.cpu().numpy()[:eng.U - e0]
        else:                                           # gloo rehearsal: host tensors
            t = torch.from_numpy(dx[e0:].copy())
            dist.all_reduce(t)
            dx[e0:] = t.numpy()
    return dx


def sharded_omega(eng, dist, device, sigma2, dx):
    """Omega = sum over ALL observation groups of (w - A dx)' P (w - A dx) (BA:472-491): every rank evaluates its own images
    (rank 0 also the scale bars and directly observed groups), the scalars are summed."""
    import torch
    t = torch.tensor([eng.omega(sigma2, dx)], dtype=torch.float64, device=device)
    dist.all_reduce(t)
    return float(t.item())
