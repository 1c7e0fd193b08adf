"""Row-tile sharding of one frame over ranks (one process per GPU) and canvas assembly on rank 0.

The reference parallelises over image rows with rayon (camera.rs:72-85) and copies the rows into the Canvas
serially (camera.rs:87-91).  Here rank r renders the interleaved row tiles t with t % world_size == r into a
compact device buffer; rank 0 gathers the buffers (RCCL over xGMI via torch.distributed, or gloo on CPU in
tests) and scatters the rows back to their canvas positions.
"""
import numpy as np


def tile_row_map(vsize: int, tile_rows: int, tile_index: int, tile_count: int) -> np.ndarray:
    """Canvas rows owned by one rank, in the order they are stored in its compact buffer.
    Mirrors rtc_local_row_map() (include/rtc.h)."""
    if tile_count <= 0:
        tile_count = 1
    if tile_rows <= 0 or tile_rows > vsize:
        tile_rows = max(vsize, 1)
    ntiles = (vsize + tile_rows - 1) // tile_rows
    rows = []
    for t in range(tile_index, ntiles, tile_count):
        rows.extend(range(t * tile_rows, min(vsize, (t + 1) * tile_rows)))
    return np.asarray(rows, dtype=np.int64)


def gather_index(vsize: int, tile_rows: int, world_size: int, padded_rows: int) -> np.ndarray:
    """Index vector for index_copy_ on a (vsize + 1)-row canvas: entry [r * padded_rows + l] is the canvas row
    of rank r's local row l; padding rows point at the extra dump row `vsize`."""
    idx = np.full((world_size, padded_rows), vsize, dtype=np.int64)
    for r in range(world_size):
        m = tile_row_map(vsize, tile_rows, r, world_size)
        idx[r, : len(m)] = m
    return idx.reshape(-1)


def max_local_rows(vsize: int, tile_rows: int, world_size: int) -> int:
    return max(len(tile_row_map(vsize, tile_rows, r, world_size)) for r in range(world_size))


def gather_canvas(local, vsize: int, tile_rows: int, rank: int, world_size: int, dist=None):
    """Assemble the full (vsize, hsize, 3) canvas on rank 0 from every rank's compact, equally padded buffer.
    `local` is a torch tensor (padded_rows, hsize, 3); returns the canvas on rank 0 and None elsewhere."""
    import torch

    if world_size == 1:
        return local[:vsize]
    padded = local.shape[0]
    gather_list = [torch.empty_like(local) for _ in range(world_size)] if rank == 0 else None
    dist.gather(local, gather_list, dst=0)
    if rank != 0:
        return None
    canvas = torch.zeros((vsize + 1,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    index = torch.from_numpy(gather_index(vsize, tile_rows, world_size, padded)).to(local.device)
    canvas.index_copy_(0, index, torch.cat(gather_list, dim=0))
    return canvas[:vsize]
