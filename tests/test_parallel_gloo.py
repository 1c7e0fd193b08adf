"""N > 1 path on the CPU: two ranks (gloo) each hold their interleaved row tiles in a compact buffer; rank 0 gathers
and de-interleaves them into the canvas exactly as bench.py does over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from scene_feed import rtc


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world_size, port, vsize, hsize, tile_rows, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        par = rtc.parallel
        rows = par.tile_row_map(vsize, tile_rows, rank, world_size)
        padded = par.max_local_rows(vsize, tile_rows, world_size)
        local = torch.zeros((padded, hsize, 3), dtype=torch.float64)
        # what a rank's render would hold: pixel (y, x) = (y, x, rank) in its compact row order
        ys = torch.from_numpy(rows).to(torch.float64)
        local[: len(rows), :, 0] = ys[:, None]
        local[: len(rows), :, 1] = torch.arange(hsize, dtype=torch.float64)[None, :]
        local[: len(rows), :, 2] = float(rank)
        canvas = par.gather_canvas(local, vsize, tile_rows, rank, world_size, dist)
        if rank == 0:
            np.save(out_path, canvas.numpy())
        else:
            assert canvas is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world_size,vsize,tile_rows", [(2, 37, 4), (3, 50, 8)])
def test_gather_canvas_over_gloo(tmp_path, world_size, vsize, tile_rows):
    hsize = 5
    out = str(tmp_path / "canvas.npy")
    mp.spawn(_worker, args=(world_size, _free_port(), vsize, hsize, tile_rows, out), nprocs=world_size, join=True)
    canvas = np.load(out)
    assert canvas.shape == (vsize, hsize, 3)
    assert np.array_equal(canvas[:, :, 0], np.repeat(np.arange(vsize, dtype=np.float64)[:, None], hsize, axis=1))
    assert np.array_equal(canvas[:, :, 1], np.repeat(np.arange(hsize, dtype=np.float64)[None, :], vsize, axis=0))
    owner = np.array([(y // tile_rows) % world_size for y in range(vsize)], dtype=np.float64)
    assert np.array_equal(canvas[:, 0, 2], owner)
