"""N>1 path on CPU: two gloo ranks split a frame into round-robin strips, gather to rank 0 and
lay the frame out.  The per-rank pixels come from the CPU oracle here (the product has no CPU
renderer), so this covers the partition, the padded gather and the assembly -- the exact host
logic bench.py runs around the frame kernels on N GPUs."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as ol

W, H = 72, 53  # 7 strips of 8 rows, the last one 5 rows: ranks get unequal row counts


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, out_path):
    sys.path.insert(0, ol.ROOT)
    from esctp1raytracer_amd import multigpu
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = ol.load_dump("two")
    rows = [h for k in multigpu.strips_of_rank(H, rank, world)
            for h in range(k * 8, min(k * 8 + 8, H))]
    assert len(rows) == multigpu.local_rows(H, rank, world)
    img, _ = ol.oracle_render_rows(d, (0, 1, 3), (0, 1, 0), W, H, rows)
    max_rows = multigpu.max_local_rows(H, world)
    local = torch.full((max_rows * W * 3,), -1.0)
    local[:len(rows) * W * 3] = torch.from_numpy(img.reshape(-1))
    gathered = torch.zeros(world, max_rows * W * 3) if rank == 0 else None
    g = multigpu.gather_to_root(local, rank, world, gathered)
    if rank == 0:
        frame = multigpu.assemble_frame_torch(g, world, W, H)
        np.save(out_path, frame.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_strip_partition_gather_assemble(tmp_path, world):
    out = str(tmp_path / "frame.npy")
    mp.spawn(_rank_main, args=(world, _free_port(), out), nprocs=world, join=True)
    full = ol.oracle_render(ol.load_dump("two"), (0, 1, 3), (0, 1, 0), W, H)
    got = np.load(out)
    assert np.array_equal(got.view(np.uint32), full.view(np.uint32))


def test_partition_covers_every_row_once():
    sys.path.insert(0, ol.ROOT)
    from esctp1raytracer_amd import multigpu
    for Hh in (2160, 1080, 77, 8, 5):
        for world in (1, 2, 4, 8):
            seen = []
            for r in range(world):
                ks = multigpu.strips_of_rank(Hh, r, world)
                seen += ks
                assert multigpu.local_rows(Hh, r, world) <= multigpu.max_local_rows(Hh, world)
            assert sorted(seen) == list(range(multigpu.n_strips(Hh)))
            assert sum(multigpu.local_rows(Hh, r, world) for r in range(world)) == Hh
