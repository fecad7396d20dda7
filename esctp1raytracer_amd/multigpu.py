"""Row-strip partition of one frame over the GPUs of a node, one process per GPU.

The reference renders rows independently (main.cpp:628-636: one scan_row call, or one
std::thread, per row), so the frame shards with no data-path communication; the only exchange
is the final framebuffer gather to rank 0 (RCCL over xGMI through torch.distributed, or gloo on
CPU in tests).

Partition: the image is cut into strips of STRIP_ROWS rows counted from h = 0 and strip k goes
to rank k % world (round-robin).  Contiguous bands would leave the ranks holding sky rows
(primary rays only) idle while the ranks holding floor rows still trace shadow rays -- on the
BASELINE c4 scene a floor row costs ~3x a sky row.
"""
import torch
import torch.distributed as dist

STRIP_ROWS = 8  # == the workgroup tile height of the frame kernels (rt_device.h kTileH)


def n_strips(H, strip_rows=STRIP_ROWS):
    return (H + strip_rows - 1) // strip_rows


def strips_of_rank(H, rank, world, strip_rows=STRIP_ROWS):
    return list(range(rank, n_strips(H, strip_rows), world))


def local_rows(H, rank, world, strip_rows=STRIP_ROWS):
    """rows rank `rank` renders (mirrors esc_strip_local_rows of the C ABI)"""
    return sum(min(strip_rows, H - k * strip_rows) for k in strips_of_rank(H, rank, world, strip_rows))


def max_local_rows(H, world, strip_rows=STRIP_ROWS):
    """rank 0 always holds the most rows; gather buffers are padded to this"""
    return local_rows(H, 0, world, strip_rows)


def gather_to_root(local, rank, world, gathered=None, dst=0, group=None, gather_list=None):
    """local: 1-D tensor of max_local_rows*W*C elements (tail rows of short ranks unused).
    gathered (root only): [world, local.numel()] tensor.  Direct peer->root transfers: on xGMI
    every peer has its own link to the root, so 7 peers send concurrently (a ring all-gather
    would push 7/8 of the frame through every link instead)."""
    if world == 1:
        return local.view(1, -1)
    if rank == dst:
        assert gathered is not None and gathered.shape == (world, local.numel())
        # gather_list: the caller's cached list(gathered.unbind(0)) (a per-frame caller saves the
        # views' construction; at N = 8 a rank's share of a frame is ~0.1 ms)
        dist.gather(local, gather_list=gather_list if gather_list is not None
                    else list(gathered.unbind(0)), dst=dst, group=group)
        return gathered
    dist.gather(local, gather_list=None, dst=dst, group=group)
    return None


def assemble_frame_torch(gathered, world, W, H, channels=3, strip_rows=STRIP_ROWS):
    """Pure-torch layout of the gathered strips as one (H, W, C) frame, h = 0 bottom row.
    Host-logic mirror of k_assemble_strips (used on CPU in the gloo tests and to check the
    HIP kernel); the GPU bench path uses Renderer.assemble_strips."""
    row = W * channels
    g = gathered.view(world, -1)
    frame = torch.empty(H * row, dtype=g.dtype, device=g.device)
    for k in range(n_strips(H, strip_rows)):
        r, j = k % world, k // world
        h0 = k * strip_rows
        rows = min(strip_rows, H - h0)
        frame[h0 * row:(h0 + rows) * row] = g[r, j * strip_rows * row:(j * strip_rows + rows) * row]
    return frame.view(H, W, channels)
