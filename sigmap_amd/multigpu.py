"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" = RCCL over
xGMI on ROCm).  The path shards by reads with a full index replica per GPU:
the only collective is the start-up broadcast of the device-resident index arrays; there is
no data-path collective.  The same functions run on CPU tensors with the gloo backend, which
is how the world_size-2 tests exercise them without GPUs.

A process that uses torch on the GPU together with this package must `import torch` BEFORE the
first DeviceMapper is created (bench.py does): torch brings its own HIP runtime, and the process has
to end up with one runtime, the one loaded first."""
import numpy as np

from . import _abi

_ARRAYS = ["nodes", "leaf_tab", "leaf_pts", "leaf_vind", "bbox", "quads"]
_SCALARS = ["dim", "root_ref", "n_nodes", "n_leaves", "m_size", "n_quads"]


def shard_range(n_items, rank, world):
    """Contiguous block of reads for `rank`: read i -> rank i*world//n."""
    lo = (n_items * rank + world - 1) // world
    hi = (n_items * (rank + 1) + world - 1) // world
    return lo, hi


def merge_ranked(per_rank_rows):
    """Merge per-rank (ref_bucket, global_read_index, line) rows into the reference's -t 1
    output order: reference index, then read index (sigmap.cc:201-213,223-238)."""
    rows = [r for part in per_rank_rows for r in part]
    rows.sort(key=lambda r: (r[0], r[1]))
    return [r[2] for r in rows]


def broadcast_meta(meta, rank, dist, device):
    import torch
    t = torch.tensor(meta if rank == 0 else [0] * len(meta), dtype=torch.int64, device=device)
    dist.broadcast(t, src=0)
    return [int(x) for x in t.tolist()]


def broadcast_host_index(ix_arrays, rank, dist):
    """gloo/CPU variant: ix_arrays = dict name -> numpy array on rank 0 (None elsewhere).
    Returns the dict on every rank."""
    import torch
    names = ["values", "positions", "vind", "nodes", "bbox"]
    dtypes = {"values": np.float32, "positions": np.uint64, "vind": np.uint32, "nodes": np.uint8, "bbox": np.float32}
    meta = [0] * len(names)
    if rank == 0:
        meta = [int(np.ascontiguousarray(ix_arrays[n]).view(np.uint8).size) for n in names]
    meta = broadcast_meta(meta, rank, dist, "cpu")
    out = {}
    for n, nbytes in zip(names, meta):
        if rank == 0:
            buf = torch.from_numpy(np.ascontiguousarray(ix_arrays[n]).view(np.uint8).reshape(-1).copy())
        else:
            buf = torch.empty(nbytes, dtype=torch.uint8)
        dist.broadcast(buf, src=0)
        out[n] = buf.numpy().view(dtypes[n])
    return out


def replicate_index(dev, ix, rank, world, local_rank, dist, device=None, copy_fn=None, via_host=False, force=False):
    """Make the index resident on this rank's GPU.  world == 1: plain upload.  Otherwise rank 0
    uploads, then each of the six device arrays is sent with ONE RCCL broadcast and the peers
    attach the received buffers (smhip_index_attach).  Returns objects to keep alive.
    Rank 0 copies an exported array into the broadcast tensor with smhip_index_export_copy (the
    library's own HIP runtime and stream); `device` / `copy_fn(array_id, dst_ptr, src_ptr, nbytes)` let
    the world_size-2 gloo test run this very function with host buffers.  Every step that can fail on
    rank 0 (upload, export, the six tensors and the copies into them) happens BEFORE the first
    collective and its outcome travels in the meta broadcast: the peers raise instead of waiting in a
    collective for a rank that has left.
    via_host=True (bench.py --backend gloo, a transport for testing the N > 1 flow where RCCL cannot run,
    e.g. several ranks on one GPU): the arrays travel as CPU tensors and every peer copies what it received
    into device memory of its own before attaching.
    force=True with world == 1 (bench.py --force-bcast: a one-GPU box): the collectives run anyway on the
    one-rank group -- the same meta + six dist.broadcast calls over the same backend -- and the call returns
    (tensors, IndexDev of the broadcast copy) so that the caller can map from the copy and compare."""
    if world == 1 and not force:
        dev.upload_index(ix)
        return None
    import torch
    on_gpu = device is None
    gpu = torch.device("cuda", local_rank) if (on_gpu or via_host) else None
    if via_host:
        device = "cpu"
        on_gpu = False
    elif on_gpu:
        device = gpu
    if copy_fn is None:
        copy_fn = lambda array_id, dst_ptr, src_ptr, nbytes: dev.export_copy(array_id, dst_ptr, nbytes)
    meta = [0] * (1 + len(_SCALARS) + len(_ARRAYS))   # [0]: 1 = rank 0 has its index, -1 = it failed
    tensors = []
    err = None
    if rank == 0:
        try:
            dev.upload_index(ix)
            d = dev.export_index()
            for k, a in enumerate(_ARRAYS):
                nbytes = int(getattr(d, a + "_bytes"))
                t = torch.empty(nbytes, dtype=torch.uint8, device=device)
                if nbytes:   # (an index without quad records: nothing to copy, nothing to send)
                    copy_fn(k, t.data_ptr(), getattr(d, a), nbytes)
                tensors.append(t)
            meta = [1] + [int(getattr(d, f)) for f in _SCALARS] + [int(getattr(d, a + "_bytes")) for a in _ARRAYS]
        except Exception as e:   # tell the peers before raising
            err = e
            meta[0] = -1
            tensors = []
    meta = broadcast_meta(meta, rank, dist, device)
    if meta[0] != 1:
        raise RuntimeError(f"rank {rank}: index upload failed on rank 0" + (f": {err}" if err else ""))
    meta = meta[1:]
    sizes = meta[len(_SCALARS):]
    for k, nbytes in enumerate(sizes):
        if rank != 0:
            tensors.append(torch.empty(nbytes, dtype=torch.uint8, device=device))
        if nbytes:   # (an index without quad records has an empty sixth array)
            dist.broadcast(tensors[k], src=0)  # RCCL broadcast over xGMI
    if on_gpu:
        torch.cuda.synchronize()
    if rank != 0 or force:
        nd = _abi.IndexDev()
        for f, v in zip(_SCALARS, meta):
            setattr(nd, f, v)
        if via_host:
            tensors = [t.to(gpu) for t in tensors]   # H2D into this rank's own device memory
            torch.cuda.synchronize()
        for a, t, nbytes in zip(_ARRAYS, tensors, sizes):
            setattr(nd, a, t.data_ptr() if nbytes else None)
            setattr(nd, a + "_bytes", nbytes)
        if rank == 0:
            return tensors, nd   # force: rank 0 keeps its own arrays; the copy is the caller's to attach
        try:
            dev.attach_index(nd)
        except Exception as e:
            raise RuntimeError(f"rank {rank}: cannot attach the broadcast index: {e}") from e
        return tensors  # the context borrows this memory
    return None
