"""N>1 path on CPU: two processes, gloo backend (world_size 2).  Exercises the same
sigmap_amd.multigpu code bench.py uses for GPUs: index replication by broadcast from rank 0,
contiguous read sharding, and the (reference index, read index) merge of per-rank rows."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from sigmap_amd import multigpu, pipeline, synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        genome = synth.make_genome(3, [20000, 12000])
        lm = synth.make_pore_model()
        arrays = None
        if rank == 0:
            ix = pipeline.HostIndex.build(genome, lm, threads=2)
            arrays = dict(values=ix.values, positions=ix.positions, vind=ix.vind, nodes=ix.nodes.view(np.uint8), bbox=ix.bbox)
        got = multigpu.broadcast_host_index(arrays, rank, dist)
        # every rank rebuilds independently and must hold byte-identical arrays after the broadcast
        mine = pipeline.HostIndex.build(genome, lm, threads=1)
        same = (np.array_equal(got["values"].view(np.uint32), mine.values.view(np.uint32))
                and np.array_equal(got["positions"], mine.positions) and np.array_equal(got["vind"], mine.vind)
                and np.array_equal(got["nodes"], mine.nodes.view(np.uint8).reshape(-1))
                and np.array_equal(got["bbox"].view(np.uint32), mine.bbox.view(np.uint32)))
        n_reads = 1001
        lo, hi = multigpu.shard_range(n_reads, rank, world)
        # fake per-rank rows: (ref bucket, global read index, line)
        rows = [((i * 7) % 3, i, "r%d" % i) for i in range(lo, hi)]
        gathered = [None] * world
        dist.all_gather_object(gathered, rows)
        merged = multigpu.merge_ranked(gathered)
        q.put((rank, same, lo, hi, merged if rank == 0 else None))
    finally:
        dist.destroy_process_group()


def test_world2_gloo_index_broadcast_sharding_and_merge():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    out.sort()
    assert all(o[1] for o in out), "broadcast index differs from a local build"
    (r0, _, lo0, hi0, merged), (r1, _, lo1, hi1, _) = out
    assert lo0 == 0 and hi0 == lo1 and hi1 == 1001          # contiguous, complete, disjoint
    exp = sorted(range(1001), key=lambda i: ((i * 7) % 3, i))
    assert merged == ["r%d" % i for i in exp]                # reference -t 1 order


def test_shard_range_partitions_everything():
    from sigmap_amd.multigpu import shard_range
    for n in (0, 1, 7, 8, 1000, 100003):
        for world in (1, 2, 4, 8):
            parts = [shard_range(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in parts) - min(b - a for a, b in parts) <= 1


class _FakeContext:
    """Stands in for pipeline.DeviceMapper on a machine without a GPU: "device" arrays are host
    buffers, export hands out their addresses, attach records what it was given."""

    def __init__(self, quads=True):
        self.exported = None
        self.attached = None
        self._keep = []
        self._quads = quads   # False: an index without quad records (a one-leaf tree, or one beyond their size limit)

    def upload_index(self, ix):
        import ctypes as C
        from sigmap_amd import _abi
        rng = np.random.default_rng(5)
        d = _abi.IndexDev()
        d.dim, d.root_ref, d.n_nodes, d.n_leaves, d.m_size, d.n_quads = 6, 0, 1000, 1001, 12345, 400 if self._quads else 0
        for name, nbytes in (("nodes", 64000), ("leaf_tab", 4004), ("leaf_pts", 70000), ("leaf_vind", 9000), ("bbox", 96),
                             ("quads", 51200 if self._quads else 0)):
            if nbytes == 0:
                continue   # (pointer and size stay null / 0)
            buf = rng.integers(0, 256, size=nbytes, dtype=np.uint8)
            self._keep.append(buf)
            setattr(d, name, buf.ctypes.data)
            setattr(d, name + "_bytes", nbytes)
        self.exported = d

    def export_index(self):
        return self.exported

    def attach_index(self, nd):
        import ctypes as C
        from sigmap_amd import multigpu
        self.attached = {f: int(getattr(nd, f)) for f in multigpu._SCALARS}
        for a in multigpu._ARRAYS:
            n = int(getattr(nd, a + "_bytes"))
            self.attached[a] = bytes((C.c_uint8 * n).from_address(getattr(nd, a))) if n else b""


def _replicate_worker(rank, world, port, q, quads=True):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import ctypes as C
    import torch.distributed as dist
    from sigmap_amd import multigpu
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = _FakeContext(quads)
        keep = multigpu.replicate_index(dev, None, rank, world, rank, dist, device="cpu",
                                        copy_fn=lambda k, dst, src, n: C.memmove(dst, src, n))
        if rank == 0:
            d = dev.exported
            want = {f: int(getattr(d, f)) for f in multigpu._SCALARS}
            for a in multigpu._ARRAYS:
                n = int(getattr(d, a + "_bytes"))
                want[a] = bytes((C.c_uint8 * n).from_address(getattr(d, a))) if n else b""
            q.put((0, want))
        else:
            q.put((rank, dev.attached if keep is not None else None))
    finally:
        dist.destroy_process_group()


def test_world2_gloo_replicate_index_hands_the_peer_identical_arrays():
    """multigpu.replicate_index itself (the function bench.py calls for N > 1), with host buffers in
    place of device memory: the peer attaches the scalars and byte-identical copies of all six arrays."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_replicate_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[1] is not None and got[1] == got[0]


def test_world2_gloo_replicate_index_without_quad_records():
    """The sixth array is empty for an index without quad records: nothing is copied or sent for it, the peer
    attaches a null pointer with n_quads == 0 and everything else as usual."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_replicate_worker, args=(r, 2, port, q, False)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[1] is not None and got[1] == got[0] and got[1]["n_quads"] == 0 and got[1]["quads"] == b""


def _forced_worker(port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    import ctypes as C
    import torch.distributed as dist
    from sigmap_amd import multigpu
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        dev = _FakeContext()
        tensors, nd = multigpu.replicate_index(dev, None, 0, 1, 0, dist, device="cpu", force=True,
                                               copy_fn=lambda k, dst, src, n: C.memmove(dst, src, n))
        d = dev.exported
        same = all(int(getattr(nd, f)) == int(getattr(d, f)) for f in multigpu._SCALARS)
        for a in multigpu._ARRAYS:
            n = int(getattr(d, a + "_bytes"))
            same = same and int(getattr(nd, a + "_bytes")) == n and getattr(nd, a) != getattr(d, a)
            same = same and bytes((C.c_uint8 * n).from_address(getattr(nd, a))) == bytes((C.c_uint8 * n).from_address(getattr(d, a)))
        q.put(bool(same and len(tensors) == len(multigpu._ARRAYS)))
    finally:
        dist.destroy_process_group()


def test_forced_one_rank_broadcast_returns_an_identical_copy():
    """bench.py --force-bcast: with ONE rank replicate_index(force=True) still runs the meta + six
    broadcasts over the group and hands back a copy of the arrays (distinct memory, same bytes)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_forced_worker, args=(_free_port(), q))
    p.start()
    assert q.get(timeout=120) is True
    p.join(timeout=60)
    assert p.exitcode == 0


def _failing_copy_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import datetime
    import torch.distributed as dist
    from sigmap_amd import multigpu
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))

    def bad_copy(k, dst, src, n):
        if k == 2:
            raise RuntimeError("export copy of array 2 failed")

    try:
        multigpu.replicate_index(_FakeContext(), None, rank, world, rank, dist, device="cpu", copy_fn=bad_copy)
        q.put((rank, "no error"))
    except RuntimeError as e:
        q.put((rank, str(e)))
    finally:
        dist.destroy_process_group()


def test_world2_rank0_failure_in_an_export_copy_reaches_the_peer():
    """A failure on rank 0 inside the per-array copies (not only upload / export) is announced in the meta
    broadcast: the peer raises at once instead of waiting in a collective until the group times out."""
    import time
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    t0 = time.time()
    procs = [ctx.Process(target=_failing_copy_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=50) for _ in range(2))
    for p in procs:
        p.join(timeout=30)
    assert "index upload failed on rank 0" in got[0] and "array 2" in got[0]
    assert "index upload failed on rank 0" in got[1]
    assert time.time() - t0 < 45


def test_shard_range_is_the_contiguous_block_partition():
    """`sigmap -m --gpus N` (host/sigmap.cc) and bench.py shard reads the same way: part r of N gets
    the contiguous block of reads i with i * N // n == r; the parts tile [0, n)."""
    import ctypes as C
    from sigmap_amd import _abi, multigpu
    L = _abi.host_lib()
    for n in (0, 1, 7, 1000, 1001, 100000, 2 ** 33 + 5):
        for world in (1, 2, 3, 8):
            prev = 0
            for r in range(world):
                lo, hi = C.c_uint64(), C.c_uint64()
                L.smhost_shard_range(n, r, world, C.byref(lo), C.byref(hi))
                assert (lo.value, hi.value) == multigpu.shard_range(n, r, world)
                assert lo.value == prev and hi.value >= lo.value
                prev = hi.value
                if n and n < 2000:
                    assert [i for i in range(n) if i * world // n == r] == list(range(lo.value, hi.value))
            assert prev == n
