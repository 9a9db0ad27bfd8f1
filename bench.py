#!/usr/bin/env python3
"""bench.py -- signal chunks/sec (4k-sample reads) mapped to a yeast-sized (12 Mb) reference.

One "step" = one pass of the hot path (event detection -> k-d radius seed lookup -> anchor sort ->
chaining -> banded sDTW refinement) over one batch of synthetic 4000-sample chunks.  At N=1 the
workload is BASELINE.json configs[1]: 10 000 reads x 1 chunk vs 16 x 750 kb.  For N>1 (launched by
torch.distributed.run, one rank per GPU) rank 0 builds the index, the device-resident index arrays
are replicated by one RCCL broadcast each, every rank maps its own 10 000-chunk shard (weak scaling,
no data-path collective), and the reported value is (chunks of all ranks) / (max over ranks of the
timed region).

Timed regions of EXACTLY K steps each, every one fenced by barrier + synchronize on both sides, the time = max
over ranks; each kind is timed --repeats times (after one discarded round, the kinds taking turns) and reported
as the MEDIAN region with min / max / all beside it:
  value          : inputs resident in HBM when the region starts (the reads were converted from their
                   ADC counts into the contexts' slots beforehand) -- the contract's definition;
  same_reads     : the same, one read set for every step (comparison only);
  pcie_inclusive : every step first copies its reads' int16 ADC counts from pinned host memory
                   (DMA on a copy stream) and converts them on the device ("reads
                   resident in host RAM").

Prints ONE JSON line on rank 0, including
  roofline     : fabric (L2-miss) traffic of the dominant kernel (seed_lookup) from the committed counter profile of
                 this workload shape / its launch duration measured live, next to the random-line ceiling, the
                 useful bytes by SURVEY 8(d)'s weights, the kernel's own bytes and the reference algorithm's
  cpu_baseline : the CPU restatement (oracle/, "port") timed on bounded samples, rank 0, N=1
"""
import argparse
import datetime
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
HBM_STREAM_GBS = 6290.0    # same guide: measured streaming ceiling (float4 copy)
RANDOM_LINES_PER_S = 52e9  # tools/micro/random_lines.hip: random 128-byte lines per second beyond L2
CHUNK = 4000
REF_CHUNKS_PER_S_PER_CORE = 19.6  # unmodified reference, -t 1, 12 Mb (survey probe)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=10000, help="chunks per GPU per step")
    ap.add_argument("--contigs", type=int, default=16)
    ap.add_argument("--contig-len", type=int, default=750000)
    ap.add_argument("--walk-blocks", type=int, default=0, help="tree-walker workgroups per CU (0 = default)")
    ap.add_argument("--fast-cap", type=int, default=0, help="fast hit slots per query (0 = default)")
    ap.add_argument("--search-mode", default="auto", choices=["auto", "exact", "set"],
                    help="smhip_params.search_mode (auto: decided from the index census)")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="target duration of each CPU-baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-streaming", action="store_true",
                    help="skip the `streaming` block (tools/stream_bench.py: multi-chunk reads through `sigmap -m`, N = 1 only)")
    ap.add_argument("--no-pcie", action="store_true", help="skip the PCIe-inclusive timed region")
    ap.add_argument("--same-reads", action="store_true", help="one read set for every step and context (rounds 1-3)")
    ap.add_argument("--no-same-reads", action="store_true", help="skip the comparison region that maps one set every step")
    ap.add_argument("--sdtw-band", type=int, default=64, help="band of the sDTW refinement (0 = skip it)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="transport of the N > 1 run: nccl = RCCL over xGMI (one rank per GPU); gloo = through host "
                         "memory, ranks may share a GPU (a way to exercise the multi-rank flow on a one-GPU box)")
    ap.add_argument("--force-bcast", action="store_true",
                    help="with one rank: run the index replication collectives anyway (one-rank process group over "
                         "--backend, six dist.broadcast calls) and map the contexts beyond the first from the broadcast "
                         "copy -- exercises the N > 1 transport on a one-GPU box")
    ap.add_argument("--repeats", type=int, default=5,
                    help="every timed region (EXACTLY --steps steps, fenced) is timed this many times after one discarded "
                         "region; value / ms_per_step are the MEDIAN region, min / max / all are printed beside it")
    ap.add_argument("--stagger", type=float, default=1.0,
                    help="the contexts of a region start this fraction of (serial step / depth) apart (0: all at once)")
    ap.add_argument("--ab-knob", default=None, help="development: NAME=V1,V2,... -- the serial pass repeated under each value of a knob")
    ap.add_argument("--depth", type=int, default=6,
                    help="batches in flight per GPU: steps are spread over this many contexts / HIP streams "
                         "that share the resident index (1 = strictly serial steps; 4 -> 6: -1 %, ~25 GB of buffers each)")
    return ap.parse_args()


def cpu_baseline(ix, raw_reads, n_refs, ref_lens, seconds):
    """The oracle (CPU restatement of the reference path) on bounded samples of the same reads: one
    thread (BASELINE configs[0] is `-t 1`), then one thread per host core (the reference parallelises
    over reads the same way, sigmap.cc:618-630).  This is the only place bench.py touches oracle/: as
    the measured CPU baseline."""
    from oracle import pyoracle as po
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    from sigmap_amd import synth
    pa_list = [po.add_signal(r, synth.DIGITISATION, synth.RANGE, synth.OFFSET) for r in raw_reads]   # A1, untimed like
    tree = po.OracleTree(ix.values.copy())                                    # the reference's load phase
    positions = np.ascontiguousarray(ix.positions)

    def run(n_threads, n_sample):
        sample = [pa_list[i % len(pa_list)] for i in range(n_sample)]
        t0 = time.perf_counter()
        po.map_reads(tree, positions, sample, n_refs, ref_lens, n_threads)   # OpenMP loop in the C oracle
        dt = time.perf_counter() - t0
        return n_sample / dt, dt

    probe, _ = run(1, 8)                                   # sizes the samples (about `seconds` per leg)
    legs = []
    for nt in sorted({1, min(32, cores), cores}):
        pilot = probe if nt == 1 else run(nt, nt * 2)[0]    # the parallel rate is not nt x the serial one
        n = max(nt * 4, 16, int(pilot * seconds))
        rate, sec = run(nt, n)
        legs.append({"threads": nt, "chunks": n, "seconds": round(sec, 1), "chunks_per_s": round(rate, 2),
                     "per_thread": round(rate / nt, 2)})
    tree.close()
    best = max(legs, key=lambda l: l["chunks_per_s"])
    t1 = legs[0]["chunks_per_s"]
    return {"value": best["chunks_per_s"], "unit": "chunks/s", "cores": best["threads"], "kind": "port",
            "sample": f"{best['chunks']} chunks (the same synthetic 4000-sample reads, same 12 Mb index) through "
                      f"oracle/sigmap_oracle.c so_map_reads (OpenMP loop over reads) on {best['threads']} threads in "
                      f"{best['seconds']} s -- the fastest of the legs below ({cores} hardware threads available)",
            "legs": legs, "host_threads_available": cores,
            "t1_value": t1, "reference_t1_chunks_per_s": REF_CHUNKS_PER_S_PER_CORE,
            "t1_vs_reference_t1": round(t1 / REF_CHUNKS_PER_S_PER_CORE, 3),
            "note": "reference_t1 = the unmodified reference with -t 1 on the survey container's Xeon 2.1 GHz "
                    "(BASELINE.md section 2); the port does the same work per read (same node visits, same hits).  "
                    "The path is a DRAM-latency-bound pointer chase over a shared 0.7 GB index: on this two-socket "
                    "host it stops scaling near 32 threads (tools/oracle_scaling.py)"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # before anything initialises HIP
    # The HIP runtime multiplexes a process's streams over 4 hardware queues by default: with `depth` contexts of three
    # streams each (compute, copy, auxiliary) the streams of different contexts share queues and wait for each other's
    # kernels (profiles/r5_hw_queues.txt: 10.04 -> 9.83 ms per step at depth 4, 9.76 at depth 6 with 16 queues).
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    import torch   # first: the process ends up with ONE HIP runtime, the one torch brings (multigpu.py)
    dist = None
    if world > 1 or args.force_bcast:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "RANK" not in os.environ:   # --force-bcast outside torch.distributed.run: a one-rank group of our own
            os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
            os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "gloo":
            local_rank = local_rank % max(torch.cuda.device_count(), 1)   # ranks may share a GPU
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=600))
        else:
            if local_rank >= torch.cuda.device_count():
                raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPUs visible")
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank),
                                    timeout=datetime.timedelta(seconds=600))

    from sigmap_amd import pipeline, synth
    from sigmap_amd.multigpu import replicate_index

    genome_lens = [args.contig_len] * args.contigs
    n_refs = len(genome_lens)
    ref_lens = np.array(genome_lens, dtype=np.uint32)
    lm = synth.make_pore_model()
    genome = synth.make_genome(7, genome_lens)  # every rank: cheap, and reads are cut from it
    pool = pipeline.MapperPool(local_rank, max(1, args.depth))
    dev = pool.first
    for dv in pool.devs:
        if args.walk_blocks:
            dv.params.walk_blocks_per_cu = args.walk_blocks
        if args.fast_cap:
            dv.params.hits_per_query_fast = args.fast_cap
        dv.params.search_mode = {"auto": 0, "exact": 1, "set": 2}[args.search_mode]
    ix = None
    t_build = 0.0
    if rank == 0:
        t0 = time.perf_counter()
        # (torch.distributed.run exports OMP_NUM_THREADS=1: ask for the cores explicitly)
        ix = pipeline.HostIndex.build(genome, lm, threads=max(1, (os.cpu_count() or 1) // max(world, 1)))
        t_build = time.perf_counter() - t0
    forced = args.force_bcast and world == 1
    keepalive = replicate_index(dev, ix, rank, world, local_rank, dist,   # RCCL broadcast when world > 1
                                via_host=(args.backend == "gloo"), force=forced)
    if forced:
        # the first context maps from the arrays it uploaded, the others from the copy that went through the
        # collectives: `contexts_agree` below then compares the two byte for byte
        for dv in pool.devs[1:]:
            dv.attach_index(keepalive[1])
    else:
        pool.share_index()  # the other contexts of this GPU borrow the same arrays

    R = args.reads
    # Read sets: one more than there are contexts, a DIFFERENT set every step (step i maps set i mod n_sets), so that
    # neither a context nor two contexts in flight ever walk the same queries twice in a row -- no cross-step or
    # cross-stream cache reuse a real run would not have.  All sets are resident in every context's slots (set s in
    # slots [s R, (s + 1) R)) when the `value` region starts; the pcie_inclusive region copies a step's set from
    # pinned host memory first.
    n_sets = max(2, args.depth + 1) if not args.same_reads else 1
    raw_sets = [synth.simulate_reads(genome, lm, R, seed=1000 + 64 * rank + s)[0] for s in range(n_sets)]
    raw = raw_sets[0]
    # the rank's reads as they sit in host RAM: int16 ADC counts in pinned memory (one DMA per batch)
    pinned = pipeline.PinnedArray((n_sets * R * CHUNK,), np.int16)
    for s_i in range(n_sets):
        pinned.array[s_i * R * CHUNK:(s_i + 1) * R * CHUNK] = raw_sets[s_i].reshape(-1)
    raw_off = (np.arange(R + 1, dtype=np.uint64) * CHUNK)
    chunk0 = np.zeros(R, dtype=np.uint32)

    def set_slots(s_i):
        return (np.arange(R, dtype=np.uint32) + np.uint32(s_i * R))

    def upload(dv, s_i, want_len=False):
        return dv.upload_raw(pinned.array[s_i * R * CHUNK:(s_i + 1) * R * CHUNK], raw_off, synth.DIGITISATION, synth.RANGE,
                             synth.OFFSET, slots=set_slots(s_i), want_len=want_len)

    pool.for_each(lambda dv: dv.configure_reads(n_sets * R, CHUNK))
    lens = [upload(dv, s_i, want_len=True) for dv in pool.devs for s_i in range(n_sets)]   # inputs resident in HBM from here on
    assert all(int(l.min()) == CHUNK for l in lens), "the generator's reads lose no sample to the (30,200) pA filter"
    if args.sdtw_band > 0:
        ref_sig, ref_off = pipeline.reference_signals(genome, lm)
        pool.first.upload_reference(ref_sig, ref_off)     # one copy per GPU: the other contexts borrow it
        for dv in pool.devs[1:]:
            dv.share_reference(pool.first)

    prefetched = {}  # context -> the read set whose copy is already on its way (pcie_inclusive region)
    n_ctx = len(pool.devs)

    def step(dv=dev, i=0, with_upload=False, prefetch_next=False, fixed_set=None):
        s_i = (i % n_sets) if fixed_set is None else fixed_set
        if with_upload and prefetched.pop(id(dv), None) != s_i:
            upload(dv, s_i)                  # async: H2D on the copy stream, conversion waits for it
        dv.select(set_slots(s_i), chunk0, CHUNK)
        dv.run(n_refs)
        if args.sdtw_band > 0:
            dv.refine(band=args.sdtw_band, pad=32)
        if prefetch_next:
            # the NEXT step of this context (static assignment: i + n_ctx) maps another set, i.e. other slots: its
            # DMA and conversion run beside this step's kernels
            nxt = (i + n_ctx) % n_sets
            upload(dv, nxt)
            prefetched[id(dv)] = nxt
        dv.sync()
        res = dv.download()
        refined = dv.refine_download() if args.sdtw_band > 0 else None
        return res, refined

    # Census (untimed): one pass with the walker restricted to exactly nanoflann's node set, so
    # that nodes_visited / points_tested count the REFERENCE's work on these inputs (SURVEY 8d:
    # "counted on the actual inputs").  The timed passes skip subtrees whose tight box is out of
    # range (same hits, same order) and therefore touch fewer bytes than that figure.
    last_set = (args.steps - 1) % n_sets      # the set of the timed region's last step: the records reported below
    dev.params.reference_walk = 1
    census, _ = step(fixed_set=last_set)
    dev.params.reference_walk = 0
    # warm-up: W steps, and one on every context (each sizes its own scratch and hit pool)
    pool.for_each(lambda dv: step(dv, with_upload=True))
    if args.warmup > len(pool.devs):
        pool.run_steps(args.warmup - len(pool.devs), step)
    # what every set maps to (untimed, one context): the timed steps are checked against these, whatever context ran them
    expected = [step(dev, fixed_set=s_i) for s_i in range(n_sets)]
    pool.for_each(lambda dv: (dv.timing(True), dv.timing_reset()))

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        pool.for_each(lambda dv: dv.sync())

    def kernel_times(devs):
        # per-kernel device time (HIP events on each context's stream)
        kern = {}
        for name in ["convert", "events", "seed_order", "seed_lookup", "sort", "chain", "sdtw"]:
            got = [dv.timing_get(name) for dv in devs]  # (mean ms, launches) per context
            n = sum(int(g[1]) for g in got)
            ms = sum(g[0] * g[1] for g in got) / max(n, 1)
            kern[name] = {"mean_ms": round(ms, 4), "launches": n}
        return kern

    def max_over_ranks(dt):
        if dist is None:
            return dt
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.backend == "gloo" else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # Reference pass (not the reported value): the same K steps strictly one after the other on one
    # context, so that every kernel has the GPU to itself -- the standalone kernel durations.
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(dev, i)
    fence()
    serial = {"ms_per_step": (time.perf_counter() - t0) / args.steps * 1e3, "kernels": kernel_times([dev])}
    pool.for_each(lambda dv: dv.timing_reset())
    if args.ab_knob:
        # development A/B on the resident index (an index of 250 Mb takes minutes to build): the serial pass again under
        # each value of one development knob, per-stage kernel times on stderr.  Needs SMHIP_DEV=1.
        name, vals = args.ab_knob.split("=")
        for v in [int(x) for x in vals.split(",")] * 2:
            dev.set_knob(name, v)
            for i in range(2):
                step(dev, i)
            fence()
            dev.timing_reset()
            for i in range(args.steps):
                step(dev, i)
            fence()
            print("ab_knob %s=%d: %s  walker %s" % (name, v, {k: x["mean_ms"] for k, x in kernel_times([dev]).items()},
                                                    walker_stats(dev)), file=sys.stderr, flush=True)
            dev.timing_reset()
        dev.set_knob(name, int(vals.split(",")[0]))

    # ---- timed regions.  Three kinds, every one EXACTLY K steps between two fences (barrier + synchronize on both sides),
    # the time taken as the max over ranks:
    #   resident (`value`)  : inputs resident in HBM, step i maps read set i mod n_sets;
    #   same_reads          : the same with ONE read set for every step and context (what rounds 1-3 timed);
    #   pcie_inclusive      : every step first copies its reads' ADC counts from pinned host memory and converts them.
    # All three assign steps to contexts the same way (static: context k runs steps k, k + depth, ...; contexts start a
    # fraction of a step apart, inside the region).  The kinds take turns -- resident, same, pcie, resident, ... -- one
    # discarded round first, then `--repeats` rounds; a kind's figure is the MEDIAN of its regions, with min / max / all.
    stagger = (args.stagger * serial["ms_per_step"] * 1e-3 / len(pool.devs)) if len(pool.devs) > 1 else 0.0
    K = args.steps

    def region(kind):
        if kind == "pcie":
            # (a step's reads are copied exactly once: by the step itself, or by the context's previous step while
            # that one's kernels ran; the last step of a context prefetches nothing)
            fn = lambda dv, i: step(dv, i, with_upload=True, prefetch_next=i + n_ctx < K)   # noqa: E731
        elif kind == "same":
            fn = lambda dv, i: step(dv, i, fixed_set=0)   # noqa: E731
        else:
            fn = step
        pool.for_each(lambda dv: dv.timing_reset())
        fence()
        t0 = time.perf_counter()
        results = pool.run_steps(K, fn, stagger_s=stagger, static=True)
        fence()
        dt = max_over_ranks(time.perf_counter() - t0)
        prefetched.clear()
        if kind == "same":
            ok = all(r.tobytes() == expected[0][0].tobytes() for r, _ in results)
        else:
            ok = all(r.tobytes() == expected[i % n_sets][0].tobytes() for i, (r, _) in enumerate(results))
        return dt, results, ok, kernel_times(pool.devs)

    kinds = ["resident"]
    if n_sets > 1 and not args.no_same_reads:
        kinds.append("same")
    if not args.no_pcie:
        kinds.append("pcie")
    times = {k: [] for k in kinds}
    agree = {k: True for k in kinds}
    last = {}
    for rep in range(max(1, args.repeats) + 1):
        for kind in kinds:
            dt_r, results_r, ok, kern_r = region(kind)
            agree[kind] = agree[kind] and ok
            if rep == 0:
                continue                      # the discarded round
            times[kind].append(dt_r)
            last[kind] = (results_r, kern_r)

    def spread(kind):
        ts = sorted(times[kind])
        med = float(np.median(ts))
        return med, {"repeats": len(ts), "median_ms_per_step": round(med / K * 1e3, 3),
                     "min_ms_per_step": round(ts[0] / K * 1e3, 3), "max_ms_per_step": round(ts[-1] / K * 1e3, 3),
                     "all_ms_per_step": [round(t / K * 1e3, 3) for t in times[kind]],   # in the order they were timed
                     "value_at_min": round(R * world * K / ts[0], 1), "value_at_max": round(R * world * K / ts[-1], 1)}

    dt, value_spread = spread("resident")
    res, refined = last["resident"][0][-1]
    contexts_agree = agree["resident"]
    kern = last["resident"][1]  # (with batches in flight: brackets include queueing behind the other streams)
    same_reads = None
    if "same" in kinds:
        dt_same, sp = spread("same")
        same_reads = {"value": round(R * world * K / dt_same, 1), "ms_per_step": round(dt_same / K * 1e3, 3),
                      "vs_value": round(dt / dt_same, 4), "results_identical": bool(agree["same"]), "spread": sp}
    pcie = None
    if "pcie" in kinds:
        dt2, sp = spread("pcie")
        pcie = {"value": round(R * world * K / dt2, 1), "ms_per_step": round(dt2 / K * 1e3, 3),
                "h2d_bytes_per_step": int(R * CHUNK * 2), "results_identical": bool(agree["pcie"]),
                "convert_kernel_ms": last["pcie"][1]["convert"]["mean_ms"],
                "vs_resident": round(dt / dt2, 4), "spread": sp,
                "what": "every step's reads are copied as int16 ADC counts from pinned host memory (hipMemcpyAsync on a "
                        "copy stream) and converted on the device before mapping; a context starts the copy of its "
                        "next step's reads (another read set, other slots) while the current step's kernels run"}

    if rank == 0:
        status_bad = int((res["status"] & ~np.uint32(1)).astype(bool).sum())
        nodes = int(census["nodes_visited"].sum())
        points = int(census["points_tested"].sum())
        same = all(np.array_equal(census[k], res[k]) for k in census.dtype.names
                   if k not in ("nodes_visited", "points_tested"))
        anchors = int(res["n_anchors"].sum())
        queries = int(res["n_queries"].sum())
        w_nodes = int(res["nodes_visited"].sum())
        w_points = int(res["points_tested"].sum())
        # B_chunk = 2*S + Q*(Nn*16 + Np*28) + 24*A + 64 with the REFERENCE's node visits and
        # point tests, counted on the actual inputs (census pass)
        ref_bytes = 2 * CHUNK * R + 16 * nodes + 28 * points + 24 * anchors + 64 * R
        # the walker's own bytes: what it reads by its own layout -- one record per node visit (order-free search:
        # a 128-byte quad record, two tree levels; searches in the reference's order: a 64-byte binary record, leaf
        # visits read none: an upper bound), 32 bytes per point tested, 16 bytes per hit written, the query
        # records -- i.e. what a perfect memory system would have to move for it
        quad_walk = int(dev.debug_counters()[6]) == 2
        rec_bytes, levels_per_rec = (128, 2) if quad_walk else (64, 1)
        own_bytes = rec_bytes * w_nodes + 32 * w_points + 16 * int(res["hits_total"].sum()) + 64 * queries
        seed_ms = serial["kernels"]["seed_lookup"]["mean_ms"]
        # SURVEY 8(d) weights on the WALKER'S OWN visits (16 B per tree level of a record it reads, 28 B per point it
        # tests, 24 B per anchor, the raw samples and the result record): the bytes a perfect layout would have to move
        useful_bytes = 2 * CHUNK * R + 16 * levels_per_rec * w_nodes + 28 * w_points + 24 * anchors + 64 * R
        traffic, traffic_src, prof_ms, read_lines = None, None, None, None
        # a committed counter profile of THIS code on THIS workload (tools/collect_profiles.sh): same shape, the same
        # walker kernel (search mode) and the same walker sources (a hash of seed_lookup.hip + common.h recorded with
        # the profile) -- bytes counted for another kernel version are never paired with this run's duration
        import glob
        # (the set walk of a batch of up to 4 096 chunks shares a query's visits between lanes: ", true>")
        walker_kernel = ("SeedWalkKernel<2, 128, %s>" % ("true" if R <= 4096 else "false")) if quad_walk else "SeedWalkKernel<0, 256, false>"
        for pmc in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
            try:
                j = json.load(open(pmc))
                sh = j.get("shape") or {}
                if (sh.get("reads"), sh.get("contigs"), sh.get("contig_len")) == (args.reads, args.contigs, args.contig_len) \
                        and args.walk_blocks == 0 and args.fast_cap == 0 and j.get("kernel") == walker_kernel \
                        and j.get("source_sha16") == walker_source_sha():
                    traffic = int(j["seed_lookup_hbm_bytes_per_launch"])
                    read_lines = float(j.get("read_requests_128B") or 0) or None
                    traffic_src = j.get("source")
                    prof_ms = (j.get("launch_avg_ns_kernel_trace_serial") or 0) * 1e-6 or None
                    break
            except Exception:
                traffic = None
        # without a counter profile of this code and workload: the SURVEY 8(d) weights on the walker's own visits
        # (what a perfect layout would have to move) -- the honest algorithmic fraction, never above the peak
        basis = traffic if traffic else useful_bytes
        achieved = basis / (seed_ms * 1e-3) / 1e9 if seed_ms > 0 else 0.0
        # (128-byte READ requests of the profile: the random-line ceiling is a read ceiling)
        lines_per_s = (read_lines or traffic / 128.0) / (seed_ms * 1e-3) if traffic and seed_ms > 0 else None
        total_chunks = R * world * args.steps
        out = {
            "metric": "signal chunks/sec (4k-sample reads) mapped to yeast ref",
            "value": round(total_chunks / dt, 1),
            "unit": "chunks/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "timing": dict(value_spread, what=f"value / ms_per_step = the MEDIAN of {len(times['resident'])} timed regions of exactly "
                           f"{args.steps} steps each (one discarded region first; the regions of the kinds resident / same_reads / "
                           "pcie_inclusive take turns; same static step-to-context assignment in all three)"),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{R} synthetic 4000-sample reads (1 chunk each) per GPU vs {n_refs} x "
                            f"{args.contig_len} bp random reference (yeast-sized), event+seed+sort+chain"
                            f"{'+banded sDTW (band %d)' % args.sdtw_band if args.sdtw_band > 0 else ''} on GPU; "
                            "inputs resident in HBM (pcie_inclusive: copied from pinned host RAM every step)",
                "chunks_per_gpu_per_step": R,
                "transport": ("rccl" if args.backend == "nccl" else "gloo (test transport; ranks may share a GPU)")
                             if world > 1 else None,
                "forced_broadcast": ({"backend": "rccl" if args.backend == "nccl" else "gloo", "ranks": 1,
                                      "contexts_on_broadcast_copy": len(pool.devs) - 1} if forced else None),
                "batches_in_flight": len(pool.devs),
                "read_sets": n_sets,
                "read_sets_note": "step i maps read set i mod read_sets (R distinct synthetic reads each, all resident): no "
                                  "step repeats the previous step's queries, contexts in flight walk different queries",
                "same_reads_every_step": same_reads,
                "contexts_agree": bool(contexts_agree),
                "index_points": int(ix.n_points) if ix is not None else None,
                "mapped_fraction": round(float((res["n_chains"] > 0).mean()), 4),
                "queries_per_chunk": round(queries / R, 2),
                "nodes_per_query": round(nodes / max(queries, 1), 1),
                "points_per_query": round(points / max(queries, 1), 1),
                "walker_nodes_per_query": round(w_nodes / max(queries, 1), 1),
                "walker_record": ("128-byte quad record (a node's four grandchildren: two tree levels per visit)" if quad_walk
                                  else "64-byte binary record"),
                "walker_points_per_query": round(w_points / max(queries, 1), 1),
                "census_results_identical": bool(same),
                "anchors_per_chunk": round(anchors / R, 1),
                "chunks_with_capacity_flags": status_bad,
                "overflow_queries": int(dev.debug_counters()[1]),
                "fast_hit_slots_per_query": int(dev.resolved_params().hits_per_query_fast),
                "search_mode": {1: "reference order (binary records)", 2: "set walk (quad records); queries beyond the hit "
                                "cap: the first num_nearest_points hits in the reference's order picked by ranking"}.get(
                                    int(dev.resolved_params().search_mode), "?"),
                "index_census_hits": (lambda h: {"q50": int(h[len(h) // 2]), "q75": int(h[len(h) * 3 // 4]),
                                                 "q90": int(h[len(h) * 9 // 10]), "max": int(h[-1]), "mean": round(float(h.mean()), 1),
                                                 "share_over_5000": round(float((h > 5000).mean()), 5)})(dev.index_census()),
                "walker_phase_stats": walker_stats(dev),
                "sdtw_refined_fraction": (round(float((refined[1] >= 0).mean()), 4)
                                          if args.sdtw_band > 0 else None),
                "index_build_s": round(t_build, 2),
                "kernels": kern,  # event brackets in the timed region (with batches in flight: incl. queueing)
            },
            "pcie_inclusive": pcie,
            "roofline": {
                "bound": "hbm",
                "kernel": "seed_lookup",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_is_from_stored_profile": bool(traffic),
                "traffic_source": traffic_src,
                "basis": ("L2-miss (fabric) bytes of the committed counter profile named in traffic_source -- 128 B x "
                          "TCC_EA0_RDREQ + WRITE_SIZE per launch (profiles/r3_calibration.json); they INCLUDE Infinity-Cache "
                          "hits, so frac is an upper bound on HBM utilisation -- divided by THIS run's launch duration "
                          "(the profile's own: profile_launch_ms)" if traffic else
                          "SURVEY 8(d) algorithmic bytes on the walker's own visits (16 B per tree level of a record read, 28 B per "
                          "point tested, 24 B per anchor, samples in, result out) -- no counter profile of this code and workload "
                          "shape under profiles/"),
                "profile_launch_ms": round(prof_ms, 4) if prof_ms else None,
                # 128-byte lines per second against the 52 G random lines / s tools/micro/random_lines.hip measures beyond L2
                "frac_of_random_line_ceiling": round(lines_per_s / RANDOM_LINES_PER_S, 4) if lines_per_s else None,
                # SURVEY 8(d) weights on the walker's own node visits and point tests: what a perfect layout would move
                "useful_bytes_per_launch": useful_bytes,
                "useful_bytes_frac": round(useful_bytes / (seed_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if seed_ms > 0 else None,
                # the WHOLE STEP against the roofline (north_star's ">= 40 % of HBM roofline"): the same useful bytes -- SURVEY
                # 8(d)'s weights on the walker's own visits, samples in, anchors written and read, result out: every stage's
                # algorithmic bytes, per GPU -- over ms_per_step of the timed region instead of the walker's launch
                "step_useful_frac": round(useful_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                "step_useful_gbs": round(useful_bytes / (dt / args.steps) / 1e9, 1),
                "frac_of_measured_streaming_ceiling": round(achieved / HBM_STREAM_GBS, 4),
                "launch_ms": round(seed_ms, 4),
                "measured": "HIP events on the context stream around the SeedWalkKernel launch, serial pass of this run "
                            "(one batch in flight)",
                "launch_ms_bracket_in_timed_region": round(kern["seed_lookup"]["mean_ms"], 4),
                "own_algorithmic_bytes_per_launch": own_bytes,
                "own_algorithmic_gbs": round(own_bytes / (seed_ms * 1e-3) / 1e9, 1) if seed_ms > 0 else None,
                "reference_algorithm_bytes_per_launch": ref_bytes,
                "reference_algorithm_equiv_gbs": round(ref_bytes / (seed_ms * 1e-3) / 1e9, 1) if seed_ms > 0 else None,
                "note": ("frac = HBM-side bytes the kernel really moves / launch duration / 8 TB/s.  "
                         "reference_algorithm_equiv_gbs prices the REFERENCE's node visits and point tests "
                         "(SURVEY 8d formula, census pass) at this kernel's speed -- the kernel returns the same "
                         "hits while pruning exactly, so that figure is not a utilisation and may exceed the peak"),
            },
            "serial_steps": {  # the reference pass above: one batch in flight, kernels run alone
                "ms_per_step": round(serial["ms_per_step"], 3),
                "value": round(R * world / (serial["ms_per_step"] * 1e-3), 1),
                "kernels": serial["kernels"],
            },
        }
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(ix, raw[:4096], n_refs, ref_lens, args.cpu_seconds)
        if world == 1 and not args.no_streaming and not args.no_cpu:
            # (this process's contexts go first: the CLI in the child sizes its read slots -- and with them its number of
            # contexts -- by the memory the GPU has left)
            pool.close()
            out["streaming"] = streaming_block(args)
        print(json.dumps(out), flush=True)
        phase_prof()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    del keepalive
    pinned.close()


def streaming_block(args):
    """The product's real mode beside the headline (never part of `value`): 10 000 reads of 8 chunks -- a
    third junk-prefixed, 2 % junk throughout -- through the `sigmap -m` CLI in a child process (tools/stream_bench.py: its
    own contexts, the reference's own "Finished mapping in" span, median of five runs + one without the CLI's warm-up round, 32 sampled reads against the
    oracle's streaming map).  A failure here never costs the bench line."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stream_bench.py"), "--reads", "10000", "--chunks-per-read", "8",
                            "--contigs", str(args.contigs), "--contig-len", str(args.contig_len), "--check", "32", "--runs", "5"],
                           capture_output=True, text=True, timeout=420, cwd=ROOT)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": (r.stderr or "no output")[-400:]}
        j = json.loads(lines[-1])
        keep = ("metric", "reads_per_s", "chunks_per_s", "span_s", "all_span_s", "warmup_s", "no_warmup", "reads", "chunks_per_read", "read_kinds",
                "mean_chunks_used", "mapped_fraction", "rounds", "lanes", "mean_batch", "host_share_of_lane_time",
                "host_ms_per_round", "device_ms_per_round", "lane_seconds", "batches_mapped_twice", "oracle_check", "cli")
        return {k: j.get(k) for k in keep}
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)[:400]}


def walker_source_sha():
    """First 16 hex digits of sha256 over the walker's sources: a counter profile is only paired with a run of the
    same code (tools/summarize_profiles.py records the same value)."""
    import hashlib
    h = hashlib.sha256()
    for f in ("sigmap_amd/csrc/seed_lookup.hip", "sigmap_amd/csrc/common.h"):
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def phase_prof():
    """Development builds only (tools/phase_prof.sh: -DSMHIP_PHASE_PROF library given by SMHIP_LIB): the clock
    ticks thread 0 of every workgroup spent between the marks of an instrumented kernel, on stderr."""
    import ctypes
    from sigmap_amd import _abi
    fw = getattr(_abi.hip_lib(), "smhip_debug_walk_prof", None)
    if fw is not None:
        wb = (ctypes.c_ulonglong * 16)()
        if fw(wb, 1) == 0 and wb[0]:
            print("walk_prof maxima over the waves (shader-clock ticks; iterations): all %d (mean %.0f)  scan %d  hand-out %d  "
                  "fetch wait %d  walk iterations %d  scan iterations %d;  first start to last end %.1f us (wall clock)" %
                  (wb[8], wb[0] / max(wb[5], 1), wb[9], wb[10], wb[11], wb[12], wb[13], (wb[15] - wb[14]) / 100.0),
                  file=sys.stderr, flush=True)
            names = ["all", "scan phase", "query hand-out", "node fetch wait", "visit logic", None, "scan set-up",
                     "scan point-load wait"]
            print("walk_prof (shader-clock ticks summed over %d waves): " % wb[5] +
                  "  ".join("%s %.3g (%.1f%%)" % (names[k], wb[k], 100.0 * wb[k] / wb[0]) for k in range(8) if names[k]),
                  file=sys.stderr, flush=True)
    for name in ("smhip_debug_phase_prof", "smhip_debug_chain_prof"):
        fn = getattr(_abi.hip_lib(), name, None)
        if fn is None:
            continue
        buf = (ctypes.c_ulonglong * 32)()
        if fn(buf, 1) == 0:
            tot = float(sum(buf)) or 1.0
            print("phase_prof %s (wall_clock64 ticks, share): " % name[12:-5] +
                  "  ".join("[%d] %.3g (%.1f%%)" % (k, buf[k], 100.0 * buf[k] / tot) for k in range(32) if buf[k]),
                  file=sys.stderr, flush=True)


def walker_stats(dev):
    """Wave-iterations and mean active lanes of the walker's two phases in the last launch."""
    c = dev.debug_counters()[8:16].view(np.uint64)
    return {"walk_iters": int(c[0]), "walk_lanes": round(float(c[1]) / max(int(c[0]), 1), 1),
            "scan_iters": int(c[2]), "scan_lanes": round(float(c[3]) / max(int(c[2]), 1), 1)}


if __name__ == "__main__":
    main()
