#!/usr/bin/env python3
"""bench.py -- whole-job throughput of the SquiggleKit hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload motifseq|segmenter] [--scaling weak|strong]

Headline (BASELINE.json `metric`): reads/s of the MotifSeq path -- scale_outliers -> medmad -> subsequence
DTW -- for 4 000-sample int16 reads against a 200-point motif (config C4: 1 000 000 reads, seed 20260929 +
rank, synthetic squiggles generated on the device).  One "step" = one pass of the hot path (prep kernel + DTW
kernels) over the whole HBM-resident batch; inputs are already in HBM when the timed region starts.

N > 1: reads are block-sharded over the GPUs, no data-path collective; each step ends with the one exchange
the path has, an RCCL all-gather of the 24-byte hit records (`sk_comm_allgather_dev`, csrc/sk_comm.hip).
Two launch shapes give the same line (squigglekit_amd/multigpu.py):
    python bench.py --gpus N                                     one process, one host thread per GPU
    python -m torch.distributed.run --nproc-per-node N bench.py --gpus N      one process per GPU; only the
                     launcher's environment is read (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT) -- torch is
                     never imported; rank 0's ncclUniqueId travels through a file store under $TMPDIR.
`--scaling strong` (default): --reads in TOTAL, block-sharded (C4 as BASELINE.json words it: 1 M reads on 1/2/4/8
GPUs); `--scaling weak`: --reads per GPU.  Every rank keeps --reads reads resident, so at N > 1 the other curve is
measured in the same launch and reported beside the headline (`weak_scaling` / `strong_scaling`).  The line carries
`ranks_seen`, `gather_backend` and `per_rank` (each rank's own ms_per_step and host-to-device GB/s).  Asking for more
ranks than there are GPUs is refused (exit 2) unless SK_OVERSUBSCRIBE / --ranks-on-device says it is a dry run.

Dry run of the N > 1 paths on a box with fewer GPUs than ranks: `--ranks-on-device D` (or SK_OVERSUBSCRIBE=1 under a
per-GPU launcher) puts every rank on device D with its own context slot; the gather then runs on the host backend
(RCCL wants one device per rank).  At N > 1 (and with --force-comm) rank 0 checks a strided sample of EVERY rank's
shard, read out of the gathered buffer, against the oracle (`parity.ranks_checked`): the device generator is
deterministic in (seed + rank, row), so rank 0 regenerates any rank's rows.

Rank 0 prints the headline JSON line LAST (extras, when asked for, as {"extra": ...} lines before it); it holds the driver's contract plus `roofline` (HIP-event kernel time vs algorithmic
bytes, and the VALU-issue view that actually binds this kernel -- DESIGN.md 4.3), `cpu_baseline` (the oracle
timed on the host: the only place it is timed), `parity` (a sample strided over the whole batch, checked
against the oracle) and, at N = 1, `secondary` (segmenter line), `exact_only_reads_per_s` and `end_to_end`.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_F64_LANEOPS = 256 * 4 * 16 * 2.4e9   # 3.93e13 f64 add/min/cmp lane-ops per second
WAVE_ISSUE_SLOTS = 256 * 4 * 64 * 2.4e9   # lane-results per second if every SIMD issued a wave64 VALU op per cycle
HIT_BYTES = 24
MAX_SEGS = 16


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="motifseq", choices=["motifseq", "segmenter"])
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="strong (default): --reads in TOTAL, block-sharded over the GPUs -- C4 as BASELINE.json words it "
                         "(1 M reads, 1/2/4/8 GPUs); weak: --reads per GPU.  At N > 1 the other one is reported as an extra")
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads in total (strong) or per GPU (weak)")
    ap.add_argument("--samples", type=int, default=4000)
    ap.add_argument("--motif", type=int, default=200, help="motif points")
    ap.add_argument("--scale", default="medmad", choices=["medmad", "zscale"])
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="1-core CPU baseline budget (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=-1,
                    help="threads of the all-cores CPU baseline (-1 = every host core, 0 = skip)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the N = 1 extras (secondary segmenter line, exact-only schemes, end-to-end ingest)")
    ap.add_argument("--no-sensitivity", action="store_true",
                    help="skip the N = 1 sensitivity block (real-signal windows, retry-fraction sweep)")
    ap.add_argument("--only-other-paths", action="store_true",
                    help="N = 1: after the timed region run only the other_paths block of the extras")
    ap.add_argument("--sweep-reads", action="store_true",
                    help="N = 1 with --no-extras: still run the reads-per-call sweep (1 M ... 31 250 reads per call; the "
                         "one-GPU prediction of the strong-scaling curve).  Part of the default extras")
    ap.add_argument("--full-json", default=None, metavar="PATH",
                    help="also write headline + every extra block as ONE JSON object to PATH (tools/)")
    ap.add_argument("--ranks-on-device", type=int, default=None, metavar="D",
                    help="dry run: all --gpus ranks share device D (own context slot each, host-backend gather)")
    ap.add_argument("--force-comm", action="store_true",
                    help="N = 1: still create the RCCL communicator and gather every step (exercises the N > 1 path)")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------------------------------
def strided_rows(total, want, run=8):
    """About `want` row indices spread over the whole batch: runs of `run` consecutive rows at evenly spaced
    positions, first and last rows included."""
    nruns = max(2, want // run)
    starts = np.unique(np.linspace(0, max(0, total - run), nruns).astype(np.int64))
    idx = (starts[:, None] + np.arange(run)[None, :]).ravel()
    return np.unique(idx[(idx >= 0) & (idx < total)])


def download_rows(L, d_base, row_bytes, rows, dtype, row_items, run=8):
    """Rows `rows` (sorted) of a device array -> numpy; consecutive rows travel in one copy."""
    from squigglekit_amd._lib import check
    base = C.cast(d_base, C.c_void_p).value
    out = np.empty((len(rows), row_items), dtype=dtype)
    k = 0
    while k < len(rows):
        j = k
        while j + 1 < len(rows) and rows[j + 1] == rows[j] + 1 and j + 1 - k < 4096:
            j += 1
        view = out[k:j + 1]
        check(L.sk_dev_download(view.ctypes.data_as(C.c_void_p), C.c_void_p(base + int(rows[k]) * row_bytes),
                                view.nbytes))
        k = j + 1
    return out


class Workload:
    """Device-resident inputs and outputs of one rank, and its step()."""

    def __init__(self, a, L, rank, world, R, workload=None, comm=None, gather_pad=None):
        from squigglekit_amd import _lib, synth
        from squigglekit_amd._lib import SegParams, check, ptr
        self.a, self.L, self.R, self.rank, self.world, self.comm = a, L, R, rank, world, comm
        self.kind = workload or a.workload
        M, N = a.samples, a.motif
        self.M, self.N = M, N
        self.stride = (M + 7) // 8 * 8
        self.motif = synth.synthetic_motif(N)
        self.seed = (synth.SEED_C4 if self.kind == "motifseq" else synth.SEED_C2) + rank
        self.mode = _lib.SK_SCALE[a.scale]
        self.bufs = []
        alloc = self._alloc
        self.d_sig = alloc(max(1, R) * self.stride * 2)
        self.d_len = alloc(max(1, R) * 4)
        self.lens = np.full(R, M if self.kind == "motifseq" else M - 1, dtype=np.int32)   # segmenter: Num = -1
        if R:
            check(L.sk_dev_upload(self.d_len, ptr(self.lens), self.lens.nbytes))
            check(L.sk_synth_squiggles_dev(self.d_sig, self.stride, R, M, self.seed, ptr(self.motif), N))
        self.pad = gather_pad if gather_pad is not None else R
        self.rec_bytes = HIT_BYTES if self.kind == "motifseq" else 4
        if self.kind == "motifseq":
            self.d_out = alloc(max(1, self.pad) * HIT_BYTES)
        else:
            self.d_segs = alloc(max(1, R) * MAX_SEGS * 2 * 4)
            self.d_out = alloc(max(1, self.pad) * 4)                  # nsegs: the fixed-size record gathered
            self.sp = SegParams()
        self.d_all = alloc(max(1, self.pad) * self.rec_bytes * world) if comm is not None else None
        self.host_rec = None
        self.host_all = None                                         # host backend: what the last gather returned

    def _alloc(self, nbytes):
        from squigglekit_amd._lib import check
        p = self.L.sk_dev_alloc(nbytes)
        if not p:
            check(-4)
        self.bufs.append(p)
        return p

    def free(self):
        for p in self.bufs:
            self.L.sk_dev_free(p)
        self.bufs = []

    def step(self):
        from squigglekit_amd._lib import check, ptr
        L = self.L
        if self.R:
            if self.kind == "motifseq":
                check(L.sk_motifseq_dev_i16(self.d_sig, self.stride, self.d_len, self.R, ptr(self.motif), self.N,
                                            self.mode, 0, 1200, self.d_out))
            else:
                check(L.sk_segment_dev_i16(self.d_sig, self.stride, self.d_len, self.R, C.byref(self.sp),
                                           self.d_segs, self.d_out, MAX_SEGS))
        if self.comm is not None and self.comm.backend == "rccl":
            # the one exchange: all-gather of the result records over RCCL, on the library's stream
            self.comm.allgather_dev(self.d_out, self.d_all, self.pad * self.rec_bytes)
        check(L.sk_sync())
        if self.comm is not None and self.comm.backend != "rccl":
            # RCCL could not be loaded / initialised: the same exchange by host concatenation
            if self.host_rec is None or self.host_rec.size != max(1, self.pad) * self.rec_bytes:
                self.host_rec = np.zeros(max(1, self.pad) * self.rec_bytes, dtype=np.uint8)
            if self.R:
                check(L.sk_dev_download(ptr(self.host_rec), self.d_out, self.R * self.rec_bytes))
            self.host_all = self.comm.allgather_host(self.host_rec)

    def gathered(self):
        """The last step's gathered records as uint8 [world, pad * rec_bytes] (out of d_all, or the host concat)."""
        from squigglekit_amd._lib import check, ptr
        if self.comm is None:
            return None
        if self.comm.backend == "rccl":
            out = np.empty((self.world, max(1, self.pad) * self.rec_bytes), dtype=np.uint8)
            check(self.L.sk_dev_download(ptr(out), self.d_all, out.nbytes))
            return out
        return np.asarray(self.host_all).reshape(self.world, -1)

    def regenerate(self, **opts):
        """Refill d_sig from the device generator (opts: _lib.SynthOpts fields; none = the default batch)."""
        from squigglekit_amd._lib import SynthOpts, check, ptr
        o = SynthOpts(**opts)
        check(self.L.sk_synth_variant_dev(self.d_sig, self.stride, self.R, self.M, self.seed, ptr(self.motif), self.N,
                                          C.byref(o)))

    def kernel_ms(self):
        from squigglekit_amd._lib import check
        p_, m_ = C.c_float(), C.c_float()
        check(self.L.sk_last_kernel_ms(C.byref(p_), C.byref(m_)))
        return p_.value, m_.value

    def dtw_profile(self):
        from squigglekit_amd._lib import check
        da, sb = C.c_float(), C.c_float()
        la, lb, rpl = C.c_int32(), C.c_int32(), C.c_int32()
        check(self.L.sk_last_dtw_profile(C.byref(da), C.byref(la), C.byref(sb), C.byref(lb), C.byref(rpl)))
        return da.value, sb.value, la.value, self.L.sk_last_dtw_retries()


def timed(w, comm, steps, warmup):
    """W warmup steps, then exactly K timed steps between barrier + device sync on both sides; returns the
    MAX over ranks of the elapsed time plus this rank's HIP-event sums."""
    from squigglekit_amd._lib import check

    def fence():
        check(w.L.sk_sync())
        if comm is not None:
            comm.barrier()
            check(w.L.sk_sync())

    for _ in range(warmup):
        w.step()
    prof = {"prep_ms": 0.0, "main_ms": 0.0, "dist_ms": 0.0, "start_ms": 0.0, "launches": 0, "retries": 0}
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        w.step()
        if w.R:
            p_, m_ = w.kernel_ms()                  # HIP events on the library's stream
            prof["prep_ms"] += p_
            prof["main_ms"] += m_
            if w.kind == "motifseq":
                da, sb, la, rt = w.dtw_profile()
                prof["dist_ms"] += da
                prof["start_ms"] += sb
                prof["launches"] += la
                prof["retries"] += rt
    fence()
    elapsed = time.perf_counter() - t0
    if w.R and w.kind == "motifseq":                       # the screening scheme's run-time guard, last timed step
        g = (C.c_int32 * 8)()
        check(w.L.sk_last_dtw_guard(g))
        prof["guard"] = {"premise_violations": int(g[0]), "audited_reads": int(g[1]), "audit_mismatches": int(g[2]),
                         "image_rejects": int(g[3]), "exact_fallback": int(g[5])}
    w.own_elapsed = elapsed                                # (this rank's; the return value is the maximum over ranks)
    if comm is not None:
        elapsed = float(comm.allgather_host(np.array([elapsed], dtype=np.float64)).max())
    return elapsed, prof


# ----------------------------------------------------------------------------------------------------
# rank 0 extras: parity, CPU baselines, secondary lines
# ----------------------------------------------------------------------------------------------------
def parity_and_cpu(a, w, want_cpu):
    """Oracle check on a sample strided over rank 0's whole batch (every chunk of the screening path), and the
    CPU baselines (N = 1 only) timed on the same reads."""
    from oracle import oracle as ora
    from squigglekit_amd._lib import HIT_DTYPE, check, ptr
    L, R = w.L, w.R
    rows = strided_rows(R, min(R, 8192))
    sample = download_rows(L, w.d_sig, w.stride * 2, rows, np.int16, w.stride)
    lens = w.lens[rows]
    cpu = None
    if w.kind == "motifseq":
        hits = np.empty(R, dtype=HIT_DTYPE)
        check(L.sk_dev_download(ptr(hits), w.d_out, hits.nbytes))
        got_all = hits[rows]
        # the oracle goes through the sample in an order that is itself strided, so that whatever part of it the
        # CPU budget covers still spans the whole batch
        import math
        step = max(1, int(len(rows) * 0.6180339887))                # golden-ratio stride, made coprime with the count:
        while math.gcd(step, len(rows)) != 1:                        # every prefix of `order` is spread over the batch
            step += 1
        order = (np.arange(len(rows), dtype=np.int64) * step) % len(rows)
        n0 = min(len(rows), 128)
        t0 = time.perf_counter()
        want = [ora.motifseq_batch_i16(sample[order[:n0]], lens[order[:n0]], w.motif, scale_mode=w.mode)]
        dt = time.perf_counter() - t0
        done = n0
        budget = a.cpu_seconds if want_cpu else 3.0
        more = int(min(len(rows) - done, max(0, (budget - dt) / (dt / n0))))
        if more > 0:
            t1 = time.perf_counter()
            want.append(ora.motifseq_batch_i16(sample[order[done:done + more]], lens[order[done:done + more]],
                                               w.motif, scale_mode=w.mode))
            dt += time.perf_counter() - t1
            done += more
        want = np.concatenate(want)
        got = got_all[order[:done]]
        covered = rows[order[:done]]
        parity = {"reads_checked": int(done),
                  "sample": "strided over the whole batch: reads %d..%d" % (int(covered.min()), int(covered.max())),
                  "start_end_exact": bool(np.array_equal(got["start"], want["start"])
                                          and np.array_equal(got["end"], want["end"])),
                  "max_abs_ddist": float(np.nanmax(np.abs(got["dist"] - want["dist"]))),
                  "dist_bit_identical": bool(np.array_equal(got["dist"], want["dist"]))}
        mean_n = float(np.mean(hits["n"]))
        if want_cpu:
            cpu = {"value": done / dt, "unit": "reads/s", "cores": 1, "kind": "port",
                   "sample": "%d of rank 0's reads (%d x %d-pt motif): oracle C restatement of filter+medmad+mlpy "
                             "dtw_subsequence (full matrix malloc per call), gcc -O2, 1 thread, %.1f s"
                             % (done, w.M, w.N, dt),
                   "host_cores_total": os.cpu_count()}
            T = (os.cpu_count() or 1) if a.cpu_threads < 0 else a.cpu_threads
            if T > 1:
                from concurrent.futures import ThreadPoolExecutor
                per = max(1, min(len(rows) // T, int(4.0 * cpu["value"])))       # a few seconds per thread
                parts = [(i * per, (i + 1) * per) for i in range(T)]
                with ThreadPoolExecutor(T) as ex:                              # ctypes calls release the GIL
                    t1 = time.perf_counter()
                    list(ex.map(lambda ab: ora.motifseq_batch_i16(sample[ab[0]:ab[1]], lens[ab[0]:ab[1]], w.motif,
                                                                  scale_mode=w.mode), parts))
                    dtt = time.perf_counter() - t1
                cpu["all_cores"] = {"value": T * per / dtt, "unit": "reads/s", "cores": T,
                                    "sample": "%d reads on each of %d threads (reads split over threads), %.1f s"
                                              % (per, T, dtt)}
            # "as shipped": the reference's own Python around the C DTW (MotifSeq.py:192-200 per-sample loop)
            k = min(48, done)
            t1 = time.perf_counter()
            for r in range(k):
                x = sample[order[r], :lens[order[r]]].astype(np.float64)
                ora.medmad_python_loop(x[(x > 0) & (x < 1200)])
            loop_ms = (time.perf_counter() - t1) / k * 1e3
            dtw_ms = 1e3 / cpu["value"]
            cpu["as_shipped"] = {"value": 1e3 / (loop_ms + dtw_ms), "unit": "reads/s", "cores": 1,
                                 "note": "estimate: the reference's per-sample Python medmad loop (%.2f ms/read, "
                                         "restated, %d reads) + the C DTW above (%.2f ms/read); TSV parsing not "
                                         "included" % (loop_ms, k, dtw_ms)}
        return parity, cpu, mean_n
    segs = np.empty((R, MAX_SEGS, 2), dtype=np.int32)
    nsegs = np.empty(R, dtype=np.int32)
    check(L.sk_dev_download(ptr(segs), w.d_segs, segs.nbytes))
    check(L.sk_dev_download(ptr(nsegs), w.d_out, nsegs.nbytes))
    t0 = time.perf_counter()
    osegs, onsegs = ora.segment_batch_i16(sample, lens, max_segs=MAX_SEGS)
    dt = time.perf_counter() - t0
    same = bool(np.array_equal(nsegs[rows], onsegs)) and all(
        np.array_equal(segs[r, :nsegs[r]], osegs[k, :onsegs[k]]) for k, r in enumerate(rows))
    parity = {"reads_checked": int(len(rows)), "sample": "strided over the whole batch",
              "segments_bit_exact": same}
    if want_cpu:
        cpu = {"value": len(rows) / dt, "unit": "reads/s", "cores": 1, "kind": "port",
               "sample": "%d of rank 0's reads: oracle C restatement of filter+get_segs, gcc -O2, 1 thread, %.2f s"
                         % (len(rows), dt), "host_cores_total": os.cpu_count()}
        k = min(64, len(rows))
        t1 = time.perf_counter()
        for r in range(k):
            x = sample[r, :lens[r]].astype(np.float64)
            ora.get_segs_python(x[(x > 0) & (x < 900)])
        cpu["as_shipped"] = {"value": k / (time.perf_counter() - t1), "unit": "reads/s", "cores": 1,
                             "note": "the reference's get_segs at interpreter speed (restated, %d reads)" % k}
    return parity, cpu, float(w.M - 1)


def regenerate_rows(w, seed, rows, run=8):
    """Rows `rows` (sorted, made of runs of consecutive rows) of the batch the device generator makes under `seed`,
    regenerated into a scratch buffer -- whatever rank holds that batch -- and downloaded."""
    from squigglekit_amd._lib import SynthOpts, check, ptr
    L = w.L
    out = np.empty((len(rows), w.stride), dtype=np.int16)
    d_tmp = L.sk_dev_alloc(4096 * w.stride * 2)
    if not d_tmp:
        check(-4)
    try:
        k = 0
        while k < len(rows):
            j = k
            while j + 1 < len(rows) and rows[j + 1] == rows[j] + 1 and j + 1 - k < 4096:
                j += 1
            cnt = j + 1 - k
            o = SynthOpts(row0=int(rows[k]))
            check(L.sk_synth_variant_dev(d_tmp, w.stride, cnt, w.M, seed, ptr(w.motif), w.N, C.byref(o)))
            view = out[k:j + 1]
            check(L.sk_dev_download(ptr(view), d_tmp, view.nbytes))
            k = j + 1
    finally:
        L.sk_dev_free(d_tmp)
    return out


def verify_gather(a, w, shard_sizes, per_rank=128):
    """Rank 0, N > 1 (or --force-comm): a strided sample of EVERY rank's shard, taken out of the gathered buffer
    (d_all after ncclAllGather, or the host concatenation), against the oracle on that rank's regenerated rows."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as ora
    from squigglekit_amd._lib import HIT_DTYPE
    blocks = w.gathered()
    base_seed = w.seed - w.rank
    T = max(1, min(16, os.cpu_count() or 1))
    res = []
    for r in range(w.world):
        Rr = int(shard_sizes[r])
        if Rr == 0:
            res.append({"rank": r, "reads": 0, "ok": True})
            continue
        rows = strided_rows(Rr, min(Rr, per_rank))
        sample = regenerate_rows(w, base_seed + r, rows)
        lens = np.full(len(rows), w.M if w.kind == "motifseq" else w.M - 1, dtype=np.int32)
        entry = {"rank": r, "reads": int(len(rows)), "rows": "%d..%d" % (int(rows[0]), int(rows[-1]))}
        if r == w.rank:                                              # the generator slice IS what sits in HBM
            resident = download_rows(w.L, w.d_sig, w.stride * 2, rows, np.int16, w.stride)
            entry["regenerated_equals_resident"] = bool(np.array_equal(resident[:, :w.M], sample[:, :w.M]))
        parts = [(i, min(len(rows), i + (len(rows) + T - 1) // T)) for i in range(0, len(rows), (len(rows) + T - 1) // T)]
        if w.kind == "motifseq":
            got = blocks[r][:Rr * HIT_BYTES].view(HIT_DTYPE)[rows]
            with ThreadPoolExecutor(T) as ex:                        # the oracle's ctypes calls release the GIL
                want = np.concatenate(list(ex.map(lambda ab: ora.motifseq_batch_i16(
                    sample[ab[0]:ab[1]], lens[ab[0]:ab[1]], w.motif, scale_mode=w.mode), parts)))
            entry["dist_bit_identical"] = bool(np.array_equal(got["dist"], want["dist"]))
            entry["start_end_exact"] = bool(np.array_equal(got["start"], want["start"])
                                            and np.array_equal(got["end"], want["end"]))
            entry["ok"] = entry["dist_bit_identical"] and entry["start_end_exact"] and \
                entry.get("regenerated_equals_resident", True)
        else:
            got = blocks[r][:Rr * 4].view(np.int32)[rows]
            _, onsegs = ora.segment_batch_i16(sample, lens, max_segs=MAX_SEGS)
            entry["segment_counts_exact"] = bool(np.array_equal(got, onsegs))
            entry["ok"] = entry["segment_counts_exact"] and entry.get("regenerated_equals_resident", True)
        res.append(entry)
    return {"ranks_checked": len(res), "every_rank_ok": bool(all(e["ok"] for e in res)),
            "source": "gathered buffer: %s" % ("d_all after ncclAllGather" if w.comm.backend == "rccl"
                                               else "host concatenation of the ranks' records"),
            "per_rank": res}


def e2e_all_ranks(a, w, comm):
    """N > 1, every rank: host arrays in -> sk_motifseq_batch_i16 (sub-batched H2D under the kernels) -> host
    records out, one feeder thread / process per GPU with pinned memory, all ranks at once between barriers.
    Returns (reads, seconds of the slowest rank) on every rank."""
    from squigglekit_amd import api
    from squigglekit_amd._lib import HIT_DTYPE, check, ptr
    L = w.L
    Rh = min(w.R, 200_000)
    hits = np.zeros(max(1, Rh), dtype=HIT_DTYPE)
    host = api.pinned_empty((max(1, Rh), w.stride), np.int16)
    lens = w.lens[:Rh]
    if Rh:
        check(L.sk_dev_download(ptr(host), w.d_sig, Rh * w.stride * 2))
    best, own = None, None
    for it in range(3):
        comm.barrier()
        t0 = time.perf_counter()
        if Rh:
            check(L.sk_motifseq_batch_i16(ptr(host), w.stride, ptr(lens), Rh, ptr(w.motif), w.N, w.mode, 0, 1200,
                                          ptr(hits)))
        mine = time.perf_counter() - t0
        dt = float(comm.allgather_host(np.array([mine], dtype=np.float64)).max())
        if it and (best is None or dt < best):
            best = dt
        if it and (own is None or mine < own):
            own = mine
    del host
    return Rh, best, (Rh * w.stride * 2 / own / 1e9 if (own and Rh) else 0.0)


def cli_block(a, L, main):
    """N = 1 extras: the drop-in command-line tools end to end, process start and text output included, on the
    batch's own reads: all of them (up to 1 M) as a packed int16 .npy (--i16) and as a BLOW5 file (--blow5), 200 000
    lines (3.2 GB) of the SquigglePull-style TSV the reference reads (-s; 256 distinct reads cycled: the tokenizer does
    not care).
    tools/cli_throughput.py is the same thing stand-alone (profiles/r03_cli_throughput.txt)."""
    import shutil
    import subprocess
    import tempfile
    from squigglekit_amd import fastio
    from squigglekit_amd._lib import check, ptr
    d = tempfile.mkdtemp()
    out = {"note": "wall clock of the whole process (interpreter start, HIP start-up, ingest, kernels, text out), best "
                   "of 2; ~0.1 s of interpreter start and ~0.1 s of process exit are in every figure, the HIP start-up "
                   "(~0.25 s) runs beside the first chunks"}
    try:
        Rp = min(main.R, 1_000_000)
        host = np.empty((Rp, main.stride), dtype=np.int16)
        check(L.sk_dev_download(ptr(host), main.d_sig, host.nbytes))
        reads = host[:, :main.M]
        model = os.path.join(ROOT, "tests", "golden", "CATCTATCCAGGGTTAAATT.model")
        seg, mot = os.path.join(ROOT, "segmenter.py"), os.path.join(ROOT, "MotifSeq.py")

        def timed_runs(runs):
            for label, n, cmd in runs:
                best, lines = None, 0
                for _ in range(2):
                    t0 = time.perf_counter()
                    p = subprocess.run([sys.executable] + cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300)
                    dt = time.perf_counter() - t0
                    if p.returncode != 0:
                        best = None
                        break
                    best = dt if best is None else min(best, dt)
                    lines = p.stdout.count(b"\n")
                    del p
                out[label] = {"reads": n, "seconds": best, "reads_per_s": n / best if best else None, "output_lines": lines}

        # one input file at a time on the scratch disk (8 GB each at C4)
        f = os.path.join(d, "r.npy")
        np.save(f, reads)
        timed_runs((("segmenter_i16", Rp, [seg, "--i16", f]), ("motifseq_i16", Rp, [mot, "--i16", f, "-m", model])))
        os.remove(f)
        f = os.path.join(d, "r.blow5")
        fastio.write_blow5(f, reads)
        timed_runs((("segmenter_blow5", Rp, [seg, "--blow5", f, "--raw_signal"]),
                    ("motifseq_blow5", Rp, [mot, "--blow5", f, "-m", model])))
        os.remove(f)
        Rt = min(Rp, 200_000)
        texts = ["\t".join(str(v) for v in reads[r].tolist()) for r in range(min(Rt, 256))]
        reads256 = np.array(reads[:min(Rt, 256)])
        del host, reads
        for label, ncols, cmd in (("segmenter_tsv", 4, [seg, "-s"]), ("motifseq_tsv", 8, [mot, "-m", model, "-s"])):
            f = os.path.join(d, label)
            with open(f, "w") as fh:
                for r in range(Rt):
                    fh.write("\t".join(["read%d.fast5" % r, "id%d" % r] + ["x"] * (ncols - 2)) + "\t"
                             + texts[r % len(texts)] + "\n")
            timed_runs(((label, Rt, cmd + [f]),))
            os.remove(f)
        # ... and as SquigglePull writes them by default: pA values with two decimals (float64 route, 100 000 lines = 2.5 GB)
        Rq = min(Rt, 100_000)
        pa = np.round((reads256.astype(np.int64) + PA_OFFSET) * (PA_RANGE / PA_DIGITISATION), 2)
        texts = ["\t".join(repr(float(v)) for v in pa[r]) for r in range(pa.shape[0])]
        for label, ncols, cmd in (("segmenter_tsv_pA", 4, [seg, "-s"]), ("motifseq_tsv_pA", 8, [mot, "-m", model, "-s"])):
            f = os.path.join(d, label)
            with open(f, "w") as fh:
                for r in range(Rq):
                    fh.write("\t".join(["read%d.fast5" % r, "id%d" % r] + ["x"] * (ncols - 2)) + "\t"
                             + texts[r % len(texts)] + "\n")
            timed_runs(((label, Rq, cmd + [f]),))
            os.remove(f)
    except Exception as e:                                            # noqa: BLE001 -- report, keep the line
        out["error"] = repr(e)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return out


def sensitivity_block(a, L, main):
    """N = 1, after everything else: what the headline is worth on less friendly data.  (i) reads that are windows of
    the one measured squiggle the reference ships (example/slow5/0.blow5, 36 978 samples; copy under tests/golden)
    plus N(0, 3) noise, against the example model (163 points) and the synthetic 200-point motif; (ii) the C4 batch
    with a given share of the reads forced through the exact retry (what a retry rate of x % costs); (iii) the C4
    batch with half of the reads also carrying the motif stretched 2 / 3 / 4 times in time (wide optimal paths: second
    tier of the window pass, then the retry).  Three steps each, best taken."""
    from squigglekit_amd import blow5
    from squigglekit_amd._lib import check, ptr
    out = {"reads": main.R, "note": "HBM-resident, kernels only (as the headline); ms = best of 3 steps"}

    def run(motif):
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            check(L.sk_motifseq_dev_i16(main.d_sig, main.stride, main.d_len, main.R, ptr(motif), motif.size,
                                        main.mode, 0, 1200, main.d_out))
            check(L.sk_sync())
            ts.append(time.perf_counter() - t0)
        return {"reads_per_s": main.R / min(ts), "ms": min(ts) * 1e3, "retried_reads": int(L.sk_last_dtw_retries()),
                "second_tier_reads": int(L.sk_last_dtw_tier2())}

    try:
        read = next(blow5.read_blow5(os.path.join(ROOT, "tests", "golden", "example_0.blow5")))
        raw = np.asarray(read["signal"], dtype=np.int16)
        import gzip
        with gzip.open(os.path.join(ROOT, "tests", "golden", "motifseq_cli.json.gz"), "rt") as fh:
            model163 = np.array(json.load(fh)["model_expanded"]["values"], dtype=np.float64)
        main.regenerate(tmpl=raw, tmpl_noise=3.0)
        out["real_signal_windows"] = {
            "source": "tests/golden/example_0.blow5 (%d samples): random %d-sample windows + N(0,3) noise" % (raw.size, main.M),
            "vs_example_model_163pt": run(model163),
            "vs_synthetic_%dpt_motif" % main.N: run(main.motif)}
    except Exception as e:                                            # noqa: BLE001 -- report, keep the line
        out["real_signal_windows"] = {"error": repr(e)}
    # (ii) what a retry costs: a given share of the reads is sent to the exact single pass whatever the window pass
    # found (SK_DTW_FORCE_RETRY_PM, a switch of the window kernel for exactly this measurement) -- on the C4 batch
    main.regenerate()
    sweep = {}
    for pm in (0, 10, 100, 500):
        os.environ["SK_DTW_FORCE_RETRY_PM"] = str(pm)
        try:
            sweep["%g%%" % (pm / 10.0)] = run(main.motif)
        finally:
            del os.environ["SK_DTW_FORCE_RETRY_PM"]
    base = sweep["0%"]["reads_per_s"]
    for v in sweep.values():
        v["vs_no_retries"] = v["reads_per_s"] / base
    out["forced_retry_sweep"] = {"what": "C4 batch; this share of the reads (by hash of the read index) takes the exact "
                                         "single-pass retry regardless of what the window pass certified",
                                 "by_share": sweep}
    # (iii) data that produces wide paths by itself: half of the reads also carry the motif stretched k times in
    # time (k x N samples); a match wider than the first look-back goes to the second tier, wider than that to the retry
    wide = {}
    for k in (2, 3, 4):
        main.regenerate(stretch_permille=500, stretch=k)
        wide["x%d" % k] = run(main.motif)
    out["stretched_motif_in_half_of_the_reads"] = wide
    main.regenerate()                                                 # the default batch again
    return out


def sweep_block(a, L, main):
    """N = 1: the headline's kernels on 1 M / 500 k / 250 k / 125 k / 62.5 k / 31 250 of the resident reads per call --
    what each GPU sees when C4 (1 M reads IN TOTAL) is block-sharded over 1 / 2 / 4 / 8 / 16 / 32 GPUs -- for MotifSeq and
    for the segmenter.  No multi-GPU node was available to any round so far: this is the one-GPU prediction of the
    strong-scaling curve (the per-rank work is exactly a call of that size; what it leaves out is the 24 B/read
    all-gather, 3 MB per GPU at N = 8, and PCIe ingest, which bench.py's end_to_end block times).  Fit: ms = fixed + per_read * R
    over the sizes; predicted efficiency at N GPUs = rate(R / N) / rate(R)."""
    from squigglekit_amd._lib import SegParams, check, ptr
    sizes = [main.R // d for d in (1, 2, 4, 8, 16, 32) if main.R // d >= 4096]

    def fit(rows):
        x = np.array([r["reads_per_call"] for r in rows], dtype=np.float64)
        y = np.array([r["ms"] for r in rows], dtype=np.float64)
        b, c = np.polyfit(x, y, 1)
        return {"fixed_ms_per_call": float(c), "us_per_1000_reads": float(b * 1e6),
                "note": "least-squares line through (reads per call, ms)"}

    def run(call, cells_per_read=None, bytes_per_read=None):
        rows = []
        clk = None
        for R in sizes:
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                call(R)
                check(L.sk_sync())
                ts.append(time.perf_counter() - t0)
            t = min(ts[1:])
            row = {"reads_per_call": R, "ms": t * 1e3, "reads_per_s": R / t}
            if cells_per_read:
                ghz = C.c_double(0.0)
                L.sk_last_dtw_clock(C.byref(ghz))
                clk = ghz.value if 0.5 < ghz.value < 3.0 else clk
                roof = WAVE_ISSUE_SLOTS / 8.0 * ((clk or 2.4) / 2.4)
                row["issue_roof_frac"] = R * cells_per_read / t / roof
            if bytes_per_read:
                row["hbm_frac"] = R * bytes_per_read / t / 1e9 / HBM_PEAK_GBS
            rows.append(row)
        base = rows[0]["reads_per_s"]
        for r in rows:
            r["vs_full_batch_rate"] = r["reads_per_s"] / base
        return rows

    out = {"note": sweep_block.__doc__.split("\n\n")[0].replace("\n    ", " ")}
    if main.kind == "motifseq":
        rows = run(lambda R: check(L.sk_motifseq_dev_i16(main.d_sig, main.stride, main.d_len, R, ptr(main.motif), main.N,
                                                          main.mode, 0, 1200, main.d_out)),
                   cells_per_read=float(main.N) * main.M)
        out["motifseq"] = {"by_reads_per_call": rows, "fit": fit(rows)}
        out["predicted_strong_scaling"] = {
            "what": "C4 (%d reads in total) over N GPUs from one GPU's rate at %d / N reads per call; gather and ingest not included" % (main.R, main.R),
            "efficiency": {str(main.R // r["reads_per_call"]): r["vs_full_batch_rate"] for r in rows},
            "reads_per_s": {str(main.R // r["reads_per_call"]): r["reads_per_s"] * (main.R // r["reads_per_call"]) for r in rows}}
    return out


def sweep_segmenter(L, w):
    """the segmenter leg of sweep_block, on a resident segmenter workload"""
    from squigglekit_amd._lib import check
    sizes = [w.R // d for d in (1, 2, 4, 8, 16, 32) if w.R // d >= 4096]
    rows = []
    for R in sizes:
        ts = []
        for _ in range(6):
            t0 = time.perf_counter()
            check(L.sk_segment_dev_i16(w.d_sig, w.stride, w.d_len, R, C.byref(w.sp), w.d_segs, w.d_out, MAX_SEGS))
            check(L.sk_sync())
            ts.append(time.perf_counter() - t0)
        t = min(ts[2:])
        rows.append({"reads_per_call": R, "ms": t * 1e3, "reads_per_s": R / t,
                     "hbm_frac": R * (2 * w.M + 4 + 16) / t / 1e9 / HBM_PEAK_GBS})
    for r in rows:
        r["vs_full_batch_rate"] = r["reads_per_s"] / rows[0]["reads_per_s"]
    x = np.array([r["reads_per_call"] for r in rows], dtype=np.float64)
    y = np.array([r["ms"] for r in rows], dtype=np.float64)
    b, c = np.polyfit(x, y, 1)
    return {"by_reads_per_call": rows, "fit": {"fixed_ms_per_call": float(c), "us_per_1000_reads": float(b * 1e6)}}


PA_OFFSET, PA_RANGE, PA_DIGITISATION = 16.0, 1493.94, 8192.0      # channel constants of the pA image (as tests/test_gpu_f64.py)


def other_paths_block(a, L, main):
    """N = 1 extras: the paths the headline does not take, each device resident, kernels only, with a roofline and a
    parity sample against the oracle.
      segmenter_f64_pA   -- the segmenter on float64 pA reads (segmenter.py:198-201; the values SquigglePull.py:183-189
                            writes for the batch's reads: np.round((raw + offset) * range / digitisation, 2))
      motifseq_f64_medmad-- MotifSeq on the same float64 reads (MotifSeq.py:270 parses every sample as float)
      motifseq_i16_zscale-- MotifSeq -l zscale on the C4 batch (MotifSeq.py:186-191,275-280)
      motifseq_multi_k4  -- four motifs against the C4 batch (the `for name in m_order` loop, MotifSeq.py:436)"""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as ora
    from squigglekit_amd import synth
    from squigglekit_amd._lib import HIT_DTYPE, SegParams, check, ptr
    out = {"note": "HBM resident, kernels only (as the headline): wall clock of the best of 3 steps after a warm-up, "
                   "HIP-event kernel times beside it"}
    M, N = main.M, main.N
    T = max(1, min(32, os.cpu_count() or 1))
    nominal = WAVE_ISSUE_SLOTS / 8.0

    def best_of(fn, n=3):
        fn()
        check(L.sk_sync())
        ts, ev = [], None
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            check(L.sk_sync())
            ts.append(time.perf_counter() - t0)
            if ts[-1] == min(ts):
                ev = main.kernel_ms()
        return min(ts), ev

    def dtw_view(R, cells_per_read, secs):
        ach = R * cells_per_read / secs
        return {"bound": "valu_issue", "achieved": ach / 1e12, "unit": "T cell-updates/s",
                "peak_at_2.4_ghz": nominal / 1e12, "frac_at_2.4_ghz": ach / nominal,
                "note": "whole step against the screening pass's 8-issue-cycles-per-cell roof (DESIGN.md 4.3)"}

    bufs = []

    def alloc(nbytes):
        q = L.sk_dev_alloc(nbytes)
        if not q:
            check(-4)
        bufs.append(q)
        return q

    try:
        # ---------------- float64 pA reads: segmenter and MotifSeq --------------------------------------------
        Rf = min(main.R, 500_000)                                    # (16 GB: enough wavefronts for the lane-per-read walk to fill the chip)
        Mf = M - 1                                                    # segmenter.py:207 with the default -n: sig[:-1]
        total = Rf * Mf
        d_pa, d_off = alloc(total * 8), alloc((Rf + 1) * 8)
        check(L.sk_synth_pa_dev(main.d_sig, main.stride, Rf, Mf, PA_OFFSET, PA_RANGE, PA_DIGITISATION, d_pa, d_off))
        d_segs, d_nsegs = alloc(Rf * MAX_SEGS * 2 * 4), alloc(Rf * 4)
        sp = SegParams()
        secs, ev = best_of(lambda: check(L.sk_segment_dev_f64(d_pa, d_off, Rf, total, Mf, C.byref(sp), d_segs, d_nsegs,
                                                              MAX_SEGS)))
        retried = L.sk_last_f64_retries()
        rows = strided_rows(Rf, 512)
        pa = download_rows(L, d_pa, Mf * 8, rows, np.float64, Mf)
        segs = np.empty((Rf, MAX_SEGS, 2), dtype=np.int32)
        nsegs = np.empty(Rf, dtype=np.int32)
        check(L.sk_dev_download(ptr(segs), d_segs, segs.nbytes))
        check(L.sk_dev_download(ptr(nsegs), d_nsegs, nsegs.nbytes))
        op = ora.SegParams(sp.error, sp.corrector, sp.window, sp.seg_dist, sp.std_scale, sp.stall_len)

        def seg_ok(k):
            want = ora.get_segs(ora.scale_outliers(pa[k], sp.lim_low, sp.lim_hi), op) or []
            r = rows[k]
            return nsegs[r] == len(want) and segs[r, :nsegs[r]].tolist() == want
        with ThreadPoolExecutor(T) as ex:
            ok = all(ex.map(seg_ok, range(len(rows))))
        alg = Rf * (8 * Mf + 4 + 16)
        out["segmenter_f64_pA"] = {
            "workload": "%d reads x %d float64 pA samples (2 decimals), default flags" % (Rf, Mf),
            "value": Rf / secs, "unit": "reads/s", "ms_per_step": secs * 1e3,
            "kernel_ms": {"statistics": ev[0], "walk": ev[1]}, "reads_redone_in_numpy_order": int(retried),
            "roofline": {"bound": "hbm", "achieved": alg / secs / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg / secs / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_step": alg,
                         "statistics_kernel_frac": (alg / (ev[0] * 1e-3) / 1e9 / HBM_PEAK_GBS) if ev[0] > 0 else None},
            "parity": {"reads_checked": int(len(rows)), "segments_bit_exact": bool(ok),
                       "segments_in_sample": int(nsegs[rows].sum())}}

        # ---------------- the same at real read lengths: 20 000 samples (C5-shaped) and 36 977 (the one measured read the
        # reference ships, example/slow5/0.blow5) -- segmenter AND MotifSeq: `MotifSeq.py --signal` parses every sample as a
        # float (MotifSeq.py:270), so this, not the int16 headline, is what the reference's default input looks like
        import gzip
        with gzip.open(os.path.join(ROOT, "tests", "golden", "motifseq_cli.json.gz"), "rt") as fh:
            model163 = np.array(json.load(fh)["model_expanded"]["values"], dtype=np.float64)
        for RL, ML, tag in ((50_000, 20_000, "20k"), (25_000, 36_978, "37k")):
            MLs = (ML + 7) // 8 * 8
            d_raw_l = alloc(RL * MLs * 2)
            check(L.sk_synth_squiggles_dev(d_raw_l, MLs, RL, ML, synth.SEED_C5, None, 0))
            MLf = ML - 1
            d_pa_l, d_off_l = alloc(RL * MLf * 8), alloc((RL + 1) * 8)
            check(L.sk_synth_pa_dev(d_raw_l, MLs, RL, MLf, PA_OFFSET, PA_RANGE, PA_DIGITISATION, d_pa_l, d_off_l))
            MAXS_L = 128
            d_segs_l, d_nsegs_l = alloc(RL * MAXS_L * 2 * 4), alloc(RL * 4)
            secs, ev = best_of(lambda: check(L.sk_segment_dev_f64(d_pa_l, d_off_l, RL, RL * MLf, MLf, C.byref(sp), d_segs_l,
                                                                  d_nsegs_l, MAXS_L)))
            retried_l = int(L.sk_last_f64_retries())
            rows_l = strided_rows(RL, 128)
            pa_l = download_rows(L, d_pa_l, MLf * 8, rows_l, np.float64, MLf)
            segs_l = np.empty((RL, MAXS_L, 2), dtype=np.int32)
            nsegs_l = np.empty(RL, dtype=np.int32)
            check(L.sk_dev_download(ptr(segs_l), d_segs_l, segs_l.nbytes))
            check(L.sk_dev_download(ptr(nsegs_l), d_nsegs_l, nsegs_l.nbytes))

            def seg_ok_l(k):
                want = ora.get_segs(ora.scale_outliers(pa_l[k], sp.lim_low, sp.lim_hi), op) or []
                r = rows_l[k]
                return nsegs_l[r] == len(want) and segs_l[r, :nsegs_l[r]].tolist() == want
            with ThreadPoolExecutor(T) as ex:
                ok_l = all(ex.map(seg_ok_l, range(len(rows_l))))
            alg = RL * (8 * MLf + 4 + 16)
            out["segmenter_f64_pA_%s" % tag] = {
                "workload": "%d reads x %d float64 pA samples (2 decimals), default flags" % (RL, MLf),
                "value": RL / secs, "unit": "reads/s", "ms_per_step": secs * 1e3,
                "kernel_ms": {"statistics": ev[0], "walk": ev[1]}, "reads_redone_in_numpy_order": retried_l,
                "roofline": {"bound": "hbm", "achieved": alg / secs / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": alg / secs / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_step": alg,
                             "statistics_kernel_frac": (alg / (ev[0] * 1e-3) / 1e9 / HBM_PEAK_GBS) if ev[0] > 0 else None},
                "parity": {"reads_checked": int(len(rows_l)), "segments_bit_exact": bool(ok_l),
                           "segments_in_sample": int(nsegs_l[rows_l].sum())}}
            # MotifSeq, float64 medmad, against the example model (163 points)
            d_hits_l = alloc(RL * HIT_BYTES)
            secs, ev = best_of(lambda: check(L.sk_motifseq_dev_f64(d_pa_l, d_off_l, RL, RL * MLf, MLf, ptr(model163),
                                                                   model163.size, 0, 0, 1200, d_hits_l)))
            g = (C.c_int32 * 8)()
            check(L.sk_last_dtw_guard(g))
            hits_l = np.empty(RL, dtype=HIT_DTYPE)
            check(L.sk_dev_download(ptr(hits_l), d_hits_l, hits_l.nbytes))
            rows_m = rows_l[::4]
            pa_m = pa_l[::4]

            def want_long(k):
                y = ora.medmad(ora.scale_outliers(pa_m[k], 0, 1200))[0]
                return ora.dtw_subsequence(model163, y)
            with ThreadPoolExecutor(T) as ex:
                want_m = list(ex.map(want_long, range(len(rows_m))))
            got_m = hits_l[rows_m]
            alg = RL * (8 * MLf + HIT_BYTES)
            cells = float(model163.size) * float(np.mean(hits_l["n"]))
            out["motifseq_f64_medmad_%s" % tag] = {
                "workload": "%d reads x %d float64 pA samples vs the example model (%d points), medmad" % (RL, MLf, model163.size),
                "value": RL / secs, "unit": "reads/s", "ms_per_step": secs * 1e3,
                "kernel_ms": {"prep": ev[0], "dtw": ev[1]},
                "guard": {"premise_violations": int(g[0]), "audited_reads": int(g[1]), "audit_mismatches": int(g[2])},
                "roofline": {"bound": "hbm", "achieved": alg / secs / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": alg / secs / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_step": alg,
                             "prep_kernel_frac": (alg / (ev[0] * 1e-3) / 1e9 / HBM_PEAK_GBS) if ev[0] > 0 else None,
                             "valu": dtw_view(RL, cells, secs)},
                "parity": {"reads_checked": int(len(rows_m)),
                           "dist_bit_identical": bool(all(got_m["dist"][k] == w[0] for k, w in enumerate(want_m))),
                           "start_end_exact": bool(all((got_m["start"][k], got_m["end"][k]) == (w[1], w[2])
                                                       for k, w in enumerate(want_m)))}}
            for q in (d_raw_l, d_pa_l, d_off_l, d_segs_l, d_nsegs_l, d_hits_l):
                L.sk_dev_free(q)
                bufs.remove(q)

        # ---------------- dRNA_segmenter.py, both branches, device resident at a size that fills the chip ----------------
        # (round 4 timed 20 000 reads = 313 wavefronts on 1 024 SIMDs: one wavefront's latency.  250 000 reads, 15 GB.)
        from squigglekit_amd import api
        from squigglekit_amd._lib import DrnaParams, RollParams
        # dRNA-shaped reads (synth.drna_reads: adapter stretch, poly(A) plateau, body; 6 000 .. 30 000 samples): 1 000
        # distinct ones, tiled -- the scans stop where the script stops ("adapter found"), which generic squiggles never reach
        base_reads = synth.drna_reads(1000, synth.SEED_C5 + 7, min_len=6000, max_len=30000)
        RD, MD = 250_000, 30_000
        NB_ = len(base_reads)
        host_b = api.pinned_empty((NB_, MD), np.int16)
        host_b[:] = 0
        lens_b = np.zeros(NB_, dtype=np.int32)
        for r, x in enumerate(base_reads):
            host_b[r, :x.size] = x
            lens_b[r] = x.size
        d_sig_d, d_len_d = alloc(RD * MD * 2), alloc(RD * 4)
        lens_d = np.tile(lens_b, RD // NB_)
        check(L.sk_dev_upload(d_len_d, ptr(lens_d), lens_d.nbytes))
        base_p = C.cast(d_sig_d, C.c_void_p).value
        for k in range(RD // NB_):                                    # the 1 000 distinct reads, 250 times
            check(L.sk_dev_upload(C.c_void_p(base_p + k * NB_ * MD * 2), ptr(host_b), host_b.nbytes))
        dp, rp = DrnaParams(), RollParams()
        d_dsegs, d_dn = alloc(RD * 32 * 2 * 4), alloc(RD * 4)
        d_xy, d_found = alloc(RD * 2 * 4), alloc(RD * 4)
        secs, ev = best_of(lambda: check(L.sk_drna_segment_dev_i16(d_sig_d, MD, d_len_d, RD, C.byref(dp), d_dsegs, d_dn, 32)))
        dsegs = np.zeros((RD, 32, 2), dtype=np.int32)
        dn = np.zeros(RD, dtype=np.int32)
        check(L.sk_dev_download(ptr(dsegs), d_dsegs, dsegs.nbytes))
        check(L.sk_dev_download(ptr(dn), d_dn, dn.nbytes))
        rows_d = strided_rows(RD, 96)
        odp = ora.DrnaParams()

        def drna_ok(r):
            b = r % NB_
            want = ora.drna_segs(ora.scale_outliers(host_b[b, :lens_b[b]].astype(float), dp.lim_low, dp.lim_hi), odp)[0]
            return dsegs[r, :dn[r]].tolist() == want
        with ThreadPoolExecutor(T) as ex:
            ok1 = all(ex.map(drna_ok, rows_d))
        kms = ev[0] + ev[1]
        alg = int(2 * lens_d.astype(np.int64).sum() + RD * 12)
        out["drna_slow5_branch"] = {
            "workload": "%d dRNA-shaped reads of 6 000 .. 30 000 int16 samples (mean %d), dRNA_segmenter.py:85-176 constants; "
                        "device resident" % (RD, int(lens_d.mean())),
            "value": RD / secs, "unit": "reads/s", "ms_per_step": secs * 1e3,
            "kernel_ms": {"statistics": ev[0], "scan": ev[1]},
            "roofline": {"bound": "hbm", "achieved": alg / secs / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg / secs / 1e9 / HBM_PEAK_GBS, "kernels_only_frac": alg / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_step": alg},
            "parity": {"reads_checked": int(len(rows_d)), "segments_bit_exact": bool(ok1)}}
        secs, ev = best_of(lambda: check(L.sk_drna_roll_dev_i16(d_sig_d, MD, d_len_d, RD, C.byref(rp), d_xy, d_found)))
        xy = np.zeros((RD, 2), dtype=np.int32)
        found = np.zeros(RD, dtype=np.int32)
        check(L.sk_dev_download(ptr(xy), d_xy, xy.nbytes))
        check(L.sk_dev_download(ptr(found), d_found, found.nbytes))
        orp = ora.RollParams()

        def roll_ok(r):
            b = r % NB_
            want = ora.drna_roll(ora.scale_outliers(host_b[b, :lens_b[b]].astype(float), rp.lim_low, rp.lim_hi), orp)
            got = (int(xy[r, 0]), int(xy[r, 1])) if found[r] else None
            return got == want
        with ThreadPoolExecutor(T) as ex:
            ok2 = all(ex.map(roll_ok, rows_d))
        kms = ev[0] + ev[1]
        out["drna_rolling_mean_branch"] = {
            "workload": "%d dRNA-shaped reads (mean %d samples), dRNA_segmenter.py:272-326, w = 2000; device resident"
                        % (RD, int(lens_d.mean())),
            "value": RD / secs, "unit": "reads/s", "ms_per_step": secs * 1e3,
            "kernel_ms": {"filter_prefix_sums_statistics": ev[0], "scan": ev[1]},
            "roofline": {"bound": "hbm", "achieved": alg / secs / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg / secs / 1e9 / HBM_PEAK_GBS, "kernels_only_frac": alg / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_step": alg},
            "parity": {"reads_checked": int(len(rows_d)), "pairs_exact": bool(ok2), "found_in_sample": int(found[rows_d].sum())}}
        del host_b
        for q in (d_sig_d, d_len_d, d_dsegs, d_dn, d_xy, d_found):
            L.sk_dev_free(q)
            bufs.remove(q)

        d_hits = alloc(max(Rf, main.R) * HIT_BYTES * 4)

        def hits_ok(rows_, got, want_fn):
            with ThreadPoolExecutor(T) as ex:
                want = list(ex.map(want_fn, range(len(rows_))))
            d_ok = all(got["dist"][k] == w[0] for k, w in enumerate(want))
            se_ok = all((got["start"][k], got["end"][k]) == (w[1], w[2]) for k, w in enumerate(want))
            return {"reads_checked": int(len(rows_)), "dist_bit_identical": bool(d_ok), "start_end_exact": bool(se_ok)}

        secs, ev = best_of(lambda: check(L.sk_motifseq_dev_f64(d_pa, d_off, Rf, total, Mf, ptr(main.motif), N, 0, 0, 1200,
                                                               d_hits)))
        hits = np.empty(Rf, dtype=HIT_DTYPE)
        check(L.sk_dev_download(ptr(hits), d_hits, hits.nbytes))
        rows2 = rows[::4]
        pa2 = pa[::4]

        def want_f64(k):
            y = ora.medmad(ora.scale_outliers(pa2[k], 0, 1200))[0]
            return ora.dtw_subsequence(main.motif, y)
        alg = Rf * (8 * Mf + HIT_BYTES)
        out["motifseq_f64_medmad"] = {
            "workload": "%d reads x %d float64 pA samples vs %d-pt motif, medmad" % (Rf, Mf, N),
            "value": Rf / secs, "unit": "reads/s", "ms_per_step": secs * 1e3,
            "kernel_ms": {"prep": ev[0], "dtw": ev[1]},
            "roofline": {"bound": "hbm", "achieved": alg / secs / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg / secs / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_step": alg,
                         "prep_kernel_frac": (alg / (ev[0] * 1e-3) / 1e9 / HBM_PEAK_GBS) if ev[0] > 0 else None,
                         "valu": dtw_view(Rf, float(N) * float(np.mean(hits["n"])), secs)},
            "parity": hits_ok(rows2, hits[rows2], want_f64)}
        for q in (d_pa, d_off, d_segs, d_nsegs):
            L.sk_dev_free(q)
            bufs.remove(q)

        # ---------------- int16, zscale ------------------------------------------------------------------------
        R = main.R
        secs, ev = best_of(lambda: check(L.sk_motifseq_dev_i16(main.d_sig, main.stride, main.d_len, R, ptr(main.motif), N,
                                                               1, 0, 1200, d_hits)))
        hits = np.empty(R, dtype=HIT_DTYPE)
        check(L.sk_dev_download(ptr(hits), d_hits, hits.nbytes))
        rows3 = strided_rows(R, 256)
        sample = download_rows(L, main.d_sig, main.stride * 2, rows3, np.int16, main.stride)
        parts = [(i, min(len(rows3), i + 8)) for i in range(0, len(rows3), 8)]

        def ora_i16(motif, mode):
            with ThreadPoolExecutor(T) as ex:
                return np.concatenate(list(ex.map(lambda ab: ora.motifseq_batch_i16(
                    sample[ab[0]:ab[1]], main.lens[rows3[ab[0]:ab[1]]], motif, scale_mode=mode), parts)))
        want = ora_i16(main.motif, 1)
        got = hits[rows3]
        alg = R * (2 * M + HIT_BYTES)
        out["motifseq_i16_zscale"] = {
            "workload": workload_name("motifseq", R, M, N, "weak", "zscale"),
            "value": R / secs, "unit": "reads/s", "ms_per_step": secs * 1e3,
            "kernel_ms": {"prep": ev[0], "dtw": ev[1]},
            "roofline": {"bound": "hbm", "achieved": alg / secs / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg / secs / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_step": alg,
                         "prep_kernel_frac": (alg / (ev[0] * 1e-3) / 1e9 / HBM_PEAK_GBS) if ev[0] > 0 else None,
                         "valu": dtw_view(R, float(N) * float(np.mean(hits["n"])), secs)},
            "parity": {"reads_checked": int(len(rows3)), "dist_bit_identical": bool(np.array_equal(got["dist"], want["dist"])),
                       "start_end_exact": bool(np.array_equal(got["start"], want["start"])
                                               and np.array_equal(got["end"], want["end"]))}}

        # ---------------- int16, medmad, four motifs ------------------------------------------------------------
        motifs = [main.motif] + [synth.synthetic_motif(n, seed=sd) for n, sd in ((N, 11), (max(8, N - 37), 12), (N + 40, 13))]
        flat = np.concatenate(motifs)
        moff = np.concatenate([[0], np.cumsum([m.size for m in motifs])]).astype(np.int32)
        K = len(motifs)
        secs, ev = best_of(lambda: check(L.sk_motifseq_multi_dev_i16(main.d_sig, main.stride, main.d_len, R, ptr(flat),
                                                                     ptr(moff), K, 0, 0, 1200, d_hits)))
        hits = np.empty((K, R), dtype=HIT_DTYPE)
        check(L.sk_dev_download(ptr(hits), d_hits, hits.nbytes))
        ok_d = ok_se = True
        for k in range(K):
            want = ora_i16(motifs[k], 0)
            got = hits[k][rows3]
            ok_d = ok_d and np.array_equal(got["dist"], want["dist"])
            ok_se = ok_se and np.array_equal(got["start"], want["start"]) and np.array_equal(got["end"], want["end"])
        cells = float(np.mean(hits[0]["n"])) * float(flat.size)
        out["motifseq_multi_k4"] = {
            "workload": "%d reads x %d int16 samples vs %d motifs (%s points), medmad" % (
                R, M, K, ", ".join(str(m.size) for m in motifs)),
            "value": R / secs, "unit": "reads/s", "read_motif_pairs_per_s": R * K / secs, "ms_per_step": secs * 1e3,
            "kernel_ms": {"prep": ev[0], "dtw_last_motif": ev[1]},
            "roofline": {"bound": "valu_issue", **dtw_view(R, cells, secs),
                         "hbm_frac": R * (2 * M + K * HIT_BYTES) / secs / 1e9 / HBM_PEAK_GBS},
            "parity": {"reads_checked": int(len(rows3)) * K, "dist_bit_identical": bool(ok_d), "start_end_exact": bool(ok_se)}}
    except Exception as e:                                            # noqa: BLE001 -- report, keep the line
        import traceback
        out["error"] = repr(e) + " | " + traceback.format_exc(limit=2).replace("\n", " / ")
    finally:
        for q in bufs:
            L.sk_dev_free(q)
    return out



def _kernels_sha():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        from kernels_sha import kernels_sha
        return kernels_sha(ROOT)
    except Exception:                                                 # noqa: BLE001 -- tools/ not shipped: no stamp
        return None


def traffic_from_profiles(workload, pattern):
    """HBM bytes per read of one kernel from the committed PMC passes (FETCH_SIZE / WRITE_SIZE, collected
    separately with rocprofv3 --pmc: a bench run cannot read hardware counters itself)."""
    tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % workload)
    try:
        tj = json.load(open(tpath))
        stamp = tj.get("kernels_sha")
        stale = " [STALE: measured on kernel sources %s, running %s]" % (stamp or "unstamped", _kernels_sha()) \
            if stamp != _kernels_sha() else ""
        if pattern is None:                                          # every kernel of the step together
            tot = sum(kk["fetch_bytes_total"] + kk["write_bytes_total"] for kk in tj["kernels"].values())
            per_read = tot / (tj["reads_per_call"] * max(1, tj.get("calls", 1)))
            return per_read, "profiles/traffic_%s.json: all kernels, %.0f B/read measured%s" % (workload, per_read, stale)
        key = [k for k in tj["kernels"] if pattern in k]
        if key and tj.get("reads_per_call"):
            kk = tj["kernels"][key[0]]
            per_read = (kk["fetch_bytes_total"] + kk["write_bytes_total"]) / (
                tj["reads_per_call"] * max(1, tj.get("calls", 1)))
            return per_read, "profiles/traffic_%s.json: %s, %.0f B/read measured%s" % (workload, key[0], per_read, stale)
    except Exception:
        pass
    return None, None


def motifseq_roofline(a, w, prof, steps, mean_n):
    R, M, N = w.R, w.M, w.N
    prep_ms, main_ms = prof["prep_ms"] / steps, prof["main_ms"] / steps
    cells = float(N) * mean_n
    alg_bytes = R * (2 * M + HIT_BYTES)                       # 2*M in + 24 out per read
    dominant, dom_ms = "k_sdtw (all launches of one call)", main_ms
    valu = {}
    if prof["launches"] > 0:
        # dominant kernel = the fixed-point screening pass k_sdtw_q<L,R,feed>; one launch per chunk
        per_step = prof["launches"] / steps
        Lg = int(os.environ.get("SK_DTW_QL", 0)) or (8 if (N <= 256 and R >= 49152) else 16 if N <= 512 else 64)   # sk_sdtwq.hip screen_layout
        Rg = (N + Lg - 1) // Lg
        dominant = "k_sdtw_q<%d,%d,0> (screening pass%s, %d launches per call)" % (
            Lg, Rg, " with the filter + medmad prologue" if prep_ms < 0.05 else "", per_step)
        dom_ms = prof["dist_ms"] / prof["launches"]
        alg_bytes = alg_bytes / per_step                      # algorithmic bytes one launch covers
        # Its roof: 2 VALU instructions per cell, v_min3_u32 + v_sad_u32, 4 shader cycles of issue each on one SIMD
        # (tools/ubench/valu_rate, profiles/r03_valu_rate.txt: cycles counted with s_memtime -- 4.30 at 4 waves per
        # SIMD, 4.15 at 8; v_add_f32 / v_add_u32 / v_and / v_mov: 2.1; v_add_f64 4.1) => at most SIMDs x 64 lanes / 8
        # cycles x clock cell-updates per second.  The clock is the one the pass ran at: its first wave counts shader
        # cycles against the 100 MHz reference (sk_last_dtw_clock); the nominal 2.4 GHz figure is given beside it.
        ghz = C.c_double(0.0)
        w.L.sk_last_dtw_clock(C.byref(ghz))
        clk = ghz.value if 0.5 < ghz.value < 3.0 else None
        q_ach = R * cells / (prof["dist_ms"] / steps * 1e-3)
        step_ach = R * cells / ((prep_ms + main_ms) * 1e-3)
        nominal = WAVE_ISSUE_SLOTS / 8.0                      # cell-updates/s at 2.4 GHz
        at_clk = nominal * (clk / 2.4) if clk else None
        valu["screening_pass"] = {
            "bound": "valu_issue", "achieved": q_ach / 1e12, "unit": "T cell-updates/s",
            "issue_cycles_per_cell": 8, "measured_clock_ghz": clk,
            "peak_at_measured_clock": at_clk / 1e12 if at_clk else None,
            "frac": q_ach / at_clk if at_clk else None,
            "peak_at_2.4_ghz": nominal / 1e12, "frac_at_2.4_ghz": q_ach / nominal,
            "note": "the pass also carries the filter + medmad prologue of its reads" if prep_ms < 0.05 else None}
        valu["whole_step"] = {"bound": "valu_issue", "achieved": step_ach / 1e12, "unit": "T cell-updates/s",
                              "peak_at_measured_clock": at_clk / 1e12 if at_clk else None,
                              "frac": step_ach / at_clk if at_clk else None,
                              "frac_at_2.4_ghz": step_ach / nominal,
                              "note": "screening (+ prologue) + pre-roll + certified window + retries, against the "
                                      "screening pass's own roof"}
        valu["passes_ms_per_call"] = {"prep": prep_ms, "screen": prof["dist_ms"] / steps,
                                      "window": prof["start_ms"] / steps,
                                      "retried_reads": prof["retries"] / steps,
                                      "second_tier_reads": int(w.L.sk_last_dtw_tier2())}
    f64_roof = VALU_F64_LANEOPS / 4.0 / cells                 # reads/s of the reference's 4-f64-op cell at full rate
    valu["exact_f64_recurrence_roof_reads_per_s"] = f64_roof
    valu["speed_vs_exact_f64_roof"] = (R / ((prep_ms + main_ms) * 1e-3)) / f64_roof
    valu["note"] = ("speed_vs_exact_f64_roof is a ratio, not a fraction of peak: 95 % of the cells are evaluated "
                    "in 32-bit fixed point (2 integer ops), only the certified window in f64")
    per_read, src = traffic_from_profiles("motifseq", "k_sdtw_q")
    traffic = per_read * (alg_bytes / (2 * M + HIT_BYTES)) if per_read else None
    step_read, step_src = traffic_from_profiles("motifseq", None)
    achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src,
            "traffic_stale": bool(src and "STALE" in src),
            "traffic_ratio": (traffic / alg_bytes) if traffic else None,
            "traffic_ratio_whole_step": (step_read / (2 * M + HIT_BYTES)) if step_read else None,
            "kernel_ms": {"prep": prep_ms, "main": main_ms, "dominant_avg_launch": dom_ms},
            "algorithmic_bytes_per_launch": alg_bytes,
            "dominant_kernel_dtype": "u32 fixed-point screening, f64 certified window",
            "binding": "valu issue rate (min-plus recurrence; HBM is not the limiter, DESIGN.md 4.3)",
            "valu": valu}


def segmenter_isolated(w, steps=3):
    """Kernel times with the two kernels run one after the other (SK_SEG_CHUNKS=1, the default): with more chunks
    the walk of one runs beside the statistics of the next, and a per-kernel roofline cannot be read off."""
    had = os.environ.get("SK_SEG_CHUNKS")
    os.environ["SK_SEG_CHUNKS"] = "1"
    try:
        _, prof = timed(w, None, steps, 1)
    finally:
        if had is None:
            del os.environ["SK_SEG_CHUNKS"]
        else:
            os.environ["SK_SEG_CHUNKS"] = had
    return prof, steps


def seg_kernel_prof(w, prof, steps):
    """The timed steps' HIP-event sums, unless the two kernels overlapped in them (SK_SEG_CHUNKS > 1; the default is one
    chunk since the jumping walk, round 4): then a short run with the kernels one after the other."""
    chunks = os.environ.get("SK_SEG_CHUNKS") or "1"
    if chunks == "1" or w.R < 65536:
        return prof, steps
    return segmenter_isolated(w)


def segmenter_roofline(w, prof, steps, step_ms=None):
    R, M = w.R, w.M
    prep_ms, main_ms = prof["prep_ms"] / steps, prof["main_ms"] / steps
    alg_bytes = R * (2 * M + 4 + 8 * 2)
    dominant, dom_ms = ("k_seg_stats (filter + statistics + in-band / kept masks)", prep_ms) if prep_ms >= main_ms \
        else ("k_seg_walk4 (run-hopping, jumping get_segs walk)", main_ms)
    per_read, src = traffic_from_profiles("segmenter", "k_seg_stats" if prep_ms >= main_ms else "k_seg_walk")
    achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
    both = alg_bytes / ((prep_ms + main_ms) * 1e-3) / 1e9
    step_ms = step_ms if step_ms else prep_ms + main_ms
    whole = alg_bytes / (step_ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": per_read * R if per_read else None, "traffic_source": src,
            "traffic_stale": bool(src and "STALE" in src),
            "traffic_ratio": (per_read * R / alg_bytes) if per_read else None,
            "kernel_ms": {"prep": prep_ms, "main": main_ms, "dominant_avg_launch": dom_ms},
            "algorithmic_bytes_per_launch": alg_bytes,
            "both_kernels": {"achieved": both, "frac": both / HBM_PEAK_GBS,
                             "note": "the two kernels run one after the other (SK_SEG_CHUNKS=1), HIP-event times added"},
            "whole_step": {"achieved": whole, "frac": whole / HBM_PEAK_GBS, "ms": step_ms,
                           "note": "as shipped (statistics kernel, then the walk); algorithmic bytes / wall time of the "
                                   "timed step"}}


def extras_single_gpu(a, L, main):
    """N = 1 only, after the timed region: the secondary (segmenter) line, the exact-only DTW schemes and the
    PCIe-inclusive host-buffer rate.  Each is a few short steps."""
    from squigglekit_amd._lib import HIT_DTYPE, check, ptr
    out = {}
    # ---- secondary metric: segmenter reads/s on 1 M x 4 000 ------------------------------------
    if a.workload == "motifseq":
        sa = argparse.Namespace(**vars(a))
        sa.workload = "segmenter"
        w = Workload(sa, L, 0, 1, a.reads, workload="segmenter")
        try:
            # (10 steps behind 3 untimed ones: a pass is 2.4 ms, and the first two after the buffers are allocated run
            # 5-10 % slower than the rest)
            el, prof = timed(w, None, 10, 3)
            par, cpu, _ = parity_and_cpu(sa, w, True)
            out["secondary"] = {"metric": "reads/sec segmenter (4k-sample read)", "value": w.R * 10 / el,
                                "unit": "reads/s", "ms_per_step": el / 10 * 1e3, "steps": 10, "warmup": 3,
                                "config": {"workload": workload_name("segmenter", w.R, w.M, None, "weak"),
                                           "seed": w.seed},
                                "roofline": segmenter_roofline(w, *seg_kernel_prof(w, prof, 10), step_ms=el / 10 * 1e3),
                                "cpu_baseline": cpu,
                                "parity": par}
            out["secondary"]["sweep"] = sweep_segmenter(L, w)
        finally:
            w.free()
    # ---- what the screening buys: the exact-only schemes on 200 000 of the same reads ------------------------
    if a.workload == "motifseq":
        ex = {}
        Rx = min(main.R, 200_000)
        for name, env in (("full_single_pass", "full"), ("exact_two_pass", "exact2")):
            os.environ["SK_DTW_SCHEME"] = env
            try:
                t = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    check(L.sk_motifseq_dev_i16(main.d_sig, main.stride, main.d_len, Rx, ptr(main.motif), main.N,
                                                main.mode, 0, 1200, main.d_out))
                    check(L.sk_sync())
                    t.append(time.perf_counter() - t0)
                ex[name] = Rx / min(t[1:])
            finally:
                del os.environ["SK_DTW_SCHEME"]
        ex["reads"] = Rx
        out["exact_only_reads_per_s"] = ex
    # ---- end to end: host buffers in, host records out, PCIe included --------------------------------------------
    # (sub-batches: the H2D copy of one runs under the kernels of the previous one -- csrc/sk_api.hip)
    from squigglekit_amd import api
    from squigglekit_amd._lib import SegParams
    Rh = min(main.R, 400_000)
    e2e = {"reads": Rh, "note": "sk_*_batch_i16 on host arrays: H2D + kernels + D2H, wall clock, best of 3 after a "
                                "warm-up call; pageable = ordinary numpy memory, pinned = api.pinned_empty() "
                                "(sk_host_alloc); PCIe Gen5 x16 ceiling ~63 GB/s = 7.9 M reads/s at 8 KB per read"}
    lens = main.lens[:Rh]
    lens_s = (lens - 1).astype(np.int32)
    hits = np.zeros(Rh, dtype=HIT_DTYPE)
    segs = np.zeros((Rh, MAX_SEGS, 2), dtype=np.int32)
    nsegs = np.zeros(Rh, dtype=np.int32)
    sp = SegParams()
    for kind in ("pageable", "pinned"):
        host = (np.empty((Rh, main.stride), dtype=np.int16) if kind == "pageable"
                else api.pinned_empty((Rh, main.stride), np.int16))
        check(L.sk_dev_download(ptr(host), main.d_sig, host.nbytes))
        ts = []
        for _ in range(4):
            t0 = time.perf_counter()
            check(L.sk_motifseq_batch_i16(ptr(host), main.stride, ptr(lens), Rh, ptr(main.motif), main.N, main.mode,
                                          0, 1200, ptr(hits)))
            ts.append(time.perf_counter() - t0)
        e2e["motifseq_%s_reads_per_s" % kind] = Rh / min(ts[1:])
        ts = []
        for _ in range(4):
            t0 = time.perf_counter()
            check(L.sk_segment_batch_i16(ptr(host), main.stride, ptr(lens_s), Rh, C.byref(sp), ptr(segs), ptr(nsegs),
                                         MAX_SEGS))
            ts.append(time.perf_counter() - t0)
        e2e["segmenter_%s_reads_per_s" % kind] = Rh / min(ts[1:])
        del host
    out["end_to_end"] = e2e
    return out


# ----------------------------------------------------------------------------------------------------
# one rank
# ----------------------------------------------------------------------------------------------------
def workload_name(kind, reads, samples, motif, scaling, scale="medmad"):
    """BASELINE.json's config label when the sizes are one of its configs, "custom" otherwise."""
    per = "per GPU" if scaling == "weak" else "in total"
    if kind == "motifseq":
        tag = {(1_000_000, 4000, 200): "C4", (10_000, 4000, 163): "C3", (100_000, 20_000, 500): "C5"}.get(
            (reads, samples, motif), "custom")
        return "MotifSeq %s: %d reads x %d int16 samples %s, %d-pt motif, %s" % (tag, reads, samples, per, motif, scale)
    tag = {(10_000, 4000): "C2", (1_000_000, 4000): "C2-1M"}.get((reads, samples), "custom")
    return "segmenter %s: %d reads x %d int16 samples %s, default flags" % (tag, reads, samples, per)



def rank_body(a, comm, rank, world, shape):
    """Runs on the rank's own thread / process with its device bound.  Returns the JSON line on rank 0."""
    from squigglekit_amd import _lib, sharding
    L = _lib.load()
    # Every rank keeps a.reads reads resident (the weak-scaling shard); the strong-scaling job -- a.reads in TOTAL,
    # block-sharded -- runs on the first hi - lo of them, so both curves come out of one launch.
    lo, hi = sharding.shard_bounds(a.reads, rank, world)
    strong_R, strong_pad = hi - lo, max(sharding.shard_sizes(a.reads, world))
    use_comm = comm if (world > 1 or a.force_comm) else None
    R_alloc = a.reads if (a.scaling == "weak" or world > 1) else strong_R
    w = Workload(a, L, rank, world, R_alloc, comm=use_comm, gather_pad=R_alloc)

    def run(scaling, steps, warmup):
        if scaling == "strong":
            w.R, w.pad = strong_R, strong_pad
        else:
            w.R, w.pad = a.reads, a.reads
        el, pf = timed(w, use_comm, steps, warmup)
        own = float(getattr(w, "own_elapsed", el))
        per_rank = ([float(v) for v in use_comm.allgather_host(np.array([own], dtype=np.float64))]
                    if use_comm is not None else [own])
        return el, pf, [v / steps * 1e3 for v in per_rank]

    elapsed, prof, per_rank_ms = run(a.scaling, a.steps, a.warmup)
    shard_sizes = sharding.shard_sizes(a.reads, world) if a.scaling == "strong" else [a.reads] * world
    ranks_seen = use_comm.ranks_seen() if use_comm is not None else 1

    other = None
    if world > 1 and a.workload == "motifseq":
        # the other curve on the data already resident (weak: C4 on every GPU; strong: C4 as BASELINE.json words it)
        oscale = "weak" if a.scaling == "strong" else "strong"
        el_o, _, pr_o = run(oscale, a.steps, 1)
        tot = a.reads * world if oscale == "weak" else a.reads
        other = {"scaling": oscale, "total_reads": tot, "value": tot * a.steps / el_o, "unit": "reads/s",
                 "ms_per_step": el_o / a.steps * 1e3, "steps": a.steps, "per_rank_ms_per_step": pr_o}
        run(a.scaling, 1, 0)                                    # every rank: d_out holds the headline's shard again
    if a.scaling == "strong":
        w.R, w.pad = strong_R, strong_pad
    e2e_multi, h2d = None, None
    if world > 1 and a.workload == "motifseq" and not a.no_extras:
        e2e_multi = e2e_all_ranks(a, w, use_comm)               # (every rank takes part)
        h2d = [float(v) for v in use_comm.allgather_host(np.array([e2e_multi[2]], dtype=np.float64))]
    if rank != 0:
        w.free()
        return None

    total_reads = a.reads * world if a.scaling == "weak" else a.reads
    ms_per_step = elapsed / a.steps * 1e3
    value = total_reads * a.steps / elapsed
    want_cpu = world == 1 and a.cpu_seconds > 0          # the CPU baseline is timed at N = 1 only
    parity, cpu, mean_n = parity_and_cpu(a, w, want_cpu)
    if use_comm is not None:
        # what did the gather gather?  a sample of every rank's shard out of the gathered buffer, against the oracle
        gv = verify_gather(a, w, shard_sizes)
        parity.update(gv)
        for key in ("dist_bit_identical", "start_end_exact", "segments_bit_exact"):
            if key in parity:
                parity[key] = bool(parity[key] and gv["every_rank_ok"])
    if a.workload == "motifseq":
        roofline = motifseq_roofline(a, w, prof, a.steps, mean_n)
        name = "reads/sec MotifSeq DTW (4k-sample read x 200-sample motif)"
        wl = workload_name("motifseq", a.reads, a.samples, a.motif, a.scaling, a.scale)
    else:
        roofline = segmenter_roofline(w, *seg_kernel_prof(w, prof, a.steps), step_ms=ms_per_step) if use_comm is None \
            else segmenter_roofline(w, prof, a.steps)
        name = "reads/sec segmenter (4k-sample read)"
        wl = workload_name("segmenter", a.reads, a.samples, None, a.scaling)
    line = {"metric": name, "value": value, "unit": "reads/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": a.scaling,
            "vs_baseline": None, "dtype": "f64" if a.workload == "motifseq" else "int16/f64",
            "data": "synthetic",
            "config": {"workload": wl, "reads_per_gpu": w.R, "total_reads": total_reads, "samples": a.samples,
                       "motif_points": a.motif if a.workload == "motifseq" else None, "seed": w.seed,
                       "generator": "device generator k_synth (csrc/sk_synth.hip): the SURVEY 8(d) squiggle model, "
                                    "counter-based RNG -- a different stream than squigglekit_amd/synth.py's "
                                    "numpy default_rng recipe the tests use under the same seed",
                       "sharding": "reads block-sharded over %d rank(s); every step ends with an RCCL all-gather "
                                   "of the result records" % world if use_comm is not None else
                                   "1 rank, no exchange",
                       "launch": {"single": "one process, one GPU", "threads": "one process, one host thread per GPU",
                                  "process": "one process per GPU (launcher environment), torch-free"}[shape],
                       "gather_backend": use_comm.backend if use_comm is not None else None,
                       "ranks_seen": ranks_seen,
                       "oversubscribed": ("every rank on device %d (dry run of the N > 1 path)" % a.ranks_on_device)
                       if a.ranks_on_device is not None else None},
            "roofline": roofline, "cpu_baseline": cpu, "parity": parity}
    line["ranks_seen"] = ranks_seen
    line["gather_backend"] = use_comm.backend if use_comm is not None else None
    line["per_rank"] = {"ms_per_step": per_rank_ms, "h2d_GBps": h2d,
                        "note": "each rank's own wall clock of the timed steps (the line's ms_per_step is their maximum); "
                                "h2d_GBps: its host-to-device rate in the every-rank end-to-end leg -- a slow PCIe root "
                                "or a rank on the wrong NUMA node shows here"}
    if other:
        line["%s_scaling" % other["scaling"]] = other
    if e2e_multi is not None:
        Rh, dt = e2e_multi[:2]
        line["end_to_end"] = {"reads_per_gpu": Rh, "motifseq_pinned_reads_per_s": world * Rh / dt if dt else None,
                              "note": "every rank at once: pinned host arrays -> sk_motifseq_batch_i16 (H2D of one "
                                      "sub-batch under the kernels of the previous one) -> host records; one feeder "
                                      "thread / process per GPU, slowest rank's wall clock, best of 2 after a warm-up"}
    if prof.get("guard") is not None:
        line["guard"] = dict(prof["guard"], note="run-time check of the screening certificate's premise, last timed step: "
                             "0 / 0 on a healthy build (sk_last_dtw_guard; DESIGN.md 4.3)")
    # Everything beyond the contract goes out as lines of its own BEFORE the headline ({"extra": name, ...}); the
    # headline is printed last and stays a few KB, so that a tail of stdout always holds it whole (round 4's record lost
    # its `secondary` and `parity` blocks to a line of 14 KB).  --full-json PATH writes headline + extras as one object.
    extras = {}
    if world == 1 and a.only_other_paths:
        extras["other_paths"] = other_paths_block(a, L, w)
    elif world == 1 and not a.no_extras:
        extras.update(extras_single_gpu(a, L, w))
        extras["sweep"] = sweep_block(a, L, w)
        if a.workload == "motifseq" and not a.no_sensitivity:
            extras["other_paths"] = other_paths_block(a, L, w)
            extras["cli"] = cli_block(a, L, w)
            extras["sensitivity"] = sensitivity_block(a, L, w)
    elif world == 1 and a.sweep_reads:
        extras["sweep"] = sweep_block(a, L, w)
    sec = extras.get("secondary")
    if sec:
        rf = sec["roofline"]
        line["secondary"] = {"metric": sec["metric"], "value": sec["value"], "unit": "reads/s", "ms_per_step": sec["ms_per_step"],
                             "workload": sec["config"]["workload"],
                             "roofline": {"bound": "hbm", "whole_step_frac": rf["whole_step"]["frac"],
                                          "dominant_kernel": rf["kernel"], "dominant_kernel_frac": rf["frac"],
                                          "kernel_ms": rf["kernel_ms"], "traffic_ratio": rf["traffic_ratio"]},
                             "cpu_baseline_reads_per_s": (sec["cpu_baseline"] or {}).get("value"),
                             "parity": sec["parity"],
                             "predicted_strong_scaling_efficiency": {
                                 str(w.R // r["reads_per_call"]): round(r["vs_full_batch_rate"], 4)
                                 for r in sec.get("sweep", {}).get("by_reads_per_call", [])},
                             "full": "the {\"extra\": \"secondary\"} line above"}
    sw = extras.get("sweep", {})
    if sw.get("predicted_strong_scaling"):
        line["predicted_strong_scaling"] = dict(sw["predicted_strong_scaling"],
                                                fixed_ms_per_call=sw["motifseq"]["fit"]["fixed_ms_per_call"])
    line["extras"] = sorted(extras)
    w.free()
    return line, extras


def main(argv=None):
    a = parse(argv)
    os.environ["SK_TUNING"] = "1"                             # the A/B legs of the extras flip tuning switches
    os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")   # RCCL logs to stdout by default: the JSON line stands alone
    from squigglekit_amd import _lib, multigpu
    _lib.load()
    if a.ranks_on_device is not None:
        os.environ["SK_OVERSUBSCRIBE"] = "d%d" % a.ranks_on_device
    elif multigpu.oversubscribed() is not None:
        a.ranks_on_device = multigpu.oversubscribed()
    shape, rank, local, world = multigpu.plan(a.gpus)
    a.gpus = world
    if world > 1 and a.ranks_on_device is None:
        have = _lib.load().sk_device_count()
        if have < world:
            sys.stderr.write("bench.py: %d GPU(s) visible, %d ranks asked for; set SK_OVERSUBSCRIBE=1 (or --ranks-on-device D) "
                             "for a dry run of the N > 1 path on one device\n" % (have, world))
            sys.exit(2)
    if shape == "process":
        with multigpu.ProcessGroup(rank, local, world) as comm:
            line = rank_body(a, comm, rank, world, shape)
    elif shape == "threads" or a.force_comm:
        devs = list(range(world)) if a.ranks_on_device is None or world == 1 else [a.ranks_on_device] * world
        g = multigpu.ThreadGroup(devs, oversubscribe=a.ranks_on_device is not None)
        try:
            line = g.run(lambda comm: rank_body(a, comm, comm.rank, world, shape))[0]
        finally:
            g.close()
    else:
        _lib.init(0)
        line = rank_body(a, None, 0, 1, shape)
    if line is not None:
        line, extras = line
        # the headline is the last thing on stdout: RCCL prints a version banner through C stdio, which would
        # otherwise be flushed at exit, after Python's own buffer
        sys.stdout.flush()
        try:
            C.CDLL(None).fflush(None)
        except Exception:
            pass
        for key in sorted(extras):
            print(json.dumps({"extra": key, key: extras[key]}), flush=True)
        if a.full_json:
            with open(a.full_json, "w") as fh:
                json.dump(dict(line, **extras), fh)
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
