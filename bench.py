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
import bench_extras        # noqa: E402  (the N = 1 extras: cli, sensitivity, sweep, other_paths)

from bench_common import (HBM_PEAK_GBS, HIT_BYTES, MAX_SEGS, VALU_F64_LANEOPS, WAVE_ISSUE_SLOTS, download_rows,  # noqa: F401
                          strided_rows, workload_name)


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
    ap.add_argument("--allow-host-gather", action="store_true",
                    help="N > 1: do not fail the run when the gather fell back to host concatenation (RCCL missing) or a "
                         "rank is not seen -- by default such a run exits non-zero instead of printing a clean-looking line")
    ap.add_argument("--no-baseline-configs", action="store_true",
                    help="N = 1: skip the extra lines for the other BASELINE.json configs (c2_10k, c3, c5)")
    ap.add_argument("--force-comm", action="store_true",
                    help="N = 1: still create the RCCL communicator and gather every step (exercises the N > 1 path)")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------------------------------
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
                         "image_rejects": int(g[3]), "exact_fallback": int(g[5]), "second_windows": int(g[6])}
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
        Lg = int(os.environ.get("SK_DTW_QL", 0)) or (8 if (N <= 256 and R >= (65536 if M > 8192 else 49152)) else 16 if N <= 512 else 64)   # sk_sdtwq.hip screen_layout
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
        # the exact window pass against ITS roof: mlpy's recurrence with start tracking is 8 instructions per cell
        # (|x - y|, two compares, two v_min_f64, two selects, one add; six of them FP64, 4.1 issue cycles each --
        # profiles/r05_valu_rate.txt), the wavefronts' steps are counted by the kernel itself (sk_last_dtw_window_steps)
        ws = (C.c_uint64 * 2)()
        w.L.sk_last_dtw_window_steps(ws)
        win_ms = prof["start_ms"] / steps
        if ws[0] and win_ms > 0:
            G = 64 // Lg
            simd_cycles = win_ms * 1e-3 * (clk or 2.4) * 1e9 * (WAVE_ISSUE_SLOTS / 2.4e9 / 64.0)   # SIMDs x cycles in the pass
            used = float(ws[0]) * Rg * 8 * 4.0
            valu["window_pass"] = {
                "bound": "valu_issue", "instructions_per_cell": 8, "wave_steps": int(ws[0]), "read_steps_asked": int(ws[1]),
                "steps_per_read": float(ws[1]) / R, "lockstep_efficiency": float(ws[1]) / (float(ws[0]) * G),
                "frac": used / simd_cycles,
                "note": "k_sdtw_w + k_sdtw_p, all tiers: cell instructions the wavefronts issued x 4 cycles / (SIMDs x cycles of "
                        "the passes' summed HIP-event time); the pre-roll (k_sdtw_p) and the per-step overhead are not in the numerator"}
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
            out["secondary"]["sweep"] = bench_extras.sweep_segmenter(L, w)
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


def baseline_config_extra(a, L, kind, reads, samples, motif, steps, warmup, label):
    """One of BASELINE.json's other configs as a bench line of its own (N = 1): the same Workload / timed / oracle
    parity / roofline code as the headline, at that config's size.  C2: segmenter 10 000 x 4 000; C3: MotifSeq
    10 000 x 4 000 vs a ~200-pt model (163 points: the size of example/CATCTATCCAGGGTTAAATT.model expanded); C5:
    MotifSeq 100 000 x 20 000 vs a 500-pt motif (on one GPU)."""
    sa = argparse.Namespace(**vars(a))
    sa.workload, sa.reads, sa.samples, sa.motif = kind, reads, samples, motif
    w = Workload(sa, L, 0, 1, reads, workload=kind)
    try:
        el, prof = timed(w, None, steps, warmup)
        par, _, mean_n = parity_and_cpu(sa, w, False)
        par["oracle_pinned"] = ORACLE_PINNED[kind]
        ms = el / steps * 1e3
        if kind == "motifseq":
            roof = motifseq_roofline(sa, w, prof, steps, mean_n)
            roof = headline_roofline_labels(roof)
        else:
            roof = segmenter_roofline(w, prof, steps, step_ms=ms)
        return {"config": label, "metric": "reads/sec " + ("MotifSeq DTW" if kind == "motifseq" else "segmenter"),
                "value": reads * steps / el, "unit": "reads/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
                "ms_per_step": ms, "workload": workload_name(kind, reads, samples, motif if kind == "motifseq" else None, "strong",
                                                             sa.scale),
                "roofline": roof, "parity": par}
    finally:
        w.free()


def baseline_configs_block(a, L):
    out = {}
    for key, args in (("c2_10k", ("segmenter", 10_000, 4000, 200, 20, 3, "BASELINE config 2: segmenter, 10 000 synthetic int16 reads x 4 000 samples")),
                      ("c3", ("motifseq", 10_000, 4000, 163, 20, 3, "BASELINE config 3: MotifSeq, 10 000 reads x 4 000 samples vs a 163-pt model")),
                      ("c5", ("motifseq", 100_000, 20000, 500, 3, 1, "BASELINE config 5: MotifSeq long-read stress, 100 000 reads x 20 000 samples vs a 500-pt motif (one GPU)"))):
        try:
            out[key] = baseline_config_extra(a, L, *args)
        except Exception as e:                                        # noqa: BLE001 -- report, keep the other lines
            out[key] = {"config": args[-1], "error": repr(e)}
    return out


# what pins the oracle each parity block compares with: the segmenter path and the normalisations are pinned
# by outputs of the reference itself (tools/gen_golden.py imports /root/reference; tests/golden/*); the DTW core restates
# mlpy 3.5.0's cdtw.c, and mlpy is in neither /root/reference nor this image -- unpinned until tools/pin_mlpy.py has run
ORACLE_PINNED = {"segmenter": True, "motifseq": False}


def headline_roofline_labels(roof):
    """SURVEY 8(d) names two fractions for the DTW path; both, side by side, with where each peak comes from"""
    v = roof.get("valu", {})
    sp, ws = v.get("screening_pass", {}), v.get("whole_step", {})
    roof["hbm_frac"] = roof["frac"]
    roof["hbm_peak_source"] = "/opt/skills/guides/MI355X_MICROARCH.md: HBM3E ~8 TB/s"
    roof["valu_frac"] = sp.get("frac") if sp.get("frac") is not None else sp.get("frac_at_2.4_ghz")
    roof["valu_frac_whole_step"] = ws.get("frac") if ws.get("frac") is not None else ws.get("frac_at_2.4_ghz")
    roof["valu_peak_source"] = ("profiles/r05_valu_rate.txt (tools/ubench/valu_rate.hip): v_min3_u32 / v_sad_u32 issue in "
                                "4.1-4.3 shader cycles on a SIMD => 1024 SIMDs x 64 lanes / 8 cycles per cell x the measured clock")
    roof["limiter"] = "valu_frac (VALU issue rate of the min-plus recurrence); hbm_frac is what SURVEY 8(d) also asks for"
    roof.pop("binding", None)
    return roof


# ----------------------------------------------------------------------------------------------------
# one rank
# ----------------------------------------------------------------------------------------------------
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
    parity["oracle_pinned"] = ORACLE_PINNED[a.workload]
    parity["oracle_pinned_note"] = ("filter / medmad / zscale are pinned by reference-made goldens; the DTW core (D1-D3) restates "
                                    "mlpy 3.5.0, absent here: *_bit_identical means identical to oracle/sk_oracle.c "
                                    "(tools/pin_mlpy.py pins it where mlpy imports)") if a.workload == "motifseq" else \
        "pinned by outputs of the reference itself (tools/gen_golden.py, tests/golden/)"
    if a.workload == "motifseq":
        roofline = headline_roofline_labels(motifseq_roofline(a, w, prof, a.steps, mean_n))
        name = "reads/sec MotifSeq DTW (4k-sample read x 200-sample motif)"
        wl = workload_name("motifseq", a.reads, a.samples, a.motif, a.scaling, a.scale)
    else:
        roofline = segmenter_roofline(w, *seg_kernel_prof(w, prof, a.steps), step_ms=ms_per_step) if use_comm is None \
            else segmenter_roofline(w, prof, a.steps)
        name = "reads/sec segmenter (4k-sample read)"
        wl = workload_name("segmenter", a.reads, a.samples, None, a.scaling)
    line = {"metric": name, "value": value, "unit": "reads/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": a.scaling,
            "vs_baseline": None,
            "dtype": "f64 results; u32 fixed-point screening" if a.workload == "motifseq" else "int16 samples; f64 thresholds",
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
    # a silent fallback must not produce a clean-looking scaling line: fewer ranks than asked for, or the gather done by
    # host concatenation because RCCL could not be used, fail the run (dry runs on one device say --allow-host-gather or
    # --ranks-on-device, where the host backend is the point)
    if world > 1 and a.ranks_on_device is None and not a.allow_host_gather:
        problems = []
        if ranks_seen != world:
            problems.append("ranks_seen = %d of %d" % (ranks_seen, world))
        if use_comm is not None and use_comm.backend != "rccl":
            problems.append("gather_backend = %s (%s)" % (use_comm.backend, getattr(use_comm, "why_host", "?")))
        if problems:
            line["failed"] = "; ".join(problems)
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
        extras["other_paths"] = bench_extras.other_paths_block(a, L, w)
    elif world == 1 and not a.no_extras:
        if a.workload == "motifseq" and not a.no_baseline_configs and a.reads == 1_000_000 and a.samples == 4000:
            extras.update(baseline_configs_block(a, L))               # c2_10k, c3, c5: lines of their own
        extras.update(extras_single_gpu(a, L, w))
        extras["sweep"] = bench_extras.sweep_block(a, L, w)
        if a.workload == "motifseq" and not a.no_sensitivity:
            extras["other_paths"] = bench_extras.other_paths_block(a, L, w)
            extras["cli"] = bench_extras.cli_block(a, L, w)
            extras["sensitivity"] = bench_extras.sensitivity_block(a, L, w)
    elif world == 1 and a.sweep_reads:
        extras["sweep"] = bench_extras.sweep_block(a, L, w)
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
        if line.get("failed"):
            sys.stderr.write("bench.py: FAILED multi-GPU sanity: %s (pass --allow-host-gather for a dry run)\n" % line["failed"])
            sys.exit(4)


if __name__ == "__main__":
    main()
