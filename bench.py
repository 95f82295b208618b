#!/usr/bin/env python3
"""bench.py -- whole-job throughput of the SquiggleKit hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload motifseq|segmenter]

Headline (BASELINE.json `metric`): reads/s of the MotifSeq path -- scale_outliers ->
medmad -> subsequence DTW -- for 4 000-sample int16 reads against a 200-point motif
(config C4: 1 000 000 reads per GPU, seed 20260929 + rank, synthetic squiggles generated
on the device).  One "step" = one pass of the hot path (prep kernel + DTW kernel) over the
whole HBM-resident batch; inputs are already in HBM when the timed region starts.

N > 1 (launched by torch.distributed.run, one rank per GPU): reads are sharded, no data-path
collective; torch.distributed (backend nccl == RCCL) provides the barrier, the MAX over
ranks of the elapsed time and the final all-gather of the 24-byte hit records.  torch is
plumbing only -- the kernels, memory and stream are the library's own (ctypes C ABI).

Rank 0 prints ONE JSON line with the driver's contract plus `roofline` (HIP-event kernel
time vs algorithmic bytes; the VALU view is inside it because this kernel is FP64-VALU bound,
see DESIGN.md) and `cpu_baseline` (the oracle's mlpy-style C restatement timed on one host
core over a bounded sample of the same reads -- the only place the oracle is timed).
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
HIT_BYTES = 24


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="motifseq", choices=["motifseq", "segmenter"])
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads per GPU")
    ap.add_argument("--samples", type=int, default=4000)
    ap.add_argument("--motif", type=int, default=200, help="motif points")
    ap.add_argument("--scale", default="medmad", choices=["medmad", "zscale"])
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline budget (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=1,
                    help="additionally time the CPU baseline over this many host threads (reads split "
                         "across threads; reported as cpu_baseline.threaded, the headline stays 1 core)")
    return ap.parse_args()


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                     % (a.gpus, a.gpus))
        a.gpus = world

    from squigglekit_amd import _lib, synth
    from squigglekit_amd._lib import SegParams, check, ptr, HIT_DTYPE

    dist = None
    torch = None
    use_dist = world > 1 or os.environ.get("SK_BENCH_FORCE_DIST") == "1"   # force: exercise the RCCL path on 1 GPU
    if use_dist:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    L = _lib.load()
    _lib.init(local)

    R, M, N = a.reads, a.samples, a.motif
    stride = (M + 7) // 8 * 8
    motif = synth.synthetic_motif(N)
    seed = (synth.SEED_C4 if a.workload == "motifseq" else synth.SEED_C2) + rank

    # ---- device-resident inputs -------------------------------------------------------
    d_sig = L.sk_dev_alloc(R * stride * 2)
    d_len = L.sk_dev_alloc(R * 4)
    if not d_sig or not d_len:
        check(-4)
    lens = np.full(R, M if a.workload == "motifseq" else M - 1, dtype=np.int32)   # segmenter: Num=-1
    check(L.sk_dev_upload(d_len, ptr(lens), lens.nbytes))
    check(L.sk_synth_squiggles_dev(d_sig, stride, R, M, seed, ptr(motif), N))

    max_segs = 16
    if a.workload == "motifseq":
        if use_dist:
            out_t = torch.empty(R * HIT_BYTES, dtype=torch.uint8, device="cuda")
            gath_t = torch.empty(world * R * HIT_BYTES, dtype=torch.uint8, device="cuda")
            d_out = C.c_void_p(out_t.data_ptr())
        else:
            d_out = L.sk_dev_alloc(R * HIT_BYTES)
        mode = _lib.SK_SCALE[a.scale]

        def step():
            check(L.sk_motifseq_dev_i16(d_sig, stride, d_len, R, ptr(motif), N, mode, 0, 1200, d_out))
            check(L.sk_sync())
            if use_dist:                       # the one exchange: gather of the hit records (RCCL)
                dist.all_gather_into_tensor(gath_t, out_t)
                torch.cuda.current_stream().synchronize()   # out_t is rewritten by the next step
    else:
        d_segs = L.sk_dev_alloc(R * max_segs * 2 * 4)
        d_nsegs = L.sk_dev_alloc(R * 4)
        sp = SegParams()

        def step():
            check(L.sk_segment_dev_i16(d_sig, stride, d_len, R, C.byref(sp), d_segs, d_nsegs, max_segs))
            check(L.sk_sync())

    def fence():
        check(L.sk_sync())
        if use_dist:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    prep_ms = main_ms = 0.0
    dtw_prof = {"dist_ms": 0.0, "start_ms": 0.0, "launches": 0, "retries": 0}
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
        p_, m_ = C.c_float(), C.c_float()
        check(L.sk_last_kernel_ms(C.byref(p_), C.byref(m_)))   # HIP events on the library's stream
        prep_ms += p_.value
        main_ms += m_.value
        if a.workload == "motifseq":           # per-launch times of the two DTW passes
            da, sb = C.c_float(), C.c_float()
            la, lb, rpl = C.c_int32(), C.c_int32(), C.c_int32()
            check(L.sk_last_dtw_profile(C.byref(da), C.byref(la), C.byref(sb), C.byref(lb), C.byref(rpl)))
            dtw_prof["dist_ms"] += da.value
            dtw_prof["start_ms"] += sb.value
            dtw_prof["launches"] += la.value
            dtw_prof["retries"] += L.sk_last_dtw_retries()
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    prep_ms /= max(1, a.steps)
    main_ms /= max(1, a.steps)

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / a.steps * 1e3
    value = world * R * a.steps / elapsed
    alg_bytes = R * (2 * M + HIT_BYTES)                       # 2*M in + 24 out per read

    # ---- parity + CPU baseline on a bounded sample of rank 0's reads -------------------
    from oracle import oracle as ora
    if world > 1:
        a.cpu_seconds = 0.0          # the CPU baseline is timed at N = 1 only; keep the parity spot check
    S = min(R, 8192)
    sample = np.empty((S, stride), dtype=np.int16)
    check(L.sk_dev_download(ptr(sample), d_sig, sample.nbytes))
    parity, cpu = {}, None
    if a.workload == "motifseq":
        hits = np.empty(S, dtype=HIT_DTYPE)
        check(L.sk_dev_download(ptr(hits), d_out, hits.nbytes))
        n0 = min(S, 128)
        t0 = time.perf_counter()
        want = [ora.motifseq_batch_i16(sample[:n0], lens[:n0], motif, scale_mode=mode)]
        dt = time.perf_counter() - t0
        done = n0
        if a.cpu_seconds > 0 and done < S:
            more = int(min(S - done, max(0, (a.cpu_seconds - dt) / (dt / n0))))
            if more > 0:
                t1 = time.perf_counter()
                want.append(ora.motifseq_batch_i16(sample[done:done + more], lens[done:done + more], motif,
                                                   scale_mode=mode))
                dt += time.perf_counter() - t1
                done += more
        want = np.concatenate(want)
        got = hits[:done]
        parity = {"reads_checked": int(done),
                  "start_end_exact": bool(np.array_equal(got["start"], want["start"])
                                          and np.array_equal(got["end"], want["end"])),
                  "max_abs_ddist": float(np.nanmax(np.abs(got["dist"] - want["dist"]))),
                  "dist_bit_identical": bool(np.array_equal(got["dist"], want["dist"]))}
        cpu = {"value": done / dt, "unit": "reads/s", "cores": 1, "kind": "port",
               "sample": "%d of rank 0's reads (%d x %d-pt motif): oracle C restatement of "
                         "filter+medmad+mlpy dtw_subsequence (full matrix malloc per call), gcc -O2, 1 thread, %.1f s"
                         % (done, M, N, dt),
               "host_cores_total": os.cpu_count()}
        if a.cpu_threads > 1 and a.cpu_seconds > 0:
            from concurrent.futures import ThreadPoolExecutor
            T = a.cpu_threads
            per = max(1, min(S // T, int(a.cpu_seconds * cpu["value"])))      # about cpu_seconds per thread
            parts = [(i * per, (i + 1) * per) for i in range(T)]
            with ThreadPoolExecutor(T) as ex:                                 # ctypes calls release the GIL
                t1 = time.perf_counter()
                list(ex.map(lambda ab: ora.motifseq_batch_i16(sample[ab[0]:ab[1]], lens[ab[0]:ab[1]], motif,
                                                              scale_mode=mode), parts))
                dtt = time.perf_counter() - t1
            cpu["threaded"] = {"value": T * per / dtt, "unit": "reads/s", "cores": T,
                               "sample": "%d reads on each of %d threads, %.1f s" % (per, T, dtt)}
        cells = float(N) * float(np.mean(got["n"]))
        # whole DTW stage expressed in the reference's arithmetic: 4 FP64 ops per cell
        valu = {"bound": "valu_f64_equivalent", "achieved": R * cells * 4 / (main_ms * 1e-3) / 1e12,
                "peak": VALU_F64_LANEOPS / 1e12,
                "unit": "T f64-lane-op/s the reference's 4-op-per-cell recurrence would need at this rate "
                        "(the screening pass replaces most of them by 2 integer ops, so this may exceed 1)"}
        valu["frac"] = valu["achieved"] / valu["peak"]
        dominant, dom_ms = "k_sdtw (all launches of one call)", main_ms
        if dtw_prof["launches"] > 0:
            # dominant kernel = the fixed-point screening pass k_sdtw_q<L,R,feed>; one launch per chunk
            per_step = dtw_prof["launches"] / a.steps
            Lg, Rg = (16, (N + 15) // 16) if N <= 256 else (64, (N + 63) // 64)
            dominant = "k_sdtw_q<%d,%d,0> (screening pass, %d launches per call)" % (Lg, Rg, per_step)
            dom_ms = dtw_prof["dist_ms"] / dtw_prof["launches"]
            alg_bytes = alg_bytes / per_step                   # algorithmic bytes one launch covers
            # its own roof: 2 half-rate VALU instructions (v_min3_u32, v_sad_u32: 4 cycles each,
            # tools/ubench/valu_rate.hip) per cell -> 256 CU x 4 SIMD x 64 lanes x 2.4 GHz / 8 cycles
            q_peak = 256 * 4 * 64 * 2.4e9 / 8.0
            q_ach = R * cells / (dtw_prof["dist_ms"] / a.steps * 1e-3)
            valu["screening_pass"] = {"bound": "valu_issue", "achieved": q_ach / 1e12, "peak": q_peak / 1e12,
                                      "unit": "T cell-updates/s", "frac": q_ach / q_peak}
            valu["passes_ms_per_call"] = {"screen": dtw_prof["dist_ms"] / a.steps,
                                          "window": dtw_prof["start_ms"] / a.steps,
                                          "retried_reads": dtw_prof["retries"] / a.steps}
    else:
        segs = np.empty((S, max_segs, 2), dtype=np.int32)
        nsegs = np.empty(S, dtype=np.int32)
        check(L.sk_dev_download(ptr(segs), d_segs, segs.nbytes))
        check(L.sk_dev_download(ptr(nsegs), d_nsegs, nsegs.nbytes))
        t0 = time.perf_counter()
        osegs, onsegs = ora.segment_batch_i16(sample, lens[:S], max_segs=max_segs)
        dt = time.perf_counter() - t0
        same = bool(np.array_equal(nsegs, onsegs)) and all(
            np.array_equal(segs[r, :nsegs[r]], osegs[r, :nsegs[r]]) for r in range(S))
        parity = {"reads_checked": int(S), "segments_bit_exact": same}
        cpu = {"value": S / dt, "unit": "reads/s", "cores": 1, "kind": "port",
               "sample": "%d of rank 0's reads: oracle C restatement of filter+get_segs, gcc -O2, 1 thread, %.2f s"
                         % (S, dt), "host_cores_total": os.cpu_count()}
        alg_bytes = R * (2 * M + 4 + 8 * 2)
        valu = None
        dominant, dom_ms = ("k_prep_i16", prep_ms) if prep_ms >= main_ms else ("k_segment_walk", main_ms)

    if world > 1:
        cpu = None
    achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
    # HBM bytes from the PMC passes (FETCH_SIZE, WRITE_SIZE; collected separately with rocprofv3 --pmc
    # and committed under profiles/ -- a bench run cannot read hardware counters itself)
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % a.workload)
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            key = [k for k in tj["kernels"] if ("k_sdtw_q" in k if a.workload == "motifseq" else "k_prep_i16" in k)]
            if key and tj.get("reads_per_call"):
                kk = tj["kernels"][key[0]]
                per_read = (kk["fetch_bytes_total"] + kk["write_bytes_total"]) / (
                    tj["reads_per_call"] * max(1, tj.get("calls", 1)))
                traffic = per_read * (alg_bytes / (2 * M + HIT_BYTES) if a.workload == "motifseq" else R)
                traffic_src = "profiles/traffic_%s.json: %s, %.0f B/read measured" % (a.workload, key[0], per_read)
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": traffic_src,
                "kernel_ms": {"prep": prep_ms, "main": main_ms, "dominant_avg_launch": dom_ms},
                "algorithmic_bytes_per_launch": alg_bytes}
    if valu:
        roofline["binding"] = "valu issue rate (min-plus recurrence; HBM is not the limiter, DESIGN.md 4.3)"
        roofline["valu"] = valu

    name = ("reads/sec MotifSeq DTW (4k-sample read x 200-sample motif)" if a.workload == "motifseq"
            else "reads/sec segmenter (4k-sample read)")
    line = {"metric": name, "value": value, "unit": "reads/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64" if a.workload == "motifseq" else "int16/f64",
            "data": "synthetic",
            "config": {"workload": ("MotifSeq C4" if a.workload == "motifseq" else "segmenter C2-1M")
                       + ": %d reads x %d int16 samples per GPU" % (R, M)
                       + (", %d-pt motif, %s" % (N, a.scale) if a.workload == "motifseq" else ", default flags"),
                       "reads_per_gpu": R, "samples": M, "motif_points": N if a.workload == "motifseq" else None,
                       "seed": seed, "sharding": "reads block-sharded, %d rank(s), result all-gather over RCCL" % world},
            "roofline": roofline, "cpu_baseline": cpu, "parity": parity}
    print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
