"""Drop-in for /root/reference/segmenter.py's command line (segmenter.py:49-297).

Same flags, same stdout (`name<TAB>s0,e0,s1,e1,...`), same stderr strings, same exit
codes; the per-read work (scale_outliers + get_segs) runs on the GPU in batches
through the C ABI.  Reads are buffered `--batch` at a time and printed in input
order.  Additive flags: --device, --batch, --blow5.  There is no CPU path.
"""
import argparse
import os
import sys
from struct import error as struct_error

import numpy as np

from . import api, fastio, tsvio
from ._warm import mark as _mark, Stats as _Stats

_KEEP = []       # input mappings / page-locked buffers of a finished reader: released with the process
from ._lib import SegParams


class _Parser(argparse.ArgumentParser):
    def error(self, message):                      # segmenter.py:40-44
        sys.stderr.write("error: %s\n" % message)
        self.print_help()
        sys.exit(2)


def build_parser():
    p = _Parser(description="segmenter (MI355X) - find stall / homopolymer stretches in squiggle data")
    src = p.add_mutually_exclusive_group()
    src.add_argument("-i", "--ind", nargs="+", help="one or more fast5 files")
    src.add_argument("-p", "--f5_path", help="directory searched recursively for fast5 files")
    src.add_argument("-s", "--signal", help="signal TSV written by SquigglePull (.gz accepted)")
    src.add_argument("--blow5", help="[extension] BLOW5 file (uncompressed or zlib records); with --raw_signal the "
                                     "records are decoded natively into int16 batches")
    src.add_argument("--i16", help="[extension] packed reads: a .npy file holding an int16 array [reads, samples] "
                                   "(read name = row index)")
    p.add_argument("--single", action="store_true", help="fast5 files hold one read each")
    p.add_argument("-n", "--Num", type=int, default=0, help="use only the first Num samples; 0 = whole read")
    p.add_argument("-e", "--error", type=int, default=5, help="out-of-band samples tolerated inside a segment")
    p.add_argument("-c", "--corrector", type=int, default=50,
                   help="window that lets the error budget recover on long segments")
    p.add_argument("-w", "--window", type=int, default=150, help="shortest segment reported")
    p.add_argument("-d", "--seg_dist", type=int, default=50, help="segments closer than this are merged")
    p.add_argument("-t", "--std_scale", type=float, default=0.75,
                   help="band half-width in standard deviations around the median")
    p.add_argument("-v", "--view", action="store_true", help="plot each result (not available in this build)")
    p.add_argument("-g", "--gap", action="store_true", help="with -u: enforce stall-to-polyT gap distance")
    p.add_argument("-b", "--gap_dist", type=int, default=3000, help="largest stall-to-polyT gap accepted")
    p.add_argument("-k", "--stall", action="store_true", help="with -u: require a stall near the read start")
    p.add_argument("-u", "--test", action="store_true", help="filter reads with the -k / -g checks")
    p.add_argument("-l", "--stall_len", type=float, default=0.25,
                   help="fraction of --window the first (stall) segment may be")
    p.add_argument("-j", "--stall_start", type=int, default=300, help="latest start accepted for the stall")
    p.add_argument("-lim_hi", "--lim_hi", type=int, default=900, help="samples >= this are dropped")
    p.add_argument("-lim_low", "--lim_low", type=int, default=0, help="samples <= this are dropped")
    p.add_argument("--raw_signal", action="store_true", help="fast5 input: keep raw ADC values (no pA conversion)")
    p.add_argument("--device", type=int, default=None, help="[extension] GPU index (default $SK_DEVICE or 0)")
    p.add_argument("--batch", type=int, default=4096, help="[extension] reads per GPU call")
    p.add_argument("--gpus", type=int, default=1,
                   help="[extension] shard every batch of reads over this many GPUs of the node")
    p.add_argument("--stats-json", dest="stats_json", default=None, metavar="PATH",
                   help="[extension] write reads / reads per second / input GB per second / GPU calls of this run to PATH "
                        "as JSON (also $SK_STATS_JSON); stdout and stderr stay the reference's")
    p.add_argument("--stats", action="store_true", help="[extension] the same as one line on stderr at the end")
    return p


_STATS = [_Stats("segmenter")]      # this run's throughput counters (--stats-json / --stats)


class _Batcher:
    """Buffers (name, signal) pairs, runs them through the GPU, emits in order."""

    def __init__(self, args):
        self.args = args
        self.params = SegParams.from_args(args)
        self.names, self.sigs = [], []
        self._pending, self._worker = None, None

    def add(self, name, sig, miss_name=None):
        self.names.append((name, miss_name if miss_name is not None else name))
        self.sigs.append(sig)
        if len(self.sigs) >= self.args.batch:
            self.flush()

    def note(self, message):
        """A stderr message that must keep its place between the reads around it."""
        self.names.append((None, message))
        self.sigs.append(None)

    def emit(self, name, miss, segs):
        """What the reference does with one read's get_segs result (segmenter.py:211-227)."""
        if not segs:
            sys.stderr.write("no segments found: {}".format(miss))            # segmenter.py:213
            return
        if self.args.test:
            segs = api.test_segs(segs, self.args)
            if not segs:
                if self.args.signal:
                    sys.stderr.write("no segs for testing: {}".format(miss))   # :219 (TSV branch only)
                return
        print("\t".join([name, ",".join(str(v) for pair in segs for v in pair)]))

    def flush(self):
        self.drain()                                  # (a pipelined block's table comes first)
        if not self.sigs:
            return
        live = [s for s in self.sigs if s is not None]
        if live:
            _STATS[0].batch(len(live))
        results = iter(api.segment_any(live, self.params) if live else [])
        for (name, miss), sig in zip(self.names, self.sigs):
            if sig is None:
                sys.stderr.write(miss)
                continue
            self.emit(name, miss, next(results))
        self.names, self.sigs = [], []

    def rows(self, rows, nsamp, name_col, name_of):
        """A block of plain int16 reads (BLOW5 --raw_signal / packed input): one GPU batch; the table through the
        native formatter unless -u asks for the per-read checks.  One block deep pipeline: this block's GPU call runs
        on a worker thread while the previous block's table is written (drain() at the end)."""
        if not len(nsamp):
            return
        Num = self.args.Num
        lens = (np.maximum(nsamp + Num, 0) if Num < 0 else np.minimum(nsamp, Num)).astype(np.int32)   # sig[:Num]
        if self._worker is None:
            from concurrent.futures import ThreadPoolExecutor
            self._worker = ThreadPoolExecutor(1)
        _mark("block of %d reads to the GPU worker" % len(nsamp))
        job = self._worker.submit(api.segment_batch, rows, lens, self.params)
        prev, self._pending = self._pending, (job, len(nsamp), name_col, name_of)
        if prev is not None:
            self._finish(prev)

    def rows_pa(self, rows, nsamp, calib, name_col, name_of):
        """A block of raw reads with their channel constants (BLOW5 records without --raw_signal): the pA conversion the
        reference applies to fast5 / slow5 input (segmenter.py:345-349) runs on the GPU, then the float64 path; one batch,
        one native table, pipelined like rows()."""
        if not len(nsamp):
            return
        Num = self.args.Num
        lens = (np.maximum(nsamp + Num, 0) if Num < 0 else np.minimum(nsamp, Num)).astype(np.int32)   # sig[:Num]
        if self._worker is None:
            from concurrent.futures import ThreadPoolExecutor
            self._worker = ThreadPoolExecutor(1)
        job = self._worker.submit(api.segment_batch_pa, rows, lens, calib, self.params)
        prev, self._pending = self._pending, (job, len(nsamp), name_col, name_of)
        if prev is not None:
            self._finish(prev)

    def rows_f64(self, fb):
        """A chunk of plain decimal (pA) lines as the float64 tokenizer leaves it (tsvio.FloatBlock: flat values +
        offsets): one GPU batch -- the sig[:Num] cut rides along as per-read lengths, nothing is repacked --, one
        native table, pipelined like rows()."""
        Num = self.args.Num
        ntok = np.diff(fb.off).astype(np.int64)
        lens = (np.maximum(ntok + Num, 0) if Num < 0 else np.minimum(ntok, Num)).astype(np.int32)    # sig[:Num]
        if self._worker is None:
            from concurrent.futures import ThreadPoolExecutor
            self._worker = ThreadPoolExecutor(1)
        _mark("block of %d float64 reads to the GPU worker" % fb.n)
        job = self._worker.submit(api.segment_ragged_f64, fb.batch_values(), fb.off, lens, self.params)
        prev, self._pending = self._pending, (job, fb.n, ("span", fb.buf, fb.spans("name")), lambda i, b=fb: b.text("name", i))
        if prev is not None:
            self._finish(prev)

    def drain(self):
        prev, self._pending = self._pending, None
        if prev is not None:
            self._finish(prev)

    def _finish(self, p):
        job, n, name_col, name_of = p
        segs, nsegs = job.result()
        _STATS[0].batch(n)
        _mark("block of %d reads back from the GPU" % n)
        if self.args.test:
            for i in range(n):
                nm = name_of(i)
                self.emit(nm, nm, segs[i, :nsegs[i]].tolist() if nsegs[i] else False)
            return
        for i in np.flatnonzero(nsegs == 0):
            sys.stderr.write("no segments found: {}".format(name_of(int(i))))          # segmenter.py:213
        off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(2 * nsegs.astype(np.int64), out=off[1:])
        keep = np.arange(segs.shape[1])[None, :] < nsegs[:, None]
        vals = segs[keep].ravel()                                                   # [start, end] pairs, read order
        text = fastio.fmt_rows(n, [name_col, ("i32list", vals, off)], skip=(nsegs == 0).astype(np.uint8))
        fastio.write_stdout(text)
        _mark("table written")

    def block(self, blk, path):
        """A parsed TSV chunk (tsvio.TsvBlock): the integer lines go to the GPU as ONE int16 batch straight from
        the tokenizer's rows; every other line takes the per-read route above, in its place."""
        Num = self.args.Num
        fast = (blk.flags & 27) == 3                                        # ALLINT | ANY, not SLOW / SHORT
        idx = np.flatnonzero(fast)
        if idx.size == blk.n and blk.n and not self.args.test:
            # every line is a plain integer read: one GPU batch, one native table, pipelined with the next chunk
            no = blk.base + blk._no.astype(np.int64)
            self.rows(blk.rows, blk.nsamp, ("span", blk.buf, np.stack([no, no + blk._nl], axis=1)), blk.name)
            return
        self.drain()                                                        # (what follows prints directly)
        res = {}
        if idx.size:
            ns = blk.nsamp[idx]
            lens = (np.maximum(ns + Num, 0) if Num < 0 else np.minimum(ns, Num)).astype(np.int32)   # sig[:Num]
            rows = blk.rows[idx] if idx.size != blk.n else blk.rows
            segs, nsegs = api.segment_batch(rows, lens, self.params)
            _STATS[0].batch(idx.size)
            res = {int(i): k for k, i in enumerate(idx)}
        for i in range(blk.n):
            k = res.get(i)
            if k is not None:
                name = blk.name(i)
                self.emit(name, name, segs[k, :nsegs[k]].tolist() if nsegs[k] else False)
                continue
            fl = int(blk.flags[i])
            if (fl & 27) == 1:                                              # integers, all zero: segmenter.py:203-205
                sys.stderr.write("No signal found in file: {} {}".format(path, blk.name(i)))
                continue
            name, sig = tsvio.parse_segmenter_line(blk.line(i).decode())    # the reference's own parse, exceptions included
            if not sig.any():
                sys.stderr.write("No signal found in file: {} {}".format(path, name))
                continue
            self.add(name, sig[:Num])
            self.flush()                                                    # keeps the output in file order


def main(argv=None):
    parser = build_parser()
    argv = sys.argv[1:] if argv is None else argv
    args = parser.parse_args(argv)
    if len(argv) == 0:                              # segmenter.py:100-102
        parser.print_help(sys.stderr)
        sys.exit(1)
    if not args.Num:                                # segmenter.py:104-105 (drops the last sample)
        args.Num = -1
    if args.view:
        sys.stderr.write("segmenter: -v/--view plotting is not part of this build; ignoring\n")

    _mark("main() entered")
    _STATS[0] = _Stats("segmenter")
    del _KEEP[:]                                     # (a previous call in this process: its buffers can go now)
    if not (args.f5_path or args.ind or args.signal or args.blow5 or args.i16):
        sys.stderr.write("Unknown file or path input")
        parser.print_help(sys.stderr)
        sys.exit(1)

    from . import _lib
    _lib.warm_start(args.device, also=())          # HIP start-up runs beside the parsing of the first chunk
    if args.gpus > 1:
        api.set_devices(range(args.gpus))
    out = _Batcher(args)

    if args.signal:
        # native tokenizer (csrc/sk_tsv.cpp): integer lines arrive as int16 rows, one GPU batch per chunk of the
        # file; a line it cannot take verbatim (odd tokens, decimals, too few columns) is parsed exactly the
        # reference's way
        # (a chunk whose first line starts with a decimal token -- pA files, SquigglePull's default output -- comes
        # straight from the float64 tokenizer: tsvio.FloatBlock)
        for blk in tsvio.iter_tsv_blocks(args.signal, 4):
            if isinstance(blk, tsvio.FloatBlock):
                fb = blk
            else:
                if blk.mostly_integer():
                    out.block(blk, args.signal)
                    continue
                fb = blk.float_block(4)                              # decimal lines after all: float64 tokenizer
                if fb is None:
                    continue
            if fb.clean() and not args.test:
                out.rows_f64(fb)                                      # the whole chunk as one batch, no Python per read
                continue
            out.flush()
            for name, _rid, vals, fl, raw in tsvio.float_block_lines(fb):
                if (fl & 24) or not (fl & 5):       # SLOW | SHORT, or neither FIRSTDOT nor ALLINT
                    name, sig = tsvio.parse_segmenter_line(raw.decode())
                else:
                    sig = vals
                if not sig.any():                   # segmenter.py:203-205
                    out.note("No signal found in file: {} {}".format(args.signal, name))
                    continue
                out.add(name, sig[:args.Num])
            out.flush()
    elif args.blow5 and args.raw_signal:
        seen = 0
        try:
            for blk in fastio.iter_blow5_blocks_i16(args.blow5, keep=_KEEP):
                ok = np.flatnonzero((blk.flags & 2) == 0)
                for i in np.flatnonzero(blk.flags & 2):
                    sys.stderr.write("segmenter: unreadable BLOW5 record {} in {}; skipped\n".format(seen + int(i), args.blow5))
                seen += blk.n
                if ok.size != blk.n:
                    blk = fastio.Blow5Block(blk.rows[ok], blk.nsamp[ok], blk.ids[ok], blk.calib[ok], blk.flags[ok])
                w = blk.ids.dtype.itemsize
                st = np.arange(blk.n, dtype=np.int64) * w
                out.rows(blk.rows, blk.nsamp, ("span", blk.ids, np.stack([st, st + np.char.str_len(blk.ids)], axis=1)),
                         lambda i, b=blk: b.ids[i].decode())
        except ValueError as e:                          # truncated file, unsupported compression: say so, no traceback
            out.drain()
            out.flush()
            sys.stderr.write("segmenter: --blow5: {}\n".format(e))
            sys.exit(1)
    elif args.i16:
        try:
            for lo, part in fastio.iter_npy_blocks_i16(args.i16, keep=_KEEP):
                ns = np.full(part.shape[0], part.shape[1], dtype=np.int32)
                out.rows(part, ns, ("i32", np.arange(lo, lo + part.shape[0], dtype=np.int32)), lambda i, lo=lo: str(lo + i))
        except ValueError as e:
            sys.stderr.write("segmenter: --i16: {}\n".format(e))
            sys.exit(1)
    elif args.blow5 and not args.test:
        # pA (the reference's default for fast5 / slow5 input): records decoded natively into int16 rows + channel
        # constants, converted on the GPU
        seen = 0
        try:
            for blk in fastio.iter_blow5_blocks_i16(args.blow5, keep=_KEEP):
                ok = np.flatnonzero((blk.flags & 2) == 0)
                for i in np.flatnonzero(blk.flags & 2):
                    sys.stderr.write("segmenter: unreadable BLOW5 record {} in {}; skipped\n".format(seen + int(i), args.blow5))
                seen += blk.n
                if ok.size != blk.n:
                    blk = fastio.Blow5Block(blk.rows[ok], blk.nsamp[ok], blk.ids[ok], blk.calib[ok], blk.flags[ok])
                w = blk.ids.dtype.itemsize
                st = np.arange(blk.n, dtype=np.int64) * w
                out.rows_pa(blk.rows, blk.nsamp, blk.calib,
                            ("span", blk.ids, np.stack([st, st + np.char.str_len(blk.ids)], axis=1)),
                            lambda i, b=blk: b.ids[i].decode())
        except ValueError as e:
            out.drain()
            out.flush()
            sys.stderr.write("segmenter: --blow5: {}\n".format(e))
            sys.exit(1)
    elif args.blow5:
        from .blow5 import read_blow5, to_pA
        try:
            for rec in read_blow5(args.blow5):
                sig = rec["signal"].astype(int)
                if not args.raw_signal:
                    sig = to_pA(sig, rec["digitisation"], rec["offset"], rec["range"])
                out.add(rec["read_id"], sig[:args.Num])
        except (ValueError, EOFError, struct_error) as e:
            out.flush()
            sys.stderr.write("segmenter: --blow5: {}\n".format(e))
            sys.exit(1)
    else:
        if args.f5_path:
            files = [os.path.join(d, f) for d, _, fs in os.walk(args.f5_path) for f in fs if f.endswith(".fast5")]
        else:
            files = list(args.ind)
        for path in files:
            label = os.path.basename(path) if args.f5_path else path
            if args.single:
                sig = tsvio.segmenter_process_fast5(path, args.raw_signal, sys.stderr)   # segmenter.py:146,262
                if not np.asarray(sig).any():
                    out.note("main():data not extracted. Moving to next file: {}".format(label))
                    continue
                out.add(label, np.array(sig[:args.Num], dtype=float))
            else:
                for read, sig in tsvio.read_multi_fast5(path, args.raw_signal).items():
                    out.add(read, np.array(sig[:args.Num], dtype=float), miss_name=label)
    out.drain()
    out.flush()
    _mark("end of main()")
    _STATS[0].finish(args, [args.signal, args.blow5, args.i16] + list(args.ind or []))
    sys.stderr.write("Done")                        # segmenter.py:297


if __name__ == "__main__":
    main()
