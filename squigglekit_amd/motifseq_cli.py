"""Drop-in for /root/reference/MotifSeq.py's command line (MotifSeq.py:80-311, 431-449).

Same flags, the same stderr banner, the same 12/13-column TSV on stdout.  Per read,
scale_outliers + medmad/zscale + dtw_subsequence run on the GPU (batched, C ABI);
the scoring of MotifSeq.py:441-445 stays in Python so the printed floats are the
reference's digit for digit.  fast5 input (-f / -p) goes through h5py when it is importable and
through the built-in reader (hdf5min.py) otherwise, with the reference's stderr messages.
Additive flags: --device, --gpus, --batch, --after_stall, --strict-compat, --blow5, --i16.
Whole chunks of plain integer reads (TSV chunks, BLOW5 / packed blocks) go to the GPU as one batch and their rows are
formatted natively (csrc/sk_io.cpp writes floats as Python does); anything unusual takes the per-read route.
"""
import argparse
import os
import sys

import numpy as np

from . import _lib, api, fastio, tsvio
from ._warm import mark as _mark, Stats as _Stats

_KEEP = []       # input mappings / page-locked buffers of a finished reader: released with the process

VERSION = "1.3.0"          # the reference's MotifSeq version string (MotifSeq.py:84)
HEADER = ["fast5", "readID", "model", "start", "end", "length", "distance_score", "model_mean",
          "model_stdev", "Z-score", "p-value", "hit_Probability"]
BANNER = ("\n\n**********************************************************\n"
          "*  z-score, p-value, probability, etc. are based on      *\n"
          "*     preliminary experimental modeling only             *\n"
          "*                Use at own risk                         *\n"
          "**********************************************************\n\n\n")


class _Parser(argparse.ArgumentParser):
    def error(self, message):                      # MotifSeq.py:73-77
        sys.stderr.write("error: %s\n" % message)
        self.print_help()
        sys.exit(2)


def build_parser():
    p = _Parser(description="MotifSeq (MI355X) - find a sequence motif's signal inside raw nanopore reads")
    src = p.add_mutually_exclusive_group()
    mod = p.add_mutually_exclusive_group()
    src.add_argument("-f", "--f5f", help="text file listing fast5 paths")
    src.add_argument("-p", "--f5_path", help="directory searched recursively for fast5 files")
    src.add_argument("-s", "--signal", help="signal TSV written by SquigglePull (.gz accepted)")
    src.add_argument("--blow5", help="[extension] BLOW5 file (stored or zlib records): raw ADC values, decoded natively")
    src.add_argument("--i16", help="[extension] packed reads: a .npy file holding an int16 array [reads, samples] "
                                   "(readID = row index)")
    p.add_argument("-l", "--scale", default="medmad", choices=["zscale", "medmad"],
                   help="per-read normalisation applied before the search")
    mod.add_argument("-i", "--fasta_input", help="fasta of motifs, turned into squiggles with scrappy")
    p.add_argument("--scrappie_model", default="squiggle_r94",
                   choices=["squiggle_r94", "squiggle_r94_rna", "squiggle_r10"],
                   help="scrappie squiggle model used for -i")
    mod.add_argument("-m", "--model", help="pre-computed motif signal: scrappie squiggle text or name/len/x/values TSV")
    p.add_argument("-x", "--sig_extract", action="store_true", help="append the matched normalised signal")
    p.add_argument("--after_stall", action="store_true",
                   help="[extension] run the segmenter first and search only the signal after its first segment "
                        "(the stall); coordinates then index that slice and a last column `search_from` gives "
                        "the raw sample index where it starts")
    p.add_argument("--slope", type=float, default=2.90, help="[experimental] distance model slope")
    p.add_argument("--intercept", type=float, default=-9.6, help="[experimental] distance model intercept")
    p.add_argument("--std_const", type=float, default=0.08468, help="[experimental] distance model stdev factor")
    p.add_argument("-v", "--view", action="store_true", help="plot each hit (not available in this build)")
    p.add_argument("--save", help="directory for hit images (not available in this build)")
    p.add_argument("--img", default="png", help="image type for --save")
    p.add_argument("-scale_hi", "--scale_hi", type=int, default=1200, help="samples >= this are dropped")
    p.add_argument("-scale_low", "--scale_low", type=int, default=0, help="samples <= this are dropped")
    p.add_argument("-V", "--version", action="store_true", help="print the version and exit")
    p.add_argument("--verbose", action="store_true", help="dump the parsed arguments to stderr")
    p.add_argument("--device", type=int, default=None, help="[extension] GPU index (default $SK_DEVICE or 0)")
    p.add_argument("--batch", type=int, default=2048, help="[extension] reads per GPU call")
    p.add_argument("--gpus", type=int, default=1,
                   help="[extension] shard every batch of reads over this many GPUs of the node")
    p.add_argument("--stats-json", dest="stats_json", default=None, metavar="PATH",
                   help="[extension] write reads / reads per second / input GB per second / GPU calls of this run to PATH "
                        "as JSON (also $SK_STATS_JSON); stdout and stderr stay the reference's")
    p.add_argument("--stats", action="store_true", help="[extension] the same as one line on stderr at the end")
    p.add_argument("--strict-compat", action="store_true",
                   help="[extension] keep the reference's -m defect (empty model order: header only)")
    return p


def norm_cdf(z):
    """scipy.stats.norm.cdf == scipy.special.ndtr (MotifSeq.py:444), the same doubles without the 0.1-0.35 s that
    importing scipy.special costs every run: fastio.ndtr (csrc/sk_io.cpp) restates the Cephes routine scipy uses."""
    return fastio.ndtr(z)


def load_models(args):
    if args.model:
        models, order, lens = tsvio.read_model_auto(args.model)
        if args.strict_compat:
            order, lens = [], []                     # MotifSeq.py:413-428 never fills them
        elif order:
            sys.stderr.write("MotifSeq: note: -m searches for the model's motif(s); the reference's read_bait_model "
                             "never registers them and prints the header only (--strict-compat reproduces that)\n")
        return models, order, lens
    if args.fasta_input:
        try:
            return tsvio.fasta_to_models(args.fasta_input, args.scrappie_model)
        except ImportError:
            side = os.path.splitext(args.fasta_input)[0] + ".model"
            if os.path.exists(side):
                sys.stderr.write("MotifSeq: scrappy is not installed; using the pre-computed scrappie "
                                 "squiggle {}\n".format(side))
                return tsvio.read_scrappie_model(side)
            sys.stderr.write("MotifSeq: -i needs the scrappy package (not installed) or a scrappie squiggle "
                             "file next to the fasta; use -m <file.model>\n")
            sys.exit(1)
    return {}, [], []


_STATS = [_Stats("MotifSeq")]       # this run's throughput counters (--stats-json / --stats)


class _Batcher:
    def __init__(self, args, models, order, lens):
        self.args, self.models, self.order, self.lens = args, models, order, lens
        self.meta, self.sigs = [], []
        self._pending, self._worker = None, None

    def add(self, fast5, read_id, sig):
        self.meta.append((fast5, read_id))
        self.sigs.append(sig)
        if len(self.sigs) >= self.args.batch:
            self.flush()

    def note(self, message):
        """A stderr message that must keep its place between the reads around it."""
        self.meta.append((None, message))
        self.sigs.append(None)

    def flush(self):
        self.drain()                                  # (a pipelined block's table comes first)
        if not self.sigs:
            return
        a = self.args
        live = [i for i, s in enumerate(self.sigs) if s is not None]
        sigs = [self.sigs[i] for i in live]
        cuts = None
        if a.after_stall and sigs:                    # [extension] search only after the segmenter's first segment
            cuts = api.stall_cuts(sigs)
            sigs = [np.asarray(s)[c:] for s, c in zip(sigs, cuts)]
            for i, s in zip(live, sigs):
                self.sigs[i] = s
        if sigs:
            _STATS[0].batch(len(sigs))
        hits = (api.motifseq_multi(sigs, [np.asarray(self.models[name], dtype=np.float64) for name in self.order],
                                   a.scale, a.scale_low, a.scale_hi) if sigs else [[] for _ in self.order])
        slot = {i: k for k, i in enumerate(live)}
        for i, (fast5, read_id) in enumerate(self.meta):
            if self.sigs[i] is None:
                sys.stderr.write(read_id)
                continue
            r = slot[i]
            self.emit(fast5, read_id, [hits[c][r] for c in range(len(self.order))], self.sigs[i],
                      None if cuts is None else int(cuts[r]))
        self.meta, self.sigs = [], []

    def emit(self, fast5, read_id, hits, sig, cut):
        """The rows of one read, one per motif (MotifSeq.py:436-449)."""
        a = self.args
        norm = None
        for c, name in enumerate(self.order):
            h = hits[c]
            if h["flags"] & 1:
                sys.stderr.write("MotifSeq: no sample of {} survived the outlier limits; skipped\n".format(read_id))
                break
            if h["flags"] & 2 and not a.strict_compat:              # SK_FLAG_DEGENERATE
                sys.stderr.write("MotifSeq: the MAD of {} is 0 (medmad divides by it, MotifSeq.py:196-199); "
                                 "skipped (--strict-compat prints the reference's nan row)\n".format(read_id))
                break
            if h["flags"] & 2:
                # the reference divides by zero and hands inf / nan to mlpy: the same division, then mlpy's C arithmetic
                # evaluated literally on the GPU (sk_dtw_subsequence_cref) -- a nan distance at the first sample that
                # equals the median
                if norm is None:
                    norm = api.normalise(sig, a.scale, a.scale_low, a.scale_hi)
                try:
                    dist, start, end = api.dtw_subsequence_cref(np.asarray(self.models[name], dtype=np.float64), norm)
                except _lib.SquiggleKitError as e:
                    if e.code != -5:                                # SK_ERR_UNSUPPORTED: more than 2^28 cells of cost matrix
                        raise
                    sys.stderr.write("MotifSeq: {} has MAD 0 and is too long for the literal evaluation of the reference's "
                                     "nan row ({} samples x {} points); skipped\n".format(read_id, len(norm), len(self.models[name])))
                    break
            else:
                dist, start, end = float(h["dist"]), int(h["start"]), int(h["end"])
            mod_mean = (a.slope * self.lens[c]) + a.intercept
            mod_stdev = mod_mean * a.std_const
            z = (dist - mod_mean) / mod_stdev
            p_value = norm_cdf(z)
            hit_p = (1 - p_value) * 100
            row = [fast5, read_id, name, start, end, end - start, dist, mod_mean, mod_stdev, z, p_value, hit_p]
            if a.sig_extract:
                if norm is None:
                    norm = api.normalise(sig, a.scale, a.scale_low, a.scale_hi)
                row.append("\t".join(str(v) for v in norm[start:end]))
            if cut is not None:
                row.append(cut)
            print("\t".join("{}".format(v) for v in row))

    def table(self, n, fast5_col, id_col, hits):
        """The rows of n reads x every motif through the native formatter (file order, read-major).  Returns False --
        nothing written -- when some read needs the general route (-x, a flagged read)."""
        a = self.args
        K = len(self.order)
        if a.sig_extract or any(bool((h["flags"] & 3).any()) for h in hits):
            return False
        names = [nm.encode() for nm in self.order]
        nblob = b"".join(names)
        noff = np.concatenate([[0], np.cumsum([len(x) for x in names])]).astype(np.int64)
        nspan = np.tile(np.stack([noff[:-1], noff[1:]], axis=1), (n, 1))
        mm = np.array([(a.slope * self.lens[c]) + a.intercept for c in range(K)], dtype=np.float64)
        ms = mm * a.std_const
        dist = np.stack([h["dist"] for h in hits], axis=1)                   # [n, K]
        start = np.stack([h["start"] for h in hits], axis=1)
        end = np.stack([h["end"] for h in hits], axis=1)
        with np.errstate(all="ignore"):
            z = (dist - mm[None, :]) / ms[None, :]                           # MotifSeq.py:441-445, the same IEEE operations
            pv = norm_cdf(z)
            hp = (1 - pv) * 100

        def rep(col):                                                        # one entry per read -> one per row
            if K == 1:
                return col
            kind = col[0]
            if kind == "span":
                return ("span", col[1], np.repeat(np.asarray(col[2]), K, axis=0))
            if kind == "i32":
                return ("i32", np.repeat(np.asarray(col[1]), K))
            return col
        cols = [rep(fast5_col), rep(id_col), ("span", nblob, nspan), ("i32", start.ravel()), ("i32", end.ravel()),
                ("i32", (end - start).ravel()), ("f64", dist.ravel()), ("f64", np.tile(mm, n)), ("f64", np.tile(ms, n)),
                ("f64", z.ravel()), ("f64", pv.ravel()), ("f64", hp.ravel())]
        text = fastio.fmt_rows(n * K, cols)
        fastio.write_stdout(text)
        return True

    def rows(self, rows, nsamp, fast5_col, id_col, name_of, id_of):
        """A block of plain int16 reads (BLOW5 / packed input): one GPU batch, native table; the per-read route only
        when a read is flagged.  One block deep pipeline: the GPU call of this block runs on a worker thread while the
        previous block's table is written (drain() at the end); the caller keeps `rows` alive one call longer."""
        if not len(nsamp):
            return
        a = self.args
        if a.after_stall:
            # get_segs first, then the search behind the stall: the per-read queue does that (api.motifseq_after_stall,
            # up to --batch reads per GPU call) and prints the search_from column; `rows` is reused by the reader
            self.drain()
            for i in range(len(nsamp)):
                self.add(name_of(i), id_of(i), np.array(rows[i, :nsamp[i]], dtype=np.float64))
            return
        motifs = [np.asarray(self.models[n_], dtype=np.float64) for n_ in self.order]
        if self._worker is None:
            from concurrent.futures import ThreadPoolExecutor
            self._worker = ThreadPoolExecutor(1)
        _mark("block of %d reads to the GPU worker" % len(nsamp))
        def call():
            _mark("GPU call starts")
            try:
                return api.motifseq_multi_batch(rows, nsamp, motifs, a.scale, a.scale_low, a.scale_hi)
            finally:
                _mark("GPU call ends")
        job = self._worker.submit(call)
        prev, self._pending = self._pending, (job, len(nsamp), fast5_col, id_col, name_of, id_of,
                                              lambda i, r=rows, ns=nsamp: r[i, :ns[i]])
        if prev is not None:
            self._finish(prev)
        if a.sig_extract:                                   # (-x normalises on the GPU from this thread: no overlap)
            self.drain()

    def rows_f64(self, fb):
        """A chunk of plain decimal (pA) lines as the float64 tokenizer leaves it (tsvio.FloatBlock: flat values +
        offsets): one GPU batch per motif, one native table, pipelined like rows()."""
        a = self.args
        motifs = [np.asarray(self.models[n_], dtype=np.float64) for n_ in self.order]
        if self._worker is None:
            from concurrent.futures import ThreadPoolExecutor
            self._worker = ThreadPoolExecutor(1)
        _mark("block of %d float64 reads to the GPU worker" % fb.n)
        job = self._worker.submit(api.motifseq_multi_ragged_f64, fb.batch_values(), fb.off, motifs, a.scale, a.scale_low, a.scale_hi)
        prev, self._pending = self._pending, (job, fb.n, ("span", fb.buf, fb.spans("name")), ("span", fb.buf, fb.spans("id")),
                                              lambda i, b=fb: b.text("name", i), lambda i, b=fb: b.text("id", i),
                                              lambda i, b=fb: b.values[b.off[i]:b.off[i + 1]])
        if prev is not None:
            self._finish(prev)

    def drain(self):
        prev, self._pending = self._pending, None
        if prev is not None:
            self._finish(prev)

    def _finish(self, p):
        job, n, fast5_col, id_col, name_of, id_of, sig_of = p
        hits = job.result()
        _STATS[0].batch(n)
        _mark("block of %d reads back from the GPU" % n)
        if self.table(n, fast5_col, id_col, hits):
            _mark("table written")
            return
        if self._pending is not None and (self.args.sig_extract or self.args.strict_compat):
            # emit() may call the GPU from THIS thread (api.normalise, api.dtw_subsequence_cref for a MAD = 0 read under
            # --strict-compat) while the worker runs the next block on the same device context -- one stream, one set
            # of scratch buffers, no lock: the next block's call has to be over first (its result stays in the future)
            self._pending[0].exception()
        for i in range(n):
            self.emit(name_of(i), id_of(i), [hits[c][i] for c in range(len(self.order))],
                      sig_of(i) if (self.args.sig_extract or self.args.strict_compat) else None, None)

    def block(self, blk):
        """A parsed TSV chunk (tsvio.TsvBlock): its integer lines go to the GPU as ONE int16 batch straight from the
        tokenizer's rows (every motif against them); any other line takes the per-read route, in its place."""
        a = self.args
        fast = (blk.flags & 27) == 3                                        # ALLINT | ANY, not SLOW / SHORT
        if a.after_stall:
            fast[:] = False                                                 # (needs the raw reads on the host)
        idx = np.flatnonzero(fast)
        if idx.size == blk.n and blk.n and not a.sig_extract:
            # every line of the chunk is a plain integer read: one GPU batch, the whole table in one native call,
            # pipelined with the next chunk (rows())
            no = blk.base + blk._no.astype(np.int64)
            io = blk.base + blk._io.astype(np.int64)
            self.rows(blk.rows, blk.nsamp, ("span", blk.buf, np.stack([no, no + blk._nl], axis=1)),
                      ("span", blk.buf, np.stack([io, io + blk._il], axis=1)), blk.name, blk.read_id)
            return
        self.drain()                                                        # (what follows prints directly)
        res, hits = {}, None
        if idx.size:
            rows = blk.rows[idx] if idx.size != blk.n else blk.rows
            hits = api.motifseq_multi_batch(rows, blk.nsamp[idx], [np.asarray(self.models[n], dtype=np.float64)
                                                                    for n in self.order],
                                            a.scale, a.scale_low, a.scale_hi)
            res = {int(i): k for k, i in enumerate(idx)}
            # the scoring of MotifSeq.py:441-445 for the whole chunk at once (the same IEEE operations as the
            # per-row arithmetic of emit(), so the same digits), then plain Python numbers for the formatting
            cols = []
            for c in range(len(self.order)):
                h = hits[c]
                mod_mean = (a.slope * self.lens[c]) + a.intercept
                mod_stdev = mod_mean * a.std_const
                with np.errstate(all="ignore"):
                    z = (h["dist"] - mod_mean) / mod_stdev
                    pv = norm_cdf(z)
                    hp = (1 - pv) * 100
                cols.append((h["flags"].tolist(), h["start"].tolist(), h["end"].tolist(), h["dist"].tolist(),
                             mod_mean, mod_stdev, z.tolist(), pv.tolist(), hp.tolist()))
        for i in range(blk.n):
            k = res.get(i)
            if k is not None:
                if a.sig_extract or any(cols[c][0][k] & 3 for c in range(len(self.order))):
                    sig = blk.rows[i, :blk.nsamp[i]] if (a.sig_extract or a.strict_compat) else None   # the general route: -x, flagged reads
                    self.emit(blk.name(i), blk.read_id(i), [hits[c][k] for c in range(len(self.order))], sig, None)
                    continue
                fast5, read_id = blk.name(i), blk.read_id(i)
                for c, name in enumerate(self.order):
                    _, st, en, dist, mm, ms, z, pv, hp = cols[c]
                    print("\t".join((fast5, read_id, name, str(st[k]), str(en[k]), str(en[k] - st[k]), repr(dist[k]),
                                     str(mm), str(ms), repr(z[k]), repr(pv[k]), repr(hp[k]))))
                continue
            fl = int(blk.flags[i])
            if a.after_stall:
                # nothing of this chunk was printed directly, so the batcher alone keeps the file order: reads queue
                # up to --batch per GPU call (segment + search) instead of one call per read
                if (fl & 27) == 1:
                    self.note("No Signal found - please check signal format\n")
                elif (fl & 27) == 3:
                    self.add(blk.name(i), blk.read_id(i), blk.rows[i, :blk.nsamp[i]].astype(np.float64))
                else:
                    fast5, read_id, sig = tsvio.parse_motifseq_line(blk.line(i).decode())
                    if sig.any():
                        self.add(fast5, read_id, sig)
                    else:
                        self.note("No Signal found - please check signal format\n")
                continue
            if (fl & 27) == 1:                                              # integers, all zero: MotifSeq.py:271-273
                sys.stderr.write("No Signal found - please check signal format\n")
                continue
            fast5, read_id, sig = tsvio.parse_motifseq_line(blk.line(i).decode())   # the reference's own parse
            if not sig.any():
                sys.stderr.write("No Signal found - please check signal format\n")
                continue
            self.add(fast5, read_id, sig)
            self.flush()                                                    # keeps the output in file order


def main(argv=None):
    parser = build_parser()
    argv = sys.argv[1:] if argv is None else argv
    args = parser.parse_args(argv)
    if len(argv) == 0:                               # MotifSeq.py:129-131
        parser.print_help(sys.stderr)
        sys.exit(1)
    if args.version:                                 # MotifSeq.py:134-136
        sys.stderr.write("SquiggleKit MotifSeq: {}\n".format(VERSION))
        sys.exit(1)
    if args.verbose:
        sys.stderr.write("Verbose mode active - dumping info to stderr\n")
        sys.stderr.write("SquiggleKit MotifSeq: {}\n".format(VERSION))
        sys.stderr.write("args: {}\n".format(args))
    sys.stderr.write(BANNER)                         # MotifSeq.py:147-151
    if args.view or args.save:
        sys.stderr.write("MotifSeq: -v/--save plotting is not part of this build; ignoring\n")

    _mark("main() entered")
    _STATS[0] = _Stats("MotifSeq")
    del _KEEP[:]                                     # (a previous call in this process: its buffers can go now)
    models, order, lens = load_models(args)
    _mark("models loaded")
    print("\t".join(HEADER + (["normalised_signal"] if args.sig_extract else [])
                    + (["search_from"] if args.after_stall else [])))                  # MotifSeq.py:160-163

    if not (args.f5f or args.f5_path or args.signal or args.blow5 or args.i16):
        sys.stderr.write("Unknown file or path input")
        parser.print_help(sys.stderr)
        sys.exit(1)
    if not order:                                    # nothing to search for: header only
        return

    from . import _lib
    _lib.warm_start(args.device)                    # HIP start-up runs beside the parsing of the first chunk
    if args.gpus > 1:
        api.set_devices(range(args.gpus))
    out = _Batcher(args, models, order, lens)
    if args.signal:
        # native tokenizer (csrc/sk_tsv.cpp): integer lines arrive as int16 rows, one GPU batch per chunk of the
        # file; decimal (pA) chunks go through the float64 tokenizer, odd lines through the reference's own parse
        for blk in tsvio.iter_tsv_blocks(args.signal, 8):
            if isinstance(blk, tsvio.FloatBlock):        # (first line decimal: straight from the float64 tokenizer)
                fb = blk
            else:
                if blk.mostly_integer():
                    out.block(blk)
                    continue
                fb = blk.float_block(8)
                if fb is None:
                    continue
            if fb.clean() and not (args.sig_extract or args.after_stall):
                out.rows_f64(fb)                         # the whole chunk as one batch, no Python per read
                continue
            out.flush()
            for fast5, read_id, vals, fl, raw in tsvio.float_block_lines(fb):
                if fl & 8 or (fl & 16 and raw is not None and raw.count(b"\t") < 1):
                    # odd tokens (or not even a readID column): the reference's own parse, exceptions included
                    fast5, read_id, sig = tsvio.parse_motifseq_line(raw.decode())
                else:
                    sig = vals
                if not sig.any():                        # MotifSeq.py:271-273
                    out.note("No Signal found - please check signal format\n")
                    continue
                out.add(fast5, read_id, sig)
            out.flush()
    elif args.blow5:
        # [extension] BLOW5: records decoded natively into int16 rows (raw ADC values, as the fast5 branches use)
        fast5 = os.path.basename(args.blow5).encode()
        seen = 0
        try:
            for blk in fastio.iter_blow5_blocks_i16(args.blow5, keep=_KEEP):
                bad = np.flatnonzero(blk.flags & 2)
                for i in bad:
                    sys.stderr.write("MotifSeq: unreadable BLOW5 record {} in {}; skipped\n".format(seen + int(i), args.blow5))
                seen += blk.n
                if bad.size:
                    ok = np.flatnonzero((blk.flags & 2) == 0)
                    blk = fastio.Blow5Block(blk.rows[ok], blk.nsamp[ok], blk.ids[ok], blk.calib[ok], blk.flags[ok])
                w = blk.ids.dtype.itemsize
                st = np.arange(blk.n, dtype=np.int64) * w
                spans = np.stack([st, st + np.char.str_len(blk.ids)], axis=1)
                out.rows(blk.rows, blk.nsamp, ("const", fast5), ("span", blk.ids, spans),
                         lambda i: fast5.decode(), lambda i, b=blk: b.ids[i].decode())
        except ValueError as e:                          # truncated file, unsupported compression: say so, no traceback
            out.drain()
            out.flush()
            sys.stdout.flush()
            sys.stderr.write("MotifSeq: --blow5: {}\n".format(e))
            sys.exit(1)
    elif args.i16:
        # [extension] packed reads: int16 [reads, samples] in a .npy file, memory mapped
        fast5 = os.path.basename(args.i16).encode()
        try:
            blocks = fastio.iter_npy_blocks_i16(args.i16, keep=_KEEP)
            for lo, part in blocks:
                ns = np.full(part.shape[0], part.shape[1], dtype=np.int32)
                out.rows(part, ns, ("const", fast5), ("i32", np.arange(lo, lo + part.shape[0], dtype=np.int32)),
                         lambda i: fast5.decode(), lambda i, lo=lo: str(lo + i))
        except ValueError as e:
            sys.stderr.write("MotifSeq: --i16: {}\n".format(e))
            sys.exit(1)
    else:
        if args.f5f:                                 # MotifSeq.py:165-184: first column = path
            with tsvio.open_text(args.f5f) as fh:
                files = [ln.strip("\n").split("\t")[0] for ln in fh]
        else:
            files = [os.path.join(d, f) for d, _, fs in os.walk(args.f5_path) for f in fs if f.endswith(".fast5")]
        for path in files:
            fast5 = path.split("/")[-1]
            sig, read_id = tsvio.motifseq_process_fast5(path, sys.stderr)          # MotifSeq.py:180,211,327-350
            if not len(sig):
                if args.f5f:
                    out.note("Failed to extract signal: {} {}\n".format(path, fast5))              # MotifSeq.py:182
                else:
                    out.note("main():data not extracted. Moving to next file - {}\n".format(path))  # MotifSeq.py:213
                continue
            out.add(fast5, read_id, np.array(sig, dtype=int))
    out.drain()
    out.flush()
    _mark("end of main()")
    _STATS[0].finish(args, [args.signal, getattr(args, "blow5", None), getattr(args, "i16", None)] + list(getattr(args, "ind", None) or []))


if __name__ == "__main__":
    main()
