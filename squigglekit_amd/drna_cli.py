"""Drop-in for /root/reference/dRNA_segmenter.py's command line (dRNA_segmenter.py:56-176).

`-f/--slow5` works (BLOW5 or ASCII SLOW5 through this package's own reader; the reference needs
pyslow5): per read scale_outliers + window statistics + the adapter scan run on the GPU, and the
first segment is printed as `readID<TAB>start<TAB>end` like the reference does.

`-s/--signal` (dRNA_segmenter.py:272-326: rolling mean of the filtered signal, segments where it stays
below mean - 0.5 std): the reference stops at its first read, because it uses the window `w` before
assigning it (:282; `# w = 2000` at :81 is commented out).  Here the branch runs with `-w/--window`
(default 2000, the commented-out value) and prints `fast5<TAB>readID<TAB>start<TAB>end` as the script
would; `--strict-compat` reproduces the reference's failure instead."""
import argparse
import sys

import numpy as np

from . import api


class _Parser(argparse.ArgumentParser):
    def error(self, message):
        sys.stderr.write("error: %s\n" % message)
        self.print_help()
        sys.exit(2)


def _signal_branch(args):
    """dRNA_segmenter.py:272-326"""
    import numpy as np
    from . import _lib
    from .tsvio import iter_tsv_native
    if args.strict_compat:
        sys.stderr.write("Traceback (most recent call last):\n  ...\n"
                         "UnboundLocalError: local variable 'w' referenced before assignment\n")
        sys.exit(1)
    if args.window <= 0:
        sys.stderr.write("error: -w/--window must be positive\n")
        sys.exit(2)
    _lib.init(args.device)
    params = _lib.RollParams(w=args.window)
    names, sigs = [], []

    def flush():
        if not sigs:
            return
        for (f5, rid), res in zip(names, api.drna_roll_reads(sigs, params)):
            if res:                                     # :318-326: the first acceptable segment, shifted
                print("{}\t{}\t{}\t{}".format(f5, rid, res[0], res[1]))
        names.clear()
        sigs.clear()

    for f5, rid, values, _flags, _raw in iter_tsv_native(args.signal, args.start_col):
        sig = api.as_int16_exact(values)                # the reference does int(i) on every token (:279)
        if sig is None:
            raise ValueError("dRNA_segmenter --signal expects raw integer samples (read %s)" % rid)
        names.append((f5, rid))
        sigs.append(np.ascontiguousarray(sig))
        if len(sigs) >= args.batch:
            flush()
    flush()


def main(argv=None):
    p = _Parser(description="dRNA_segmenter (MI355X) - locate the adapter stretch at the start of dRNA reads")
    p.add_argument("-s", "--signal", help="signal TSV (rolling-mean branch)")
    p.add_argument("-w", "--window", type=int, default=2000,
                   help="[extension] rolling window of the --signal branch (the reference leaves it undefined; "
                        "its commented-out default is 2000)")
    p.add_argument("--strict-compat", action="store_true",
                   help="[extension] --signal fails the way the reference does (w used before assignment)")
    p.add_argument("-f", "--slow5", help="SLOW5 / BLOW5 file")
    p.add_argument("-c", "--start_col", type=int, default=4, help="first signal column of a TSV")
    p.add_argument("-p", "--plot", action="store_true", help="plot each read (not available in this build)")
    p.add_argument("--device", type=int, default=None, help="[extension] GPU index")
    p.add_argument("--batch", type=int, default=512, help="[extension] reads per GPU call")
    argv = sys.argv[1:] if argv is None else argv
    args = p.parse_args(argv)
    if len(argv) == 0:
        p.print_help(sys.stderr)
        sys.exit(1)
    if not args.slow5:
        if not args.signal:
            p.print_help(sys.stderr)
            sys.exit(1)
        return _signal_branch(args)
    if args.plot:
        sys.stderr.write("dRNA_segmenter: -p/--plot is not part of this build; ignoring\n")
    from . import _lib
    from .blow5 import read_slow5
    _lib.init(args.device)
    ids, sigs = [], []

    def flush():
        if not sigs:
            return
        for rid, segs in zip(ids, api.drna_segment_reads(sigs)):
            if segs:                                    # dRNA_segmenter.py:173-176: first segment only
                print("{}\t{}\t{}".format(rid, segs[0][0], segs[0][1]))
        ids.clear()
        sigs.clear()

    with open(args.slow5, "rb") as fh:
        binary = fh.read(6) == b"BLOW5\x01"
    if binary:
        # BLOW5: records decoded natively into int16 rows, a block (16 384 reads) per GPU call
        from . import fastio
        from ._lib import DrnaParams
        try:
            for blk in fastio.iter_blow5_blocks_i16(args.slow5):
                for i in np.flatnonzero(blk.flags & 2):
                    sys.stderr.write("dRNA_segmenter: unreadable BLOW5 record in {}; skipped\n".format(args.slow5))
                segs, nsegs = api.drna_segment_batch(blk.rows, blk.nsamp, DrnaParams())
                for i in np.flatnonzero((nsegs > 0) & ((blk.flags & 2) == 0)):
                    print("{}\t{}\t{}".format(blk.ids[i].decode(), segs[i, 0, 0], segs[i, 0, 1]))
        except ValueError as e:
            sys.stderr.write("dRNA_segmenter: -f: {}\n".format(e))
            sys.exit(1)
        return
    for rec in read_slow5(args.slow5):
        ids.append(rec["read_id"])
        sigs.append(rec["signal"])
        if len(sigs) >= args.batch:
            flush()
    flush()


if __name__ == "__main__":
    main()
