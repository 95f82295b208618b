"""Drop-in for /root/reference/dRNA_segmenter.py's command line (dRNA_segmenter.py:56-176).

`-f/--slow5` works (BLOW5 or ASCII SLOW5 through this package's own reader; the reference needs
pyslow5): per read scale_outliers + window statistics + the adapter scan run on the GPU, and the
first segment is printed as `readID<TAB>start<TAB>end` like the reference does.  The reference's
`-s/--signal` branch cannot run as shipped (it uses an undefined `w`, dRNA_segmenter.py:281 with
:81 commented out), so here it reports that instead of guessing a window."""
import argparse
import sys

from . import api


class _Parser(argparse.ArgumentParser):
    def error(self, message):
        sys.stderr.write("error: %s\n" % message)
        self.print_help()
        sys.exit(2)


def main(argv=None):
    p = _Parser(description="dRNA_segmenter (MI355X) - locate the adapter stretch at the start of dRNA reads")
    p.add_argument("-s", "--signal", help="signal TSV (the reference's branch for it is broken; see --help text)")
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
        sys.stderr.write("dRNA_segmenter: the reference's TSV branch uses an undefined window `w` "
                         "(dRNA_segmenter.py:281) and cannot run; use -f <slow5/blow5>\n")
        sys.exit(1)
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

    for rec in read_slow5(args.slow5):
        ids.append(rec["read_id"])
        sigs.append(rec["signal"])
        if len(sigs) >= args.batch:
            flush()
    flush()


if __name__ == "__main__":
    main()
