"""Start the HIP runtime on a background thread before anything heavy is imported (command-line launchers only).

Creating the GPU context costs a few hundred milliseconds, about as long as `import numpy`; the two run side by side
when the launcher calls start() first.  Nothing here is required for correctness: the library initialises itself on
first use whether or not this ran."""
import ctypes
import os
import threading


def _device_from_argv(argv):
    """The --device N (or --device=N) the tool is about to parse, if any: warming device 0 for a job bound elsewhere
    would create a second context (memory, start-up time; a failure on a busy or exclusive GPU 0)."""
    for i, a in enumerate(argv):
        if a == "--device" and i + 1 < len(argv):
            return argv[i + 1]
        if a.startswith("--device="):
            return a.split("=", 1)[1]
    return None


def start(argv=None):
    import sys
    want = _device_from_argv(sys.argv[1:] if argv is None else argv)
    if want is None:
        want = os.environ.get("SK_DEVICE", os.environ.get("LOCAL_RANK", "0")) or 0
    try:
        dev = int(want)
    except ValueError:
        return None                                              # the parser will complain; nothing to warm

    def run():
        try:
            hip = ctypes.CDLL("libamdhip64.so")
            hip.hipInit(0)
            hip.hipSetDevice(dev)
            hip.hipFree(None)                                    # forces the primary context into existence
        except Exception:                                        # noqa: BLE001 -- no runtime here: first use reports it
            pass
    t = threading.Thread(target=run, name="sk-hip-warm", daemon=True)
    t.start()
    return t


def mark(label):
    """Stage timestamps on stderr when SK_T0 (launch time, seconds since the epoch) is set: tools/cli_throughput.py."""
    t0 = os.environ.get("SK_T0")
    if t0:
        import sys
        import time
        sys.stderr.write("[t+%.3f s] %s\n" % (time.time() - float(t0), label))


class Stats:
    """Throughput of a command-line run (SURVEY section 5 "metrics"): reads, GPU calls, wall time, input bytes -> one JSON
    object in the file --stats-json PATH (or $SK_STATS_JSON) names, and one line on stderr with --stats.  The reference
    prints nothing of the kind (its loops, segmenter.py:189-230 / MotifSeq.py:261-298, are one read at a time); the
    side file keeps stdout / stderr byte-identical to the reference's."""

    def __init__(self, tool):
        import time
        self.tool, self.reads, self.calls, self.t0 = tool, 0, 0, time.time()

    def batch(self, nreads):
        self.reads += int(nreads)
        self.calls += 1

    def finish(self, args, inputs):
        import json
        import sys
        import time
        path = getattr(args, "stats_json", None) or os.environ.get("SK_STATS_JSON")
        if not path and not getattr(args, "stats", False):
            return
        dt = time.time() - self.t0
        nbytes = 0
        for f in inputs:
            try:
                nbytes += os.path.getsize(f) if f and os.path.isfile(f) else 0
            except OSError:
                pass
        t_launch = os.environ.get("SK_T0")
        rec = {"tool": self.tool, "reads": self.reads, "gpu_calls": self.calls,
               "reads_per_gpu_call": (self.reads / self.calls) if self.calls else None,
               "seconds_in_main": dt, "reads_per_s": (self.reads / dt) if dt > 0 else None,
               "input_bytes": nbytes, "input_GB_per_s": (nbytes / dt / 1e9) if dt > 0 else None,
               "seconds_since_launch": (time.time() - float(t_launch)) if t_launch else None}
        if path:
            with open(path, "w") as fh:
                json.dump(rec, fh)
        if getattr(args, "stats", False):
            sys.stderr.write("\n[stats] %s: %d reads in %.3f s = %.0f reads/s, %.2f GB/s in, %d GPU calls\n"
                             % (self.tool, self.reads, dt, rec["reads_per_s"] or 0, rec["input_GB_per_s"] or 0, self.calls))


def fast_exit(code=0):
    """Leave a finished command-line tool at once: flush the text streams, then os._exit.  The interpreter's and the
    HIP runtime's orderly teardown (unloading code objects, destroying the context, unmapping the input) costs
    0.1-0.2 s and gives back nothing the exiting process does not give back anyway."""
    import sys
    for st in (sys.stdout, sys.stderr):
        try:
            st.flush()
        except Exception:                                        # noqa: BLE001 -- closed pipe: nothing left to say
            code = code or 1
    os._exit(code)
