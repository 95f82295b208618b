"""ctypes binding of libsquigglekit_hip.so (the C ABI in include/squigglekit_hip.h).

There is no CPU fallback: if the shared library is missing, or no gfx950 device is
visible, every compute call raises.  Nothing here imports torch or the oracle.
"""
import ctypes as C
import os
import subprocess
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("SK_LIB_PATH") or os.path.join(_HERE, "libsquigglekit_hip.so")   # (override: A/B builds)
CSRC = os.path.join(_HERE, "csrc")


class SquiggleKitError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("squigglekit_hip error %d: %s" % (code, msg))
        self.code = code


# sk_status (include/squigglekit_hip.h)
SK_OK, SK_ERR_INVALID, SK_ERR_NO_DEVICE, SK_ERR_HIP, SK_ERR_NOMEM, SK_ERR_UNSUPPORTED, SK_ERR_OVERFLOW = 0, -1, -2, -3, -4, -5, -6
SK_SCALE = {"medmad": 0, "zscale": 1}
SK_FLAG_EMPTY, SK_FLAG_DEGENERATE, SK_FLAG_RECENTRE = 1, 2, 4


class SegParams(C.Structure):
    """sk_seg_params; defaults are the argparse defaults of segmenter.py:65-96."""
    _fields_ = [("error", C.c_int32), ("corrector", C.c_int32), ("window", C.c_int32),
                ("seg_dist", C.c_int32), ("std_scale", C.c_double), ("stall_len", C.c_double),
                ("lim_low", C.c_int32), ("lim_hi", C.c_int32)]

    def __init__(self, error=5, corrector=50, window=150, seg_dist=50, std_scale=0.75,
                 stall_len=0.25, lim_low=0, lim_hi=900):
        super().__init__(error, corrector, window, seg_dist, std_scale, stall_len, lim_low, lim_hi)

    @classmethod
    def from_args(cls, args):
        """Build from an argparse Namespace shaped like segmenter.py's."""
        return cls(args.error, args.corrector, args.window, args.seg_dist, args.std_scale,
                   args.stall_len, getattr(args, "lim_low", 0), getattr(args, "lim_hi", 900))


class DrnaParams(C.Structure):
    """sk_drna_params; defaults are the constants hard-coded at dRNA_segmenter.py:80-104."""
    _fields_ = [("error", C.c_int32), ("no_err_thresh", C.c_int32), ("w", C.c_int32),
                ("window", C.c_int32), ("seg_dist", C.c_int32), ("t_start", C.c_int32),
                ("t_end", C.c_int32), ("std_scale", C.c_double), ("lim_low", C.c_int32),
                ("lim_hi", C.c_int32)]

    def __init__(self, error=5, no_err_thresh=2500, w=1200, window=100, seg_dist=1200, t_start=1000,
                 t_end=5000, std_scale=0.8, lim_low=0, lim_hi=1200):
        super().__init__(error, no_err_thresh, w, window, seg_dist, t_start, t_end, std_scale,
                         lim_low, lim_hi)


class RollParams(C.Structure):
    """sk_roll_params: the constants of dRNA_segmenter.py:288-295,322 and the rolling window `w` the
    script reads before assigning (its commented-out default, :81, is 2000)."""
    _fields_ = [("w", C.c_int32), ("seg_dist", C.c_int32), ("hi_thresh", C.c_int32), ("lo_thresh", C.c_int32),
                ("shift", C.c_int32), ("std_scale", C.c_double), ("lim_low", C.c_int32), ("lim_hi", C.c_int32)]

    def __init__(self, w=2000, seg_dist=1500, hi_thresh=200000, lo_thresh=2000, shift=1000, std_scale=0.5,
                 lim_low=0, lim_hi=1200):
        super().__init__(w, seg_dist, hi_thresh, lo_thresh, shift, std_scale, lim_low, lim_hi)


class SynthOpts(C.Structure):
    """sk_synth_opts (bench tooling): slice / variant of the device generator's batch."""
    _fields_ = [("row0", C.c_int64), ("hit_permille", C.c_int32), ("stretch_permille", C.c_int32),
                ("stretch", C.c_int32), ("tmpl", C.c_void_p), ("ntmpl", C.c_int32), ("tmpl_noise", C.c_double)]

    def __init__(self, row0=0, hit_permille=500, stretch_permille=0, stretch=1, tmpl=None, tmpl_noise=0.0):
        self._keep = None if tmpl is None else np.ascontiguousarray(tmpl, dtype=np.int16)
        super().__init__(row0, hit_permille, stretch_permille, stretch,
                         None if tmpl is None else self._keep.ctypes.data, 0 if tmpl is None else self._keep.size,
                         tmpl_noise)


class Hit(C.Structure):
    _fields_ = [("dist", C.c_double), ("start", C.c_int32), ("end", C.c_int32),
                ("n", C.c_int32), ("flags", C.c_int32)]


HIT_DTYPE = np.dtype([("dist", "<f8"), ("start", "<i4"), ("end", "<i4"),
                      ("n", "<i4"), ("flags", "<i4")])

# every symbol include/squigglekit_hip.h declares: name -> (restype, argtypes)
_vp, _i16p, _i32p, _i64p, _dp = (C.c_void_p, C.POINTER(C.c_int16), C.POINTER(C.c_int32),
                                 C.POINTER(C.c_int64), C.POINTER(C.c_double))
ABI = {
    "sk_version": (C.c_char_p, []),
    "sk_last_error": (C.c_char_p, []),
    "sk_device_count": (C.c_int, []),
    "sk_init": (C.c_int, [C.c_int]),
    "sk_init_slot": (C.c_int, [C.c_int, C.c_int]),
    "sk_shutdown": (C.c_int, []),
    "sk_sync": (C.c_int, []),
    "sk_device_name": (C.c_int, [C.c_char_p, C.c_int]),
    "sk_device_pci_bus_id": (C.c_int, [C.c_char_p, C.c_int]),
    "sk_dev_alloc": (_vp, [C.c_size_t]),
    "sk_dev_free": (C.c_int, [_vp]),
    "sk_dev_upload": (C.c_int, [_vp, _vp, C.c_size_t]),
    "sk_dev_download": (C.c_int, [_vp, _vp, C.c_size_t]),
    "sk_host_alloc": (_vp, [C.c_size_t]),
    "sk_host_free": (C.c_int, [_vp]),
    "sk_segment_batch_i16": (C.c_int, [_vp, C.c_int64, _vp, C.c_int32, C.POINTER(SegParams),
                                       _vp, _vp, C.c_int32]),
    "sk_segment_batch_f64": (C.c_int, [_vp, _vp, C.c_int32, C.POINTER(SegParams), _vp, _vp, C.c_int32]),
    "sk_segment_batch_i16_pa": (C.c_int, [_vp, C.c_int64, _vp, C.c_int32, _vp, C.POINTER(SegParams), _vp, _vp, C.c_int32]),
    "sk_pa_calib": (C.c_int, [_vp, C.c_int32, _vp]),
    "sk_segment_dev_i16_pa": (C.c_int, [_vp, C.c_int64, _vp, C.c_int32, _vp, C.POINTER(SegParams), _vp, _vp, C.c_int32]),
    "sk_last_pa_retries": (C.c_int, []),
    "sk_segment_batch_f64_len": (C.c_int, [_vp, _vp, _vp, C.c_int32, C.POINTER(SegParams), _vp, _vp, C.c_int32]),
    "sk_segment_batch_centi_len": (C.c_int, [_vp, _vp, _vp, C.c_int32, C.POINTER(SegParams), _vp, _vp, C.c_int32]),
    "sk_segment_dev_i16": (C.c_int, [_vp, C.c_int64, _vp, C.c_int32, C.POINTER(SegParams),
                                     _vp, _vp, C.c_int32]),
    "sk_segment_dev_f64": (C.c_int, [_vp, _vp, C.c_int32, C.c_int64, C.c_int64, C.POINTER(SegParams), _vp, _vp, C.c_int32]),
    "sk_drna_segment_batch_i16": (C.c_int, [_vp, C.c_int64, _vp, C.c_int32, _vp, _vp, _vp, C.c_int32]),
    "sk_drna_roll_batch_i16": (C.c_int, [_vp, C.c_int64, _vp, C.c_int32, _vp, _vp, _vp]),
    "sk_drna_segment_dev_i16": (C.c_int, [_vp, C.c_int64, _vp, C.c_int32, _vp, _vp, _vp, C.c_int32]),
    "sk_drna_roll_dev_i16": (C.c_int, [_vp, C.c_int64, _vp, C.c_int32, _vp, _vp, _vp]),
    "sk_motifseq_batch_i16": (C.c_int, [_vp, C.c_int64, _vp, C.c_int32, _vp, C.c_int32, C.c_int32,
                                        C.c_int32, C.c_int32, _vp]),
    "sk_motifseq_multi_batch_i16": (C.c_int, [_vp, C.c_int64, _vp, C.c_int32, _vp, _vp, C.c_int32, C.c_int32,
                                              C.c_int32, C.c_int32, _vp]),
    "sk_motifseq_batch_f64": (C.c_int, [_vp, _vp, C.c_int32, _vp, C.c_int32, C.c_int32,
                                        C.c_int32, C.c_int32, _vp]),
    "sk_motifseq_multi_batch_f64": (C.c_int, [_vp, _vp, C.c_int32, _vp, _vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _vp]),
    "sk_motifseq_multi_batch_centi": (C.c_int, [_vp, _vp, C.c_int32, _vp, _vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _vp]),
    "sk_motifseq_dev_i16": (C.c_int, [_vp, C.c_int64, _vp, C.c_int32, _vp, C.c_int32, C.c_int32,
                                      C.c_int32, C.c_int32, _vp]),
    "sk_motifseq_multi_dev_i16": (C.c_int, [_vp, C.c_int64, _vp, C.c_int32, _vp, _vp, C.c_int32, C.c_int32,
                                            C.c_int32, C.c_int32, _vp]),
    "sk_motifseq_dev_f64": (C.c_int, [_vp, _vp, C.c_int32, C.c_int64, C.c_int64, _vp, C.c_int32, C.c_int32,
                                      C.c_int32, C.c_int32, _vp]),
    "sk_dtw_subsequence_batch": (C.c_int, [_vp, C.c_int32, _vp, _vp, C.c_int32, _vp]),
    "sk_dtw_subsequence": (C.c_int, [_vp, C.c_int32, _vp, C.c_int32, _dp, _i32p, _i32p, _vp]),
    "sk_dtw_subsequence_cref": (C.c_int, [_vp, C.c_int32, _vp, C.c_int32, _dp, _i32p, _i32p]),
    "sk_normalise_i16": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _vp, _i32p]),
    "sk_normalise_f64": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _vp, _i32p]),
    "sk_tsv_count_lines": (C.c_int64, [_vp, C.c_size_t]),
    "sk_tsv_count_tokens": (C.c_int, [_vp, C.c_size_t, C.c_int32, C.c_int64, _vp, C.c_int32]),
    "sk_tsv_parse": (C.c_int, [_vp, C.c_size_t, C.c_int32, C.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                               C.c_int32]),
    "sk_tsv_parse_centi": (C.c_int, [_vp, C.c_size_t, C.c_int32, C.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                     C.c_int32]),
    "sk_tsv_parse_i16": (C.c_int, [_vp, C.c_size_t, C.c_int32, C.c_int64, C.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                   _vp, C.c_int32]),
    "sk_fmt_rows": (_vp, [C.c_int64, C.c_int32, _vp, _vp, C.c_int32, _i64p]),
    "sk_fmt_free": (None, [_vp]),
    "sk_ndtr": (None, [_vp, _vp, C.c_int64]),
    "sk_blow5_index": (C.c_int64, [_vp, C.c_int64, C.c_int64, _vp, _vp, C.c_int64]),
    "sk_blow5_index_some": (C.c_int64, [_vp, C.c_int64, C.c_int64, C.c_int64, _vp, _vp, _vp]),
    "sk_blow5_rows_i16": (C.c_int, [_vp, C.c_int64, _vp, _vp, C.c_int64, C.c_int32, C.c_int64, _vp, _vp, _vp, C.c_int32, _vp, _vp,
                                    C.c_int32]),
    "sk_comm_unique_id": (C.c_int, [_vp]),
    "sk_comm_init_rank": (C.c_int, [_vp, C.c_int, C.c_int]),
    "sk_comm_init_all": (C.c_int, [_i32p, C.c_int]),
    "sk_comm_info": (C.c_int, [_i32p, _i32p]),
    "sk_comm_allgather_dev": (C.c_int, [_vp, _vp, C.c_size_t]),
    "sk_comm_allgather_host": (C.c_int, [_vp, _vp, C.c_size_t]),
    "sk_comm_destroy": (C.c_int, []),
    "sk_tunables": (C.c_int, [C.c_char_p, C.c_int]),
    "sk_last_kernel_ms": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "sk_last_dtw_retries": (C.c_int, []),
    "sk_last_dtw_tier2": (C.c_int, []),
    "sk_last_dtw_guard": (C.c_int, [_i32p]),
    "sk_last_dtw_window_steps": (C.c_int, [_vp]),
    "sk_last_dtw_premise_violations": (C.c_int, []),
    "sk_last_dtw_audit_mismatches": (C.c_int, []),
    "sk_last_f64_retries": (C.c_int, []),
    "sk_last_dtw_clock": (C.c_int, [_dp]),
    "sk_last_dtw_profile": (C.c_int, [C.POINTER(C.c_float), _i32p, C.POINTER(C.c_float), _i32p, _i32p]),
    "sk_synth_squiggles_dev": (C.c_int, [_vp, C.c_int64, C.c_int32, C.c_int32, C.c_uint64, _vp, C.c_int32]),
    "sk_synth_variant_dev": (C.c_int, [_vp, C.c_int64, C.c_int32, C.c_int32, C.c_uint64, _vp, C.c_int32, _vp]),
    "sk_synth_pa_dev": (C.c_int, [_vp, C.c_int64, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, _vp, _vp]),
}


def build(force=False):
    """Compile the HIP sources for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, "-s", "-j8"]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return SO_PATH


_lib = None


def load():
    """dlopen the library and bind every ABI symbol.  Raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise SquiggleKitError(-2, "%s not built (run `python -c 'import __graft_entry__ as g; "
                                   "g.build()'` or `make -C squigglekit_amd/csrc`); there is no CPU "
                                   "fallback" % SO_PATH)
        L = C.CDLL(SO_PATH)
        for name, (res, argt) in ABI.items():
            fn = getattr(L, name)            # AttributeError here == header/library drift
            fn.restype = res
            fn.argtypes = argt
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise SquiggleKitError(rc, load().sk_last_error().decode(errors="replace"))


_tls = threading.local()      # the library binds a device per host thread (thread_local in sk_runtime.hip)


def init(device=None, slot=None):
    """Bind the calling thread to a GPU (default: $SK_DEVICE, else LOCAL_RANK, else 0).  `slot`: an explicit
    context slot (sk_init_slot) -- several threads sharing one GPU each need their own."""
    L = load()
    if device is None:
        device = int(os.environ.get("SK_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    if slot is None:
        check(L.sk_init(int(device)))
    else:
        check(L.sk_init_slot(int(slot), int(device)))
    _tls.device = int(device)
    _ready.set()
    return _tls.device


_ready = threading.Event()


def is_ready():
    """Has some thread of this process already bound a GPU (so that init() costs nothing now)?"""
    return _ready.is_set()


def warm_start(device=None, also=()):
    """Start binding the GPU on a background thread (the HIP runtime's start-up costs a few hundred milliseconds that a
    command-line tool can spend parsing its first chunk of input) and import the modules named in `also` there too.
    Every thread binds for itself later (init / ensure_init), by then at no cost.  Returns the thread."""
    if device is not None:
        os.environ["SK_DEVICE"] = str(int(device))

    def run():
        try:
            init(device)
            for mod in also:
                __import__(mod)
        except Exception:                                            # noqa: BLE001 -- the foreground call reports it
            pass
    t = threading.Thread(target=run, name="sk-warm", daemon=True)
    t.start()
    return t


def ensure_init():
    if getattr(_tls, "device", None) is None:
        init()
    return load()


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


# Host-side tuning switches (the native ones live in csrc/sk_runtime.hip: SK_TUNABLES): name -> (values the parity
# test flips it to, description).  Like the native ones they are read only when SK_TUNING=1 is set as well.
PY_TUNABLES = {
    "SK_BLOW5_PIN": ("1", "BLOW5 reader: page-locked streaming buffers"),
    "SK_BLOW5_BLOCK": ("64 5000", "BLOW5 reader: records per block"),
    "SK_BLOW5_ZAP": ("0", "BLOW5 reader: keep the consumed pages of the file map"),
    "SK_I16_PIN": ("1", "--i16 reader: page-locked streaming buffers"),
    "SK_I16_BLOCK_MB": ("1 8", "--i16 reader: block size in MB"),
    "SK_TSV_THREADS": ("4 64", "TSV reader: worker threads of the native tokenizer (default 32)"),
    "SK_TSV_NO_CENTI": ("1", "TSV reader: decimal lines through the float64 tokenizer even when every token has at most two decimals"),
}


def tune(name, default=None):
    """Value of a host-side tuning switch, or `default` when it is unset or SK_TUNING=1 is not set."""
    if os.environ.get("SK_TUNING", "")[:1] != "1" or name not in PY_TUNABLES:
        return default
    return os.environ.get(name, default)


def tunables():
    """Every tuning switch: {name: (test values, description)} -- the native table plus PY_TUNABLES."""
    L = load()
    n = L.sk_tunables(None, 0)
    buf = C.create_string_buffer(n)
    L.sk_tunables(buf, n)
    out = {}
    for line in buf.value.decode().splitlines():
        name, vals, what = line.split("\t")
        out[name] = (vals, what)
    out.update(PY_TUNABLES)
    return out
