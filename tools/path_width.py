"""Width of the optimal path (end - start + 1) / N and distance quantiles on the C4 batch with and without implanted motif copies:
what a shorter first look-back of the window pass would cost.   python tools/path_width.py   (GPU box, repo root)"""
import sys, numpy as np, ctypes as C
sys.path.insert(0, ".")
from squigglekit_amd import _lib, synth
from squigglekit_amd._lib import HIT_DTYPE, check, ptr, SynthOpts
L = _lib.load(); _lib.init(0)
R, M, N = 200000, 4000, 200
motif = synth.synthetic_motif(N)
stride = M
d_sig = L.sk_dev_alloc(R * stride * 2); d_len = L.sk_dev_alloc(R * 4); d_out = L.sk_dev_alloc(R * 24)
lens = np.full(R, M, dtype=np.int32); check(L.sk_dev_upload(d_len, ptr(lens), lens.nbytes))
for label, opts in (("C4 synthetic", dict()), ("no implants", dict(hit_permille=0))):
    o = SynthOpts(**opts)
    check(L.sk_synth_variant_dev(d_sig, stride, R, M, synth.SEED_C4, ptr(motif), N, C.byref(o)))
    check(L.sk_motifseq_dev_i16(d_sig, stride, d_len, R, ptr(motif), N, 0, 0, 1200, d_out)); check(L.sk_sync())
    hits = np.empty(R, dtype=HIT_DTYPE); check(L.sk_dev_download(ptr(hits), d_out, hits.nbytes))
    w = (hits["end"] - hits["start"] + 1) / N
    d = hits["dist"]
    print(label, "width/N quantiles 5,25,50,75,90,95,99,99.9:", np.round(np.quantile(w, [.05,.25,.5,.75,.9,.95,.99,.999]), 3))
    print("   dist quantiles:", np.round(np.quantile(d, [.01,.05,.25,.5,.75,.95,.99]), 2))
    lo = d < np.median(d) * 0.6
    print("   share with small dist:", lo.mean(), "their width q50/q99:", np.round(np.quantile(w[lo], [.5,.99]),3) if lo.any() else None,
          "others width q50/q90/q99:", np.round(np.quantile(w[~lo], [.5,.9,.99]), 3))
    for s in (0.5, 0.6, 0.7, 0.8, 0.9):
        print("   non-small-dist reads with width > %.1f N: %.3f" % (s, (w[~lo] > s).mean()))
