"""Reads sharded over the GPUs of one node, and the one exchange the path has: the final gather of the
fixed-size result records over RCCL (xGMI).

Reads are independent in both scripts (segmenter.py:189-230, MotifSeq.py:261-298 carry no state from one
read to the next), so N GPUs take contiguous blocks of the reads (`sharding.shard_bounds`) and never talk
while computing.  Two launch shapes:

  * one process, one host thread per GPU (`ThreadGroup`): what `api.*(devices=[...])`, the CLIs' `--gpus` and
    a plain `python bench.py --gpus N` use.  The library binds a device per thread; ctypes releases the GIL.
  * one process per GPU (`ProcessGroup`): what `python -m torch.distributed.run ... bench.py` gives.  Only the
    launcher's environment (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR / MASTER_PORT) is read; rank 0's ncclUniqueId
    travels through a small file store under $TMPDIR (one node), no torch import anywhere.

The gather is `ncclAllGather` inside the library (`sk_comm_*`, csrc/sk_comm.hip).  If librccl cannot be loaded
or a communicator cannot be created, the same interface is served by host concatenation (`backend == "host"`).
"""
import ctypes as C
import os
import shutil
import tempfile
import threading
import time

import numpy as np

from . import _lib, sharding
from ._lib import check, ptr

UID_BYTES = 128


# ------------------------------------------------------------------------------------------------
# launch-shape detection
# ------------------------------------------------------------------------------------------------
def launch_env(environ=None):
    """(rank, local_rank, world) when a one-process-per-GPU launcher set the usual variables, else None."""
    e = os.environ if environ is None else environ
    try:
        world = int(e.get("WORLD_SIZE", "1"))
    except ValueError:
        return None
    if world <= 1 or "RANK" not in e:
        return None
    try:
        rank = int(e["RANK"])
        return rank, int(e.get("LOCAL_RANK", rank)), world
    except ValueError:
        return None


def plan(gpus, environ=None):
    """How a job asking for `gpus` GPUs runs here: ("single", 0, 0, 1), ("threads", 0, 0, N) -- this process
    drives N devices -- or ("process", rank, local_rank, world) under a per-GPU launcher (whose WORLD_SIZE wins
    over a conflicting --gpus)."""
    le = launch_env(environ)
    if le is not None:
        return ("process",) + le
    e = os.environ if environ is None else environ
    if e.get("SK_FORCE_PROCESS_SHAPE") == "1" and "RANK" in e:       # tests: the per-GPU-process shape with one rank
        try:
            return ("process", int(e["RANK"]), int(e.get("LOCAL_RANK", "0")), max(1, int(e.get("WORLD_SIZE", "1"))))
        except ValueError:
            pass
    if gpus and int(gpus) > 1:
        return ("threads", 0, 0, int(gpus))
    return ("single", 0, 0, 1)


# ------------------------------------------------------------------------------------------------
# a tiny file store: rendezvous for the process-per-GPU shape, and the host fallback exchange
# ------------------------------------------------------------------------------------------------
def oversubscribed(environ=None):
    """The device every rank shares when SK_OVERSUBSCRIBE is set (SK_OVERSUBSCRIBE=1: device 0; =dN: device N),
    else None.  Oversubscription is the dry run of the N > 1 paths on a box with fewer GPUs than ranks: every rank
    gets its own context slot (stream, scratch) on the shared device; the gather runs on the host backend because
    RCCL wants one device per rank."""
    e = os.environ if environ is None else environ
    v = e.get("SK_OVERSUBSCRIBE", "")
    if not v or v == "0":
        return None
    if v[:1] == "d" and v[1:].isdigit():
        return int(v[1:])
    return 0


def store_dir(environ=None):
    """One directory per job on this node: keyed on what every rank of a launch shares whatever started it --
    the rendezvous address and port (unique among the jobs running on a node) and the launcher's run id.  A stale
    directory of an earlier job with the same key is harmless: nothing in it carries this job's session id
    (FileStore.rendezvous)."""
    e = os.environ if environ is None else environ
    if e.get("SK_RDZV_DIR"):
        return e["SK_RDZV_DIR"]
    tag = "sk_rdzv_%d_%s_%s_%s" % (os.getuid(), e.get("MASTER_ADDR", "local").replace("/", "_"),
                                  e.get("MASTER_PORT", "0"), e.get("TORCHELASTIC_RUN_ID", "none").replace("/", "_"))
    return os.path.join(tempfile.gettempdir(), tag)


class FileStore:
    """Atomic small-file exchange between the ranks of one node.

    rendezvous() first agrees on a session id so that files a crashed earlier job left under the same key are never
    taken for this job's: every rank publishes a fresh nonce, rank 0 publishes {session, payload, the nonces it has
    seen} and repeats that until every rank has acknowledged a version carrying its own current nonce."""

    def __init__(self, path, rank, world, timeout=300.0):
        self.path, self.rank, self.world, self.timeout = path, rank, world, timeout
        # the directory sits in a shared $TMPDIR under a predictable name: it must be ours, private and not a link
        try:
            os.makedirs(path, mode=0o700, exist_ok=True)
            st = os.lstat(path)
        except OSError as e:
            raise RuntimeError("rendezvous directory %s cannot be used: %s" % (path, e))
        import stat as _stat
        if not _stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o022):
            raise RuntimeError("rendezvous directory %s is not a private directory of uid %d (set SK_RDZV_DIR)"
                               % (path, os.getuid()))
        self._seq = 0
        self.session = None

    def put(self, key, data):
        tmp = os.path.join(self.path, ".%s.%d.tmp" % (key, self.rank))
        with open(tmp, "wb") as fh:
            fh.write(data)
        os.replace(tmp, os.path.join(self.path, key))

    def peek(self, key):
        try:
            with open(os.path.join(self.path, key), "rb") as fh:
                return fh.read()
        except FileNotFoundError:
            return None

    def get(self, key, accept=None):
        t0, nap = time.monotonic(), 0.0005
        while True:
            data = self.peek(key)
            if data is not None and (accept is None or accept(data)):
                return data
            if time.monotonic() - t0 > self.timeout:
                raise TimeoutError("rank %d: nothing usable at %s/%s after %.0f s"
                                   % (self.rank, self.path, key, self.timeout))
            time.sleep(nap)
            nap = min(nap * 2, 0.02)

    def rendezvous(self, payload=b""):
        """Agree on a session; rank 0's `payload` (e.g. the ncclUniqueId) reaches every rank.  Returns it."""
        nonce = os.urandom(8).hex().encode()
        self.put("hello.%d" % self.rank, nonce)
        if self.rank == 0:
            session = os.urandom(8).hex().encode()
            t0 = time.monotonic()
            while True:
                seen = [self.peek("hello.%d" % r) or b"-" for r in range(self.world)]
                self.put("session", b"|".join([session, payload.hex().encode()] + seen))
                time.sleep(0.002)
                acks = [self.peek("ack.%d" % r) for r in range(1, self.world)]
                if all(a == session for a in acks):
                    break
                if time.monotonic() - t0 > self.timeout:
                    raise TimeoutError("rank 0: ranks %s never joined %s"
                                       % ([r + 1 for r, a in enumerate(acks) if a != session], self.path))
            self.session = session.decode()
            return payload
        blob = self.get("session", accept=lambda d: d.split(b"|")[2 + self.rank:3 + self.rank] == [nonce])
        parts = blob.split(b"|")
        self.put("ack.%d" % self.rank, parts[0])
        self.session = parts[0].decode()
        return bytes.fromhex(parts[1].decode())

    def allgather(self, payload):
        if self.session is None:
            self.rendezvous()
        self._seq += 1
        self.put("ag.%s.%d.%d" % (self.session, self._seq, self.rank), payload)
        return [self.get("ag.%s.%d.%d" % (self.session, self._seq, r)) for r in range(self.world)]

    def close(self):
        """Every rank leaves a note; rank 0 waits for all of them (nobody reads the store any more), then removes
        the directory.  The other ranks wait for nothing, so the removal cannot strand them."""
        if self.session is None:
            return
        self.put("bye.%s.%d" % (self.session, self.rank), b"")
        if self.rank == 0:
            for r in range(self.world):
                self.get("bye.%s.%d" % (self.session, r))
            shutil.rmtree(self.path, ignore_errors=True)


# ------------------------------------------------------------------------------------------------
# communicators (one per rank); both launch shapes give the same interface
# ------------------------------------------------------------------------------------------------
class RankComm:
    """rank / world / backend ("rccl" or "host"), barrier(), allgather_host(array) -> [world, ...],
    allgather_dev(d_send, d_recv, nbytes) (RCCL only), ranks_seen()."""

    def __init__(self, rank, world, backend, host_exchange):
        self.rank, self.world, self.backend = rank, world, backend
        self._hx = host_exchange                                     # bytes -> list of bytes (rank order)

    def ranks_seen(self):
        """The communicator size as RCCL reports it (ncclCommCount); for the host fallback, the ranks that
        actually answered an exchange."""
        if self.backend == "rccl":
            n, r = C.c_int32(), C.c_int32()
            check(_lib.load().sk_comm_info(C.byref(n), C.byref(r)))
            assert r.value == self.rank
            return n.value
        return len(self._hx(b"x"))

    def all_ok(self, ok):
        """True when every rank reports ok.  Always a host-side exchange (thread barrier / file store), so it can be
        called right before a device collective: a rank that failed must not leave the others inside ncclAllGather."""
        return all(p == b"\x01" for p in self._hx(b"\x01" if ok else b"\x00"))

    def allgather_host(self, arr):
        a = np.ascontiguousarray(arr)
        if self.backend == "rccl":
            out = np.empty((self.world,) + a.shape, dtype=a.dtype)
            check(_lib.load().sk_comm_allgather_host(ptr(a), ptr(out), a.nbytes))
            return out
        parts = self._hx(a.tobytes())
        return np.stack([np.frombuffer(p, dtype=a.dtype).reshape(a.shape) for p in parts])

    def barrier(self):
        self.allgather_host(np.zeros(1, dtype=np.int64))

    def allgather_dev(self, d_send, d_recv, nbytes):
        """Every rank's `nbytes` at d_send -> d_recv[world * nbytes] on every rank, enqueued on the library's
        stream (wait with sk_sync).  Device pointers; RCCL backend only."""
        if self.backend != "rccl":
            raise _lib.SquiggleKitError(-5, "device all-gather needs the RCCL backend (this group fell back to "
                                            "host concatenation)")
        check(_lib.load().sk_comm_allgather_dev(d_send, d_recv, nbytes))

    def close(self):
        if self.backend == "rccl":
            check(_lib.load().sk_comm_destroy())


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def bind_thread_near_device():
    """Pin the calling thread (and the threads it starts: the tokenizer's, the formatter's) to the CPUs the bound GPU
    hangs off -- /sys/bus/pci/devices/<bus id>/local_cpulist -- so that a rank's host buffers are first touched, and
    its copies fed, from the GPU's own NUMA node instead of wherever the scheduler put the thread.  Best effort: no
    such file, an empty intersection with the allowed CPUs, SK_NUMA_BIND=0 -> nothing happens.  Returns the set used."""
    if os.environ.get("SK_NUMA_BIND", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        L = _lib.load()
        buf = C.create_string_buffer(64)
        if L.sk_device_pci_bus_id(buf, 64) != 0:
            return None
        with open("/sys/bus/pci/devices/%s/local_cpulist" % buf.value.decode().lower()) as fh:
            near = _parse_cpulist(fh.read())
        allowed = os.sched_getaffinity(0)
        use = near & allowed
        if not use or use == allowed:
            return None
        os.sched_setaffinity(0, use)                                 # (pid 0: the calling thread)
        return use
    except (OSError, ValueError):
        return None


def _want_rccl():
    return os.environ.get("SK_COMM", "rccl").lower() != "host"


def _comm_timeout():
    try:
        return max(1.0, float(os.environ.get("SK_COMM_TIMEOUT", "180")))
    except ValueError:
        return 180.0


def _bounded(fn, what):
    """fn() on a helper thread, waited for at most $SK_COMM_TIMEOUT seconds (180): (True, result) or (False, why).
    Creating an RCCL communicator is the one step of the multi-GPU path that can block for ever (a wedged peer, a
    fabric that never answers); the job is better served by the host gather than by a hung launch.  A helper that
    never returns is a daemon thread and dies with the process."""
    box = {}

    def run():
        try:
            box["r"] = fn()
        except BaseException as e:                                   # noqa: BLE001 -- reported to the caller
            box["e"] = e
    t = threading.Thread(target=run, name="sk-comm-init", daemon=True)
    t.start()
    t.join(_comm_timeout())
    if t.is_alive():
        return False, "%s did not return within %.0f s (SK_COMM_TIMEOUT)" % (what, _comm_timeout())
    if "e" in box:
        raise box["e"]
    return True, box["r"]


class ThreadGroup:
    """One process driving `devices`, one host thread each.  run(fn) calls fn(comm) on every rank's thread
    (bound to its device) and returns the results in rank order."""

    def __init__(self, devices, rccl=True, bind=True, oversubscribe=False):
        self.bind = bind                                             # False: host logic only (CPU tests)
        rccl = rccl and bind
        self.devices = [int(d) for d in devices]
        if not self.devices:
            raise ValueError("devices must be non-empty")
        self.shared = len(set(self.devices)) != len(self.devices)
        if self.shared and not oversubscribe:
            raise ValueError("devices must be distinct (oversubscribe=True lets ranks share one: dry runs)")
        n = len(self.devices)
        if self.shared and n > 16:
            raise ValueError("at most 16 ranks can share devices (one context slot each)")
        self.world = n
        self.backend, self.why_host = "host", None
        if self.shared:
            self.why_host = "ranks share a device (oversubscribed dry run): RCCL needs one device per rank"
        elif not rccl:
            self.why_host = "not requested"
        elif _want_rccl():
            L = _lib.load()
            arr = np.array(self.devices, dtype=np.int32)
            done, rc = _bounded(lambda: (L.sk_comm_init_all(arr.ctypes.data_as(C.POINTER(C.c_int32)), n),
                                         L.sk_last_error().decode(errors="replace")), "ncclCommInitAll")
            if not done:
                self.why_host = rc
            elif rc[0] == 0:
                self.backend = "rccl"
            elif rc[0] == -5:                                        # SK_ERR_UNSUPPORTED: no RCCL here
                self.why_host = rc[1]
            else:
                raise _lib.SquiggleKitError(rc[0], rc[1])
        else:
            self.why_host = "SK_COMM=host"
        self._bar = threading.Barrier(n)
        self._slots = [None] * n

    def _exchange(self, rank):
        def hx(payload):
            self._bar.wait()                                         # previous round fully read
            self._slots[rank] = payload
            self._bar.wait()
            return list(self._slots)
        return hx

    def run(self, fn):
        """fn(comm) on every rank's own thread; the communicators stay alive until close()."""
        n = self.world
        res, err = [None] * n, [None] * n

        def body(rank):
            try:
                if self.bind:
                    # ranks sharing a device each take their own context slot, counted down from the top so that
                    # they never collide with a plain sk_init(device) (slot == device)
                    _lib.init(self.devices[rank], slot=(15 - rank) if self.shared else None)
                    if self.world > 1 and not self.shared:
                        bind_thread_near_device()
                res[rank] = fn(RankComm(rank, n, self.backend, self._exchange(rank)))
            except BaseException as e:                               # noqa: BLE001 -- re-raised below
                err[rank] = e
                self._bar.abort()                                    # do not leave the other ranks waiting

        ts = [threading.Thread(target=body, args=(r,), name="sk-gpu%d" % self.devices[r]) for r in range(n)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if any(e is not None for e in err):
            self._bar.reset()                                        # an aborted barrier stays broken otherwise
        first = [e for e in err if e is not None and not isinstance(e, threading.BrokenBarrierError)]
        if first:
            raise first[0]
        if any(e is not None for e in err):
            raise [e for e in err if e is not None][0]
        return res

    def close(self):
        """Destroy the communicators (each on a thread bound to its device)."""
        if self.backend == "rccl":
            self.run(lambda comm: comm.close())
            self.backend = "host"


class ProcessGroup:
    """This process is one rank of a per-GPU launch.  `with ProcessGroup(rank, local, world) as comm: ...`"""

    def __init__(self, rank, local_rank, world, environ=None, bind=True):
        self.rank, self.local_rank, self.world = rank, local_rank, world
        self.store = FileStore(store_dir(environ), rank, world)
        self.backend, self.why_host = "host", None
        shared = oversubscribed(environ) if world > 1 else None      # every rank on one device (dry run)
        if bind:
            _lib.init(local_rank if shared is None else shared)
            if world > 1 and shared is None:
                bind_thread_near_device()
        want = _want_rccl() and shared is None
        if shared is not None:
            self.why_host = "ranks share device %d (oversubscribed dry run): RCCL needs one device per rank" % shared
        payload = b"\0"                                             # [0]: 1 = an ncclUniqueId follows
        if want and bind and rank == 0:
            L = _lib.load()
            uid = C.create_string_buffer(UID_BYTES)
            rc = L.sk_comm_unique_id(uid)
            if rc == 0:
                payload = b"\1" + uid.raw
            elif rc == -5:                                           # SK_ERR_UNSUPPORTED: no RCCL here
                self.why_host = L.sk_last_error().decode(errors="replace")
            else:
                check(rc)
        blob = self.store.rendezvous(payload)
        if shared is not None:
            pass
        elif not _want_rccl():
            self.why_host = "SK_COMM=host"
        elif bind and blob[:1] == b"\1":
            L = _lib.load()
            dev = local_rank if shared is None else shared

            def init_rank():                                         # (the helper thread binds the device itself)
                _lib.init(dev)
                return L.sk_comm_init_rank(blob[1:1 + UID_BYTES], world, rank), L.sk_last_error().decode(errors="replace")
            done, rc = _bounded(init_rank, "ncclCommInitRank")
            mine = 1 if (done and rc[0] == 0) else 0
            if not mine:
                self.why_host = rc if not done else rc[1]
            votes = self.store.allgather(bytes([mine]))               # all or nothing
            if all(v == b"\x01" for v in votes):
                self.backend = "rccl"
            elif mine:
                L.sk_comm_destroy()
        self.comm = RankComm(rank, world, self.backend, self.store.allgather)

    def __enter__(self):
        return self.comm

    def __exit__(self, *exc):
        try:
            self.comm.close()
        finally:
            if exc[0] is None:
                self.store.close()
        return False


# ------------------------------------------------------------------------------------------------
# product entry: a host batch over several GPUs
# ------------------------------------------------------------------------------------------------
_groups = {}
_groups_mu = threading.Lock()


def group_for(devices, rccl=False):
    """The (cached) ThreadGroup of a device list; created with RCCL communicators when `rccl` is asked for
    (an RCCL group also serves the host-gather calls).  A list naming a device twice is only accepted under
    SK_OVERSUBSCRIBE (dry runs of the sharded path on a box with fewer GPUs than ranks)."""
    key = tuple(int(d) for d in devices)
    with _groups_mu:
        g = _groups.get(key)
        if g is not None and rccl and g.backend != "rccl" and g.why_host == "not requested":
            g = None                                                 # upgrade: build the communicators now
        if g is None:
            g = _groups[key] = ThreadGroup(key, rccl=rccl, oversubscribe=oversubscribed() is not None)
        return g


def drop_group(g):
    """Forget a group whose run failed: its communicators are destroyed and the next call builds a fresh one."""
    with _groups_mu:
        for k in [k for k, v in _groups.items() if v is g]:
            del _groups[k]
    try:
        g.close()
    except Exception:                                               # noqa: BLE001 -- already failing
        pass


def close_groups():
    with _groups_mu:
        for g in _groups.values():
            g.close()
        _groups.clear()


_fault_hook = None          # tests: callable(rank) run at the start of a rank's shard (raises to play a failing device)


def run_sharded(devices, total, fn, rccl=False, reshard=None):
    """fn(lo, hi, comm) on one thread per device, over the block split of `total` reads.  Results are written
    by fn into views of caller-owned host arrays (that IS the host-side gather).  Returns the group.

    A rank whose shard fails with a library error (a device that dropped out, ran out of memory, ...) does not take the
    job down while other devices are healthy ("never abort the stream of reads"): its block is cut up
    over the surviving devices and run again there, with one line on stderr.  Only for shards that do not talk to each
    other (reshard defaults to `not rccl`); invalid arguments (SK_ERR_INVALID), unsupported shapes and max_segs overflows
    fail on every device alike and are raised at once.
    If every rank failed, or the second attempt fails too, the first error is raised."""
    import sys
    g = group_for(devices, rccl=rccl)
    reshard = (not rccl) if reshard is None else reshard
    failed = []                                                      # (rank, lo, hi, error)
    mu = threading.Lock()

    def body(comm):
        lo, hi = sharding.shard_bounds(total, comm.rank, comm.world)
        try:
            if _fault_hook is not None:
                _fault_hook(comm.rank)
            return fn(lo, hi, comm)
        except _lib.SquiggleKitError as e:
            # the caller's arguments / a shape no device covers fail everywhere alike: raised.  A lost device
            # (SK_ERR_NO_DEVICE), a failed HIP call, an allocation failure: the block goes to the survivors.
            if not reshard or e.code in (_lib.SK_ERR_INVALID, _lib.SK_ERR_UNSUPPORTED, _lib.SK_ERR_OVERFLOW):
                raise
            with mu:
                failed.append((comm.rank, lo, hi, e))
            return None

    try:
        g.run(body)
    except BaseException:
        drop_group(g)
        raise
    if failed:
        bad = {r for r, _, _, _ in failed}
        good = [d for k, d in enumerate(g.devices) if k not in bad]
        drop_group(g)                                                # (a context that failed is not reused)
        if not good:
            raise failed[0][3]
        for r, lo, hi, e in sorted(failed):
            sys.stderr.write("squigglekit: device %d (rank %d) failed on reads %d..%d (%s); re-running them on device(s) %s\n"
                             % (g.devices[r], r, lo, hi - 1, e, ", ".join(str(d) for d in good)))
            n = hi - lo
            if n <= 0:
                continue
            g2 = group_for(good, rccl=False)

            def again(comm, lo=lo, n=n):
                a, b = sharding.shard_bounds(n, comm.rank, comm.world)
                return fn(lo + a, lo + b, comm) if b > a else None
            try:
                g2.run(again)
            except BaseException:
                drop_group(g2)
                raise
        return group_for(good, rccl=False)
    return g


def motifseq_sharded(sig, lens, motif, scale_mode, scale_low, scale_hi, devices, gather="host"):
    """sk_motifseq over `devices`: rank r uploads rows [lo_r, hi_r), runs the device-resident path and
      gather == "host"  downloads its records into its slice of the result (no inter-GPU traffic at all);
      gather == "rccl"  all-gathers the padded record blocks over RCCL so that every GPU holds the complete
                        result, and rank 0 downloads it (the shape a device-resident pipeline uses).
    Returns (hits, info)."""
    L = _lib.load()
    R, stride = sig.shape
    out = np.zeros(R, dtype=_lib.HIT_DTYPE)
    sizes = sharding.shard_sizes(R, len(devices))
    pad = max(sizes) if sizes else 0
    info = {"devices": list(devices), "shards": sizes, "gather": gather}

    def body(lo, hi, comm):
        n = hi - lo
        use_rccl = gather == "rccl" and comm.backend == "rccl"
        hb = _lib.HIT_DTYPE.itemsize
        d_sig = L.sk_dev_alloc(max(1, n) * stride * 2)
        d_len = L.sk_dev_alloc(max(1, n) * 4)
        d_out = L.sk_dev_alloc(max(1, pad) * hb)
        d_all = L.sk_dev_alloc(max(1, pad) * hb * comm.world) if use_rccl else None
        try:
            failed = None
            try:
                if not d_sig or not d_len or not d_out or (use_rccl and not d_all):
                    check(-4)
                if n:
                    check(L.sk_dev_upload(d_sig, ptr(sig[lo:hi]), n * stride * 2))
                    check(L.sk_dev_upload(d_len, ptr(lens[lo:hi]), n * 4))
                    check(L.sk_motifseq_dev_i16(d_sig, stride, d_len, n, ptr(motif), motif.size, scale_mode,
                                                int(scale_low), int(scale_hi), d_out))
                    check(L.sk_sync())                               # a kernel error surfaces here, not in the gather
            except Exception as e:                                   # noqa: BLE001 -- re-raised after the vote
                failed = e
            if use_rccl and not comm.all_ok(failed is None):
                # some rank failed before the collective: nobody enters ncclAllGather (it would never return)
                raise failed if failed is not None else _lib.SquiggleKitError(
                    -3, "another rank failed before the gather; rank %d skipped the collective" % comm.rank)
            if failed is not None:
                raise failed
            if use_rccl:
                comm.allgather_dev(d_out, d_all, pad * hb)
                check(L.sk_sync())
                if comm.rank == 0:
                    full = np.empty(comm.world * pad, dtype=_lib.HIT_DTYPE)
                    check(L.sk_dev_download(ptr(full), d_all, full.nbytes))
                    for r in range(comm.world):
                        a, b = sharding.shard_bounds(R, r, comm.world)
                        out[a:b] = full[r * pad:r * pad + (b - a)]
            elif n:
                check(L.sk_sync())
                check(L.sk_dev_download(ptr(out[lo:hi]), d_out, n * hb))
        finally:
            for d in (d_sig, d_len, d_out, d_all):
                if d:
                    L.sk_dev_free(d)

    g = run_sharded(devices, R, body, rccl=(gather == "rccl"))
    info["backend"] = g.backend
    if gather == "rccl" and g.backend != "rccl":
        info["gather"] = "host (RCCL unavailable: %s)" % g.why_host
    return out, info
