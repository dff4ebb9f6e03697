"""
Process group of the multi-GPU path without PyTorch: one process per GPU, RCCL over xGMI through ctypes.

The hot path shards by independent channel: no collective touches the data.  What ranks exchange is
  * control values - barriers, the MAX of the elapsed time, the SUM of the symbol-error counters, the rank count -
  * and, in the optional split-capture mode (qampy_amd/distributed.py), the end taps of the tier-b segments per pass.
Both go through ``librccl.so`` (``ncclCommInitRank`` / ``ncclAllReduce`` / ``ncclBroadcast``), the collectives enqueued on the
library's own HIP stream (``qh_stream_handle``), so a tap exchange sits between the kernels of a pass like any other launch.

Rendezvous.  The launcher (``bench.py --gpus N`` itself, or ``torch.distributed.run`` as the driver uses it) provides
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT.  Rank 0 listens on an ephemeral TCP port of MASTER_ADDR and
publishes it in a file keyed by (MASTER_PORT, parent pid) - MASTER_PORT itself belongs to the launcher's own store - the other
ranks connect; this star of sockets carries the ``ncclUniqueId`` and is also a complete (slow, host-side) implementation of
the collectives: the ``tcp`` backend, used where there is no GPU (CPU tests, ``--dry-run``), where several ranks share one
GPU (RCCL refuses duplicate devices), and as the fallback when RCCL cannot initialise - the JSON line says which ran.
"""
import ctypes as C
import os
import socket
import struct
import sys
import tempfile
import time

import numpy as np

NCCL_UNIQUE_ID_BYTES = 128
_NCCL_DTYPE = {np.dtype(np.int8): 0, np.dtype(np.uint8): 1, np.dtype(np.int32): 2, np.dtype(np.uint32): 3, np.dtype(np.int64): 4,
               np.dtype(np.uint64): 5, np.dtype(np.float32): 7, np.dtype(np.float64): 8}
_NCCL_OP = {"sum": 0, "prod": 1, "max": 2, "min": 3}
_NP_OP = {"sum": np.add, "prod": np.multiply, "max": np.maximum, "min": np.minimum}


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * NCCL_UNIQUE_ID_BYTES)]


def rank_info(env=None):
    """(rank, local_rank, world_size) from the launcher's environment; (0, 0, 1) when launched plainly."""
    env = os.environ if env is None else env
    return int(env.get("RANK", "0")), int(env.get("LOCAL_RANK", "0")), int(env.get("WORLD_SIZE", "1"))


def launch(script, argv, nproc, env=None):
    """Start ``nproc`` ranks of ``script`` on this node (one per GPU) with the usual environment and relay their exit codes:
    what ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N`` does, without the dependency."""
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = dict(os.environ if env is None else env)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    base.setdefault("OMP_NUM_THREADS", "1")
    base.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(nproc), LOCAL_WORLD_SIZE=str(nproc))
    procs = [subprocess.Popen([sys.executable, script] + list(argv), env=dict(base, RANK=str(r), LOCAL_RANK=str(r))) for r in range(nproc)]
    rc = 0
    for p in procs:
        rc = p.wait() or rc
    return rc


# ------------------------------------------------------------------------------------------------------------ sockets
# Wire format of the star: typed frames, no pickle - what travels is None, an integer (a rank), bytes (the ncclUniqueId, the rendezvous
# token) or a numeric ndarray (dtype name from a fixed list + shape + raw bytes), and nothing read from a socket is ever executed.
_WIRE_DTYPES = ("int8", "uint8", "int32", "uint32", "int64", "uint64", "float32", "float64", "complex64", "complex128")
_MAX_FRAME = 1 << 32


def _rd(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(n - len(buf), 1 << 20))
        if not chunk:
            raise ConnectionError("peer closed the rendezvous socket")
        buf += chunk
    return bytes(buf)


def _send(sock, obj):
    if obj is None:
        sock.sendall(b"N")
    elif isinstance(obj, (int, np.integer)) and not isinstance(obj, bool):
        sock.sendall(b"I" + struct.pack("<q", int(obj)))
    elif isinstance(obj, (bytes, bytearray)):
        sock.sendall(b"B" + struct.pack("<Q", len(obj)) + bytes(obj))
    elif isinstance(obj, np.ndarray) and obj.dtype.name in _WIRE_DTYPES:
        a = np.ascontiguousarray(obj)
        name = a.dtype.name.encode()
        sock.sendall(b"A" + struct.pack("<BB", len(name), a.ndim) + name + struct.pack("<%dq" % a.ndim, *a.shape) + a.tobytes())
    else:
        raise TypeError("qampy_amd.comm: cannot send %r over the rendezvous socket" % type(obj))


def _recv(sock):
    tag = _rd(sock, 1)
    if tag == b"N":
        return None
    if tag == b"I":
        return struct.unpack("<q", _rd(sock, 8))[0]
    if tag == b"B":
        (n,) = struct.unpack("<Q", _rd(sock, 8))
        if n > _MAX_FRAME:
            raise ConnectionError("rendezvous socket: oversized frame")
        return _rd(sock, n)
    if tag == b"A":
        ln, nd = struct.unpack("<BB", _rd(sock, 2))
        name = _rd(sock, ln).decode("ascii", "replace")
        if name not in _WIRE_DTYPES or nd > 8:
            raise ConnectionError("rendezvous socket: unexpected array header")
        shape = struct.unpack("<%dq" % nd, _rd(sock, 8 * nd)) if nd else ()
        count = 1
        for d in shape:
            if d < 0:
                raise ConnectionError("rendezvous socket: negative dimension")
            count *= d
        dt = np.dtype(name)
        if count * dt.itemsize > _MAX_FRAME:
            raise ConnectionError("rendezvous socket: oversized frame")
        return np.frombuffer(_rd(sock, count * dt.itemsize), dtype=dt).reshape(shape).copy()
    raise ConnectionError("rendezvous socket: unknown frame type %r" % tag)


def _rendezvous_dir():
    """Per-user directory (mode 0700, owned by this user - checked, not assumed) for the rendezvous files."""
    d = os.path.join(tempfile.gettempdir(), "qampy_comm_%d" % os.getuid())
    try:
        os.mkdir(d, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    import stat
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise RuntimeError("qampy_amd.comm: %s is not a private directory of this user" % d)
    return d


class Comm:
    """World group of the launch.  ``backend``: "auto" (RCCL when a device is given and every rank can initialise it, else
    tcp), "rccl", "tcp"."""

    def __init__(self, device=None, backend="auto", env=None, timeout=120.0):
        env = os.environ if env is None else env
        self.rank, self.local_rank, self.world = rank_info(env)
        self.device = device
        self.backend = "single" if self.world == 1 else "tcp"
        self.note = None
        self.op_timeout = 900.0                            # a collective may wait for a rank that is still computing; a dead rank must not hang the job
        self._peers, self._root, self._srv, self._file = [], None, None, None
        self._nccl, self._ncomm, self._scratch = None, None, None
        self.stuck = False
        if self.world > 1:
            self._rendezvous(env, timeout)
        want_rccl = backend == "rccl" or (backend == "auto" and device is not None and self.world > 1)
        if want_rccl:
            ok, why = self._init_rccl_guarded(float(env.get("QAMPY_COMM_RCCL_TIMEOUT", "120")))
            # every rank must take the same path
            flags = self._tcp_allreduce(np.array([1.0 if ok else 0.0]), "min") if self.world > 1 else np.array([1.0 if ok else 0.0])
            if flags[0] > 0:
                self.backend = "rccl"
            else:
                self._drop_rccl()
                self.note = "rccl unavailable (%s): host-side tcp collectives" % (why or "another rank failed")
                if backend == "rccl":
                    raise RuntimeError(self.note)

    # ---------------------------------------------------------------------------------------------- rendezvous (tcp star)
    def _rendezvous(self, env, timeout):
        addr = env.get("MASTER_ADDR", "127.0.0.1")
        key = "%s_%s_%d" % (addr.replace(":", "_").replace("/", "_"), env.get("MASTER_PORT", "0"), os.getppid())
        path = os.path.join(_rendezvous_dir(), key)
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, int(env.get("QAMPY_COMM_PORT", "0"))))
            srv.listen(self.world)
            token = os.urandom(16)                       # only readers of the (0600, own directory) file can join the star
            tmp = path + ".%d" % os.getpid()
            try:
                os.remove(tmp)
            except OSError:
                pass
            fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
            with os.fdopen(fd, "w") as f:
                f.write("%s %d %d %s" % (addr, srv.getsockname()[1], os.getpid(), token.hex()))
            os.replace(tmp, path)
            self._srv, self._file = srv, path
            srv.settimeout(timeout)
            peers = {}
            t0 = time.time()
            while len(peers) < self.world - 1:
                if time.time() - t0 > timeout:
                    raise TimeoutError("rank 0: %d of %d ranks joined the rendezvous" % (len(peers) + 1, self.world))
                c, _ = srv.accept()
                try:
                    c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    c.settimeout(timeout)
                    tok, r = _recv(c), _recv(c)
                    if tok != token or not isinstance(r, int) or not 1 <= r < self.world or r in peers:
                        raise ConnectionError("bad hello")
                    peers[r] = c
                except (ConnectionError, OSError, struct.error, ValueError):
                    c.close()                            # not one of ours: ignored
            self._peers = [peers[r] for r in range(1, self.world)]
            for c in self._peers:
                c.settimeout(self.op_timeout)
        else:
            t0 = time.time()
            while True:
                try:
                    st = os.stat(path)
                    if st.st_uid != os.getuid():
                        raise ValueError("rendezvous file of another user")
                    host, port, pid, tok = open(path).read().split()
                    os.kill(int(pid), 0)                 # a file left behind by a dead launch of the same key is ignored
                    s = socket.create_connection((host, int(port)), timeout=timeout)
                    break
                except (OSError, ValueError):
                    if time.time() - t0 > timeout:
                        raise TimeoutError("rank %d: no rendezvous at %s" % (self.rank, path))
                    time.sleep(0.05)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.settimeout(timeout)
            _send(s, bytes.fromhex(tok))
            _send(s, self.rank)
            s.settimeout(self.op_timeout)
            self._root = s

    def _tcp_allreduce(self, a, op):
        a = np.ascontiguousarray(a)
        if self.world == 1:
            return a.copy()
        f = _NP_OP[op]
        if self.rank == 0:
            acc = a.copy()
            for p in self._peers:
                acc = f(acc, _recv(p))
            for p in self._peers:
                _send(p, acc)
            return acc
        _send(self._root, a)
        return _recv(self._root)

    def _tcp_bcast_obj(self, obj, root=0):
        if self.world == 1:
            return obj
        if root != 0:                                  # via rank 0
            if self.rank == root:
                _send(self._root, obj)
            if self.rank == 0:
                obj = _recv(self._peers[root - 1])
        if self.rank == 0:
            for p in self._peers:
                _send(p, obj)
            return obj
        return _recv(self._root)

    # ---------------------------------------------------------------------------------------------- RCCL
    def _init_rccl_guarded(self, timeout):
        """RCCL's bootstrap (sockets between the ranks, then the xGMI / PCIe topology).  Everything that talks over the rendezvous star -
        the broadcast of the ncclUniqueId - happens HERE, on the calling thread; only ``ncclCommInitRank`` itself runs in a helper
        thread: if it does not come back within `timeout` seconds (an interface it cannot use, a peer that died) this rank reports
        failure, all ranks agree on the socket backend and the job goes on (the collectives here carry a few scalars).  The helper never
        touches the star, and stdout - which RCCL's banner must not reach: it carries bench.py's JSON line - is redirected and restored
        by this thread, so a helper that hangs leaves both intact.  `stuck` makes close() leave without joining it."""
        import threading
        # single node: the bootstrap may use the loopback interface (the container's hostname need not resolve); xGMI / PCIe carry the data
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        try:
            from . import _lib
            lib = None
            for name in ("librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"):
                try:
                    lib = C.CDLL(name)
                    break
                except OSError:
                    continue
            ok_local = lib is not None
            if ok_local:
                lib.ncclGetErrorString.restype = C.c_char_p
                lib.ncclGetErrorString.argtypes = [C.c_int]
                lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
                lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
                lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
                lib.ncclBroadcast.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
                lib.ncclCommDestroy.argtypes = [C.c_void_p]
                _lib.init(self.device)                   # the library's device (one per process)
            uid = _UniqueId()
            raw = None
            if self.rank == 0 and ok_local:
                rc = lib.ncclGetUniqueId(C.byref(uid))
                raw = None if rc else C.string_at(C.addressof(uid), NCCL_UNIQUE_ID_BYTES)      # (the whole structure: it may hold NULs)
            raw = self._tcp_bcast_obj(raw)               # rank 0's failure reaches everybody: no rank waits in vain
            if not ok_local:
                return False, "librccl.so not found"
            if raw is None:
                return False, "ncclGetUniqueId failed on rank 0"
            C.memmove(C.addressof(uid), raw, NCCL_UNIQUE_ID_BYTES)
        except Exception as e:                           # noqa: BLE001 - any failure means "use the fallback"
            return False, "%s: %s" % (type(e).__name__, e)
        box = {}

        def work():
            box["r"] = self._init_rccl(lib, uid)

        # RCCL's own messages: warnings only, to stderr (rounds 2-4 pointed file descriptor 1 at stderr around the call instead - a process-wide
        # side effect on every thread that happened to print meanwhile)
        os.environ.setdefault("NCCL_DEBUG", "WARN")
        os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")
        t = threading.Thread(target=work, daemon=True)
        t.start()
        t.join(timeout)
        if t.is_alive():
            self.stuck = True
            return False, "ncclCommInitRank did not return within %g s" % timeout
        return box.get("r", (False, "rccl initialisation raised"))

    def _init_rccl(self, lib, uid):
        try:
            from . import _lib
            # the current device is per THREAD and this runs in a helper thread: the communicator must be created on the rank's GPU
            hip = C.CDLL("libamdhip64.so")
            hip.hipSetDevice.argtypes = [C.c_int]
            if hip.hipSetDevice(int(self.device)) != 0:
                return False, "hipSetDevice(%d) failed" % int(self.device)
            comm = C.c_void_p()
            rc = lib.ncclCommInitRank(C.byref(comm), self.world, uid, self.rank)
            if rc:
                return False, "ncclCommInitRank: %s" % lib.ncclGetErrorString(rc).decode()
            self._nccl, self._ncomm = lib, comm
            self._scratch = _lib.DeviceArray((64,), np.float64)
            return True, None
        except Exception as e:                             # noqa: BLE001 - any failure means "use the fallback"
            return False, "%s: %s" % (type(e).__name__, e)

    def _drop_rccl(self):
        if self._nccl is not None and self._ncomm is not None:
            try:
                self._nccl.ncclCommDestroy(self._ncomm)
            except Exception:                              # noqa: BLE001
                pass
        self._nccl, self._ncomm, self._scratch = None, None, None

    def _stream(self):
        from . import _lib
        s = C.c_void_p()
        _lib.call("qh_stream_handle", C.byref(s))
        return s

    def _ck(self, rc, what):
        if rc:
            raise RuntimeError("%s: %s" % (what, self._nccl.ncclGetErrorString(rc).decode()))

    # ---------------------------------------------------------------------------------------------- collectives: host values
    def allreduce(self, values, op="sum"):
        """Element-wise reduction over the ranks of a small table of host values (float64); returns an ndarray."""
        a = np.ascontiguousarray(np.asarray(values, dtype=np.float64))
        if self.world == 1:
            return a.copy()
        if self.backend == "rccl":
            from . import _lib
            flat = a.ravel()
            if flat.size > self._scratch.shape[0]:
                self._scratch = _lib.DeviceArray((flat.size,), np.float64)
            _lib.call("qh_memcpy_h2d", self._scratch.ptr, _lib.ptr(flat), flat.nbytes)
            self._ck(self._nccl.ncclAllReduce(self._scratch.ptr, self._scratch.ptr, flat.size, 8, _NCCL_OP[op], self._ncomm, self._stream()), "ncclAllReduce")
            out = np.empty_like(flat)
            _lib.call("qh_memcpy_d2h", _lib.ptr(out), self._scratch.ptr, flat.nbytes)      # synchronises the library stream
            return out.reshape(a.shape)
        return self._tcp_allreduce(a, op)

    def barrier(self):
        self.allreduce([0.0])

    # ---------------------------------------------------------------------------------------------- collectives: device buffers
    def allreduce_dev(self, ptr, count, dtype, op="sum"):
        """In-place all-reduce of a DEVICE buffer of ``count`` elements.  RCCL: enqueued on the library stream, returns at once;
        tcp: staged through the host (synchronises)."""
        dt = np.dtype(dtype)
        if self.world == 1:
            return
        if self.backend == "rccl":
            self._ck(self._nccl.ncclAllReduce(C.c_void_p(ptr), C.c_void_p(ptr), int(count), _NCCL_DTYPE[dt], _NCCL_OP[op], self._ncomm, self._stream()), "ncclAllReduce")
            return
        from . import _lib
        host = np.empty(int(count), dt)
        _lib.call("qh_memcpy_d2h", _lib.ptr(host), C.c_void_p(ptr), host.nbytes)
        host = self._tcp_allreduce(host, op)
        _lib.call("qh_memcpy_h2d", C.c_void_p(ptr), _lib.ptr(host), host.nbytes)

    def broadcast_dev(self, ptr, nbytes, root=0):
        """Broadcast of a device buffer from ``root`` (the north star's "broadcast of converged taps": 1312 B at 2 x 2 x 41)."""
        if self.world == 1:
            return
        if self.backend == "rccl":
            self._ck(self._nccl.ncclBroadcast(C.c_void_p(ptr), C.c_void_p(ptr), int(nbytes), 1, int(root), self._ncomm, self._stream()), "ncclBroadcast")
            return
        from . import _lib
        host = np.empty(int(nbytes), np.uint8)
        _lib.call("qh_memcpy_d2h", _lib.ptr(host), C.c_void_p(ptr), host.nbytes)
        host = self._tcp_bcast_obj(host if self.rank == root else None, root)
        _lib.call("qh_memcpy_h2d", C.c_void_p(ptr), _lib.ptr(host), host.nbytes)

    @property
    def on_stream(self):
        """Device collectives are enqueued on the library stream (no host synchronisation around them)."""
        return self.backend == "rccl"

    def close(self):
        try:
            if self.world > 1:
                self._tcp_allreduce(np.array([0.0]), "sum")      # nobody tears the star down while another rank still needs it
        except Exception:                                  # noqa: BLE001
            pass
        self._drop_rccl()
        for s in self._peers + ([self._root] if self._root else []) + ([self._srv] if self._srv else []):
            try:
                s.close()
            except OSError:
                pass
        if self._file:
            try:
                os.remove(self._file)
            except OSError:
                pass
        self._peers, self._root, self._srv, self._file = [], None, None, None
