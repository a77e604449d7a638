"""Communicators of the sharded (one process per GPU) run.

``RcclComm``    RCCL over xGMI through the C-ABI (``shp_comm_*`` in include/shepseg_hip.h):
                device-to-device strips, collectives on device staging.  Bootstrap: rank 0 makes
                the 128-byte unique id and leaves it in a rendezvous directory the ranks of one
                launch share.
``SocketComm``  plain TCP between the ranks of one host, for the CPU tests (oracle engine) and for
                ranks that share one GPU; same interface, strips staged through host memory.
``LocalComm``   world size 1.

The reference has nothing of the kind: it farms tiles to workers over a
``multiprocessing.managers`` TCP channel and ships whole pickled results back
(tiling.py:1799-1912).  No torch, no MPI.

Ranks find each other through the environment the launcher sets (RANK, WORLD_SIZE, LOCAL_RANK,
MASTER_PORT: what ``bench.py --gpus N`` and ``python -m torch.distributed.run`` both provide).
"""
import ctypes
import hashlib
import hmac
import os
import pickle
import secrets
import socket
import stat
import struct
import threading
import time

import numpy

from . import _lib


class CommError(RuntimeError):
    pass


def rendezvousDir():
    """A directory private to this launch and to this user: the ranks are children of one launcher
    process.  Created with mode 0700; an existing one must be a real directory owned by this user and
    closed to everybody else -- on a shared host another user could otherwise plant the files the ranks
    trust (ports, the RCCL unique id, the handshake key)."""
    d = os.environ.get('SHEPSEG_COMM_DIR')
    if not d:
        # (TORCHELASTIC_RESTART_COUNT: a relaunch by the same agent must not find the last attempt's files)
        d = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'shepseg_comm_%s_%d_%s' % (
            os.environ.get('MASTER_PORT', '0'), os.getppid(), os.environ.get('TORCHELASTIC_RESTART_COUNT', '0')))
    try:
        os.makedirs(d, mode=0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise CommError("rendezvous directory %s is not a private directory of this user "
                        "(mode %o, uid %d)" % (d, st.st_mode & 0o7777, st.st_uid))
    return d


def launchTag():
    """Distinguishes the rendezvous files of THIS launch from what an earlier, crashed launch left in the same
    directory: a nonce the launcher put into every rank's environment (bench.py's spawn_ranks and the test
    helpers set SHEPSEG_LAUNCH_NONCE; torchrun sets TORCHELASTIC_RUN_ID).  Without one the names are plain and
    rank 0's clean-up at start is all there is."""
    t = os.environ.get('SHEPSEG_LAUNCH_NONCE') or os.environ.get('TORCHELASTIC_RUN_ID') or ''
    t = ''.join(ch for ch in t if ch.isalnum() or ch in '-_')[:48]
    return ('.' + t) if t and t != 'none' else ''


def launchKey(d, rank):
    """32 random bytes shared by the ranks of one launch (rank 0 makes them): the key of the socket
    transport's connection handshake."""
    path = os.path.join(d, 'key' + launchTag())
    if rank == 0:
        key = secrets.token_bytes(32)
        _publish(path, key)
        return key
    return _await(path)


def _publish(path, data):
    tmp = '%s.tmp%d' % (path, os.getpid())
    with open(tmp, 'wb') as f:
        f.write(data)
    os.replace(tmp, path)


def _await(path, timeout=180.0):
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > timeout:
            raise CommError("rendezvous: %s did not appear within %.0f s" % (path, timeout))
        time.sleep(0.01)
    with open(path, 'rb') as f:
        return f.read()


class LocalComm(object):
    def allgather_arrays(self, arrays):
        return [[numpy.ascontiguousarray(a).copy() for a in arrays]]

    """World size 1: every collective is the identity."""
    rank = 0
    world = 1
    transport = 'none (one rank)'
    onDevice = False

    def allgather_obj(self, obj):
        return [obj]

    def bcast_obj(self, obj, src=0):
        return obj

    def allreduce_sum_i64(self, arr):
        return arr

    def max_f64(self, v):
        return v

    def barrier(self):
        pass

    def close(self):
        pass


def _packArrays(arrays):
    """Several numpy arrays as ONE byte string with a fixed-size header per array (dtype char code, length):
    the data path of the statistics exchange travels as raw bytes, not as pickled objects."""
    parts = [numpy.array([len(arrays)], dtype=numpy.int64).tobytes()]
    for a in arrays:
        a = numpy.ascontiguousarray(a)
        parts.append(numpy.array([ord(a.dtype.char), a.size], dtype=numpy.int64).tobytes())
        parts.append(a.tobytes())
    return b''.join(parts)


def _unpackArrays(data):
    data = memoryview(data)
    n = int(numpy.frombuffer(data[:8], dtype=numpy.int64)[0])
    off = 8
    out = []
    for _ in range(n):
        (code, size) = numpy.frombuffer(data[off:off + 16], dtype=numpy.int64)
        dt = numpy.dtype(chr(int(code)))
        off += 16
        nbytes = int(size) * dt.itemsize
        out.append(numpy.frombuffer(data[off:off + nbytes], dtype=dt).copy())
        off += nbytes
    return out


class _ObjCollectives(object):
    def allgather_arrays(self, arrays):
        """Every rank's list of numpy arrays on every rank: [rank][i].  Raw bytes point to point (rank r sends to
        every other rank); built on send_bytes / recv_bytes of the transport."""
        mine = _packArrays(arrays)
        out = [None] * self.world
        out[self.rank] = [numpy.ascontiguousarray(a).copy() for a in arrays]
        # a fixed schedule without deadlock: in round d rank r sends to r + d and receives from r - d
        for d in range(1, self.world):
            dst = (self.rank + d) % self.world
            src = (self.rank - d) % self.world
            if (self.rank // d) % 2 == 0:
                self.send_bytes(mine, dst)
                out[src] = _unpackArrays(self.recv_bytes(src))
            else:
                got = self.recv_bytes(src)
                self.send_bytes(mine, dst)
                out[src] = _unpackArrays(got)
        return out


    """Collectives on Python objects built from send_obj / recv_obj (gather to rank 0, fan out)."""
    def allgather_obj(self, obj):
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            out = [obj] + [self.recv_obj(r) for r in range(1, self.world)]
            for r in range(1, self.world):
                self.send_obj(out, r)
            return out
        self.send_obj(obj, 0)
        return self.recv_obj(0)

    def bcast_obj(self, obj, src=0):
        if self.world == 1:
            return obj
        if self.rank == src:
            for r in range(self.world):
                if r != src:
                    self.send_obj(obj, r)
            return obj
        return self.recv_obj(src)

    def allreduce_sum_i64(self, arr):
        parts = self.allgather_obj(numpy.asarray(arr, dtype=numpy.int64))
        out = parts[0].copy()
        for p in parts[1:]:
            out += p
        return out

    def max_f64(self, v):
        return max(self.allgather_obj(float(v)))

    def barrier(self):
        self.allgather_obj(None)


class SocketComm(_ObjCollectives):
    """TCP between the ranks of one host.  Every rank listens on an ephemeral port and publishes it
    in the rendezvous directory; a directed connection per (sender, receiver) pair is opened on the
    first send.  Messages are length-prefixed."""
    transport = 'socket: TCP on the loopback interface, strips staged through host memory'
    onDevice = False

    def __init__(self, rank=None, world=None):
        self.rank = int(os.environ.get('RANK', '0')) if rank is None else rank
        self.world = int(os.environ.get('WORLD_SIZE', '1')) if world is None else world
        self.dir = rendezvousDir()
        self.out = {}
        self.inc = {}
        self.cond = threading.Condition()
        self.tag = launchTag()
        if self.rank == 0:          # leftovers of an earlier launch that shared this directory (any tag but ours)
            for f in os.listdir(self.dir):
                if (f.startswith('port') or f.startswith('key') or f.startswith('rccl_unique_id')) and \
                        (not self.tag or not f.endswith(self.tag)):
                    try:
                        os.remove(os.path.join(self.dir, f))
                    except OSError:
                        pass
        self.key = launchKey(self.dir, self.rank)
        self.srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        self.srv.bind(('127.0.0.1', 0))
        self.srv.listen(self.world + 4)
        self.closing = False
        self.acceptor = threading.Thread(target=self._accept, daemon=True)
        self.acceptor.start()
        _publish(os.path.join(self.dir, 'port%d%s' % (self.rank, self.tag)), str(self.srv.getsockname()[1]).encode())

    def _accept(self):
        """Registers a peer only after it proved that it holds the launch key (HMAC over a fresh nonce
        and the rank it claims): anything else that reaches the loopback port is dropped, and a
        connection that dies half way does not stop the acceptor."""
        while not self.closing:
            try:
                (conn, _addr) = self.srv.accept()
            except OSError:
                return
            try:
                conn.settimeout(30.0)
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                nonce = secrets.token_bytes(16)
                conn.sendall(nonce)
                hello = self._read(conn, 4 + 32)
                src = struct.unpack('<i', hello[:4])[0]
                want = hmac.new(self.key, nonce + hello[:4], hashlib.sha256).digest()
                if not (0 <= src < self.world) or not hmac.compare_digest(want, hello[4:]):
                    raise CommError("handshake failed")
                conn.sendall(b'\x01')                   # accepted: the connecting side waits for this
                conn.settimeout(None)
            except (CommError, OSError, struct.error):
                try:
                    conn.close()
                except OSError:
                    pass
                continue
            with self.cond:
                self.inc[src] = conn
                self.cond.notify_all()

    @staticmethod
    def _read(conn, n):
        buf = bytearray(n)
        mv = memoryview(buf)
        got = 0
        while got < n:
            k = conn.recv_into(mv[got:], n - got)
            if k == 0:
                raise CommError("peer closed the connection")
            got += k
        return bytes(buf)

    def _conn_to(self, dst):
        c = self.out.get(dst)
        if c is None:
            port = int(_await(os.path.join(self.dir, 'port%d%s' % (dst, self.tag))).decode())
            c = socket.create_connection(('127.0.0.1', port), timeout=180)
            c.settimeout(60.0)
            c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            try:
                nonce = self._read(c, 16)
                me = struct.pack('<i', self.rank)
                c.sendall(me + hmac.new(self.key, nonce + me, hashlib.sha256).digest())
                if self._read(c, 1) != b'\x01':
                    raise CommError("bad acknowledgement")
            except (CommError, OSError) as e:
                # fail HERE, not in a receive that never returns: the peer dropped us (a stale key or port file of
                # an earlier launch in a shared rendezvous directory, or something else on that port)
                c.close()
                raise CommError("rank %d: rank %d refused the connection handshake (%s): stale rendezvous files in %s?"
                                % (self.rank, dst, e, self.dir))
            c.settimeout(None)
            self.out[dst] = c
        return c

    def send_bytes(self, data, dst):
        c = self._conn_to(dst)
        mv = memoryview(data).cast('B')
        c.sendall(struct.pack('<q', len(mv)))
        c.sendall(mv)

    def recv_bytes(self, src, timeout=600.0):
        with self.cond:
            t0 = time.time()
            while src not in self.inc:
                if time.time() - t0 > timeout:
                    raise CommError("rank %d: nothing from rank %d within %.0f s" % (self.rank, src, timeout))
                self.cond.wait(timeout=1.0)
            conn = self.inc[src]
        n = struct.unpack('<q', self._read(conn, 8))[0]
        return self._read(conn, n)

    def send_obj(self, obj, dst):
        self.send_bytes(pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL), dst)

    def recv_obj(self, src):
        return pickle.loads(self.recv_bytes(src))

    def close(self):
        try:
            self.barrier()
        except Exception:
            pass
        self.closing = True
        for c in list(self.out.values()) + list(self.inc.values()):
            try:
                c.close()
            except OSError:
                pass
        try:
            self.srv.close()
        except OSError:
            pass
        for f in ['port%d%s' % (self.rank, self.tag)] + (['key' + self.tag] if self.rank == 0 else []):
            try:
                os.remove(os.path.join(self.dir, f))
            except OSError:
                pass
        try:
            os.rmdir(self.dir)          # (the last rank out succeeds)
        except OSError:
            pass


class RcclComm(_ObjCollectives):
    """RCCL communicator of this rank's GPU (device = LOCAL_RANK), bound to a context of its own (on pooled streams).
    Device buffers go straight through ncclSend / ncclRecv; Python objects are pickled into a device
    staging buffer and travel by the same calls."""
    onDevice = True

    def __init__(self, rank=None, world=None, device=None):
        self.rank = int(os.environ.get('RANK', '0')) if rank is None else rank
        self.world = int(os.environ.get('WORLD_SIZE', '1')) if world is None else world
        if device is None:
            device = int(os.environ.get('LOCAL_RANK', os.environ.get('SHEPSEG_DEVICE', '0')))
        # (its stream comes from the library's pool: a process that is over its ~24 hardware queues pays for it on every
        #  launch -- RCCL's own streams plus a context's two took the tiled driver from 462 to 522 ms per step)
        self.c = _lib.Context(device=device, sharedStreams=True)
        self.L = self.c._L
        self.dir = rendezvousDir()
        idpath = os.path.join(self.dir, 'rccl_unique_id' + launchTag())
        if self.rank == 0:
            try:
                os.remove(idpath)       # (an earlier launch's id must not be taken for this one's)
            except OSError:
                pass
            buf = (ctypes.c_uint8 * 128)()
            if self.L.shp_comm_unique_id(buf) != 0:
                raise CommError("ncclGetUniqueId failed")
            _publish(idpath, bytes(buf))
            uid = bytes(buf)
        else:
            uid = _await(idpath)
        h = ctypes.c_void_p()
        # RCCL greets on stdout ("Hostname : ... Librccl path : ..."): keep the caller's stdout clean
        # (bench.py prints one JSON line there) by pointing fd 1 at stderr while it initialises
        import sys
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            rc = self.L.shp_comm_create(self.c.handle, self.rank, self.world, ctypes.c_char_p(uid), ctypes.byref(h))
        finally:
            os.dup2(saved, 1)
            os.close(saved)
        self.c.check(rc)
        self.h = h
        self.stage = None
        self.stageBytes = 0

    def _staging(self, nbytes):
        if self.stageBytes < nbytes:
            if self.stage is not None:
                self.c.check(self.L.shp_dev_free(self.c.handle, self.stage))
            p = ctypes.c_void_p()
            self.stageBytes = max(int(nbytes), 1 << 20)
            self.c.check(self.L.shp_dev_alloc(self.c.handle, self.stageBytes, ctypes.byref(p)))
            self.stage = p
        return self.stage

    # ---- device buffers ----
    def send_dev(self, dptr, nbytes, dst):
        self.c.check(self.L.shp_comm_send(self.h, ctypes.c_void_p(dptr), nbytes, dst))

    def recv_dev(self, dptr, nbytes, src):
        self.c.check(self.L.shp_comm_recv(self.h, ctypes.c_void_p(dptr), nbytes, src))

    # ---- asynchronous strips: ordered on the device against the given context's stream, no host wait ----
    def isend_dev(self, dptr, nbytes, dst, producer):
        self.c.check(self.L.shp_comm_isend(self.h, ctypes.c_void_p(dptr), nbytes, dst, producer.handle))

    def irecv_dev(self, dptr, nbytes, src, consumer):
        self.c.check(self.L.shp_comm_irecv(self.h, ctypes.c_void_p(dptr), nbytes, src,
                                           consumer.handle if consumer is not None else None))

    def wait_dev(self, consumer):
        self.c.check(self.L.shp_comm_wait(self.h, consumer.handle))

    def group(self, begin):
        self.c.check(self.L.shp_comm_group(self.h, 1 if begin else 0))

    def drain(self):
        self.c.check(self.L.shp_comm_drain(self.h))

    def count(self):
        """ncclCommCount: the number of ranks RCCL itself says this communicator spans"""
        n = ctypes.c_int(0)
        self.c.check(self.L.shp_comm_count(self.h, ctypes.byref(n)))
        return n.value

    transport = 'rccl: ncclSend / ncclRecv of device buffers (xGMI inside a node)'

    # ---- collectives on device buffers (the statistics exchange: nothing crosses the host) ----
    def allgather_dev(self, d_send, d_recv, bytesPerRank):
        self.c.check(self.L.shp_comm_allgather(self.h, ctypes.c_void_p(d_send), ctypes.c_void_p(d_recv), bytesPerRank))

    def allreduce_dev_i64(self, d_buf, count):
        self.c.check(self.L.shp_comm_allreduce(self.h, ctypes.c_void_p(d_buf), count, 0))

    # ---- host data through the staging buffer ----
    def send_bytes(self, data, dst):
        a = numpy.frombuffer(memoryview(data).cast('B'), dtype=numpy.uint8)
        hdr = numpy.array([len(a)], dtype=numpy.int64)
        st = self._staging(max(len(a), 8))
        self.c.check(self.L.shp_dev_upload(self.c.handle, st, _lib.ptr(hdr), 8))
        self.c.check(self.L.shp_comm_send(self.h, st, 8, dst))
        if len(a):
            self.c.check(self.L.shp_dev_upload(self.c.handle, st, _lib.ptr(numpy.ascontiguousarray(a)), len(a)))
            self.c.check(self.L.shp_comm_send(self.h, st, len(a), dst))

    def recv_bytes(self, src):
        hdr = numpy.zeros(1, dtype=numpy.int64)
        st = self._staging(8)
        self.c.check(self.L.shp_comm_recv(self.h, st, 8, src))
        self.c.check(self.L.shp_dev_download(self.c.handle, _lib.ptr(hdr), st, 8))
        n = int(hdr[0])
        out = numpy.empty(n, dtype=numpy.uint8)
        if n:
            st = self._staging(n)
            self.c.check(self.L.shp_comm_recv(self.h, st, n, src))
            self.c.check(self.L.shp_dev_download(self.c.handle, _lib.ptr(out), st, n))
        return out.tobytes()

    def send_obj(self, obj, dst):
        self.send_bytes(pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL), dst)

    def recv_obj(self, src):
        return pickle.loads(self.recv_bytes(src))

    # ---- collectives that have an RCCL form ----
    def allreduce_sum_i64(self, arr):
        a = numpy.ascontiguousarray(arr, dtype=numpy.int64)
        if self.world == 1:
            return a
        st = self._staging(a.nbytes)
        self.c.check(self.L.shp_dev_upload(self.c.handle, st, _lib.ptr(a), a.nbytes))
        self.c.check(self.L.shp_comm_allreduce(self.h, st, a.size, 0))
        out = numpy.empty_like(a)
        self.c.check(self.L.shp_dev_download(self.c.handle, _lib.ptr(out), st, a.nbytes))
        return out

    def max_f64(self, v):
        if self.world == 1:
            return v
        a = numpy.array([v], dtype=numpy.float64)
        st = self._staging(8)
        self.c.check(self.L.shp_dev_upload(self.c.handle, st, _lib.ptr(a), 8))
        self.c.check(self.L.shp_comm_allreduce(self.h, st, 1, 1))
        self.c.check(self.L.shp_dev_download(self.c.handle, _lib.ptr(a), st, 8))
        return float(a[0])

    def bcast_obj(self, obj, src=0):
        if self.world == 1:
            return obj
        hdr = numpy.zeros(1, dtype=numpy.int64)
        data = None
        if self.rank == src:
            data = numpy.frombuffer(pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL), dtype=numpy.uint8)
            hdr[0] = len(data)
        st = self._staging(8)
        self.c.check(self.L.shp_dev_upload(self.c.handle, st, _lib.ptr(hdr), 8))
        self.c.check(self.L.shp_comm_bcast(self.h, st, 8, src))
        self.c.check(self.L.shp_dev_download(self.c.handle, _lib.ptr(hdr), st, 8))
        n = int(hdr[0])
        st = self._staging(n)
        if self.rank == src:
            self.c.check(self.L.shp_dev_upload(self.c.handle, st, _lib.ptr(numpy.ascontiguousarray(data)), n))
        self.c.check(self.L.shp_comm_bcast(self.h, st, n, src))
        if self.rank == src:
            return obj
        out = numpy.empty(n, dtype=numpy.uint8)
        self.c.check(self.L.shp_dev_download(self.c.handle, _lib.ptr(out), st, n))
        return pickle.loads(out.tobytes())

    def allgather_obj(self, obj):
        if self.world == 1:
            return [obj]
        data = numpy.frombuffer(pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL), dtype=numpy.uint8)
        sizes = numpy.zeros(self.world, dtype=numpy.int64)
        sizes[self.rank] = len(data)
        sizes = self.allreduce_sum_i64(sizes)
        slot = int(sizes.max())
        st = self._staging(slot * (self.world + 1))
        mine = ctypes.c_void_p(st.value + slot * self.world)
        self.c.check(self.L.shp_dev_upload(self.c.handle, mine, _lib.ptr(numpy.ascontiguousarray(data)), len(data)))
        self.c.check(self.L.shp_comm_allgather(self.h, mine, st, slot))
        allb = numpy.empty(slot * self.world, dtype=numpy.uint8)
        self.c.check(self.L.shp_dev_download(self.c.handle, _lib.ptr(allb), st, allb.nbytes))
        return [pickle.loads(allb[r * slot:r * slot + int(sizes[r])].tobytes()) for r in range(self.world)]

    def barrier(self):
        self.allreduce_sum_i64(numpy.zeros(1, dtype=numpy.int64))

    def close(self):
        if self.h is not None:
            if self.stage is not None:
                self.c.check(self.L.shp_dev_free(self.c.handle, self.stage))
                self.stage = None
            self.L.shp_comm_destroy(self.h)
            self.h = None
            if self.rank == 0:
                try:
                    os.remove(os.path.join(self.dir, 'rccl_unique_id' + launchTag()))
                except OSError:
                    pass
            self.c.close()


class HostStagedDev(object):
    """The device collectives of the statistics exchange on top of a transport that is not on the device
    (SocketComm: ranks that share one GPU, the rehearsal runs and tests of a one-GPU box): buffers are
    downloaded, travel as raw arrays and are uploaded again.  Same interface as RcclComm's."""
    onDevice = True

    def __init__(self, comm, ctx):
        (self.comm, self.c) = (comm, ctx)
        (self.rank, self.world) = (comm.rank, comm.world)
        self.transport = getattr(comm, 'transport', type(comm).__name__) + ' (device buffers staged through the host)'

    def allgather_obj(self, obj):
        return self.comm.allgather_obj(obj)

    def allgather_dev(self, d_send, d_recv, bytesPerRank):
        mine = numpy.empty(bytesPerRank, dtype=numpy.uint8)
        self.c.check(self.c._L.shp_dev_download(self.c.handle, _lib.ptr(mine), ctypes.c_void_p(d_send), bytesPerRank))
        parts = self.comm.allgather_arrays([mine])
        allb = numpy.concatenate([p[0] for p in parts])
        self.c.check(self.c._L.shp_dev_upload(self.c.handle, ctypes.c_void_p(d_recv), _lib.ptr(allb), allb.nbytes))

    def allreduce_dev_i64(self, d_buf, count):
        mine = numpy.empty(count, dtype=numpy.int64)
        self.c.check(self.c._L.shp_dev_download(self.c.handle, _lib.ptr(mine), ctypes.c_void_p(d_buf), mine.nbytes))
        tot = self.comm.allreduce_sum_i64(mine)
        self.c.check(self.c._L.shp_dev_upload(self.c.handle, ctypes.c_void_p(d_buf), _lib.ptr(tot), tot.nbytes))

    def barrier(self):
        self.comm.barrier()

    def max_f64(self, v):
        return self.comm.max_f64(v)


def fromEnvironment(transport=None):
    """The communicator the launcher's environment asks for: LocalComm at world size 1, else
    ``transport`` ('rccl' or 'socket'; default SHEPSEG_COMM or 'rccl')."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if transport is None:
        transport = os.environ.get('SHEPSEG_COMM', 'rccl')
    if world <= 1 and os.environ.get('SHEPSEG_FORCE_DIST', '0') != '1':
        return LocalComm()
    if transport == 'socket':
        return SocketComm()
    return RcclComm()
