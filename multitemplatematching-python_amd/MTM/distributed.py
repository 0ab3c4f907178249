"""
Multi-GPU template sharding, one process per GPU: every rank holds the whole image and a subset of the units
(templates / rotations / scales); the only exchange step is an all-gather of the per-rank hit lists (24-byte
records), after which every rank runs the same global NMS.  (Several GPUs in ONE process need none of this:
``MTM.matchTemplates(..., devices="all")`` / ``_lib.Group``.)

The reference has no distributed code at all: its only parallelism is one thread-pool task per template
(MTM/__init__.py:172-175), which is the same independence this module exploits.

Exchange backends of :class:`HitExchange`
  "rccl"   : ncclAllGather inside libmtm_hip.so (mtm_comm_*; RCCL over xGMI).  The 128-byte unique id travels
             through a :class:`TcpStore` (rank 0 listens on MASTER_ADDR : MTM_STORE_PORT, default MASTER_PORT + 1).
  "tcp"    : the same fixed-slot protocol over the store's sockets - no GPU, no third-party package; control-plane
             fallback and CPU tests.
  "custom" : the caller supplies ``allgather_bytes(payload: bytes) -> list[bytes]`` (one entry per rank, rank
             order) built on whatever collective library it already runs (gloo, MPI ...).

Nothing here imports torch: the package's only dependencies are numpy and libmtm_hip.so.
"""
import contextlib
import os
import socket
import struct
import sys
import time
from typing import List, Sequence

import numpy as np

from . import _lib


def unit_cost(template, image_shape, masked=False) -> float:
    """Multiply-accumulates of the direct method: out_px * w * h * C (* 2 with a mask)."""
    th, tw = template.shape[:2]
    ch = 1 if template.ndim == 2 else template.shape[2]
    out = max(image_shape[0] - th + 1, 0) * max(image_shape[1] - tw + 1, 0)
    return float(out) * th * tw * ch * (2.0 if masked else 1.0)


def shard_units(costs: Sequence[float], world_size: int) -> List[List[int]]:
    """Longest-processing-time-first partition of unit indices over ranks (deterministic; the same rule as
    mtm_group_shards)."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0.0] * world_size
    shards = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        shards[r].append(i)
        loads[r] += costs[i]
    return [sorted(s) for s in shards]


@contextlib.contextmanager
def _stdout_to_stderr():
    """librccl prints a version banner on stdout when it is first initialised; programs whose stdout
    is a protocol (bench.py prints one JSON line) get it on stderr instead."""
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def _send(sock, payload: bytes):
    sock.sendall(struct.pack("<q", len(payload)) + payload)


_MAX_MESSAGE = 1 << 30          # bytes; a hit list of 2^25 records - anything longer is not one of ours


def _recv(sock) -> bytes:
    def exactly(n):
        buf = bytearray()
        while len(buf) < n:
            part = sock.recv(n - len(buf))
            if not part:
                raise ConnectionError("MTM TcpStore: peer closed the connection")
            buf += part
        return bytes(buf)
    (n,) = struct.unpack("<q", exactly(8))
    if not 0 <= n <= _MAX_MESSAGE:
        raise ConnectionError("MTM TcpStore: implausible message length %d" % n)
    return exactly(n)


class TcpStore:
    """Star-shaped rendezvous over plain sockets: rank 0 listens, every other rank keeps one connection to it.
    Carries the RCCL unique id (``broadcast``) and, for the "tcp" backend, the hit records (``allgather``).
    Address and port default to MASTER_ADDR / MTM_STORE_PORT (else MASTER_PORT + 1), the variables every
    launcher (torch.distributed.run, mpirun wrappers, srun) already exports."""

    def __init__(self, rank, world_size, addr=None, port=None, timeout=120.0, collective_timeout=None):
        """timeout: the rendezvous (listen / connect / announce).  collective_timeout: how long a later broadcast or
        allgather may wait for the slowest rank - None (default, also MTM_STORE_TIMEOUT unset) waits for ever, as a rank
        that is behind (warm-up, disk) is not an error; MTM_STORE_TIMEOUT=<seconds> or the argument bounds it."""
        self.rank, self.world_size = int(rank), int(world_size)
        if collective_timeout is None and os.environ.get("MTM_STORE_TIMEOUT"):
            collective_timeout = float(os.environ["MTM_STORE_TIMEOUT"])
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        if port is None:
            port = int(os.environ.get("MTM_STORE_PORT", int(os.environ.get("MASTER_PORT", "29500")) + 1))
        self.peers = {}
        self.sock = None
        if self.world_size == 1:
            return
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr if addr not in ("localhost",) else "127.0.0.1", int(port)))
            srv.listen(self.world_size)
            srv.settimeout(timeout)
            try:
                while len(self.peers) < self.world_size - 1:
                    conn, _ = srv.accept()
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    conn.settimeout(timeout)
                    (r,) = struct.unpack("<i", _recv(conn))
                    if not 1 <= r < self.world_size or r in self.peers:
                        conn.close()
                        raise ConnectionError("MTM TcpStore: a peer announced rank %d (world size %d, ranks seen %s)"
                                              % (r, self.world_size, sorted(self.peers)))
                    self.peers[r] = conn
            except BaseException:
                # a failed rendezvous (bad rank, time-out) must not leave the ranks that DID connect waiting in their
                # first collective for ever (collective_timeout=None): they see their connection closed instead
                for c in self.peers.values():
                    try:
                        c.close()
                    except OSError:
                        pass
                self.peers.clear()
                raise
            finally:
                srv.close()
            for conn in self.peers.values():
                conn.settimeout(collective_timeout)
        else:
            deadline = time.time() + timeout
            while True:
                try:
                    s = socket.create_connection((addr, int(port)), timeout=timeout)
                    break
                except OSError:
                    if time.time() > deadline:
                        raise
                    time.sleep(0.05)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            _send(s, struct.pack("<i", self.rank))
            s.settimeout(collective_timeout)
            self.sock = s

    def broadcast(self, payload=None) -> bytes:
        """rank 0's payload on every rank"""
        if self.world_size == 1:
            return payload
        if self.rank == 0:
            for r in sorted(self.peers):
                _send(self.peers[r], payload)
            return payload
        return _recv(self.sock)

    def allgather(self, payload: bytes) -> List[bytes]:
        if self.world_size == 1:
            return [payload]
        if self.rank == 0:
            parts = [payload] + [_recv(self.peers[r]) for r in range(1, self.world_size)]
            blob = struct.pack("<%dq" % len(parts), *[len(p) for p in parts]) + b"".join(parts)
            for r in sorted(self.peers):
                _send(self.peers[r], blob)
            return parts
        _send(self.sock, payload)
        blob = _recv(self.sock)
        lens = struct.unpack_from("<%dq" % self.world_size, blob)
        out, off = [], 8 * self.world_size
        for n in lens:
            out.append(blob[off:off + n])
            off += n
        return out

    def close(self):
        for c in self.peers.values():
            c.close()
        if self.sock is not None:
            self.sock.close()
        self.peers, self.sock = {}, None


class HitExchange:
    """All-gather of structured hit arrays (dtype _lib.HIT_DTYPE) across ranks."""

    def __init__(self, backend="tcp", rank=0, world_size=1, context=None, store=None, allgather_bytes=None):
        if backend not in ("rccl", "tcp", "custom"):
            raise ValueError("backend must be 'rccl', 'tcp' or 'custom'")
        self.backend = backend
        self.rank, self.world_size = rank, world_size
        self.ctx = context
        self.store = store
        self._allgather_bytes = allgather_bytes
        if backend == "custom" and world_size > 1 and allgather_bytes is None:
            raise ValueError("backend 'custom' needs allgather_bytes(payload) -> [bytes per rank]")
        if world_size > 1 and backend in ("rccl", "tcp") and store is None:
            self.store = TcpStore(rank, world_size)
        if backend == "rccl" and world_size > 1:
            self.ctx = context or _lib.default_context()
            with _stdout_to_stderr():
                uid = self.store.broadcast(_lib.comm_unique_id() if rank == 0 else None)     # bootstrap only
                self.ctx.comm_init(uid, world_size, rank)

    def allgather(self, hits: np.ndarray) -> np.ndarray:
        hits = np.ascontiguousarray(hits, dtype=_lib.HIT_DTYPE)
        if self.world_size == 1:
            return hits
        if self.backend == "rccl":
            out, _ = self.ctx.allgather_hits(hits)
            return out
        # host backends: one exchange of the raw records (variable length), concatenated in rank order
        fn = self._allgather_bytes if self.backend == "custom" else self.store.allgather
        parts = fn(hits.tobytes())
        if len(parts) != self.world_size:
            raise _lib.MtmError("hit exchange returned %d parts for %d ranks" % (len(parts), self.world_size))
        return np.frombuffer(b"".join(parts), dtype=_lib.HIT_DTYPE).copy()


def merge_and_nms(raw_all: np.ndarray, listTemplates, method, N_object, score_threshold, maxOverlap,
                  xOffset=0, yOffset=0):
    """Global NMS over the gathered hits; identical on every rank.  The gathered list is first put
    in the single-process order (template index, then the per-template order each rank produced)."""
    from . import _nms_raw, _to_hit_list
    idx = raw_all["templ_idx"]
    if len(idx) > 1 and bool((idx[1:] < idx[:-1]).any()):       # one rank: already in template order
        raw_all = raw_all[np.argsort(idx, kind="stable")]
    kept = _nms_raw(raw_all, score_threshold, method == 1, N_object, maxOverlap)
    return _to_hit_list(kept, listTemplates, xOffset, yOffset)


def _u8_units(sub, image, method):
    """[(template, mask or None)] if every template of `sub` takes the 8-bit path of MTM._raw_matches against `image`
    (uint8, same number of dimensions and channels, masks only where the method uses them and only well-formed ones),
    else None - then the general per-template pixel policy applies and the caller takes the step-by-step route."""
    if image.dtype != np.uint8 or image.ndim not in (2, 3):
        return None
    ichans = image.shape[2] if image.ndim == 3 else 1
    if ichans > 4:
        return None
    units = []
    for t in sub:
        a = t[1]
        if a.dtype != np.uint8 or a.ndim != image.ndim or (a.ndim == 3 and a.shape[2] != ichans):
            return None
        mask = None
        if len(t) >= 3:
            if method not in (0, 3):
                return None                 # (the reference warns about the ignored mask: the general route does)
            mask = t[2]
            if not (mask.shape == a.shape and mask.dtype == np.uint8):
                return None
        units.append((a, mask))
    return units


def matchTemplates_sharded(listTemplates, image, exchange: HitExchange, method=5, N_object=float("inf"),
                           score_threshold=0.5, maxOverlap=0.25, searchBox=None, find_local=None):
    """matchTemplates with the units sharded over exchange.world_size ranks.  Collective: every rank
    calls it with the same arguments and gets the same list back.  ``find_local`` (tests) replaces
    the GPU step: callable(sub_list, image) -> structured hits with LOCAL template indices."""
    from . import _raw_matches, _validate_search
    if maxOverlap < 0 or maxOverlap > 1:
        raise ValueError("Maximal overlap between bounding box is in range [0-1]")
    image, xOffset, yOffset = _validate_search(listTemplates, image, N_object, searchBox)
    costs = [unit_cost(t[1], image.shape, len(t) >= 3 and method in (0, 3)) for t in listTemplates]
    mine = shard_units(costs, exchange.world_size)[exchange.rank]
    sub = [listTemplates[i] for i in mine]
    # 8-bit inputs over the RCCL exchange: search, index remap, all-gather, merge and NMS in ONE native call
    # (mtm_find_matches_image_sharded_nms, round 5) - the same collective, the same selection
    if find_local is None and exchange.backend == "rccl" and exchange.ctx is not None and N_object != 1:
        units = _u8_units(sub, image, method)
        if units is not None:
            n_obj = -1 if N_object == float("inf") else int(N_object)
            with exchange.ctx.lock:
                raw = exchange.ctx.search_sharded_nms(units, image, method, score_threshold, maxOverlap, n_obj, mine)
            if method == 0:
                raise ValueError("The method TM_SQDIFF is not supported. Use TM_SQDIFF_NORMED instead.")
            from . import _to_hit_list
            return _to_hit_list(raw, listTemplates, xOffset, yOffset)
    if find_local is not None:
        raw = find_local(sub, image)
    elif sub:
        raw = _raw_matches(sub, image, method, N_object, score_threshold)
    else:
        raw = np.zeros(0, dtype=_lib.HIT_DTYPE)
    raw = raw.copy()
    raw["templ_idx"] = np.asarray(mine, dtype=np.int32)[raw["templ_idx"]] if len(raw) else raw["templ_idx"]
    gathered = exchange.allgather(raw)
    if method == 0:
        raise ValueError("The method TM_SQDIFF is not supported. Use TM_SQDIFF_NORMED instead.")
    return merge_and_nms(gathered, listTemplates, method, N_object, score_threshold, maxOverlap, xOffset, yOffset)
