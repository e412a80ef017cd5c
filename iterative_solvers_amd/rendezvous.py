"""Single-node control plane for the one-process-per-GPU launch (``torch.distributed.run`` or
any launcher that exports RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT).

Only three things cross this channel: the 128-byte RCCL unique id (rank 0 -> all), barriers,
and small float reductions (max of the timed region).  All solver data moves over RCCL/xGMI
inside ``libits_hip.so``.

Why not ``torch.distributed``: PyTorch's wheel bundles its own ROCm runtime (libamdhip64 /
librccl, ROCm 7.0) next to the system one (/opt/rocm, 7.2) that ``libits_hip.so`` is built
against; importing torch into the process that drives the library puts two HIP runtimes in one
address space (observed: segfault on the first launch, double free at exit).  So the processes
that load the library stay torch-free and rendezvous over a plain TCP socket on 127.0.0.1.

Protocol: rank 0 listens on an ephemeral port and publishes it in a file keyed by MASTER_PORT and
the launcher's pid; every other rank connects and keeps one socket to rank 0.  Collectives are
rank-0-rooted gather + broadcast of pickled-free, length-prefixed byte strings.
"""
from __future__ import annotations

import os
import socket
import struct
import time


def _recvn(s: socket.socket, n: int) -> bytes:
    buf = bytearray()
    while len(buf) < n:
        chunk = s.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("rendezvous peer closed the connection")
        buf += chunk
    return bytes(buf)


def _send_msg(s: socket.socket, payload: bytes):
    s.sendall(struct.pack("<I", len(payload)) + payload)


def _recv_msg(s: socket.socket) -> bytes:
    (n,) = struct.unpack("<I", _recvn(s, 4))
    return _recvn(s, n)


class Rendezvous:
    def __init__(self, rank: int | None = None, world: int | None = None, timeout: float = 300.0):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else rank
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
        self.local_rank = int(os.environ.get("LOCAL_RANK", str(self.rank)))
        self.peers: list[socket.socket] = []
        self.root: socket.socket | None = None
        self._file = None
        if self.world == 1:
            return
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        if addr in ("localhost", socket.gethostname()):
            addr = "127.0.0.1"
        key = f"{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'x')}_{os.getppid()}"
        path = os.path.join(os.environ.get("ITS_RDZV_DIR", "/tmp"), f"its_rdzv_{key}.port")
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, 0))
            srv.listen(self.world)
            port = srv.getsockname()[1]
            tmp = path + f".{os.getpid()}"
            with open(tmp, "w") as f:
                f.write(str(port))
            os.replace(tmp, path)
            self._file = path
            srv.settimeout(timeout)
            slots: dict[int, socket.socket] = {}
            while len(slots) < self.world - 1:
                c, _ = srv.accept()
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                (r,) = struct.unpack("<I", _recvn(c, 4))
                slots[r] = c
            srv.close()
            self.peers = [slots[r] for r in range(1, self.world)]
        else:
            deadline = time.time() + timeout
            while True:
                try:
                    with open(path) as f:
                        port = int(f.read().strip())
                    s = socket.create_connection((addr, port), timeout=5.0)
                    break
                except (OSError, ValueError):
                    if time.time() > deadline:
                        raise TimeoutError(f"rendezvous: rank 0 never published {path}")
                    time.sleep(0.05)
            s.settimeout(timeout)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.sendall(struct.pack("<I", self.rank))
            self.root = s

    # ---- collectives (rank 0 rooted)
    def broadcast(self, payload: bytes | None) -> bytes:
        if self.world == 1:
            return payload or b""
        if self.rank == 0:
            for p in self.peers:
                _send_msg(p, payload or b"")
            return payload or b""
        return _recv_msg(self.root)

    def gather(self, payload: bytes) -> list[bytes] | None:
        if self.world == 1:
            return [payload]
        if self.rank == 0:
            return [payload] + [_recv_msg(p) for p in self.peers]
        _send_msg(self.root, payload)
        return None

    def allgather_bytes(self, payload: bytes) -> list[bytes]:
        parts = self.gather(payload)
        if self.rank == 0:
            hdr = struct.pack(f"<{self.world}Q", *[len(p) for p in parts])
            blob = self.broadcast(hdr + b"".join(parts))
        else:
            blob = self.broadcast(None)
        lens = struct.unpack(f"<{self.world}Q", blob[:8 * self.world])
        out, o = [], 8 * self.world
        for n in lens:
            out.append(blob[o:o + n]); o += n
        return out

    def barrier(self):
        self.gather(b"b")
        self.broadcast(b"b")

    def allreduce_max(self, x: float) -> float:
        parts = self.gather(struct.pack("<d", x))
        if self.rank == 0:
            m = max(struct.unpack("<d", p)[0] for p in parts)
            return struct.unpack("<d", self.broadcast(struct.pack("<d", m)))[0]
        return struct.unpack("<d", self.broadcast(None))[0]

    def allgather_f64(self, xs: list[float]) -> list[list[float]]:
        fmt = f"<{len(xs)}d"
        parts = self.gather(struct.pack(fmt, *xs))
        if self.rank == 0:
            blob = self.broadcast(b"".join(parts))
        else:
            blob = self.broadcast(None)
        k = struct.calcsize(fmt)
        return [list(struct.unpack(fmt, blob[i * k:(i + 1) * k])) for i in range(self.world)]

    def close(self):
        for p in self.peers:
            p.close()
        if self.root is not None:
            self.root.close()
        if self._file:
            try:
                os.remove(self._file)
            except OSError:
                pass
