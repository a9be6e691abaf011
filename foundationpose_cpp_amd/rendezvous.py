"""Control plane of the one-process-per-GPU runs (bench.py --gpus N): a localhost TCP rendezvous with ONE primitive,
an all-gather of small byte strings, from which the ranks build what they need -- broadcast of the 128-byte ncclUniqueId, barriers, the
max-over-ranks of the step time.  No torch, no second HIP runtime in the rank processes: the data plane is the library's own
ncclAllGather (fp_register_sharded), this file only carries a few hundred bytes per run.

Who serves: `python bench.py --gpus N` (no WORLD_SIZE in the environment) spawns the ranks itself and serves from the parent
(`Server`, address handed down in FP_RDV_ADDR); under `python -m torch.distributed.run` rank 0 serves from a thread and publishes
its port in a file keyed by the launcher's pid and start time (the same for every rank of one launch, different for every launch).
"""
from __future__ import annotations

import os
import socket
import struct
import tempfile
import threading
import time


def _recv_exact(sock: socket.socket, n: int) -> bytes:
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("rendezvous: peer closed the connection")
        buf += chunk
    return bytes(buf)


def _send_msg(sock: socket.socket, payload: bytes) -> None:
    sock.sendall(struct.pack("<I", len(payload)) + payload)


def _recv_msg(sock: socket.socket) -> bytes:
    (n,) = struct.unpack("<I", _recv_exact(sock, 4))
    return _recv_exact(sock, n)


class Server:
    """Accepts `world` connections (each opens with its rank), then serves rounds: one message from every rank in, the rank-ordered
    list of all of them out to every rank.  Ends when every rank has closed its connection (or one of them breaks: the others get an
    error on their next round instead of waiting for ever)."""

    def __init__(self, world: int, host: str = "127.0.0.1", port: int = 0, timeout: float = 600.0):
        self.world, self.timeout = world, timeout
        self.sock = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        self.sock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        self.sock.bind((host, port))
        self.sock.listen(world)
        self.host, self.port = host, self.sock.getsockname()[1]
        self.error: Exception | None = None
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()

    @property
    def address(self) -> str:
        return f"{self.host}:{self.port}"

    def _run(self) -> None:
        conns: dict[int, socket.socket] = {}
        try:
            self.sock.settimeout(self.timeout)
            while len(conns) < self.world:
                c, _ = self.sock.accept()
                c.settimeout(self.timeout)
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                (rank,) = struct.unpack("<i", _recv_exact(c, 4))
                if not 0 <= rank < self.world or rank in conns:
                    raise ConnectionError(f"rendezvous: unexpected rank {rank}")
                conns[rank] = c
            while True:
                msgs = []
                for r in range(self.world):
                    try:
                        msgs.append(_recv_msg(conns[r]))
                    except ConnectionError:
                        if r == 0 and not msgs:
                            return              # rank 0 closed between rounds: the run is over
                        raise
                out = struct.pack("<I", self.world) + b"".join(struct.pack("<I", len(m)) + m for m in msgs)
                for r in range(self.world):
                    _send_msg(conns[r], out)
        except Exception as e:      # noqa: BLE001 -- kept for the owner; the ranks see their sockets close
            self.error = e
        finally:
            for c in conns.values():
                try:
                    c.close()
                except OSError:
                    pass
            self.sock.close()

    def join(self, timeout: float | None = None) -> None:
        self._thread.join(timeout)


class Client:
    def __init__(self, address: str, rank: int, world: int, timeout: float = 600.0):
        host, port = address.rsplit(":", 1)
        self.rank, self.world = rank, world
        deadline = time.monotonic() + timeout
        while True:
            try:
                self.sock = socket.create_connection((host, int(port)), timeout=timeout)
                break
            except OSError:
                if time.monotonic() > deadline:
                    raise
                time.sleep(0.05)
        self.sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
        self.sock.sendall(struct.pack("<i", rank))

    def all_gather(self, payload: bytes) -> list[bytes]:
        _send_msg(self.sock, payload)
        data = _recv_msg(self.sock)
        (n,) = struct.unpack_from("<I", data, 0)
        out, off = [], 4
        for _ in range(n):
            (ln,) = struct.unpack_from("<I", data, off)
            out.append(data[off + 4:off + 4 + ln])
            off += 4 + ln
        return out

    def broadcast(self, payload: bytes | None, root: int = 0) -> bytes:
        return self.all_gather(payload if self.rank == root and payload is not None else b"")[root]

    def barrier(self) -> None:
        self.all_gather(b"")

    def all_max(self, x: float) -> float:
        return max(struct.unpack("<d", b)[0] for b in self.all_gather(struct.pack("<d", x)))

    def all_min_int(self, x: int) -> int:
        return min(struct.unpack("<q", b)[0] for b in self.all_gather(struct.pack("<q", x)))

    def gather_floats(self, x: float) -> list[float]:
        return [struct.unpack("<d", b)[0] for b in self.all_gather(struct.pack("<d", x))]

    def close(self) -> None:
        try:
            self.sock.close()
        except OSError:
            pass


def _launch_key() -> str:
    """the same string in every rank of ONE launch, another one in every other launch: the launcher's pid and its start time"""
    ppid = os.getppid()
    start = "0"
    try:
        with open(f"/proc/{ppid}/stat") as f:
            start = f.read().rsplit(")", 1)[1].split()[19]
    except (OSError, IndexError):
        pass
    return f"{ppid}_{start}_{os.environ.get('MASTER_PORT', '0')}"


def connect(rank: int, world: int, timeout: float = 600.0) -> tuple[Client, Server | None]:
    """-> (client, server or None).  FP_RDV_ADDR set (bench.py's own launcher): just connect.  Otherwise (torch.distributed.run or any other
    launcher that sets RANK / WORLD_SIZE): rank 0 serves from a thread and publishes host:port in $TMPDIR/fp_rdv_<launch key>."""
    addr = os.environ.get("FP_RDV_ADDR")
    if addr:
        return Client(addr, rank, world, timeout), None
    path = os.path.join(tempfile.gettempdir(), f"fp_rdv_{_launch_key()}")
    server = None
    if rank == 0:
        server = Server(world, os.environ.get("MASTER_ADDR", "127.0.0.1") if os.environ.get("MASTER_ADDR", "127.0.0.1")[0].isdigit() else "127.0.0.1", 0, timeout)
        tmp = f"{path}.{os.getpid()}"
        with open(tmp, "w") as f:
            f.write(server.address)
        os.replace(tmp, path)
        addr = server.address
    else:
        deadline = time.monotonic() + timeout
        while not os.path.exists(path):
            if time.monotonic() > deadline:
                raise TimeoutError(f"rendezvous: rank 0 never published {path}")
            time.sleep(0.02)
        with open(path) as f:
            addr = f.read().strip()
    client = Client(addr, rank, world, timeout)
    client.barrier()                      # everyone has read the file ...
    if rank == 0:
        try:
            os.unlink(path)               # ... so it can go
        except OSError:
            pass
    return client, server
