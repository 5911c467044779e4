"""Host-side message mesh between the rank hosts of one multi-GPU job.

One process per GPU hosts a block of the scenario's nodes; the control plane of those nodes (runtime/host.py) must
span the ranks: a BEAT / ROLE of a node on rank 3 has to reach the TCP peers of a gateway on rank 0, START_LEARNING
that arrives at one rank starts the learning loops of all. That traffic is a few hundred bytes per message and has
nothing to do on the RCCL stream: the rank hosts keep one TCP socket per rank pair (full mesh, at most 8 x 7 / 2
sockets per node) and exchange length-prefixed frames on it. Addresses are exchanged once, at start-up, through the
job's host-side (gloo) process group.

frame:  u32 payload length | u8 kind | i32 sender (node index or -1) | payload
"""
from __future__ import annotations

import logging
import os
import socket
import struct
import threading
from typing import Callable, Dict, Optional

import torch.distributed as dist

HEADER = struct.Struct("!IBi")
MAX_PAYLOAD = 1 << 20      # control messages are a few hundred bytes; anything larger is a broken stream

# frame kinds
BUS = 1          # a wire-format control message of node `sender` (BEAT / ROLE / MODELS_READY / ...), payload = the message
START = 2        # learning starts on every rank: payload = json {rounds, epochs, init_src}
STOP = 3         # STOP_LEARNING reached one rank
NAME = 4         # node `sender` is now called payload (a gateway listens on another port than the node's name said)
XBEAT = 5        # an external (TCP) node beat at one of the sender rank's gateways: payload = its name


class RankBridge:
    def __init__(self, rank: int, world: int, group, on_frame: Callable[[int, int, int, bytes], None],
                 listen_addr: Optional[str] = None, timeout: float = 60.0):
        """Collective over ``group``: every rank constructs its bridge at the same point of the program."""
        self.rank, self.world = rank, world
        self.on_frame = on_frame
        self._peers: Dict[int, socket.socket] = {}
        self._locks: Dict[int, threading.Lock] = {}
        self._closed = threading.Event()
        addr = listen_addr or os.environ.get("P2PFL_BRIDGE_ADDR", "127.0.0.1")
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind((addr, 0))
        srv.listen(max(world, 8))
        srv.settimeout(timeout)
        addrs = [None] * world
        dist.all_gather_object(addrs, (addr, srv.getsockname()[1], os.urandom(16)), group=group)
        # the job's token (rank 0's draw, known to the ranks only - it travelled over the process group): whoever dials
        # a bridge port has to present it, so another process of the machine cannot pose as a rank
        token = addrs[0][2]
        # lower ranks are dialled, higher ranks dial in (the listen backlog holds them until the accepts below)
        for r in range(rank):
            s = socket.create_connection(tuple(addrs[r][:2]), timeout=timeout)
            s.sendall(struct.pack("!I", rank) + token)
            self._add(r, s)
        while len(self._peers) < world - 1:
            s, _peer = srv.accept()
            s.settimeout(timeout)
            try:
                hello = self._recv_exact(s, 4 + len(token))
            except (ConnectionError, OSError):
                s.close()
                continue
            r = struct.unpack("!I", hello[:4])[0]
            if hello[4:] != token or not rank < r < world or r in self._peers:
                logging.warning("[BRIDGE] rank %d: refused a connection that did not present the job's token", rank)
                s.close()
                continue
            self._add(r, s)
        srv.close()
        self._threads = [threading.Thread(target=self._reader, args=(r, s), name=f"rank-bridge-{rank}<-{r}", daemon=True)
                         for r, s in self._peers.items()]
        for t in self._threads:
            t.start()

    def _add(self, r: int, s: socket.socket) -> None:
        s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
        s.settimeout(None)
        self._peers[r] = s
        self._locks[r] = threading.Lock()

    @staticmethod
    def _recv_exact(s: socket.socket, n: int) -> bytes:
        buf = bytearray()
        while len(buf) < n:
            chunk = s.recv(n - len(buf))
            if not chunk:
                raise ConnectionError("rank bridge: peer closed the link")
            buf += chunk
        return bytes(buf)

    def _reader(self, r: int, s: socket.socket) -> None:
        try:
            while not self._closed.is_set():
                n, kind, sender = HEADER.unpack(self._recv_exact(s, HEADER.size))
                if n > MAX_PAYLOAD:
                    raise ConnectionError(f"rank bridge: frame of {n} bytes")
                payload = self._recv_exact(s, n) if n else b""

                try:
                    self.on_frame(r, kind, sender, payload)
                except Exception as e:      # noqa: BLE001 - a faulty handler must not take the link down
                    logging.exception(e)
        except (ConnectionError, OSError):
            if not self._closed.is_set():
                logging.info("[BRIDGE] rank %d: link to rank %d ended", self.rank, r)

    def send_all(self, kind: int, sender: int, payload: bytes = b"") -> None:
        frame = HEADER.pack(len(payload), kind, int(sender)) + payload
        for r, s in list(self._peers.items()):
            try:
                with self._locks[r]:
                    s.sendall(frame)
            except OSError:
                if not self._closed.is_set():
                    logging.info("[BRIDGE] rank %d: send to rank %d failed", self.rank, r)

    def close(self) -> None:
        self._closed.set()
        for s in self._peers.values():
            try:
                s.shutdown(socket.SHUT_RDWR)
            except OSError:
                pass
            s.close()
        self._peers.clear()
