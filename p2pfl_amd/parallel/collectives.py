"""Exchange primitives of the batched federation: RCCL over xGMI on GPUs (``torch.distributed`` backend
"nccl" on ROCm), gloo on the CPU for tests.

What crosses a GPU boundary in a round is decided by :mod:`p2pfl_amd.parallel.planner`; this module
moves it:

* :meth:`DeviceExchange.exchange_rows` - one variable-split ``all_to_all_single`` of exactly the
  parameter rows each peer needs (one launch, every xGMI link busy at once; gloo: batched
  isend/irecv);
* :meth:`DeviceExchange.allreduce_partial` - all-reduce(sum) of the per-GPU weighted partial sum when
  all aggregating nodes share one contributor set;
* :meth:`DeviceExchange.broadcast` - initial-model diffusion from the start node's GPU
  (reference: node.py:298-299,702-724).

With ``encrypt=True`` every row that leaves the GPU is AES-128-ECB encrypted on the device with the
key of its (src, dst) link and decrypted on arrival (reference: per-link AES on the socket stream,
node_connection.py:382-386). An all-reduce has no per-link ciphertext form, so the encrypted variant
is an all-to-all of per-link ciphertexts followed by a local sum. Link keys are RANDOM: the lower rank of
every pair draws 16 bytes from the OS and hands them to its peer point-to-point over the process group at
start-up (:meth:`DeviceExchange.negotiate_link_keys`; the reference likewise draws a random key per link and
sends it over the fresh socket, base_node.py:248-256). ``P2PFL_LINK_KEYS=derived`` (or a group-less exchange)
falls back to keys derived from the scenario name - reproducible, a COST emulation only, no secrecy.

A process group of ONE rank still goes through the collectives (self-split all-to-all, all-reduce, broadcast):
that is how the RCCL path is exercised on a single-GPU box (tests/test_rccl_gpu.py).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence

import torch
import torch.distributed as dist

from .planner import link_key


class DeviceExchange:
    def __init__(self, rank: int = 0, world_size: int = 1, process_group=None, device="cpu",
                 encrypt: bool = False, scenario: str = "scenario"):
        self.rank, self.world = rank, world_size
        self.pg = process_group
        self.device = torch.device(device)
        self.encrypt = encrypt
        self.scenario = scenario
        self._rk_cache: Dict[tuple, torch.Tensor] = {}
        self._link_keys: Dict[tuple, bytes] = {}     # negotiated random keys, (lo, hi) -> 16 bytes
        self._idx_cache: Dict[tuple, torch.Tensor] = {}
        self._send_buf: Optional[torch.Tensor] = None
        self._recv_buf: Optional[torch.Tensor] = None      # ciphertext of the rows received (encrypted exchange)
        self._wire_buf: Optional[torch.Tensor] = None      # per-peer ciphertexts / receive slots of the encrypted all-reduce
        self._crypt_buf: Dict[int, torch.Tensor] = {}
        self.bytes_sent = 0                     # payload bytes this rank pushed to peers (metrics)
        self.bytes_encrypted = 0                # bytes that went through the link cipher (both directions)
        if encrypt and process_group is not None and world_size > 1 and \
                os.environ.get("P2PFL_LINK_KEYS", "random") != "derived":
            self.negotiate_link_keys()

    # ------------------------------------------------------------------ link cipher
    def negotiate_link_keys(self) -> None:
        """Random 16-byte key per rank pair: drawn by the lower rank, sent point-to-point to the higher one."""
        on_dev = self._backend() == "nccl"
        mine = {r: torch.frombuffer(bytearray(os.urandom(16)), dtype=torch.uint8).clone()
                for r in range(self.rank + 1, self.world)}
        theirs = {r: torch.zeros(16, dtype=torch.uint8) for r in range(self.rank)}
        if on_dev:
            mine = {r: t.to(self.device) for r, t in mine.items()}
            theirs = {r: t.to(self.device) for r, t in theirs.items()}
        ops = [dist.P2POp(dist.isend, t, r, group=self.pg) for r, t in mine.items()]
        ops += [dist.P2POp(dist.irecv, t, r, group=self.pg) for r, t in theirs.items()]
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for r, t in list(mine.items()) + list(theirs.items()):
            self._link_keys[(min(self.rank, r), max(self.rank, r))] = bytes(t.cpu().tolist())
        self._rk_cache.clear()

    def link_key(self, a: int, b: int) -> bytes:
        k = (min(a, b), max(a, b))
        return self._link_keys.get(k) or link_key(self.scenario, a, b)

    def _round_keys(self, a: int, b: int) -> torch.Tensor:
        key = (min(a, b), max(a, b))
        t = self._rk_cache.get(key)
        if t is None:
            from ..utils.aes import expand_key
            t = torch.tensor(list(expand_key(self.link_key(a, b))), dtype=torch.uint8, device=self.device)
            self._rk_cache[key] = t
        return t

    def crypt(self, buf: torch.Tensor, peer: int, decrypt: bool, out: Optional[torch.Tensor] = None,
              local: Optional[int] = None) -> torch.Tensor:
        """AES-128-ECB over a [rows, P] fp32 buffer with the key of link (rank, peer); ``out`` (same shape,
        contiguous) receives the result, else a per-direction scratch buffer of this object is reused.
        ``local``: the near end of the link when it is not this rank (virtual devices sharing one GPU)."""
        if buf.numel() == 0:
            return buf
        me = self.rank if local is None else local
        self.bytes_encrypted += buf.numel() * 4
        raw = buf.contiguous().view(torch.uint8).reshape(-1)
        if buf.is_cuda:
            from ..ops import hip
            if out is None:
                scratch = self._crypt_buf.get(int(decrypt))
                if scratch is None or scratch.numel() < raw.numel():
                    scratch = self._crypt_buf[int(decrypt)] = torch.empty(raw.numel(), dtype=torch.uint8, device=buf.device)
                dst = scratch[:raw.numel()]
            else:
                dst = out.view(torch.uint8).reshape(-1)
            hip.aes128_ecb(raw, dst, self._round_keys(me, peer), decrypt)
            return dst.view(torch.float32).reshape(buf.shape)
        from ..utils.aes import ecb_host
        data = ecb_host(self.link_key(me, peer), raw.numpy().tobytes(), decrypt)
        res = torch.frombuffer(bytearray(data), dtype=torch.float32).reshape(buf.shape)
        if out is not None:
            out.copy_(res)
            return out
        return res

    def _backend(self) -> str:
        return dist.get_backend(self.pg)

    # ------------------------------------------------------------------ primitives
    def all_to_all(self, outs: List[torch.Tensor], ins: List[torch.Tensor]) -> None:
        if self._backend() == "nccl":
            dist.all_to_all(outs, ins, group=self.pg)
            return
        ops, staged = [], {}
        for r in range(self.world):
            if r == self.rank:
                outs[r].copy_(ins[r])
                continue
            staged[r] = torch.empty(outs[r].shape, dtype=outs[r].dtype)
            ops.append(dist.P2POp(dist.irecv, staged[r], r, group=self.pg))
            ops.append(dist.P2POp(dist.isend, ins[r].contiguous().cpu(), r, group=self.pg))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        for r, t in staged.items():
            outs[r].copy_(t)

    def exchange_rows(self, params: torch.Tensor, send_rows: Sequence[Sequence[int]],
                      recv_counts: Sequence[int], staging: torch.Tensor) -> int:
        """Send ``params[send_rows[r]]`` to every peer r, receive ``recv_counts[r]`` rows from it into
        ``staging`` (peer-major order). Returns the number of rows received."""
        n_recv = int(sum(recv_counts))
        if self.pg is None:            # no process group at all: a single shard, nothing to move
            return 0
        P = params.shape[1]
        send_counts = [len(l) for l in send_rows]
        n_send = sum(send_counts)
        # ONE gather builds the peer-major send buffer (rows of a 26 MB model: no per-peer temporaries); the index
        # tensor of a plan and the buffer itself are kept, so a steady-state round allocates nothing
        key = tuple(tuple(int(j) for j in l) for l in send_rows)
        idx = self._idx_cache.get(key)
        if idx is None:
            if len(self._idx_cache) > 64:
                self._idx_cache.clear()
            idx = self._idx_cache[key] = torch.as_tensor([j for l in key for j in l], dtype=torch.long,
                                                         device=self.device)
        if self._send_buf is None or self._send_buf.shape[0] < max(n_send, 1) or self._send_buf.shape[1] != P:
            self._send_buf = torch.empty(max(n_send, 1), P, device=self.device)
        send_cat = self._send_buf[:n_send]
        if n_send:
            torch.index_select(params, 0, idx, out=send_cat)
        self.bytes_sent += send_cat.numel() * 4
        if self.encrypt:
            off = 0
            for r, c in enumerate(send_counts):
                if c:       # in place: AES-ECB maps 16-byte blocks independently, so source == destination is fine
                    self.crypt(send_cat[off:off + c], r, decrypt=False, out=send_cat[off:off + c])
                    off += c
        if self._backend() == "nccl":
            # plaintext rows land straight in the staging buffer the reduce kernel reads; ciphertext in a scratch that
            # is kept between rounds (a steady-state round allocates nothing)
            if self.encrypt:
                if self._recv_buf is None or self._recv_buf.shape[0] < max(n_recv, 1) or self._recv_buf.shape[1] != P:
                    self._recv_buf = torch.empty(max(n_recv, 1), P, device=self.device)
                recv_cat = self._recv_buf[:n_recv]
            else:
                recv_cat = staging[:n_recv]
            dist.all_to_all_single(recv_cat, send_cat, output_split_sizes=list(recv_counts),
                                   input_split_sizes=send_counts, group=self.pg)
        else:
            # gloo: point-to-point on host memory (device tensors are staged through the CPU, which also lets
            # several ranks share ONE GPU in tests - RCCL refuses two ranks on a device)
            host_send = send_cat.cpu()
            host_recv = torch.empty(n_recv, P)
            ops, off, soff = [], 0, 0
            for r in range(self.world):
                if recv_counts[r]:
                    ops.append(dist.P2POp(dist.irecv, host_recv[off:off + recv_counts[r]], r, group=self.pg))
                    off += recv_counts[r]
                if send_counts[r]:
                    ops.append(dist.P2POp(dist.isend, host_send[soff:soff + send_counts[r]], r, group=self.pg))
                    soff += send_counts[r]
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
            recv_cat = host_recv.to(self.device)
        if self.encrypt:
            off = 0
            for r in range(self.world):
                if recv_counts[r]:
                    self.crypt(recv_cat[off:off + recv_counts[r]], r, True, out=staging[off:off + recv_counts[r]])
                    off += recv_counts[r]
        elif recv_cat.data_ptr() != staging.data_ptr():
            staging[:n_recv] = recv_cat
        return n_recv

    def allreduce_partial(self, partial: torch.Tensor) -> torch.Tensor:
        """Sum of every rank's [1, P] partial (weights already normalised by the global weight sum)."""
        if self.pg is None:
            return partial
        self.bytes_sent += partial.numel() * 4 * (self.world > 1)
        if not self.encrypt:
            if self._backend() != "nccl" and partial.is_cuda:
                host = partial.cpu()
                dist.all_reduce(host, group=self.pg)
                partial.copy_(host)
            else:
                dist.all_reduce(partial, group=self.pg)
            return partial
        # per-peer ciphertexts and receive slots live in one buffer kept between rounds: [2, world, *partial.shape]
        shape = (2, self.world) + tuple(partial.shape)
        if self._wire_buf is None or tuple(self._wire_buf.shape) != shape or self._wire_buf.device != partial.device:
            self._wire_buf = torch.empty(shape, dtype=partial.dtype, device=partial.device)
        enc = [self.crypt(partial, r, False, out=self._wire_buf[0, r]) if r != self.rank else partial
               for r in range(self.world)]
        outs = [self._wire_buf[1, r] for r in range(self.world)]
        self.all_to_all(outs, enc)
        total = torch.zeros_like(partial)
        for r, o in enumerate(outs):
            total += self.crypt(o, r, True) if r != self.rank else o
        return total

    def broadcast(self, t: torch.Tensor, src: int) -> torch.Tensor:
        """In-place broadcast of ``t`` from rank ``src`` (initial-model diffusion, node.py:298-299,702-724)."""
        if self.pg is not None:
            if self._backend() != "nccl" and t.is_cuda:
                host = t.cpu()
                dist.broadcast(host, src=src, group=self.pg)
                t.copy_(host)
            else:
                dist.broadcast(t, src=src, group=self.pg)
        return t

    def barrier(self) -> None:
        if self.pg is not None:
            dist.barrier(group=self.pg)
