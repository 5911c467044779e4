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
is an all-to-all of per-link ciphertexts followed by a local sum.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

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
        self.bytes_sent = 0                     # payload bytes this rank pushed to peers (metrics)

    # ------------------------------------------------------------------ link cipher
    def _round_keys(self, a: int, b: int) -> torch.Tensor:
        key = (min(a, b), max(a, b))
        t = self._rk_cache.get(key)
        if t is None:
            from ..utils.aes import expand_key
            t = torch.tensor(list(expand_key(link_key(self.scenario, a, b))), dtype=torch.uint8, device=self.device)
            self._rk_cache[key] = t
        return t

    def crypt(self, buf: torch.Tensor, peer: int, decrypt: bool) -> torch.Tensor:
        """AES-128-ECB over a [rows, P] fp32 buffer with the key of link (rank, peer)."""
        if buf.numel() == 0:
            return buf
        raw = buf.view(torch.uint8).reshape(-1)
        if buf.is_cuda:
            from ..ops import hip
            out = torch.empty_like(raw)
            hip.aes128_ecb(raw, out, self._round_keys(self.rank, peer), decrypt)
            return out.view(torch.float32).reshape(buf.shape)
        from ..utils.aes import ecb_host
        data = ecb_host(link_key(self.scenario, self.rank, peer), raw.numpy().tobytes(), decrypt)
        return torch.frombuffer(bytearray(data), dtype=torch.float32).reshape(buf.shape)

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
        if self.world == 1:
            return 0
        P = params.shape[1]
        send_counts = [len(l) for l in send_rows]
        flat = [int(j) for l in send_rows for j in l]
        # ONE gather builds the peer-major send buffer (rows of a 26 MB model: no per-peer temporaries)
        if flat:
            send_cat = params.index_select(0, torch.as_tensor(flat, device=self.device))
        else:
            send_cat = params.new_zeros(0, P)
        self.bytes_sent += send_cat.numel() * 4
        if self.encrypt:
            off = 0
            for r, c in enumerate(send_counts):
                if c:
                    send_cat[off:off + c] = self.crypt(send_cat[off:off + c], r, decrypt=False)
                    off += c
        if self._backend() == "nccl":
            recv_cat = staging[:n_recv] if not self.encrypt else torch.empty(n_recv, P, device=self.device)
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
                    staging[off:off + recv_counts[r]] = self.crypt(recv_cat[off:off + recv_counts[r]].contiguous(), r, True)
                    off += recv_counts[r]
        elif recv_cat.data_ptr() != staging.data_ptr():
            staging[:n_recv] = recv_cat
        return n_recv

    def allreduce_partial(self, partial: torch.Tensor) -> torch.Tensor:
        """Sum of every rank's [1, P] partial (weights already normalised by the global weight sum)."""
        if self.world == 1:
            return partial
        self.bytes_sent += partial.numel() * 4
        if not self.encrypt:
            if self._backend() != "nccl" and partial.is_cuda:
                host = partial.cpu()
                dist.all_reduce(host, group=self.pg)
                partial.copy_(host)
            else:
                dist.all_reduce(partial, group=self.pg)
            return partial
        enc = [self.crypt(partial, r, False) if r != self.rank else partial for r in range(self.world)]
        outs = [torch.empty_like(partial) for _ in range(self.world)]
        self.all_to_all(outs, enc)
        return sum(self.crypt(o, r, True) if r != self.rank else o for r, o in enumerate(outs))

    def broadcast(self, t: torch.Tensor, src: int) -> torch.Tensor:
        if self.world > 1:
            dist.broadcast(t, src=src, group=self.pg)
        return t

    def barrier(self) -> None:
        if self.world > 1:
            dist.barrier(group=self.pg)
