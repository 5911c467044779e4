"""Shard engines: local training / evaluation of ALL federation nodes resident on one device.

The unit of execution is the *shard* (every virtual node of one GPU), not the node: parameters,
gradients and optimizer moments are ``[K, P]`` slabs and every kernel carries a leading client
dimension. Two interchangeable engines implement the same contract:

* :class:`HipEngine`   – the MI355X path: hand-written gfx950 kernels (``p2pfl_amd/csrc``), a whole
  local-training round (epochs x batches x {fwd, loss, bwd, optimizer}) captured into ONE hipGraph
  and replayed per federated round. No autograd, no per-client Python.
* :class:`EagerEngine` – one ``torch.nn`` module + ``torch.optim`` per client, looped in Python,
  fp32. This is the behaviour-equivalent of the reference's per-node ``LightningLearner.fit``
  (fedstellar/learning/pytorch/lightninglearner.py:113-143), serves as the CPU implementation,
  as the numerical oracle of the HIP engine, and as the measured eager baseline of BASELINE.md.

Both share the slab layout, so aggregation / exchange code is engine-agnostic.
"""
from __future__ import annotations

import math
import atexit
import logging
import os
import weakref
from collections.abc import Mapping
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch

from ..models.specs import ModelSpec
from ..models.torch_models import module_for_spec
from .slab import ParamLayout

PHASES = ("Train", "Validation", "Test")


class LazySummary(Mapping):
    """Read-only mapping of host tensors whose device-to-host copies are in flight until the first access."""

    def __init__(self, dev: Dict[str, torch.Tensor]):
        self._host = {k: torch.empty(v.shape, dtype=v.dtype, pin_memory=True) for k, v in dev.items()}
        for k, v in dev.items():
            self._host[k].copy_(v, non_blocking=True)
        self._event = torch.cuda.Event()
        self._event.record()

    def _wait(self) -> None:
        if self._event is not None:
            self._event.synchronize()
            self._event = None

    def __getitem__(self, key):
        self._wait()
        return self._host[key]

    def __iter__(self):
        return iter(self._host)

    def __len__(self):
        return len(self._host)


@dataclass
class PhaseStats:
    """Per-client accumulators of one phase (device tensors)."""
    loss_sum: torch.Tensor     # [K] fp32
    correct: torch.Tensor      # [K] int32
    count: torch.Tensor        # [K] int32
    confusion: torch.Tensor    # [K, C, C] int32 (row = target, col = prediction)

    @staticmethod
    def alloc(K: int, C: int, device) -> "PhaseStats":
        return PhaseStats(torch.zeros(K, device=device), torch.zeros(K, dtype=torch.int32, device=device),
                          torch.zeros(K, dtype=torch.int32, device=device),
                          torch.zeros(K, C, C, dtype=torch.int32, device=device))

    def zero_(self) -> None:
        self.loss_sum.zero_(); self.correct.zero_(); self.count.zero_(); self.confusion.zero_()

    def summary(self) -> "Dict[str, torch.Tensor]":
        """Host copies of the phase metrics. On a GPU the copies are ENQUEUED (pinned buffers, stream order: they see
        the counters as they are now, whatever is launched afterwards) and awaited at the first access, so a round
        can enqueue evaluation, training and aggregation back to back instead of idling the GPU at every report
        (1000 FEMNIST clients: 6.9 % of the round had no kernel running)."""
        cnt = self.count.clamp(min=1).float()
        dev = {"loss": self.loss_sum / cnt, "accuracy": self.correct.float() / cnt, "count": self.count,
               "confusion": self.confusion}
        if not self.loss_sum.is_cuda:
            return {k: v.clone() for k, v in dev.items()}
        return LazySummary(dev)


class ShardEngine:
    """Common state + contract of the two engines."""

    def __init__(self, spec: ModelSpec, n_clients: int, device="cpu", batch_size: int = 32, seed: int = 0,
                 persist_optimizer: bool = False, client_offset: int = 0):
        self.spec = spec
        self.K = int(n_clients)
        self.device = torch.device(device)
        self.B = int(batch_size)
        self.seed = seed
        self.persist_optimizer = persist_optimizer   # reference re-creates the optimizer every round
        self.layout = ParamLayout(spec)
        self.params = self.layout.alloc(self.K, self.device)
        self.params_alt = None                        # aggregation output (allocated on first use)
        self.stats = {ph: PhaseStats.alloc(self.K, spec.num_classes, self.device) for ph in PHASES}
        self.client_offset = int(client_offset)       # global id of client 0 (this rank's first node)
        self._perm_calls = 0
        self.train_x = self.train_y = self.val_x = self.val_y = self.test_x = self.test_y = None
        self.opt_steps = 0
        self._init_step_log(False)

    # ------------------------------------------------------------------ per-step Train/* series
    LOG_EVERY = 20

    def _init_step_log(self, log_steps: bool) -> None:
        """log_steps: keep the running (loss sum, sample count) of every optimizer step of a ``fit`` in a device table
        and, every LOG_EVERY steps, the confusion counters of that step and of the step before it - what the
        reference's models log per step (Train/Loss + macro metrics of the batch, mnist/models/mlp.py:33-106;
        log_every_n_steps=20, lightninglearner.py:188). Read back with step_losses / step_confusions / step_metrics."""
        self.log_steps = bool(log_steps)
        self._step_log = None
        self._step_cm = None
        self._cm_slot: Dict[int, int] = {}
        self._step_pos = 0

    def _alloc_step_tables(self, total: int) -> bool:
        """Tables for a fit of ``total`` optimizer steps; True when they were (re)allocated."""
        if not self.log_steps or (self._step_log is not None and self._step_log.shape[0] == total):
            return False
        self._step_log = torch.zeros(total, 2, self.K, device=self.device)
        logged = [p for p in range(total) if (p + 1) % self.LOG_EVERY == 0]
        self._cm_slot = {p: i for i, p in enumerate(sorted({q for p in logged for q in (p - 1, p)}))}
        if getattr(self.spec, "loss", "ce") != "ce":
            self._cm_slot = {}                        # no class metrics to derive: the loss series only
        C = self.stats["Train"].confusion.shape[-1]
        self._step_cm = torch.zeros(max(len(self._cm_slot), 1), self.K, C, C, dtype=torch.int32, device=self.device)
        return True

    def _record_client_step(self, pos: int, k: int) -> None:
        """Per-client engines (one replica at a time): snapshot client k's running Train counters after step ``pos``."""
        if self._step_log is None or pos >= self._step_log.shape[0]:
            return
        st = self.stats["Train"]
        self._step_log[pos, 0, k] = st.loss_sum[k]
        self._step_log[pos, 1, k] = st.count[k]
        slot = self._cm_slot.get(pos)
        if slot is not None:
            self._step_cm[slot, k].copy_(st.confusion[k])

    def step_confusions(self) -> Dict[int, torch.Tensor]:
        """{step (1-based, every LOG_EVERY-th): [K, C, C] confusion matrix of THAT step's batch} of the last ``fit``
        (needs ``log_steps``): the difference of the counters snapshotted inside the graph after this step and after
        the one before it."""
        if self._step_cm is None:
            return {}
        cms = self._step_cm.cpu().to(torch.int64)
        out = {}
        for pos, slot in sorted(self._cm_slot.items()):
            if (pos + 1) % self.LOG_EVERY:
                continue
            cur = cms[slot]
            d = cur - (cms[self._cm_slot[pos - 1]] if pos - 1 in self._cm_slot else 0)
            reset = d.flatten(1).min(dim=1).values < 0        # the per-epoch statistics were reset between the two steps
            d[reset] = cur[reset]
            out[pos + 1] = d
        return out

    def step_metrics(self) -> Dict[int, List[Dict[str, float]]]:
        """{step: per-client macro Accuracy / Precision / Recall / F1Score of that step's batch} - the per-step
        ``Train/*`` series of the reference's models (mnist/models/mlp.py:33-106), derived on the host."""
        from ..learning.metrics import metrics_from_confusion
        out = {}
        for step, cm in self.step_confusions().items():
            rows = []
            for k in range(cm.shape[0]):
                m = metrics_from_confusion(cm[k])
                m.pop("MicroAccuracy", None)
                rows.append(m)
            out[step] = rows
        return out

    def step_losses(self) -> Optional[torch.Tensor]:
        """[steps, K] mean training loss of every optimizer step of the last ``fit`` (needs ``log_steps``)."""
        if self._step_log is None:
            return None
        log = self._step_log.cpu()
        prev = torch.zeros_like(log[0])
        out = []
        for t in range(log.shape[0]):
            cur = log[t]
            if (cur[1] < prev[1]).any():          # the per-epoch statistics were reset before this step
                prev = torch.zeros_like(prev)
            d = cur - prev
            out.append(d[0] / d[1].clamp(min=1))
            prev = cur
        return torch.stack(out)

    # ------------------------------------------------------------------ parameters
    def init_params(self, seed: Optional[int] = None, identical: bool = True) -> None:
        """PyTorch default initialisation; ``identical`` = every node starts from the start node's
        weights (the reference's init-model diffusion, node.py:298-299,372-379)."""
        seed = self.seed if seed is None else seed
        if identical:
            sd = module_for_spec(self.spec, seed=seed).state_dict()
            row = torch.zeros(self.layout.row_stride)
            self.layout.state_dict_to_row(sd, row)
            self.params.copy_(row.to(self.device).unsqueeze(0).expand(self.K, -1))
        else:
            for k in range(self.K):
                sd = module_for_spec(self.spec, seed=seed + k).state_dict()
                self.layout.state_dict_to_row(sd, self.params[k])

    def get_state_dict(self, k: int):
        return self.layout.row_to_state_dict(self.params[k])

    def set_state_dict(self, k: int, sd) -> None:
        self.layout.state_dict_to_row(sd, self.params[k])

    def get_arrays(self, k: int):
        """Payload of client k: numpy arrays in state_dict order (reference lightninglearner.py:73-89)."""
        return self.layout.row_to_arrays(self.params[k])

    def set_arrays(self, k: int, arrays) -> None:
        self.layout.arrays_to_row(arrays, self.params[k])

    def alt_slab(self) -> torch.Tensor:
        if self.params_alt is None:
            self.params_alt = torch.zeros_like(self.params)
        return self.params_alt

    def swap_slabs(self) -> None:
        """Make the aggregation output the live parameters. Engines that captured pointers into
        ``params`` override this with a copy."""
        self.params.copy_(self.alt_slab())

    # ------------------------------------------------------------------ data
    def set_data(self, train_x, train_y, val_x=None, val_y=None, test_x=None, test_y=None) -> None:
        """``*_x`` [K, n, 784] fp32 (flattened 28x28, already normalised), ``*_y`` [K, n] integer."""
        def put(x, y):
            if x is None:
                return None, None
            x = x.reshape(self.K, x.shape[1], -1).to(self.device, torch.float32).contiguous()
            y = y.to(self.device, torch.int32).contiguous()
            # the loss kernel indexes logits and the confusion matrix with the label: an out-of-range one (a foreign
            # FEMNIST / SYSCALL file) must fail here, like torch's CrossEntropyLoss does, not corrupt device memory
            if y.numel() and (int(y.min()) < 0 or int(y.max()) >= self.spec.num_classes):
                raise ValueError(f"labels must lie in [0, {self.spec.num_classes}): got "
                                 f"[{int(y.min())}, {int(y.max())}]")
            return x, y
        self.train_x, self.train_y = put(train_x, train_y)
        self.val_x, self.val_y = put(val_x, val_y)
        self.test_x, self.test_y = put(test_x, test_y)
        self._on_data_changed()

    def _on_data_changed(self) -> None:
        pass

    def num_samples(self):
        n_tr = 0 if self.train_x is None else self.train_x.shape[1]
        n_te = 0 if self.test_x is None else self.test_x.shape[1]
        return n_tr, n_te

    def steps_per_epoch(self) -> int:
        return math.ceil(self.train_x.shape[1] / self.B)

    def _epoch_perms(self, epochs: int) -> torch.Tensor:
        """[epochs, K, steps*B] int32 shuffles (padded with wrapped indices; padded rows are masked).

        The shuffle of a client is a pure function of (seed, GLOBAL client id, fit call, epoch): a
        counter-based hash instead of a stateful generator, so a federation gives the same result however
        its nodes are split over ranks."""
        n = self.train_x.shape[1]
        steps = self.steps_per_epoch()
        i64 = torch.int64
        dev = self.device          # integer hash + argsort run where the engine lives (3.7 ms of host time per
        #                            round for 100 x 540 x 3 on the CPU; wrap-around int64 arithmetic is identical)
        e = torch.arange(epochs, dtype=i64, device=dev).view(-1, 1, 1)
        k = (torch.arange(self.K, dtype=i64, device=dev) + self.client_offset).view(1, -1, 1)
        i = torch.arange(n, dtype=i64, device=dev).view(1, 1, -1)
        x = i * -7046029254386353131                                                      # 0x9E3779B97F4A7C15 as int64
        c = (self._perm_calls * 1442695040888963407 + int(self.seed) * 3935559000370003845) & 0xFFFFFFFFFFFFFFFF
        c = c - (1 << 64) if c >= (1 << 63) else c                                          # python int -> int64 wrap
        x = x + k * -4658895280553007687 + e * 6364136223846793005 + c
        for mul in (-4658895280553007687, -7723592293110705685):                         # splitmix64 finaliser
            x = (x ^ (x >> 30)) * mul
        x = x ^ (x >> 31)
        self._perm_calls += 1
        perm = x.argsort(dim=2).to(torch.int32)
        if steps * self.B != n:   # wrap around: the extra rows are masked out through n_valid
            perm = perm[:, :, torch.arange(steps * self.B, device=dev) % n]
        return perm.contiguous()

    # ------------------------------------------------------------------ contract
    def reset_optimizer(self) -> None:
        raise NotImplementedError

    def fit(self, epochs: int, active: Optional[torch.Tensor] = None) -> Dict[str, Dict[str, torch.Tensor]]:
        raise NotImplementedError

    def evaluate(self) -> Dict[str, torch.Tensor]:
        raise NotImplementedError

    def synchronize(self) -> None:
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)


# ======================================================================================
# Eager (reference-equivalent) engine
# ======================================================================================
class EagerEngine(ShardEngine):
    """One ``nn.Module`` per client whose parameters alias the slab rows."""

    def __init__(self, *a, log_steps: bool = False, **kw):
        super().__init__(*a, **kw)
        self._init_step_log(log_steps)
        self.modules: List[torch.nn.Module] = []
        for k in range(self.K):
            m = module_for_spec(self.spec).to(self.device)
            for name, p in m.named_parameters():
                p.data = self.layout.view(self.params, name)[k]
            self.modules.append(m)
        self.optims = None

    def reset_optimizer(self) -> None:
        self.optims = [m.configure_optimizers() for m in self.modules]
        self.opt_steps = 0

    def _run_phase(self, k, x, y, phase, train, optim=None, order=None, step0: int = 0):
        m = self.modules[k]
        st = self.stats[phase]
        n = x.shape[0]
        C = self.spec.num_classes
        for s in range(0, n, self.B):
            idx = order[s:s + self.B] if order is not None else torch.arange(s, min(s + self.B, n), device=self.device)
            xb, yb = x[idx], y[idx].long()
            if train:                                        # (training_loss: criterion + predictions of the family -
                loss, pred = m.training_loss(xb, yb)         #  arg-max of the logits, p > 0.5 for the one-logit WADI MLP)
                optim.zero_grad(set_to_none=True)
                loss.backward()
                optim.step()
            else:
                with torch.no_grad():
                    loss, pred = m.training_loss(xb, yb)
            st.loss_sum[k] += loss.detach() * len(idx)
            st.count[k] += len(idx)
            if pred is not None:                             # (None: a family without classes - the auto-encoder)
                st.correct[k] += (pred == yb).sum().to(torch.int32)
                st.confusion[k].view(-1).index_add_(0, (yb * C + pred), torch.ones_like(pred, dtype=torch.int32))
            if train:
                self._record_client_step(step0 + s // self.B, k)

    def fit(self, epochs, active=None):
        if self.optims is None or not self.persist_optimizer:
            self.reset_optimizer()
        perms = self._epoch_perms(epochs).to(self.device).long()
        n = self.train_x.shape[1]
        self._alloc_step_tables(epochs * self.steps_per_epoch())
        hist = []
        for e in range(epochs):
            self.stats["Train"].zero_(); self.stats["Validation"].zero_()
            for k in range(self.K):
                if active is not None and not bool(active[k]):
                    continue
                self.modules[k].train()
                self._run_phase(k, self.train_x[k], self.train_y[k], "Train", True, self.optims[k], perms[e, k, :n],
                                step0=e * self.steps_per_epoch())
                if self.val_x is not None:
                    self.modules[k].eval()
                    self._run_phase(k, self.val_x[k], self.val_y[k], "Validation", False)
            hist.append({"Train": self.stats["Train"].summary(), "Validation": self.stats["Validation"].summary()})
        self.opt_steps += epochs * self.steps_per_epoch()
        return hist[-1] if hist else {}

    def evaluate(self):
        self.stats["Test"].zero_()
        for k in range(self.K):
            self.modules[k].eval()
            self._run_phase(k, self.test_x[k], self.test_y[k], "Test", False)
        return self.stats["Test"].summary()


# ======================================================================================
# HIP engine
# ======================================================================================
# Engines alive in this process: weak references to the ENGINES (plain Python objects), never to their streams or
# graphs. Round 2 kept weak references to the ``torch.cuda.Stream`` objects themselves and big shards then died in the
# interpreter's FINAL collection ("Fatal Python error: Segmentation fault ... Garbage-collecting"): at that point
# nothing orders the destructors - captured graphs (hipGraphExecDestroy), their streams and the weak-reference
# callbacks run in whatever order the collector finds them, possibly with replays still in flight and with parts of
# torch already torn down. What the trace shows is only "device objects destroyed during finalization"; which
# destructor faults was not isolated. The fix therefore removes the situation instead of one suspect: ownership is
# explicit - ``close()`` synchronises the device, drops the graphs, then the side stream - and ``_close_live_engines``
# runs it from ``atexit``, i.e. BEFORE finalization starts and with the HIP runtime alive, so the final collection
# only ever sees engines that hold no device objects.
_LIVE_ENGINES: "weakref.WeakSet" = weakref.WeakSet()


def _close_live_engines() -> None:
    for eng in list(_LIVE_ENGINES):
        try:
            eng.close()
        except Exception:     # noqa: BLE001 - shutting down: nothing useful to do with an error here
            pass


atexit.register(_close_live_engines)


def pending_stream_events():
    """(engine id, event name) of the first unsignalled debug events of every live engine (P2PFL_STREAM_DEBUG=1)."""
    out = []
    for obj in list(_LIVE_ENGINES):
        if obj._stream_debug:
            names = [(n, ev.query()) for n, ev in obj._stream_debug]
            done = sum(1 for _, q in names if q)
            out.append((id(obj), done, len(names), [n for n, q in names if not q][:4]))
    return out


def live_aux_streams():
    """(role, stream) of every auxiliary HIP stream of the engines alive in this process - for the stall
    diagnostics (tests/conftest.py, tools/diag_streams.py)."""
    return [("side", obj._side) for obj in list(_LIVE_ENGINES)
            if getattr(obj, "overlap", False) and getattr(obj, "_side", None) is not None]


class HipEngine(ShardEngine):
    """Client-batched gfx950 kernels + hipGraph replay (see module docstring)."""

    def __init__(self, *a, use_graph: bool = True, fuse_optimizer: bool = True, overlap: bool = True,
                 precision: Optional[str] = None, log_steps: bool = False, **kw):
        super().__init__(*a, **kw)
        # log_steps: keep the running (loss sum, sample count) of every optimizer step in a device table written
        # from inside the captured graph, so per-step Train/Loss can be reported like the reference's
        # log_every_n_steps=20 (lightninglearner.py:188) without leaving the graph
        # ... and, every LOG_EVERY steps, the confusion counters of that step and of the step before it: their
        # difference is the confusion matrix of ONE batch, from which the host derives the per-step
        # Train/{Accuracy,Precision,Recall,F1Score} series of the reference's models (mnist/models/mlp.py:33-106:
        # torchmetrics on the step's batch, logged every 20 steps)
        self._init_step_log(log_steps)
        self._in_prepare = False
        if self.device.type != "cuda":
            raise RuntimeError("HipEngine needs a GPU device")
        from ..ops import hip
        hip.load()   # fail loudly if the kernels were not built
        # precision policy of every matrix product of this engine: "fp32" (default; the reference trains 32-bit,
        # fedstellar/learning/pytorch/lightninglearner.py:182-193) or "bf16" (bf16 operands, fp32 accumulation)
        self.precision = precision or os.environ.get("P2PFL_PRECISION") or hip.default_precision()
        self.hip = hip = hip.bound(self.precision)
        self.use_graph = use_graph
        K, B, L = self.K, self.B, self.layout
        dev = self.device
        self.grads = None            # allocated below, once the fused layers (whose gradients never reach memory) are known
        o = self.spec.optim
        self.m = L.alloc(K, dev)
        self.v = L.alloc(K, dev) if o.kind == "adam" else None
        # activations / activation gradients / pool codes per layer
        self.act, self.code = self._alloc_acts(B)
        self.dact = [torch.zeros_like(a) for a in self.act]
        # evaluation runs with its own, larger batch (forward only: no gradient buffers) so that the big
        # dense layer streams its weights once per up to EVAL_B rows instead of once per 32
        self._eval_bufs: Dict[int, tuple] = {}
        self._fwd_scratch: Dict[tuple, Optional[torch.Tensor]] = {}
        # tap-major weight copies (in the engine's precision) of the stage-2 convolutions (refreshed at every forward pass)
        self.wprep = {i: hip.conv2_prep_alloc(K, l.ksize, dev) for i, l in enumerate(self.spec.layers)
                      if l.kind == "conv2pool"}
        # partial-sum scratch of the batch-split stage-2 weight gradient (small shards only)
        self.w2_scratch = {i: hip.conv2_wgrad_scratch(K, B, l.ksize, dev) for i, l in enumerate(self.spec.layers)
                           if l.kind == "conv2pool"}
        self.karange = torch.arange(K, device=dev).unsqueeze(1)
        self._nv_cache: Dict[int, torch.Tensor] = {}
        self._tail_cache: Dict[tuple, torch.Tensor] = {}
        # big linear layers: weight gradient fused with the optimizer update (gradient never hits HBM)
        self._fused = set()
        # The fused update of a big layer can also emit that layer's data gradient (one pass over the weights less),
        # at the price of the convolution backward no longer running BESIDE the update of the same clients - which the
        # two-group schedule gives back (the other group's convolution chain runs there). Both fp32 policies take it:
        # "fp32" 310 -> 290 ms per headline round when it was introduced; "fp32-mfma" lost with its first convolution
        # kernels (336 -> 340 ms) and wins since their tile / tap counts are exact (293.9 -> 291.2 ms, with the
        # small-footprint convolution variants inside the groups 290.4; tools/probe/mfma_fuse.sh). "bf16": the short
        # convolutions hide completely. P2PFL_FUSE_DGRAD=0/1 overrides for the fp32 policies.
        env = os.environ.get("P2PFL_FUSE_DGRAD")
        self._fuse_dgrad = self.precision != "bf16" and (self.precision in ("fp32", "fp32-mfma") if env is None else env != "0")
        self._adam_ranges = [(0, L.row_stride)]      # slab ranges the element-wise optimizer kernel owns
        if o.kind in ("adam", "sgd") and fuse_optimizer and not os.environ.get("P2PFL_NO_FUSE"):
            cut = []
            for i, l in enumerate(self.spec.layers):
                ws, bs = L.by_name[l.weight], L.by_name[l.bias]
                # per client >= 1 M weights, or a shard of them that no longer fits the 256 MB infinity cache -
                # or the L2s, when the layer fills whole 128-row tiles. (100 MLP clients, 784 -> 256: 17.7 ->
                # 16.4 ms per round fused; 50 FEMNIST clients, 3136 -> 62, i.e. half-empty tiles: the fused
                # kernel is 22 us per step SLOWER than wgrad + SGD.)
                shard = K * ws.numel
                big = ws.numel >= (1 << 20) or (ws.numel >= (1 << 16) and (
                    shard >= (1 << 26) or (shard >= (1 << 24) and l.out_features % 128 == 0)))
                if l.kind == "linear" and big and bs.offset == ws.offset + -(-ws.numel // 64) * 64:
                    self._fused.add(i)
                    cut.append((ws.offset, bs.offset + -(-bs.numel // 64) * 64))
            ranges, pos = [], 0
            for a, b in sorted(cut):
                if a > pos:
                    ranges.append((pos, a))
                pos = b
            if pos < L.row_stride:
                ranges.append((pos, L.row_stride))
            self._adam_ranges = ranges
        # Gradient slab: the columns of the fused layers are LEFT OUT (their gradient tile lives in the accumulators of
        # the fused kernel and is never stored) - for the MNIST-CNN that is 96 % of the row, 25.7 MB per client, i.e.
        # a quarter of the optimizer-state memory of a shard sized for the 288 GB of the card.
        self._gcut = sorted((a, b) for a, b in (cut if self._fused else []))
        self.grads = torch.zeros((K, L.row_stride - sum(b - a for a, b in self._gcut)), dtype=torch.float32, device=dev)
        # everything the big fused layers do not own is cleared when a round starts a new optimizer
        self._zero_ranges = list(self._adam_ranges)
        # Small layers: the optimizer step runs in the EPILOGUE of the kernel that holds the layer's final gradient
        # (fixed-order reduction passes of the convolution weight gradients; the one-kernel class head), so their
        # gradients never reach memory and the element-wise optimizer launches (4 per step, each waiting for wave
        # slots beside the other group's HBM-bound update: 20 us alone, 270 us in the schedule) are gone.
        # P2PFL_NO_FOLD=1 keeps the separate launches (A/B runs).
        self._fold = set()
        self._head = None
        last = len(self.spec.layers) - 1
        ll = self.spec.layers[last]
        # Two forms (csrc/head.hip). Wide heads (In >= 1024) take two kernels over (client, column slice) grids at any
        # client count. Narrow ones take the one-kernel form, one workgroup per client: below ~32 clients that leaves
        # most of the chip idle and the four small kernels it replaces are faster (P2PFL_HEAD_MIN_K overrides).
        self._head_scratch = None
        if (ll.kind == "linear" and last > 0 and not ll.relu and last not in self._fused and self.spec.loss == "ce"
                and not os.environ.get("P2PFL_NO_HEAD") and hip.linear_head_accepts(B, ll.in_features, ll.out_features)):
            self._head_scratch = hip.linear_head_scratch(K, B, ll.in_features, ll.out_features, dev)
            if self._head_scratch is not None or K >= int(os.environ.get("P2PFL_HEAD_MIN_K", "32")):
                self._head = last
        if o.kind in ("adam", "sgd") and fuse_optimizer and not os.environ.get("P2PFL_NO_FUSE") \
                and not os.environ.get("P2PFL_NO_FOLD"):
            for i, l in enumerate(self.spec.layers):
                if l.kind in ("conv1pool", "conv2pool") or i == self._head:
                    self._fold.add(i)
            cut = []
            for i in sorted(self._fold):
                l = self.spec.layers[i]
                for slot in (L.by_name[l.weight], L.by_name[l.bias]):
                    cut.append((slot.offset, slot.offset + -(-slot.numel // 64) * 64))
            ranges = []
            for a, b in self._adam_ranges:
                pos = a
                for ca, cb in sorted(cut):
                    if cb <= pos or ca >= b:
                        continue
                    if ca > pos:
                        ranges.append((pos, ca))
                    pos = max(pos, cb)
                if pos < b:
                    ranges.append((pos, b))
            self._adam_ranges = ranges
        self.c1_scratch = {i: hip.conv1_wgrad_scratch(K, B, l.ksize, dev) for i, l in enumerate(self.spec.layers)
                           if l.kind == "conv1pool"}
        for i in self.w2_scratch:
            if i in self._fold:
                self.w2_scratch[i] = hip.conv2_wgrad_scratch_opt(K, B, self.spec.layers[i].ksize, dev)
        self._graphs: Dict[tuple, tuple] = {}
        self._capturing = False
        self._stream_debug = [] if os.environ.get("P2PFL_STREAM_DEBUG") else None
        self._perm_buf = None
        self._bc_buf = None
        self._eval_graph = None
        # Two-stream software pipeline (see _train_body_pipelined): only pays when the first dense layer
        # behind the convolutions is a big fused wgrad+Adam layer (HBM-bound for milliseconds) under
        # which the MFMA/LDS-bound convolution kernels can hide.
        self.nc = sum(1 for l in self.spec.layers if l.kind != "linear")
        # (FEMNIST-CNN: the head has 0.19 M weights, its fused update takes 0.6 ms for 1000 clients next to
        # 3.5 ms of MFMA-bound convolutions - overlapped it ran 38..61 ms per round from run to run against a
        # steady 42 ms in order, so the pipeline is kept for update-dominated models only.)
        first_dense = self.spec.layers[self.nc] if self.nc < len(self.spec.layers) else None
        self.overlap = (bool(overlap) and self.nc > 0 and self.nc in self._fused
                        and L.by_name[first_dense.weight].numel >= (1 << 20))
        if self.overlap:
            # SAME priority as the launch stream, on purpose. Root cause of the round-1 "eager two-stream stall"
            # (tools/diag_stall.py, docs/kernels.rst): with a HIGH-priority side stream (priority -1) the join
            # `main.wait_stream(side)` - a normal-priority stream waiting for an event of a high-priority one - never
            # resolves on the 4th high-priority stream a process creates (2nd engine of the 2nd batch of engines;
            # deterministic on ROCm 7.2, GPU_MAX_HW_QUEUES 4 or 8 alike): side's event IS signalled, main's
            # barrier on it is not released, both streams stay BUSY for good. With equal priorities the same
            # schedule ran 100 x 2 engines, in turns and from two threads at once, without a stall - and the
            # priority bought nothing anyway (headline 231.4 vs 231.4 ms, cnn10-fully 303.7 vs 306.0,
            # femnist50-ring 37.4 vs 37.5). P2PFL_SIDE_PRIO overrides for experiments.
            self._side = torch.cuda.Stream(device=dev, priority=int(os.environ.get('P2PFL_SIDE_PRIO', '0')))
            # (Tried for few clients, where the side chain is as long as the fused update: the stage-2 weight
            # gradient on a third stream, and on the main stream in front of the fused update - no gain over the
            # side stream: 10 clients 313 vs 302 ms, 50 FEMNIST clients 54.6 vs 55.3.)
            self._act_pp = [self.act[self.nc - 1], torch.zeros_like(self.act[self.nc - 1])]
        # exact-split policy, update without fused data gradient (small shards): the stage-2 weight gradient runs on the
        # launch stream in front of the fused update (P2PFL_WGRAD_MAIN=0/1 overrides)
        # Models whose step is a chain of short convolution kernels (FEMNIST-CNN: no HBM-bound update to hide under, so
        # no two-stream pipeline) still have two INDEPENDENT chains in their backward pass: the weight gradients of the
        # dense head and of the stage-2 convolution (+ their optimizer steps) need nothing from the data-gradient chain
        # conv2 dgrad -> conv1 wgrad. They run on a second stream beside it (fork after the loss, join before the next
        # forward). P2PFL_BWD_FORK=0 keeps the in-order backward.
        convs = [i for i, l in enumerate(self.spec.layers) if l.kind != "linear"]
        self._bwd_fork = (not self.overlap and self._head is None and o.kind in ("adam", "sgd")
                          and any(l.kind == "conv2pool" for l in self.spec.layers)
                          and len(self.spec.layers) == len(convs) + 1 and self.spec.layers[-1].kind == "linear"
                          and len(self.spec.layers) - 1 not in self._fused
                          and all(i in self._fold for i in convs)     # their steps ride in their gradient kernels
                          and os.environ.get("P2PFL_BWD_FORK", "1") != "0")
        if self._bwd_fork:
            self._side = torch.cuda.Stream(device=dev)
        env = os.environ.get("P2PFL_WGRAD_MAIN")
        self._wgrad_on_main = (self.overlap and self.nc >= 2 and self.spec.layers[self.nc - 1].kind == "conv2pool"
                               and (self.precision == "fp32" if env is None else env != "0"))
        self._wgrad_ev = None
        _LIVE_ENGINES.add(self)

    def close(self) -> None:
        """Release the captured graphs and then the side stream now (while the HIP runtime is certainly alive, after a
        device synchronise) instead of at interpreter shutdown; idempotent, also run from ``atexit``."""
        if self.device.type == "cuda" and (self._graphs or self._eval_graph is not None
                                           or getattr(self, "_side", None) is not None):
            torch.cuda.synchronize(self.device)
        self._graphs.clear()
        self._eval_graph = None
        self._wgrad_ev = None
        if self._stream_debug:
            self._stream_debug.clear()
        if getattr(self, "_side", None) is not None:
            self._side = None
            self.overlap = False          # (a closed engine that is used again runs the single-stream schedule)
            self._bwd_fork = False
        _LIVE_ENGINES.discard(self)

    EVAL_B = 256

    def _alloc_acts(self, B: int):
        act, code = [], []
        for l in self.spec.layers:
            if l.kind == "conv1pool":
                shp = (self.K, B, l.cout, 196)
            elif l.kind == "conv2pool":
                shp = (self.K, B, l.cout, 49)
            else:
                shp = (self.K, B, l.out_features)
            act.append(torch.zeros(shp, device=self.device))
            code.append(torch.zeros(shp, dtype=torch.uint8, device=self.device) if l.kind != "linear" else None)
        return act, code

    def _eval_batch(self, n: int) -> int:
        """Rows per evaluation pass over a split of n samples: a multiple of 32, at least the training batch."""
        return max(self.B, min(self.EVAL_B, -(-n // 32) * 32))

    def _bufs(self, B: int):
        """(act, code, B) for batch size B (the training buffers, or lazily made evaluation ones)."""
        if B == self.B:
            return self.act, self.code, B
        if B not in self._eval_bufs:
            self._eval_bufs[B] = self._alloc_acts(B) + (B,)
        return self._eval_bufs[B]

    def _linear_fwd(self, x, W, b, y, relu, sidx=None) -> None:
        """hip.linear_fwd with an engine-owned scratch buffer per problem shape (made at the first, eager, call)."""
        key = (y.shape[1], W.shape[2], W.shape[1])
        if key not in self._fwd_scratch:
            self._fwd_scratch[key] = self.hip.linear_fwd_scratch(self.K, *key, self.device)
        self.hip.linear_fwd(x, W, b, y, relu, sidx=sidx, scratch=self._fwd_scratch[key])

    def _linear_dgrad(self, dy, W, mask, dx) -> None:
        """hip.linear_dgrad with an engine-owned scratch buffer per problem shape (made at the first, eager, call)."""
        key = ("dgrad", dy.shape[1], W.shape[2], W.shape[1])
        if key not in self._fwd_scratch:
            self._fwd_scratch[key] = self.hip.linear_dgrad_scratch(self.K, *key[1:], self.device)
        self.hip.linear_dgrad(dy, W, mask, dx, scratch=self._fwd_scratch[key])

    # ------------------------------------------------------------------ helpers
    def _w(self, slab, l):
        return self.layout.view2d(slab, l.weight), self.layout.view(slab, l.bias)

    def _gcol(self, off: int) -> int:
        """Column of the (compact) gradient slab that holds slab column ``off`` (outside the fused layers' ranges)."""
        return off - sum(b - a for a, b in self._gcut if b <= off)

    def _gcols(self, a: int, b: int) -> torch.Tensor:
        lo = self._gcol(a)
        return self.grads[:, lo:lo + (b - a)]

    def grad_view(self, name: str) -> torch.Tensor:
        """``[K, *shape]`` view of the gradient of parameter ``name`` (not kept for the layers of ``_fused``)."""
        s_ = self.layout.by_name[name]
        if any(a <= s_.offset < b for a, b in self._gcut):
            raise KeyError(f"{name}: the gradient of a fused layer is never stored")
        return self._gcols(s_.offset, s_.offset + s_.numel).unflatten(1, s_.shape)

    def _gw(self, l):
        """(weight gradient as [K, out, in...] matrix view, bias gradient) of layer ``l``; (None, None): fused."""
        ws = self.layout.by_name[l.weight]
        if any(a <= ws.offset < b for a, b in self._gcut):
            return None, None
        gw = self._gcols(ws.offset, ws.offset + ws.numel).unflatten(1, (ws.shape[0], ws.numel // ws.shape[0]))
        return gw, self.grad_view(l.bias)

    def reset_optimizer(self) -> None:
        # moments of the fused layers are not cleared when every round starts a new optimizer: their first
        # fused step overwrites them without reading (saves two passes over the biggest tensors per round)
        ranges = self._zero_ranges if (self._fused and not self.persist_optimizer) else [(0, self.layout.row_stride)]
        for a, b in ranges:
            self.m[:, a:b].zero_()
            if self.v is not None:
                self.v[:, a:b].zero_()
        self.opt_steps = 0

    def swap_slabs(self) -> None:
        # captured graphs hold pointers into self.params: copy instead of swapping storages
        self.params.copy_(self.alt_slab())

    def _on_data_changed(self) -> None:
        self._graphs.clear()
        self._eval_graph = None

    # ------------------------------------------------------------------ one batch
    def _forward(self, data, sidx, labels, n_valid, phase: str, train: bool, bufs=None, prep: bool = True) -> None:
        hip, layers = self.hip, self.spec.layers
        act, code, B = bufs if bufs is not None else (self.act, self.code, self.B)
        x = None
        for i, l in enumerate(layers):
            if train and i == self._head:      # forward, loss and backward of the head are one kernel: _backward starts with it
                self._head_in = (x, labels, sidx, n_valid)
                return
            W, b = self._w(self.params, l)
            if l.kind == "conv1pool":
                hip.conv1_fwd(data, sidx, W, b, act[i], code[i], l.ksize)
            elif l.kind == "conv2pool":
                if prep:     # (the weights do not change between the passes of one evaluation)
                    hip.conv2_prep(W, self.wprep[i], l.ksize)
                hip.conv2_fwd(x, W, b, act[i], code[i], l.ksize, wprep=self.wprep[i])
            else:
                if x is None:      # first layer: the mini-batch rows are gathered by the kernel's operand loader
                    self._linear_fwd(data if sidx is not None else data[:, :B], W, b, act[i], l.relu, sidx=sidx)
                else:
                    self._linear_fwd(x.flatten(2) if x.dim() > 3 else x, W, b, act[i], l.relu)
            x = act[i]
        st = self.stats[phase]
        if self.spec.loss == "mse_input":      # auto-encoder: reconstruction error against the batch's own input rows
            hip.mse_input(x, data if sidx is not None else data[:, :B], sidx, self.dact[-1] if train else None,
                          st.loss_sum, st.count, n_valid)
        else:
            hip.softmax_ce(x, labels, sidx, self.dact[-1] if train else None, st.loss_sum, st.correct, st.count,
                           st.confusion, n_valid)
        if train:
            self._record_step()

    def _record_step(self) -> None:
        if self._step_log is not None and self._step_pos < self._step_log.shape[0]:
            st = self.stats["Train"]
            self._step_log[self._step_pos, 0].copy_(st.loss_sum)
            self._step_log[self._step_pos, 1].copy_(st.count)
            slot = self._cm_slot.get(self._step_pos)
            if slot is not None and self._step_cm is not None:
                self._step_cm[slot].copy_(st.confusion)
            self._step_pos += 1

    def _backward(self, data, sidx, step_index: int = 1) -> None:
        hip, layers = self.hip, self.spec.layers
        o = self.spec.optim
        dy = self.dact[-1]
        if self._bwd_fork:
            self._backward_forked(data, sidx, step_index)
            return
        for i in range(len(layers) - 1, -1, -1):
            l = layers[i]
            W, _ = self._w(self.params, l)
            gW, gb = self._gw(l)
            if i == self._head:
                x, labels, lsidx, n_valid = self._head_in
                self._head_step(x, labels, lsidx, n_valid, step_index)
                dy = self.dact[i - 1]
            elif l.kind == "linear":
                gather = None
                if i == 0:
                    xin, gather = (data, sidx) if sidx is not None else (data[:, :self.B], None)
                else:
                    xin = self.act[i - 1].flatten(2) if self.act[i - 1].dim() > 3 else self.act[i - 1]
                dy_next = None
                took = False
                if i > 0:   # data gradient first: it needs the weights BEFORE the fused update below
                    prev = layers[i - 1]
                    mask = self.act[i - 1] if (prev.kind == "linear" and prev.relu) else None
                    dx = self.dact[i - 1].flatten(2) if self.dact[i - 1].dim() > 3 else self.dact[i - 1]
                    if i in self._fused and mask is None and self._fuse_dgrad:
                        took = self._fused_update(l, dy, xin, step_index, sidx=gather, dx=dx)
                    if not took:
                        self._linear_dgrad(dy, W, mask, dx)
                    dy_next = self.dact[i - 1]
                if took:
                    pass
                elif i in self._fused:
                    self._fused_update(l, dy, xin, step_index, sidx=gather)
                else:
                    hip.linear_wgrad(dy, xin, gW, gb, sidx=gather)
                if dy_next is not None:
                    dy = dy_next
            elif l.kind == "conv2pool":
                hip.conv2_wgrad(dy, self.code[i], self.act[i - 1], gW, gb, l.ksize, scratch=self.w2_scratch[i],
                                opt=self._opt_epilogue(i, step_index))
                hip.conv2_dgrad(dy, self.code[i], W, self.dact[i - 1], l.ksize, wprep=self.wprep[i])
                dy = self.dact[i - 1]
            else:  # conv1pool (first layer: no data gradient)
                hip.conv1_wgrad(dy, self.code[i], data, sidx, gW, gb, l.ksize, scratch=self.c1_scratch[i],
                                opt=self._opt_epilogue(i, step_index))

    def _backward_forked(self, data, sidx, step_index: int) -> None:
        """Backward pass + optimizer step with the weight-gradient chain on a second stream (see ``_bwd_fork``):

            launch stream :  head dgrad --ev--> conv2 dgrad -> conv1 wgrad (+ folded step) ............ join
            side stream   :  head wgrad ......--ev--> head optimizer step -> conv2 wgrad (+ folded step) --^

        The head's optimizer step waits for the head's data gradient (it reads the weights the step overwrites); the
        stage-2 data gradient reads the PREPARED weight copy of the forward pass, so the stage-2 step may run beside it."""
        hip, layers = self.hip, self.spec.layers
        main, side = torch.cuda.current_stream(), self._side
        last = len(layers) - 1
        l = layers[last]
        W, _ = self._w(self.params, l)
        gW, gb = self._gw(l)
        dy = self.dact[-1]
        xin = self.act[last - 1].flatten(2) if self.act[last - 1].dim() > 3 else self.act[last - 1]
        side.wait_stream(main)
        with torch.cuda.stream(side):
            hip.linear_wgrad(dy, xin, gW, gb)
        prev = layers[last - 1]
        mask = self.act[last - 1] if (prev.kind == "linear" and prev.relu) else None
        dx = self.dact[last - 1].flatten(2) if self.dact[last - 1].dim() > 3 else self.dact[last - 1]
        self._linear_dgrad(dy, W, mask, dx)
        ev = torch.cuda.Event()
        ev.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ev)
            self._optim_step(step_index)          # everything no gradient kernel updates itself: the head
        dy = self.dact[last - 1]
        for i in range(last - 1, -1, -1):
            l = layers[i]
            W, _ = self._w(self.params, l)
            gW, gb = self._gw(l)
            if l.kind == "conv2pool":
                with torch.cuda.stream(side):
                    hip.conv2_wgrad(dy, self.code[i], self.act[i - 1], gW, gb, l.ksize, scratch=self.w2_scratch[i],
                                    opt=self._opt_epilogue(i, step_index))
                hip.conv2_dgrad(dy, self.code[i], W, self.dact[i - 1], l.ksize, wprep=self.wprep[i])
                dy = self.dact[i - 1]
            elif l.kind == "conv1pool":
                hip.conv1_wgrad(dy, self.code[i], data, sidx, gW, gb, l.ksize, scratch=self.c1_scratch[i],
                                opt=self._opt_epilogue(i, step_index))
            else:
                raise RuntimeError("forked backward: dense layers below the head are not covered")
        main.wait_stream(side)

    def _fused_update(self, l, dy, xin, step_index: int, sidx=None, dx=None) -> bool:
        """Weight gradient of a big linear layer + its optimizer update in one kernel. With ``dx`` the kernel is
        also asked for the layer's data gradient (one pass over the weights less, fp32 policies + Adam); False: it
        declined and NOTHING ran - the caller runs ``linear_dgrad`` and calls again without ``dx``."""
        o = self.spec.optim
        Wb, Mb = self._w(self.params, l), self._w(self.m, l)
        # the reference builds a new optimizer every round: in its first step the moments are zero, so the
        # kernel neither reads them nor do they have to be cleared beforehand (see reset_optimizer)
        fresh = step_index == 1 and not self.persist_optimizer
        if dx is not None:
            if o.kind != "adam":
                return False
            Vb = self._w(self.v, l)
            return self.hip.linear_wgrad_adam_dgrad(
                dy, xin, Wb[0], Mb[0], Vb[0], Wb[1], Mb[1], Vb[1], dx, lr=o.lr, beta1=o.beta1, beta2=o.beta2, eps=o.eps,
                weight_decay=o.weight_decay, t=step_index,
                bc=self._bc_buf[step_index - 1] if self._bc_buf is not None else None, fresh=fresh, sidx=sidx)
        if o.kind == "adam":
            Vb = self._w(self.v, l)
            self.hip.linear_wgrad_adam(dy, xin, Wb[0], Mb[0], Vb[0], Wb[1], Mb[1], Vb[1], lr=o.lr, beta1=o.beta1,
                                       beta2=o.beta2, eps=o.eps, weight_decay=o.weight_decay, t=step_index,
                                       bc=self._bc_buf[step_index - 1] if self._bc_buf is not None else None,
                                       fresh=fresh, sidx=sidx)
        else:
            self.hip.linear_wgrad_sgdm(dy, xin, Wb[0], Mb[0], Wb[1], Mb[1], lr=o.lr, momentum=o.momentum,
                                       weight_decay=o.weight_decay, fresh=fresh, sidx=sidx)
        return True

    def _opt_epilogue(self, i: int, step_index: int):
        """Optimizer epilogue (ops/hip.py: fold_opt) of layer ``i`` for step ``step_index``; None: not folded."""
        if i not in self._fold:
            return None
        o, l = self.spec.optim, self.spec.layers[i]
        (W, b), (Mw, Mb) = self._w(self.params, l), self._w(self.m, l)
        Vw, Vb = self._w(self.v, l) if o.kind == "adam" else (None, None)
        return self.hip.fold_opt(o.kind, W, Mw, Vw, b, Mb, Vb, lr=o.lr, beta1=o.beta1, beta2=o.beta2, eps=o.eps,
                                 weight_decay=o.weight_decay, momentum=o.momentum, t=step_index,
                                 bc=self._bc_buf[step_index - 1] if self._bc_buf is not None else None)

    def _head_step(self, x, labels, sidx, n_valid, step_index: int) -> None:
        """Class head of a training step in one kernel (csrc/head.hip): forward, loss + counters, data gradient for
        the layer below, weight gradient and - folded - the optimizer step of the head."""
        i = self._head
        l, prev = self.spec.layers[i], self.spec.layers[i - 1]
        W, b = self._w(self.params, l)
        st = self.stats["Train"]
        opt = self._opt_epilogue(i, step_index)
        gW, gb = (None, None) if opt is not None else self._gw(l)
        dx = self.dact[i - 1].flatten(2) if self.dact[i - 1].dim() > 3 else self.dact[i - 1]
        self.hip.linear_head(x.flatten(2) if x.dim() > 3 else x, W, b, labels, sidx, st.loss_sum, st.correct, st.count,
                             st.confusion, n_valid, dx=dx, dw=gW, db=gb,
                             relu_mask=prev.kind == "linear" and prev.relu, opt=opt, scratch=self._head_scratch)
        self._record_step()

    def _optim_step(self, step_index: int) -> None:
        o = self.spec.optim
        if not self._adam_ranges:       # every layer's update is fused / folded into its gradient kernel
            return
        # (one launch over both ranges - hip.adam_step_ranges - is 2.8 ms per headline round SLOWER beside the other
        # group's update than these two: 256.4 vs 253.5 ms, measured twice alternating)
        if o.kind == "adam":
            for a, b in self._adam_ranges:   # everything the fused wgrad+Adam kernels do not own
                self.hip.adam_step(self.params[:, a:b], self._gcols(a, b), self.m[:, a:b], self.v[:, a:b],
                                   lr=o.lr, beta1=o.beta1, beta2=o.beta2, eps=o.eps, weight_decay=o.weight_decay,
                                   t=step_index,
                                   bc=self._bc_buf[step_index - 1] if self._bc_buf is not None else None)
        else:
            for a, b in self._adam_ranges:
                self.hip.sgdm_step(self.params[:, a:b], self._gcols(a, b), self.m[:, a:b], lr=o.lr,
                                   momentum=o.momentum, weight_decay=o.weight_decay)

    # ------------------------------------------------------------------ epochs
    def _nv(self, value: int) -> torch.Tensor:
        """Cached [K] int32 constant (static address: safe to bake into a captured graph)."""
        t = self._nv_cache.get(value)
        if t is None:
            t = torch.full((self.K,), value, dtype=torch.int32, device=self.device)
            self._nv_cache[value] = t
        return t

    def _tail_sidx(self, n: int, nb: int, B: int) -> torch.Tensor:
        """Index list of the last partial evaluation batch: nb real rows, then clamped padding rows
        (masked through n_valid) so every access stays inside the allocation."""
        key = (n, nb, B)
        t = self._tail_cache.get(key)
        if t is None:
            idx = torch.arange(n - nb, n - nb + B, device=self.device).clamp(max=n - 1).to(torch.int32)
            t = idx.unsqueeze(0).expand(self.K, -1).contiguous()
            self._tail_cache[key] = t
        return t

    def _eval_pass(self, x, y, phase: str) -> None:
        n = x.shape[1]
        B = self._eval_batch(n)
        bufs = self._bufs(B)
        for s in range(0, n, B):
            nb = min(B, n - s)
            if nb == B:
                self._forward(x[:, s:], None, y[:, s:], self._nv(B), phase, train=False, bufs=bufs, prep=s == 0)
            else:
                self._forward(x, self._tail_sidx(n, nb, B), y, self._nv(nb), phase, train=False, bufs=bufs,
                              prep=s == 0)

    def _train_body(self, epochs: int, steps: int, n: int) -> None:
        """Everything a local-training round launches (captured into one hipGraph)."""
        nv_last = self._nv(n - (steps - 1) * self.B)
        for ph in ("Train", "Validation"):
            self.stats[ph].zero_()
        t = 0
        for e in range(epochs):
            if e == epochs - 1 and epochs > 1:
                for ph in ("Train", "Validation"):   # report the last epoch, like the per-epoch logs
                    self.stats[ph].zero_()
            for s in range(steps):
                sidx = self._perm_buf[e, :, s * self.B:(s + 1) * self.B]
                nv = nv_last if s == steps - 1 else self._nv(self.B)
                self._forward(self.train_x, sidx, self.train_y, nv, "Train", train=True)
                t += 1
                self._backward(self.train_x, sidx, t)
                if not self._bwd_fork:             # (the forked backward runs the step on its side stream)
                    self._optim_step(t)
            if self.val_x is not None:
                self._eval_pass(self.val_x, self.val_y, "Validation")

    # ------------------------------------------------------------------ two-stream pipeline
    def _conv_forward(self, data, sidx, out_last, coresident: bool = False) -> None:
        """Convolution stages of one batch; the last stage writes ``out_last``."""
        hip, x = self.hip, None
        for i in range(self.nc):
            l = self.spec.layers[i]
            W, b = self._w(self.params, l)
            out = out_last if i == self.nc - 1 else self.act[i]
            if l.kind == "conv1pool":
                hip.conv1_fwd(data, sidx, W, b, out, self.code[i], l.ksize)
            else:
                hip.conv2_prep(W, self.wprep[i], l.ksize)
                hip.conv2_fwd(x, W, b, out, self.code[i], l.ksize, wprep=self.wprep[i], coresident=coresident)
            x = out

    def _conv_backward(self, data, sidx, coresident: bool = False, step_index: int = 1) -> None:
        """Backward pass of the convolution stages. Under the fused update (``coresident``) the kernels that fit on a
        CU beside its two workgroups go first - stage-2 data gradient, stage-1 weight gradient - and the stage-2
        weight gradient (512 threads, 104 KB of LDS in the exact-split policy: it waits for whole free CUs) last, so
        the chain behind it no longer waits the update out (10 clients: see docs/kernels.rst)."""
        hip = self.hip
        deferred = []
        late_wgrad = coresident and not os.environ.get("P2PFL_WGRAD_FIRST")
        for i in range(self.nc - 1, -1, -1):
            l = self.spec.layers[i]
            W, _ = self._w(self.params, l)
            gW, gb = self._gw(l)
            if l.kind == "conv2pool":
                def wgrad(i=i, l=l, gW=gW, gb=gb):
                    hip.conv2_wgrad(self.dact[i], self.code[i], self.act[i - 1], gW, gb, l.ksize,
                                    scratch=self.w2_scratch[i], coresident=coresident,
                                    opt=self._opt_epilogue(i, step_index))
                if i == self.nc - 1 and getattr(self, "_wgrad_ev", None) is not None:
                    ev = self._wgrad_ev                       # ran on the launch stream (see _dense_step)
                    deferred.append(lambda ev=ev: torch.cuda.current_stream().wait_event(ev))
                elif late_wgrad:
                    deferred.append(wgrad)
                else:
                    wgrad()
                hip.conv2_dgrad(self.dact[i], self.code[i], W, self.dact[i - 1], l.ksize, wprep=self.wprep[i],
                                coresident=coresident)
            else:
                hip.conv1_wgrad(self.dact[i], self.code[i], data, sidx, gW, gb, l.ksize, scratch=self.c1_scratch[i],
                                opt=self._opt_epilogue(i, step_index))
        for fn in deferred:
            fn()

    def _dense_step(self, conv_out, labels, sidx, n_valid, step_index: int, side) -> None:
        """Dense layers of one training step on the current stream: forward, loss, backward. ``side`` is
        released (it may start the convolution backward pass) as soon as the data gradient of the first
        dense layer exists, i.e. BEFORE that layer's long fused wgrad+optimizer kernel is enqueued.
        """
        hip, layers, nc, o = self.hip, self.spec.layers, self.nc, self.spec.optim
        main = torch.cuda.current_stream()
        x = conv_out.flatten(2)
        head = self._head if (self._head is not None and self._head > nc) else None
        for i in range(nc, len(layers)):
            if i == head:
                break
            W, b = self._w(self.params, layers[i])
            self._linear_fwd(x, W, b, self.act[i], layers[i].relu)
            x = self.act[i]
        if head is not None:
            self._head_step(x, labels, sidx, n_valid, step_index)
        else:
            st = self.stats["Train"]
            hip.softmax_ce(x, labels, sidx, self.dact[-1], st.loss_sum, st.correct, st.count, st.confusion, n_valid)
            self._record_step()
        for i in range(len(layers) - 1 if head is None else head - 1, nc - 1, -1):
            l = layers[i]
            W, _ = self._w(self.params, l)
            xin = conv_out.flatten(2) if i == nc else self.act[i - 1]
            prev = layers[i - 1]
            mask = self.act[i - 1] if (prev.kind == "linear" and prev.relu) else None
            took = False
            if i in self._fused and mask is None and self._fuse_dgrad:
                # the update kernel hands out the data gradient itself (fp32: the convolutions behind it cannot
                # hide under the update anyway, see docs/kernels.rst), so the side stream is released after it
                took = self._fused_update(l, self.dact[i], xin, step_index, dx=self.dact[i - 1].flatten(2))
            if not took:
                self._linear_dgrad(self.dact[i], W, mask, self.dact[i - 1].flatten(2))
            if i == nc:
                side.wait_stream(main)
                self._wgrad_ev = None
                if not took and self._wgrad_on_main:
                    # exact-split stage-2 weight gradient: a 512-thread / 104 KB block needs a CU to itself and would
                    # wait the whole update out on the side stream (and the rest of the chain behind it)
                    j = nc - 1
                    lj = layers[j]
                    gWj, gbj = self._gw(lj)
                    hip.conv2_wgrad(self.dact[j], self.code[j], self.act[j - 1], gWj, gbj, lj.ksize,
                                    scratch=self.w2_scratch[j], opt=self._opt_epilogue(j, step_index))
                    self._wgrad_ev = torch.cuda.Event()
                    self._wgrad_ev.record(main)
            if took:
                pass
            elif i in self._fused:
                self._fused_update(l, self.dact[i], xin, step_index)
            else:
                gW, gb = self._gw(l)
                hip.linear_wgrad(self.dact[i], xin, gW, gb)

    def _train_body_pipelined(self, epochs: int, steps: int, n: int) -> None:
        """Same work as :meth:`_train_body`, scheduled on two HIP streams (graph branches when captured):

            main : [dense fwd + loss + dense bwd ....... fused wgrad+Adam of the big layer (HBM-bound)] ...
            side :              [conv bwd | Adam of the small layers | conv fwd of the NEXT batch]

        The convolution kernels are MFMA/LDS-bound and short, the fused update streams weights and both
        Adam moments of every client through HBM for milliseconds with idle matrix cores, so the two
        overlap. The output of the last convolution stage is double-buffered because the fused update of
        step t still reads it while the side stream already produces the one of step t+1.
        """
        main, side = torch.cuda.current_stream(), self._side
        dbg = self._stream_debug
        if dbg is not None:
            dbg.clear()

        def mark(stream, name):          # P2PFL_STREAM_DEBUG=1: named events, so a stall report can say which wait hangs
            if dbg is not None and not self._capturing:
                ev = torch.cuda.Event()
                ev.record(stream)
                dbg.append((name, ev))
        nv_last = self._nv(n - (steps - 1) * self.B)
        for ph in ("Train", "Validation"):
            self.stats[ph].zero_()
        t = 0
        for e in range(epochs):
            if e == epochs - 1 and epochs > 1:
                for ph in ("Train", "Validation"):
                    self.stats[ph].zero_()
            self._conv_forward(self.train_x, self._perm_buf[e, :, :self.B], self._act_pp[t & 1])
            for s in range(steps):
                sidx = self._perm_buf[e, :, s * self.B:(s + 1) * self.B]
                nv = nv_last if s == steps - 1 else self._nv(self.B)
                t += 1
                self._dense_step(self._act_pp[(t - 1) & 1], self.train_y, sidx, nv, t, side)
                mark(main, f"step {t}: main after fused update")
                with torch.cuda.stream(side):     # under the fused update: kernel variants that fit beside it on a CU
                    self._conv_backward(self.train_x, sidx, coresident=True, step_index=t)
                    self._optim_step(t)
                    if s + 1 < steps:
                        self._conv_forward(self.train_x, self._perm_buf[e, :, (s + 1) * self.B:(s + 2) * self.B],
                                           self._act_pp[t & 1], coresident=True)
                mark(side, f"step {t}: side after conv bwd/fwd")
                main.wait_stream(side)
                mark(main, f"step {t}: main after join")
            if self.val_x is not None:
                self._eval_pass(self.val_x, self.val_y, "Validation")

    # ------------------------------------------------------------------ client groups beside each other
    def _group_views(self, G: int):
        """``G`` shallow copies of this engine that each see one contiguous 1/G of the clients (every per-client
        buffer sliced along the client dimension), so the per-step methods can be run per group."""
        import copy
        keep = self.__dict__.setdefault("_group_keep", {})
        bounds = [(g * self.K) // G for g in range(G + 1)]
        views = []
        for g in range(G):
            sl = slice(bounds[g], bounds[g + 1])
            v = copy.copy(self)
            v.K = sl.stop - sl.start
            v._rows = sl
            for name in ("params", "grads", "m", "v", "karange", "train_x", "train_y"):
                t = getattr(self, name, None)
                if t is not None:
                    setattr(v, name, t[sl])
            v.act = [a[sl] for a in self.act]
            v.dact = [a[sl] for a in self.dact]
            v.code = [None if c is None else c[sl] for c in self.code]
            v._act_pp = [a[sl] for a in self._act_pp]
            v.wprep = {i: w[sl] for i, w in self.wprep.items()}
            v.stats = {ph: PhaseStats(st.loss_sum[sl], st.correct[sl], st.count[sl], st.confusion[sl])
                       for ph, st in self.stats.items()}
            v._step_log = None if self._step_log is None else self._step_log[:, :, sl]
            v._step_cm = None if self._step_cm is None else self._step_cm[:, sl]
            # buffers with static addresses of their own (baked into the captured graph): kept on the engine
            own = keep.setdefault((G, g), {"nv": {}, "fwd": {}, "w2": None, "c1": None})
            if own["w2"] is None:
                own["w2"] = {i: (self.hip.conv2_wgrad_scratch_opt if i in self._fold else self.hip.conv2_wgrad_scratch)(
                                 v.K, self.B, l.ksize, self.device)
                             for i, l in enumerate(self.spec.layers) if l.kind == "conv2pool"}
                own["c1"] = {i: self.hip.conv1_wgrad_scratch(v.K, self.B, l.ksize, self.device)
                             for i, l in enumerate(self.spec.layers) if l.kind == "conv1pool"}
            v._nv_cache, v._fwd_scratch, v.w2_scratch, v.c1_scratch = own["nv"], own["fwd"], own["w2"], own["c1"]
            v._wgrad_ev = None
            views.append(v)
        return views

    def _train_body_groups(self, epochs: int, steps: int, n: int, G: int) -> None:
        """The clients as ``G`` independent groups whose steps alternate on the two streams:

            main : [A dense fwd + loss + fused update/dgrad][B dense ...........................][A ...
            side :                                          [A conv bwd, small Adam, conv fwd   ][B conv ...

        No client depends on another during local training, so while the HBM-bound dense chain of one group streams
        its weights, the matrix-pipe-bound convolution chain of the other group has the matrix cores - an overlap the
        update and the convolutions of the SAME clients cannot have once the update hands out the data gradient itself
        (tools/probe/halves_probe.py: 4.93 -> 4.62 ms per step for 100 MNIST-CNN clients; CU-masked streams lose).
        """
        main, side = torch.cuda.current_stream(), self._side
        B = self.B
        views = self._group_views(G)
        last_valid = n - (steps - 1) * B
        for ph in ("Train", "Validation"):
            self.stats[ph].zero_()
        # small-footprint convolution variants inside the groups: the fp32-MFMA kernels (290.4 vs 291.2 ms); not the
        # exact-split ones. P2PFL_GROUPS_CORES=0/1 overrides.
        env = os.environ.get("P2PFL_GROUPS_CORES")
        cores = self.precision == "fp32-mfma" if env is None else env not in ("", "0")
        done = [None] * G                    # side stream finished the convolution chain (next forward included) of group g
        t = 0
        for e in range(epochs):
            if e == epochs - 1 and epochs > 1:
                for ph in ("Train", "Validation"):
                    self.stats[ph].zero_()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for g, v in enumerate(views):
                    v._conv_forward(v.train_x, self._perm_buf[e, v._rows, :B], v._act_pp[t & 1])
                    done[g] = torch.cuda.Event()
                    done[g].record(side)
            for s in range(steps):
                t += 1
                for g, v in enumerate(views):
                    sidx = self._perm_buf[e, v._rows, s * B:(s + 1) * B]
                    nv = v._nv(last_valid if s == steps - 1 else B)
                    main.wait_event(done[g])
                    v._dense_step(v._act_pp[(t - 1) & 1], v.train_y, sidx, nv, t, side)     # releases side after the update
                    with torch.cuda.stream(side):
                        v._conv_backward(v.train_x, sidx, coresident=cores, step_index=t)
                        v._optim_step(t)
                        if s + 1 < steps:
                            v._conv_forward(v.train_x, self._perm_buf[e, v._rows, (s + 1) * B:(s + 2) * B], v._act_pp[t & 1],
                                            coresident=cores)
                        done[g] = torch.cuda.Event()
                        done[g].record(side)
            main.wait_stream(side)
            if self.val_x is not None:
                self._eval_pass(self.val_x, self.val_y, "Validation")
        self._step_pos = views[0]._step_pos

    def _groups(self) -> int:
        """Client groups of the two-stream schedule: 2 when the SMALLEST group still takes the fused update + data
        gradient - asked of the kernel's own chip-filling rule (``p2pfl_linear_wgrad_adam_dgrad_accepts``: MNIST-CNN l1
        on 256 CUs = 31 clients per group; measured from 2 x 32 clients up, tools/probe/groups_k.sh), so another layer
        width or CU count moves the threshold with it. ``P2PFL_GROUPS`` overrides (1: one group, the plain pipeline); a
        forced split whose groups fall back to data gradient + separate update is logged once."""
        first_dense = self.spec.layers[self.nc] if self.nc < len(self.spec.layers) else None

        def accepted(G: int) -> bool:
            return (first_dense is not None and first_dense.kind == "linear" and self.K // G >= 1
                    and self.hip.linear_wgrad_adam_dgrad_accepts(self.K // G, self.B, first_dense.in_features,
                                                                 first_dense.out_features))
        env = os.environ.get("P2PFL_GROUPS")
        if env is not None:
            G = max(1, min(int(env), self.K))
            if G > 1 and self._fuse_dgrad and not accepted(G) and not getattr(self, "_groups_warned", False):
                self._groups_warned = True
                logging.warning("[HipEngine] P2PFL_GROUPS=%d: groups of %d clients are below what the fused update + data "
                                "gradient takes; every group runs linear_dgrad + the separate update (measured slower)",
                                G, self.K // G)
            return G
        return 2 if (self._fuse_dgrad and self.spec.optim.kind == "adam" and accepted(2)) else 1

    def _train(self, epochs: int, steps: int, n: int) -> None:
        self._step_pos = 0
        if self.overlap and not os.environ.get("P2PFL_INORDER") and self._groups() > 1:
            self._train_body_groups(epochs, steps, n, self._groups())
            return
        # The two-stream schedule runs eagerly (warm-up, use_graph=False) exactly as it is captured: the stall that
        # once forced eager launches onto the in-order schedule came from the side stream's priority (see __init__).
        if self.overlap and not os.environ.get("P2PFL_INORDER"):
            self._train_body_pipelined(epochs, steps, n)
        else:
            self._train_body(epochs, steps, n)

    def fit(self, epochs, active=None):
        if not self.persist_optimizer:
            self.reset_optimizer()
        n = self.train_x.shape[1]
        steps = self.steps_per_epoch()
        total = epochs * steps
        perms = self._epoch_perms(epochs)
        if self._perm_buf is None or self._perm_buf.shape != perms.shape:
            self._perm_buf = torch.zeros(perms.shape, dtype=torch.int32, device=self.device)
            self._graphs.clear()
        self._perm_buf.copy_(perms, non_blocking=True)
        # Adam bias corrections for steps t0+1 .. t0+total live in a device table, so one captured
        # graph serves both the reference policy (fresh optimizer per round) and persistent moments.
        o = self.spec.optim
        if o.kind == "adam":
            ts = torch.arange(self.opt_steps + 1, self.opt_steps + total + 1, dtype=torch.float64)
            bc = torch.stack([1.0 / (1.0 - o.beta1 ** ts), 1.0 / torch.sqrt(1.0 - o.beta2 ** ts)], dim=1).float()
            if self._bc_buf is None or self._bc_buf.shape != bc.shape:
                self._bc_buf = torch.zeros(bc.shape, device=self.device)
                self._graphs.clear()
            self._bc_buf.copy_(bc, non_blocking=True)
        if self._alloc_step_tables(total):
            self._graphs.clear()
        saved = None
        if active is not None and not bool(torch.as_tensor(active).all()):
            inactive = (~torch.as_tensor(active, dtype=torch.bool)).nonzero().flatten().to(self.device)
            saved = [inactive, self.params[inactive].clone()]
            if self.persist_optimizer:       # a node that sits the round out keeps its optimizer state as well
                saved += [self.m[inactive].clone(), None if self.v is None else self.v[inactive].clone()]
        if self.use_graph:
            key = (epochs, steps, n, self.persist_optimizer)
            if key not in self._graphs:
                # warm-up on a side stream (lazy module loads, allocator), then capture
                # (a fresh stream per capture: with one long-lived warm-up stream the parallel branch of the
                # captured graph ended up on the hardware queue of the launch stream - headline 237 -> 259 ms)
                # (the snapshot BEFORE the warm-up stream picks up its dependency on this one: taken after
                # `s.wait_stream`, the clones raced the warm-up - the class head, whose update is the first thing a step
                # writes, saw second moments that already contained the warm-up's gradients)
                snap = None if self._in_prepare else self._snapshot_state()   # (prepare() restores its own snapshot)
                s = torch.cuda.Stream(device=self.device)
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    self._train(1, min(steps, 2), n)
                torch.cuda.current_stream().wait_stream(s)
                torch.cuda.synchronize()
                self._restore_state(snap)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):   # RCCL's watchdog thread may query events meanwhile
                    self._capturing = True
                    try:
                        self._train(epochs, steps, n)
                    finally:
                        self._capturing = False
                self._graphs[key] = (g,)
                self._restore_state(snap)
            self._graphs[key][0].replay()
        else:
            self._train(epochs, steps, n)
        self.opt_steps += total
        if saved is not None:
            self.params[saved[0]] = saved[1]
            if len(saved) > 2:
                self.m[saved[0]] = saved[2]
                if saved[3] is not None:
                    self.v[saved[0]] = saved[3]
        out = {"Train": self.stats["Train"].summary()}
        if self.val_x is not None:
            out["Validation"] = self.stats["Validation"].summary()
        return out

    def prepare(self, epochs: int) -> None:
        """Build everything a round needs ahead of time - lazy kernel loading, buffers, the captured training
        and evaluation graphs - by running one round and putting parameters, optimizer state, step and shuffle
        counters back, so the first real round costs what every other round costs."""
        snap, counters = self._snapshot_state(), (self.opt_steps, self._perm_calls)
        self._in_prepare = True
        try:
            self.evaluate()
            self.fit(epochs)
        finally:
            self._in_prepare = False
        self.synchronize()
        self._restore_state(snap)
        self.opt_steps, self._perm_calls = counters
        for st in self.stats.values():
            st.zero_()

    def _snapshot_state(self):
        """Copies of parameters and optimizer state (warm-up steps and the prepare() round are undone with them): on the
        device while a second copy of the state fits beside what a round allocates, else in host memory - a shard sized
        for the 288 GB of the card (2000 MNIST-CNN nodes: 145 GB of p / m / v) must not need its state twice."""
        state = [t for t in (self.params, self.m, self.v) if t is not None]
        need = sum(t.numel() * t.element_size() for t in state)
        free = torch.cuda.mem_get_info(self.device)[0]
        cached = torch.cuda.memory_reserved(self.device) - torch.cuda.memory_allocated(self.device)
        where = self.device if 2 * need < free + cached else torch.device("cpu")
        return tuple(None if t is None else t.to(where, copy=True) for t in (self.params, self.m, self.v))

    def _restore_state(self, snap) -> None:
        if snap is None:
            return
        for dst, src in zip((self.params, self.m, self.v), snap):
            if dst is not None:
                dst.copy_(src)

    def _eval_body(self) -> None:
        self.stats["Test"].zero_()
        self._eval_pass(self.test_x, self.test_y, "Test")

    def evaluate(self):
        if self.use_graph:
            if self._eval_graph is None:
                s = torch.cuda.Stream(device=self.device)
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    self._eval_body()
                torch.cuda.current_stream().wait_stream(s)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):   # RCCL's watchdog thread may query events meanwhile
                    self._eval_body()
                self._eval_graph = g
            self._eval_graph.replay()
        else:
            self._eval_body()
        return self.stats["Test"].summary()


def make_engine(spec: ModelSpec, n_clients: int, device, impl: str = "auto", **kw) -> ShardEngine:
    """``impl``: "hip" | "eager" | "auto" (hip on a GPU, eager on the CPU)."""
    device = torch.device(device)
    if not isinstance(spec, ModelSpec):     # a ModuleSpec: model family without kernel-level spec (stock ops)
        from .module_engine import ModuleEngine
        kw.pop("use_graph", None)
        kw.pop("overlap", None)
        kw.pop("precision", None)
        return ModuleEngine(spec, n_clients, device, **kw)
    if impl == "auto":
        impl = "hip" if device.type == "cuda" else "eager"
    if impl == "hip":
        return HipEngine(spec, n_clients, device, **kw)
    kw.pop("use_graph", None)
    kw.pop("overlap", None)
    kw.pop("precision", None)      # the eager engine is fp32 by construction
    return EagerEngine(spec, n_clients, device, **kw)
