"""Batched engine for ARBITRARY model families (BatchNorm / dropout / depthwise / residual / auto-encoders / SVMs).

The spec models (MNIST MLP / CNN, FEMNIST CNN) train through the hand-written client-batched HIP kernels
(:class:`HipEngine`). Every other family of the reference's factory (node_start.py:48-85: CIFAR-10 ResNets and
MobileNets, SYSCALL MLP / auto-encoder / SVM, WADI MLP) keeps its ``nn.Module`` and trains with stock PyTorch-ROCm
ops (MIOpen / hipBLASLt) - but inside the same one-process-per-GPU runtime: all floating-point state of all
co-resident nodes (parameters AND buffers such as BatchNorm running statistics) aliases rows of one ``[K, P]`` slab,
so model exchange stays "pass a row index" / one RCCL all-to-all of rows, and FedAvg stays the grouped row-reduce
kernel (``csrc/fedavg.hip``), exactly as for the spec models. Integer buffers (``num_batches_tracked``) stay
node-local.

Client-batched execution: the slab IS a stack of the K models (column range of an entry = ``[K, *shape]``), so one
``torch.func.vmap`` over ``functional_call`` runs a training step of ALL co-resident nodes at once - a batched GEMM per
linear layer, a grouped convolution per conv layer, BatchNorm with per-node statistics - instead of K Python-looped
steps of launch-bound kernels, and ONE optimizer over the stacked tensors replaces K (Adam / SGD are element-wise, so
it is the same arithmetic per node). Families whose step vmap cannot express fall back to the per-node loop.

Reference behaviour kept: a new optimizer per round (lightninglearner.py:148-160), per-epoch validation, evaluation
before training, loss / confusion-matrix metrics for classifiers, loss only for the others.
"""
from __future__ import annotations

import logging
import os
from collections import OrderedDict
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

from .engine import PHASES, PhaseStats, ShardEngine
from .slab import ALIGN, ParamLayout, ParamSlot


@dataclass
class ModuleSpec:
    """What :class:`BatchedFederation` needs to know about a model family that has no kernel-level spec."""
    name: str
    factory: Callable[[], torch.nn.Module]      # () -> FederatedModule (reference model class, default arguments)
    num_classes: int
    input_shape: tuple                           # per-sample shape the module's forward expects
    augment: Optional[Callable] = None           # training-batch augmentation of the data module (CIFAR-10 crop / flip)


class StateLayout(ParamLayout):
    """Slab columns for the floating-point entries of a ``state_dict`` (parameters and buffers, in order)."""

    def __init__(self, sd):
        self.spec = None
        self.keys: List[str] = list(sd.keys())
        self.int_keys: List[str] = [k for k, t in sd.items() if not torch.is_floating_point(t)]
        self.int_shapes = {k: tuple(sd[k].shape) for k in self.int_keys}
        slots, off = [], 0
        for name, t in sd.items():
            if torch.is_floating_point(t):
                slots.append(ParamSlot(name, tuple(t.shape), off, t.numel()))
                off += -(-t.numel() // ALIGN) * ALIGN
        self.slots = slots
        self.by_name = {s.name: s for s in slots}
        self.row_stride = max(off, ALIGN)
        self.num_params = sum(s.numel for s in slots)

    def full_arrays(self, row: torch.Tensor, ints: Dict[str, torch.Tensor]) -> List[np.ndarray]:
        """Payload of one node: every state_dict entry in order (reference lightninglearner.py:73-89)."""
        host = row.detach().to("cpu")
        out = []
        for k in self.keys:
            if k in self.by_name:
                s = self.by_name[k]
                out.append(host[s.offset:s.offset + s.numel].reshape(s.shape).numpy().copy())
            else:
                out.append(ints[k].detach().cpu().numpy().copy())
        return out

    def arrays_to_row(self, arrays: Sequence, row: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Accepts the full payload (every entry) or only the floating-point ones; returns the integer entries."""
        if len(arrays) == len(self.keys):
            named = dict(zip(self.keys, arrays))
            super().arrays_to_row([named[s.name] for s in self.slots], row)
            return {k: torch.as_tensor(np.asarray(named[k])) for k in self.int_keys}
        super().arrays_to_row(arrays, row)
        return {}


class _LossCall(torch.nn.Module):
    """``forward`` = the wrapped module's ``training_loss`` (``functional_call`` only calls ``forward``)."""

    def __init__(self, m: torch.nn.Module):
        super().__init__()
        self.m = m

    def forward(self, x, y):
        loss, pred = self.m.training_loss(x, y)
        if pred is None:
            pred = torch.zeros(x.shape[0], dtype=torch.long, device=x.device)
        return loss, pred


class ModuleEngine(ShardEngine):
    """K independent ``nn.Module`` replicas on one device whose float state aliases the slab rows.

    ``batched``: train / evaluate all K replicas per step through ``torch.func.vmap`` (default; None = unless
    ``P2PFL_MODULE_VMAP=0``). The per-node loop remains as the fallback and as the oracle of the batched path."""

    def __init__(self, spec: ModuleSpec, n_clients: int, device="cpu", batch_size: int = 32, seed: int = 0,
                 persist_optimizer: bool = False, client_offset: int = 0, batched: Optional[bool] = None,
                 log_steps: bool = False):
        self.spec = spec
        self.K = int(n_clients)
        self.device = torch.device(device)
        self.B = int(batch_size)
        self.seed = seed
        self.persist_optimizer = persist_optimizer
        self.client_offset = int(client_offset)
        self._perm_calls = 0
        self.train_x = self.train_y = self.val_x = self.val_y = self.test_x = self.test_y = None
        self.opt_steps = 0
        self.params_alt = None
        torch.manual_seed(seed)
        self.modules: List[torch.nn.Module] = [spec.factory().to(self.device) for _ in range(self.K)]
        self.layout = StateLayout(self.modules[0].state_dict())
        self.params = self.layout.alloc(self.K, self.device)
        self.classifier = bool(getattr(self.modules[0], "is_classifier", True))
        self.C = max(int(spec.num_classes), 2)
        self.stats = {ph: PhaseStats.alloc(self.K, self.C, self.device) for ph in PHASES}
        # per-step Train/* series (ShardEngine._init_step_log). The batched pass of ONE epoch is what gets captured and
        # replayed, so its steps write epoch-local tables (same addresses every epoch) that are copied into the tables
        # of the whole fit after each epoch.
        self._init_step_log(log_steps)
        self._ep_log = self._ep_cm = None
        self._ep_row: Dict[int, int] = {}
        for k, m in enumerate(self.modules):
            tensors = dict(m.named_parameters())
            tensors.update(dict(m.named_buffers()))
            for s in self.layout.slots:
                t = tensors[s.name]
                view = self.params[k, s.offset:s.offset + s.numel].view(s.shape)
                view.copy_(t.detach())
                t.data = view                                    # parameter / buffer now lives in the slab row
        self.optims = None
        if batched is None:
            batched = os.environ.get("P2PFL_MODULE_VMAP", "1") != "0"
        self.batched = bool(batched) and self.K > 1
        self._stack = None            # name -> [K, *shape] view of the slab (floats) / stacked integer buffers
        self._stack_opt = None
        # hipGraph replay of a whole batched pass (an epoch of steps incl. backward and optimizer, or an evaluation
        # pass): small families are launch-bound even client-batched (50 SYSCALL MLPs: ~50 kernels of microseconds per
        # step). A pass runs eagerly the first time (lazy initialisation, MIOpen search), is captured at its second
        # occurrence and replayed from then on. P2PFL_MODULE_GRAPH=0 keeps the passes eager.
        # A pass is captured as SEGMENTS of at most ``graph_steps`` steps: hipStreamEndCapture of ROCm 7.2 dies with a
        # segmentation fault on very long captures - bisected with tools/probe/capture_augment.py: every piece of the
        # CIFAR-10 crop / flip augmentation (pad, device RNG, index arithmetic, the 4-index gather) captures fine, and
        # so do 8- and 64-step SimpleMobileNet passes WITH the hook; the 390-step pass of a 4-node CIFAR-10 split
        # (~10^5 kernel nodes in one graph) is what faults, with or without augmentation.
        self.use_graph = self.device.type == "cuda" and os.environ.get("P2PFL_MODULE_GRAPH", "1") != "0"
        self.graph_steps = max(1, int(os.environ.get("P2PFL_MODULE_GRAPH_STEPS", "32")))
        self._graphs: Dict[tuple, object] = {}
        self._seen: set = set()
        self._gstate = None
        self._perm_static = None
        if self.batched:
            try:
                self._build_stack()
            except Exception as exc:                      # noqa: BLE001 - any family vmap cannot take: per-node loop
                logging.warning("[ModuleEngine] %s: client-batched execution unavailable (%s); per-node loop", spec.name, exc)
                self.batched = False

    # ------------------------------------------------------------------ parameters
    def init_params(self, seed: Optional[int] = None, identical: bool = True) -> None:
        """Modules were initialised by their constructors; ``identical``: every node starts from node 0's
        weights (the reference's init-model diffusion, node.py:298-299,372-379)."""
        if identical and self.K > 1:
            self.params[1:] = self.params[0:1]

    def _ints(self, k: int) -> Dict[str, torch.Tensor]:
        sd = self.modules[k].state_dict()
        return {name: sd[name] for name in self.layout.int_keys}

    def get_state_dict(self, k: int):
        return OrderedDict((n, t.detach().clone()) for n, t in self.modules[k].state_dict().items())

    def set_state_dict(self, k: int, sd) -> None:
        self.set_arrays(k, [sd[n].detach().cpu().numpy() for n in self.layout.keys])

    def get_arrays(self, k: int) -> List[np.ndarray]:
        return self.layout.full_arrays(self.params[k], self._ints(k))

    def set_arrays(self, k: int, arrays) -> None:
        ints = self.layout.arrays_to_row(arrays, self.params[k])
        own = self.modules[k].state_dict()
        for name, v in ints.items():
            own[name].copy_(v.to(own[name].device).reshape(own[name].shape))

    # ------------------------------------------------------------------ data
    def set_data(self, train_x, train_y, val_x=None, val_y=None, test_x=None, test_y=None) -> None:
        """``*_x`` [K, n, ...] (reshaped to the module's input shape), ``*_y`` [K, n] (dtype kept)."""
        def put(x, y):
            if x is None:
                return None, None
            x = x.reshape(self.K, x.shape[1], *self.spec.input_shape).to(self.device, torch.float32).contiguous()
            return x, y.to(self.device).contiguous()
        self.train_x, self.train_y = put(train_x, train_y)
        self.val_x, self.val_y = put(val_x, val_y)
        self.test_x, self.test_y = put(test_x, test_y)
        if getattr(self, "_graphs", None):
            torch.cuda.synchronize(self.device)
            self._graphs.clear()
        if hasattr(self, "_seen"):
            self._seen.clear()

    # ------------------------------------------------------------------ client-batched execution (torch.func)
    def _build_stack(self) -> None:
        m0 = self.modules[0]
        pnames = {n for n, _ in m0.named_parameters()}
        stack: Dict[str, torch.Tensor] = {}
        for s_ in self.layout.slots:
            t = self.params[:, s_.offset:s_.offset + s_.numel].view(self.K, *s_.shape)
            if s_.name in pnames:
                t.requires_grad_(True)                   # a leaf that IS the slab's column range
            stack["m." + s_.name] = t
        # integer buffers (BatchNorm's num_batches_tracked) are not in the slab: one stacked tensor per entry whose
        # rows the replicas' own buffers alias from now on
        for name in self.layout.int_keys:
            own = [dict(m.named_buffers())[name] for m in self.modules]
            st = torch.stack([b.detach() for b in own])
            for k, b in enumerate(own):
                b.data = st[k]
            stack["m." + name] = st
        self._stack = stack
        self._param_keys = ["m." + n for n, _ in m0.named_parameters()]
        # floating-point buffers (BatchNorm running statistics): vmap's batch-norm rule drops the in-place update of a
        # NON-contiguous batched buffer (a slab column range has the row stride of the slab), so a pass works on
        # contiguous copies and writes them back (see _state / _writeback)
        self._fbuf_keys = ["m." + s_.name for s_ in self.layout.slots if s_.name not in pnames]
        self._call = _LossCall(m0)
        from torch.func import functional_call, vmap

        def one(state, x, y):
            return functional_call(self._call, state, (x, y))
        self._vstep = vmap(one, in_dims=(0, 0, 0), randomness="different")

    def _state(self) -> Dict[str, torch.Tensor]:
        st = dict(self._stack)
        for k in self._fbuf_keys:
            st[k] = self._stack[k].detach().contiguous().clone()
        return st

    def _writeback(self, st: Dict[str, torch.Tensor]) -> None:
        with torch.no_grad():
            for k in self._fbuf_keys:
                self._stack[k].copy_(st[k])

    def _stacked_optimizer(self):
        """The replicas' optimizer, built once over the stacked tensors (same class, same groups and hyper-parameters
        as ``configure_optimizers`` of one replica)."""
        m0 = self.modules[0]
        opt0 = m0.configure_optimizers()
        if not isinstance(opt0, torch.optim.Optimizer):
            raise TypeError("configure_optimizers() did not return a plain optimizer")
        name_of = {id(p): "m." + n for n, p in m0.named_parameters()}
        groups = []
        for g in opt0.param_groups:
            ng = {k: v for k, v in g.items() if k != "params"}
            if "capturable" in ng and self.use_graph:
                ng["capturable"] = True                     # step counter on the device: the step can live in a graph
            ng["params"] = [self._stack[name_of[id(p)]] for p in g["params"]]
            groups.append(ng)
        return type(opt0)(groups)

    def _fresh_optimizer(self):
        """A new optimizer per round (reference lightninglearner.py:148-160). With graph replay the optimizer OBJECT
        has to stay (the captured steps hold its state tensors): its state - moments, momentum buffers, the device
        step counter - is zeroed in place instead, which is what a new Adam / SGD(dampening 0) starts from."""
        if self._stack_opt is not None and self.use_graph:
            with torch.no_grad():
                for st in self._stack_opt.state.values():
                    for v in st.values():
                        if torch.is_tensor(v):
                            v.zero_()
            return self._stack_opt
        return self._stacked_optimizer()

    def _metrics(self, phase: str, losses, preds, yb, nb: int) -> None:
        st, C, K = self.stats[phase], self.C, self.K
        st.loss_sum += losses.detach().float() * nb
        st.count += nb
        if self.classifier:
            pred = preds.long().clamp(0, C - 1)
            yc = yb.long().clamp(0, C - 1)
            st.correct += (pred == yc).sum(dim=1).to(torch.int32)
            off = torch.arange(K, device=self.device).view(K, 1) * (C * C)
            st.confusion.view(-1).index_add_(0, (off + yc * C + pred).reshape(-1),
                                             torch.ones(K * nb, dtype=torch.int32, device=self.device))

    def _phase_batched(self, x, y, phase: str, train: bool, optim=None, order=None) -> None:
        """One pass over [K, n, ...] data for all replicas at once; ``order`` [K, n] sample order (training)."""
        if self.use_graph and self._phase_graphed(x, y, phase, train, optim, order):
            return
        state = self._state()
        try:
            self._phase_batched_inner(state, x, y, phase, train, optim, order)
        finally:
            if train:
                self._writeback(state)

    def _phase_graphed(self, x, y, phase: str, train: bool, optim, order) -> bool:
        """Replay (or capture at the second occurrence of) this pass as one hipGraph; False: run it eagerly."""
        key = (phase, train, x.data_ptr(), tuple(x.shape), None if optim is None else id(optim))
        if key not in self._seen:                  # first occurrence: eager (warm-up of everything lazy)
            self._seen.add(key)
            return False
        if self._gstate is None:                   # static tensors the graphs read / update: BN statistics, shuffles
            self._gstate = dict(self._stack)
            for k in self._fbuf_keys:
                self._gstate[k] = self._stack[k].detach().contiguous().clone()
        with torch.no_grad():
            for k in self._fbuf_keys:
                self._gstate[k].copy_(self._stack[k])
        order_static = None
        if order is not None:
            if self._perm_static is None or self._perm_static.shape != order.shape:
                self._perm_static = torch.empty_like(order)
                self._graphs = {k: g for k, g in self._graphs.items() if not k[1]}      # training graphs read it
            self._perm_static.copy_(order)
            order_static = self._perm_static
        segs = self._graphs.get(key)
        if segs is None:
            n, span = x.shape[1], self.graph_steps * self.B
            segs = []
            try:
                torch.cuda.synchronize(self.device)
                for lo in range(0, n, span):                # one graph per <= graph_steps steps (see __init__)
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        self._phase_batched_inner(self._gstate, x, y, phase, train, optim, order_static,
                                                  lo, min(n, lo + span))
                    segs.append(g)
            except Exception as exc:  # noqa: BLE001 - a family / optimizer that cannot be captured stays eager
                logging.warning("[ModuleEngine] %s: hipGraph capture of the %s pass failed (%s); eager from here on",
                                self.spec.name, phase, str(exc).splitlines()[0] if str(exc) else type(exc).__name__)
                self.use_graph = False
                self._graphs.clear()
                torch.cuda.synchronize(self.device)
                return False
            self._graphs[key] = segs
            # (the capture itself executed nothing: the replay below is the pass)
        for g in segs:
            g.replay()
        if train:
            self._writeback(self._gstate)
        return True

    def _phase_batched_inner(self, state, x, y, phase: str, train: bool, optim, order, lo: int = 0,
                             hi: Optional[int] = None) -> None:
        """Steps of the pass whose first sample index lies in [lo, hi) (default: the whole pass)."""
        n = x.shape[1]
        m0 = self.modules[0]
        m0.train(train)
        kar = torch.arange(self.K, device=self.device).view(self.K, 1)
        for s in range(lo, n if hi is None else min(hi, n), self.B):
            nb = min(self.B, n - s)
            if train and nb < 2:
                continue                                     # BatchNorm cannot train on a single sample
            if order is not None:
                idx = order[:, s:s + nb]
                xb, yb = x[kar, idx], y[kar, idx]
            else:
                xb, yb = x[:, s:s + nb], y[:, s:s + nb]
            if self.classifier:
                yb = yb.long()
            if train:
                if self.spec.augment is not None:
                    xb = self.spec.augment(xb.reshape(self.K * nb, *xb.shape[2:])).reshape(xb.shape)
                losses, preds = self._vstep(state, xb, yb)
                optim.zero_grad(set_to_none=True)
                losses.sum().backward()
                optim.step()
            else:
                with torch.no_grad():
                    losses, preds = self._vstep(state, xb, yb)
            self._metrics(phase, losses, preds, yb, nb)
            if train and self._ep_log is not None:
                pos, st = s // self.B, self.stats[phase]
                self._ep_log[pos, 0].copy_(st.loss_sum)
                self._ep_log[pos, 1].copy_(st.count)
                row = self._ep_row.get(pos)
                if row is not None:
                    self._ep_cm[row].copy_(st.confusion)

    def _alloc_epoch_tables(self, epochs: int) -> None:
        spe = self.steps_per_epoch()
        if not self._alloc_step_tables(epochs * spe) and self._ep_log is not None:
            return
        if not self.log_steps:
            return
        if not self.classifier:
            self._cm_slot = {}                        # loss only: nothing to derive class metrics from
        self._ep_log = torch.zeros(spe, 2, self.K, device=self.device)
        self._ep_row = {p: i for i, p in enumerate(sorted({q % spe for q in self._cm_slot}))}
        self._ep_cm = torch.zeros(max(len(self._ep_row), 1), self.K, self.C, self.C, dtype=torch.int32, device=self.device)
        if self._graphs:                              # captured training passes do not write the new tables
            torch.cuda.synchronize(self.device)
            self._graphs = {k: g for k, g in self._graphs.items() if not k[1]}

    def _collect_epoch_tables(self, e: int) -> None:
        if self._ep_log is None:
            return
        spe = self._ep_log.shape[0]
        self._step_log[e * spe:(e + 1) * spe].copy_(self._ep_log)
        for p, row in self._ep_row.items():
            slot = self._cm_slot.get(e * spe + p)
            if slot is not None:
                self._step_cm[slot].copy_(self._ep_cm[row])

    def _fit_batched(self, epochs: int, active) -> Dict:
        if self._stack_opt is None or not self.persist_optimizer:
            self._stack_opt = self._fresh_optimizer()
        saved = None
        if active is not None and not bool(torch.as_tensor(active).all()):
            inactive = (~torch.as_tensor(active, dtype=torch.bool)).nonzero().flatten().to(self.device)
            ints = {k: self._stack["m." + k][inactive].clone() for k in self.layout.int_keys}
            saved = (inactive, self.params[inactive].clone(), ints)
        perms = self._epoch_perms(epochs).to(self.device).long()
        n = self.train_x.shape[1]
        self._alloc_epoch_tables(epochs)
        out = {}
        for e in range(epochs):
            self.stats["Train"].zero_(); self.stats["Validation"].zero_()
            self._phase_batched(self.train_x, self.train_y, "Train", True, self._stack_opt, perms[e, :, :n])
            self._collect_epoch_tables(e)
            if self.val_x is not None:
                self._phase_batched(self.val_x, self.val_y, "Validation", False)
            out = {"Train": self.stats["Train"].summary(), "Validation": self.stats["Validation"].summary()}
        if saved is not None:                                # nodes that sat the round out keep their model
            with torch.no_grad():
                self.params[saved[0]] = saved[1]
                for k, v in saved[2].items():
                    self._stack["m." + k][saved[0]] = v
        self.opt_steps += epochs * self.steps_per_epoch()
        return out

    # ------------------------------------------------------------------ training / evaluation
    def reset_optimizer(self) -> None:
        self.optims = [m.configure_optimizers() for m in self.modules]
        self._stack_opt = None
        self.opt_steps = 0

    def _run_phase(self, k, x, y, phase, train, optim=None, order=None, step0: int = 0) -> None:
        m, st, C = self.modules[k], self.stats[phase], self.C
        n = x.shape[0]
        for s in range(0, n, self.B):
            idx = order[s:s + self.B] if order is not None else torch.arange(s, min(s + self.B, n), device=self.device)
            if train and len(idx) < 2:
                continue                                     # BatchNorm cannot train on a single sample
            xb, yb = x[idx], y[idx]
            if self.classifier:
                yb = yb.long()
            if train:
                if self.spec.augment is not None:
                    xb = self.spec.augment(xb)
                loss, pred = m.training_loss(xb, yb)
                optim.zero_grad(set_to_none=True)
                loss.backward()
                optim.step()
            else:
                with torch.no_grad():
                    loss, pred = m.training_loss(xb, yb)
            st.loss_sum[k] += loss.detach() * len(idx)
            st.count[k] += len(idx)
            if self.classifier and pred is not None:
                pred = pred.long().clamp(0, C - 1)
                yc = yb.clamp(0, C - 1)
                st.correct[k] += (pred == yc).sum().to(torch.int32)
                st.confusion[k].view(-1).index_add_(0, yc * C + pred, torch.ones_like(pred, dtype=torch.int32))
            if train:
                self._record_client_step(step0 + s // self.B, k)

    def fit(self, epochs, active=None):
        if self.batched:
            try:
                return self._fit_batched(epochs, active)
            except (RuntimeError, NotImplementedError, TypeError) as exc:
                if self.opt_steps > 0 or os.environ.get("P2PFL_MODULE_VMAP_STRICT"):
                    raise                                     # it worked before: a real error, not an unsupported op
                logging.warning("[ModuleEngine] %s: client-batched step failed (%s); per-node loop from here on",
                                self.spec.name, exc)
                self.batched = False
                self._perm_calls = max(self._perm_calls - 1, 0)
        if self.optims is None or not self.persist_optimizer:
            self.reset_optimizer()
        perms = self._epoch_perms(epochs).to(self.device).long()
        n = self.train_x.shape[1]
        self._ep_log = None
        if self._alloc_step_tables(epochs * self.steps_per_epoch()) and not self.classifier:
            self._cm_slot = {}
        out = {}
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
            out = {"Train": self.stats["Train"].summary(), "Validation": self.stats["Validation"].summary()}
        self.opt_steps += epochs * self.steps_per_epoch()
        return out

    def evaluate(self):
        self.stats["Test"].zero_()
        if self.batched:
            try:
                self._phase_batched(self.test_x, self.test_y, "Test", False)
                return self.stats["Test"].summary()
            except (RuntimeError, NotImplementedError, TypeError) as exc:
                if self.opt_steps > 0 or os.environ.get("P2PFL_MODULE_VMAP_STRICT"):
                    raise
                logging.warning("[ModuleEngine] %s: client-batched evaluation failed (%s); per-node loop from here on",
                                self.spec.name, exc)
                self.batched = False
                self.stats["Test"].zero_()
        for k in range(self.K):
            self.modules[k].eval()
            self._run_phase(k, self.test_x[k], self.test_y[k], "Test", False)
        return self.stats["Test"].summary()
