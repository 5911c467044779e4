"""Scenario launcher for the batched (one process per GPU) execution model.

``python -m p2pfl_amd.runtime.launch <scenario config dir> [--device cuda|cpu] [--encrypt]`` — or under
``torch.distributed.run`` with one rank per GPU — runs a whole scenario that the controller
prepared as ``participant_<i>.json`` files (the same files the per-process ``node_start`` consumes,
reference controller.py:272-298): topology from the ``neighbors`` strings, roles / federation /
rounds / epochs / dataset / model from the participant configs. Every rank hosts a contiguous block
of the nodes in one :class:`BatchedFederation`.

Per node it keeps the reference's observable artefacts: ``<log_dir>/<scenario>/participant_<i>.log``,
``metrics/participant_<i>/metrics.csv`` (Test/Train/Validation Loss + macro metrics per round), a
status POST of the live participant config to the controller, and (optionally) a final checkpoint
in the payload format.
"""
from __future__ import annotations

import argparse
import glob
import json
import logging
import os
import time
from typing import List

import numpy as np
import torch
import torch.distributed as dist

from ..config.config import _deep_update, default_participant
from ..learning.codec import save_checkpoint
from ..learning.metrics import metrics_from_confusion
from ..learning.pytorch.statisticslogger import CSVLogger
from ..models.specs import get_spec
from ..node_start import build_dataset_and_model, build_model
from ..parallel.planner import locality_placement, placement_order
from .federation import BatchedFederation

_SPEC_OF = {("MNIST", "MLP"): "mnist-mlp", ("MNIST", "CNN"): "mnist-cnn", ("FEMNIST", "CNN"): "femnist-cnn",
            ("WADI", "MLP"): "wadi-mlp", ("SYSCALL", "Autoencoder"): "syscall-autoencoder"}
# on the CPU these families stay on the client-batched stock-op engine (one vmap step for all nodes) instead of the
# per-node loop that is the CPU side of a kernel-level spec
_GPU_ONLY_SPECS = {("WADI", "MLP"), ("SYSCALL", "Autoencoder")}


def load_participants(config_dir: str) -> List[dict]:
    """Participant documents of a scenario: the ``participant_<i>.json`` files, or - when there are none - the single
    ``scenario.json`` (config/scenario.py: one file for all N nodes, so 1000 co-resident clients need not be 1000
    files)."""
    files = sorted(glob.glob(os.path.join(config_dir, "participant_*.json")),
                   key=lambda p: int(os.path.basename(p).split("_")[1].split(".")[0]))
    if not files:
        from ..config.scenario import SCENARIO_FILE, read_scenario
        if os.path.exists(os.path.join(config_dir, SCENARIO_FILE)):
            return [_deep_update(default_participant(), p) for p in read_scenario(config_dir)]
        raise ValueError(f"No participant files (or {SCENARIO_FILE}) found in {config_dir}")
    out = []
    for f in files:
        with open(f) as fh:
            out.append(_deep_update(default_participant(), json.load(fh)))
    return out


def adjacency_from_participants(parts: List[dict]) -> np.ndarray:
    names = [f"{p['network_args']['ip']}:{p['network_args']['port']}" for p in parts]
    index = {n: i for i, n in enumerate(names)}
    adj = np.zeros((len(parts), len(parts)), dtype=np.float32)
    for i, p in enumerate(parts):
        for nb in p["network_args"]["neighbors"].split():
            if nb in index and index[nb] != i:
                adj[i, index[nb]] = adj[index[nb], i] = 1     # links are bidirectional sockets
    return adj


def load_faults(config_dir: str, parts: List[dict]) -> List[dict]:
    """Fault schedule of a scenario: ``faults.json`` in the config directory and/or ``scenario_args.faults`` of
    participant 0 - a list of ``{"round": r, "action": "kill|revive|pause|resume|drop_link|restore_link",
    "node": i[, "peer": j]}``."""
    out = list(parts[0]["scenario_args"].get("faults") or [])
    path = os.path.join(config_dir, "faults.json")
    if os.path.exists(path):
        with open(path) as f:
            out += json.load(f)
    return out


def _parse_gateway(text: str):
    """"IDX", "IDX:PORT" or "IDX:HOST:PORT" -> (virtual node index, listen host or None, port or None)."""
    bits = str(text).split(":")
    if len(bits) == 1:
        return int(bits[0]), None, None
    if len(bits) == 2:
        return int(bits[0]), None, int(bits[1])
    return int(bits[0]), bits[1], int(bits[2])


def run_scenario(config_dir: str, device: str = "auto", encrypt: bool = False, save_models: bool = False,
                 rounds: int = None, resume: str = None, virtual_devices: int = 1, control_plane: bool = False,
                 gateways=(), wait_start: float = None, placement: str = None) -> dict:
    """Run the scenario of ``config_dir`` on this rank's GPU.

    ``control_plane``: the co-resident nodes get the reference's protocol behaviour from a
    :class:`~p2pfl_amd.runtime.host.FederationHost` per rank - BEAT / ROLE / MODELS_READY on an in-process bus (joined
    across ranks by the host-side bridge), status POSTs per node, and for every ``gateways`` entry (``IDX[:PORT]``) a
    TCP endpoint - on the rank that hosts node IDX - through which reference-protocol ``Node`` processes peer with
    virtual node IDX. ``wait_start`` (seconds): do not start by ourselves, wait that long for a START_LEARNING from
    such a peer (fedstellar/node.py:224-271); whichever rank receives it starts all of them.
    """
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    control_plane = bool(control_plane or gateways)      # a gateway implies the control plane (one status POST path)
    if device == "auto":
        device = "cuda" if torch.cuda.is_available() else "cpu"
    dev = torch.device(device, local_rank) if device == "cuda" else torch.device("cpu")
    if device == "cuda":
        torch.cuda.set_device(local_rank)
    pg = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl" if device == "cuda" else "gloo", rank=rank, world_size=world,
                                **({"device_id": dev} if device == "cuda" else {}))
        pg = dist.group.WORLD

    parts = load_participants(config_dir)
    n = len(parts)
    # placement: ``device_args.placement`` = the rank (GPU) that hosts a node; ``scenario_args.placement ==
    # "locality"`` (or ``placement``) lets the planner keep neighbours on one GPU. Internally the nodes are ordered so
    # that every rank's nodes are one block of slab rows; ``ext[k]`` is the participant index of internal node k and
    # names everything that leaves the process (files, logs, fault schedules, gateways, events).
    wanted = [p["device_args"].get("placement") for p in parts]
    wanted = [None if w in (None, "", "auto") else int(w) for w in wanted]          # "auto" (the default): anywhere
    mode = placement or parts[0]["scenario_args"].get("placement")
    if mode == "locality" and world > 1 and all(w is None for w in wanted):
        wanted = locality_placement(adjacency_from_participants(parts), world)
    ext, block_sizes = placement_order(wanted, world)
    pos = {e: k for k, e in enumerate(ext)}
    parts = [parts[e] for e in ext]
    p0 = parts[pos[0]]
    key = (p0["data_args"]["dataset"], p0["model_args"]["model"])
    root_dir = p0["data_args"].get("root_dir")
    if key in _SPEC_OF and (str(dev).startswith("cuda") or key not in _GPU_ONLY_SPECS):
        spec = get_spec(_SPEC_OF[key])               # client-batched HIP kernels
    else:
        # every other family of the reference's factory: its nn.Module on stock PyTorch-ROCm ops, state in the
        # same [K, P] slab (runtime/module_engine.py). Raises ValueError for unknown dataset / model names.
        from .module_engine import ModuleSpec
        probe = build_dataset_and_model(key[0], key[1], 0, n, root_dir)[0]
        spec = ModuleSpec(f"{key[0]}-{key[1]}", lambda: build_model(key[0], key[1]), int(probe.num_classes),
                          tuple(probe.input_shape), augment=getattr(probe, "augment", None))
    scenario = p0["scenario_args"]["name"] or os.path.basename(os.path.normpath(config_dir))
    federation = p0["scenario_args"]["federation"]
    roles = [p["device_args"]["role"] for p in parts]
    total_rounds = rounds or int(p0["scenario_args"]["rounds"])     # of the whole experiment (a resumed run does the rest)
    epochs = int(p0["training_args"]["epochs"])

    fed = BatchedFederation(spec, adjacency_from_participants(parts), federation=federation, roles=roles, device=dev,
                            epochs=epochs, seed=int(p0["scenario_args"].get("random_seed", 0)), rank=rank,
                            world_size=world, process_group=pg, encrypt=encrypt, scenario=scenario,
                            heartbeat_period=float(p0.get("HEARTBEAT_PERIOD", 4)),
                            node_timeout=float(p0.get("NODE_TIMEOUT", 20)),
                            aggregation_timeout=float(p0.get("AGGREGATION_TIMEOUT", 60)),
                            # the reference's failure semantics: a dead node is waited for (AGGREGATION_TIMEOUT, on
                            # the virtual clock) until the heartbeat table evicts it NODE_TIMEOUT later
                            instant_detection=False, virtual_devices=virtual_devices, block_sizes=block_sizes,
                            log_steps=True)
    for f in load_faults(config_dir, parts):
        fed.schedule_fault(int(f["round"]), f["action"], pos[int(f["node"])],
                           None if f.get("peer") is None else pos[int(f["peer"])], f.get("phase", "start"))
    # data shards of the local nodes (real files when present, synthetic otherwise)
    shards = [build_dataset_and_model(key[0], key[1], ext[i], n, root_dir)[0] for i in fed.local]

    # rows per split: the smallest shard of the WHOLE scenario (splits can differ by a row between nodes), agreed
    # across ranks so that every rank stacks the same shape
    lens = [[len(t) if t is not None else 0 for t in s.tensors()] for s in shards]
    mins = torch.tensor([min(l[i] for l in lens) for i in range(6)], dtype=torch.int64)
    n_train_local = torch.tensor([l[0] for l in lens], dtype=torch.int64)
    n_train_all = n_train_local
    if pg is not None:
        work = mins.to(dev) if device == "cuda" else mins
        dist.all_reduce(work, op=dist.ReduceOp.MIN, group=pg)
        mins = work.cpu()
        gathered = [None] * world
        dist.all_gather_object(gathered, n_train_local.tolist(), group=pg)
        n_train_all = torch.tensor([v for g in gathered for v in g], dtype=torch.int64)

    def stack(idx):
        ts = [s.tensors()[idx] for s in shards]
        if ts[0] is None:
            return None
        return torch.stack([t[:int(mins[idx])] for t in ts])
    fed.init_params(start_node=next((i for i, p in enumerate(parts) if p["device_args"].get("start")), 0))
    # FedAvg weights = the true per-node training-set sizes of the whole scenario (node.py:447)
    fed.set_local_data(*(stack(i) for i in range(6)), n_train_all=n_train_all.tolist())
    first_round = 0
    if resume:
        # resume = every node restarts from its checkpoint (the reference payload bytes) and the
        # experiment continues where it stopped: round counter, shuffle counter, roles (the SDFL leader) and the
        # membership state (faults scheduled for the rounds already done are replayed without training)
        from ..learning.codec import load_checkpoint
        metas = []
        for k, i in enumerate(fed.local):
            arrays, _, _, meta = load_checkpoint(os.path.join(resume, f"participant_{ext[i]}_final.ckpt"))
            fed.engine.set_arrays(k, arrays)
            metas.append(meta or {})
            first_round = max(first_round, int((meta or {}).get("round", 0)))
        roles_saved = {int(i): m.get("role") for i, m in zip(fed.local, metas) if m.get("role")}
        if pg is not None:
            allr = [None] * world
            dist.all_gather_object(allr, roles_saved, group=pg)
            roles_saved = {k: v for d in allr for k, v in d.items()}
        for i, role in roles_saved.items():
            fed.roles[i] = role
        fed.fast_forward(first_round)

    fed.prepare()       # captured graphs / exchange plan before round 0 (models and counters stay as they are)
    log_root = os.path.join(p0["tracking_args"].get("log_dir") or "logs", scenario)
    os.makedirs(log_root, exist_ok=True)
    loggers = {i: CSVLogger(log_root, name="metrics", version=f"participant_{ext[i]}") for i in fed.local}
    node_logs = {i: open(os.path.join(log_root, f"participant_{ext[i]}.log"), "a") for i in fed.local}

    def node_log(i, msg):
        node_logs[i].write(f"[{time.strftime('%Y-%m-%d %H:%M:%S')}] {msg}\n")
        node_logs[i].flush()

    def post_status(i):
        ctrl = parts[i]["scenario_args"].get("controller")
        if not ctrl:
            return
        try:
            import requests
            cfgp = dict(parts[i])
            cfgp["device_args"] = dict(cfgp["device_args"], role=fed.roles[i])
            requests.post(f"http://{ctrl}/scenario/{scenario}/node/update", data=json.dumps(cfgp),
                          headers={"Content-Type": "application/json", "X-Scenario-Token":
                                   str(parts[i]["scenario_args"].get("controller_token") or "")}, timeout=1)
        except Exception:
            pass

    t0 = time.time()
    total_samples = 0
    rounds = max(total_rounds - first_round, 0)
    if rounds == 0:
        logging.warning("resumed at round %d of %d: nothing left to do", first_round, total_rounds)
    done = [0]

    def after_round(r, rep):
        nonlocal total_samples
        done[0] += 1
        total_samples += rep.train_samples
        test = fed.engine.stats["Test"].summary()
        train = fed.engine.stats["Train"].summary()
        # the reference's per-step series (Train/Loss + macro metrics of the step's batch every 20 optimizer steps,
        # lightninglearner.py:188, mnist/models/mlp.py:33-106) on the global step axis, from the snapshots the HIP
        # engine takes inside its captured graph
        per_loss = fed.engine.step_losses() if hasattr(fed.engine, "step_losses") else None
        if per_loss is not None:
            per_cm = fed.engine.step_metrics()          # {} for models without classes (autoencoders): loss only
            base = r * per_loss.shape[0]
            every = fed.engine.LOG_EVERY
            for t in range(every, per_loss.shape[0] + 1, every):
                for k, i in enumerate(fed.local):
                    if fed.alive[i]:
                        row = {"Train/Loss": float(per_loss[t - 1, k])}
                        if t in per_cm:
                            row.update({f"Train/{name}": v for name, v in per_cm[t][k].items()})
                        loggers[i].log_metrics_direct(row, base + t)
        for k, i in enumerate(fed.local):
            # (with the per-step series above, "Train/Loss" is the step axis; the epoch mean goes to TrainEpoch/Loss)
            m = {"Round": r, "Test/Loss": float(test["loss"][k]),
                 ("TrainEpoch/Loss" if per_loss is not None else "Train/Loss"): float(train["loss"][k]),
                 "Federation/RoundSeconds": rep.seconds, "Federation/ExchangedRows": rep.exchanged_rows,
                 "Federation/AggregationTimeouts": rep.timeouts, "Federation/EncryptedBytes": rep.encrypted_bytes}
            for ph, sec in rep.phase_seconds.items():
                m[f"Federation/Phase_{ph}_seconds"] = sec
            for ph, st in (("Test", test), ("Train", train)):
                for name, v in metrics_from_confusion(st["confusion"][k]).items():
                    if name != "MicroAccuracy":
                        m[f"{ph}Epoch/{name}"] = v
            loggers[i].log_metrics_direct(m, r)
            node_log(i, f"round {r} role {fed.roles[i]} alive {bool(fed.alive[i])} test_loss {m['Test/Loss']:.4f} "
                        f"test_acc {m['TestEpoch/Accuracy']:.4f} ({rep.seconds:.3f}s)")
            if r % 2 == 0 and not control_plane:
                post_status(i)

    if control_plane or gateways:
        # one host per rank for the nodes that rank hosts; the hosts are joined by the rank bridge (runtime/bridge.py)
        from .host import FederationHost
        names = [f"{p['network_args']['ip']}:{p['network_args']['port']}" for p in parts]
        host = FederationHost(fed, names=names, scenario=scenario, participant=p0,
                              status_poster=lambda i, view: post_status(i))
        host.round_callback = lambda rep: after_round(first_round + len(host.reports) - 1, rep)
        host.start()
        for g in gateways:
            vidx, gh, gp = _parse_gateway(g)
            vidx = pos[vidx]
            if vidx not in fed.local:
                continue                                  # the rank that hosts the node opens its gateway
            gw = host.gateway(vidx, gh or parts[vidx]["network_args"]["ip"], gp)
            node_log(vidx, f"TCP gateway of virtual node {ext[vidx]} listening on {gw.get_name()}")
        if wait_start is not None:
            if not host.wait_started(wait_start):
                logging.warning("no START_LEARNING within %.0f s: starting locally", wait_start)
                host.start_learning(rounds, epochs)
        elif rank == 0:
            host.start_learning(rounds, epochs)          # (the bridge starts the other ranks' loops)
        host.wait_started(30 if wait_start is None else wait_start + 30)
        host.wait_finished()
        host.stop()
        rounds = done[0]
    else:
        for r in range(first_round, total_rounds):
            after_round(r, fed.run_round())
    final = fed.engine.evaluate()
    elapsed = time.time() - t0
    if save_models:
        for k, i in enumerate(fed.local):
            save_checkpoint(os.path.join(log_root, f"participant_{ext[i]}_final.ckpt"),
                            fed.engine.get_arrays(k),
                            [f"{parts[i]['network_args']['ip']}:{parts[i]['network_args']['port']}"],
                            int(fed.weights[i]), meta={"round": first_round + rounds, "role": fed.roles[i]})
    for f in node_logs.values():
        f.close()
    for lg in loggers.values():
        lg.finalize()
    summary = {"scenario": scenario, "rank": rank, "world": world, "nodes": n, "local_nodes": fed.n_local,
               "rounds": rounds, "seconds": elapsed, "rounds_per_sec": rounds / max(elapsed, 1e-9),
               "samples_per_sec": total_samples / elapsed, "final_test_accuracy": float(final["accuracy"].mean()),
               "final_test_loss": float(final["loss"].mean()), "first_round": first_round,
               "membership_events": [[round(t, 3), what, ext[int(node)]]
                                     for t, what, node in fed.membership.events],
               "placement": None if block_sizes is None else {"block_sizes": block_sizes, "order": ext}}
    if rank == 0:
        with open(os.path.join(log_root, "summary.json"), "w") as f:
            json.dump(summary, f, indent=2)
    if world > 1:
        try:                      # leave together: a rank that tears its communicator down while a peer still
            dist.barrier()        # finishes a collective makes that peer abort
        except Exception as e:    # noqa: BLE001 - a peer that already died must not keep this rank from exiting
            logging.warning("final barrier failed: %s", e)
        dist.destroy_process_group()
    return summary


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("config_dir")
    ap.add_argument("--device", default="auto", choices=["auto", "cuda", "cpu"])
    ap.add_argument("--encrypt", action="store_true")
    ap.add_argument("--save-models", action="store_true")
    ap.add_argument("--rounds", type=int, default=None,
                    help="total rounds of the experiment (default: scenario_args.rounds); a --resume run does the remaining ones")
    ap.add_argument("--resume", default=None, help="directory with participant_<i>_final.ckpt files to start from")
    ap.add_argument("--control-plane", action="store_true",
                    help="serve the reference's node protocol for the co-resident nodes (heartbeats, roles, "
                         "MODELS_READY, status POSTs; START/STOP_LEARNING honoured) - one host per rank, joined by a "
                         "host-side TCP bridge")
    ap.add_argument("--gateway", action="append", default=[], metavar="IDX[:PORT]",
                    help="open a TCP endpoint for virtual node IDX so reference-protocol nodes can peer with it "
                         "(implies --control-plane; repeatable)")
    ap.add_argument("--wait-start", type=float, default=None, metavar="SECONDS",
                    help="with a gateway: wait this long for START_LEARNING from an external node instead of starting")
    ap.add_argument("--placement", default=None, choices=["locality"],
                    help="multi-GPU: let the planner place neighbours on the same GPU (fewer rows over xGMI); explicit "
                         "placements are read from device_args.placement")
    ap.add_argument("--virtual-devices", type=int, default=1,
                    help="one-GPU runs: split the nodes into this many virtual devices whose exchange takes the "
                         "inter-GPU path (staging + per-link AES with --encrypt)")
    a = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO)
    print(json.dumps(run_scenario(a.config_dir, a.device, a.encrypt, a.save_models, a.rounds, a.resume,
                                  a.virtual_devices, a.control_plane, a.gateway, a.wait_start, a.placement)), flush=True)


if __name__ == "__main__":
    main()
