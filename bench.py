#!/usr/bin/env python
"""Headline benchmark: federated rounds/sec + aggregate client samples/sec, MNIST CNN, 100 nodes
(BASELINE.json). One "step" = one complete federated round of every node hosted by the job:

    evaluate (test shard) -> local training (epochs x ceil(n_train/32) Adam steps, batch 32,
    shuffled, + per-epoch validation) -> model exchange -> FedAvg over each node's train set.

Launch:  python bench.py --gpus N --steps K --warmup W        (N > 1: this process starts N rank processes itself,
                                                              one per GPU over RCCL, BEFORE touching the GPU)
         python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
                                                             (an external launcher that sets RANK/WORLD_SIZE)
A rank exits non-zero when the box has fewer GPUs than ranks (unless --cpu / --rehearse-on-one-gpu); ``n_gpus`` in the
JSON line is the number of ranks of the process group that really ran (``config.rccl_ranks`` / ``config.backend``).

Scaling is weak by default: every GPU hosts ``--nodes-per-gpu`` (100) federation nodes of one
N x 100-node random-topology scenario, so N = 1 is exactly the named 100-node configuration.
Data is synthetic MNIST-shaped (reference shard sizes for a 100-node split: 540 train / 60 val /
100 test per node), weights are random-init. Compute precision is fp32 end to end by default, like the
reference (its Trainer runs 32-bit: fedstellar/learning/pytorch/lightninglearner.py:182-193): fp32 master weights,
fp32 Adam moments, fp32 accumulation, fp32 operands on v_mfma_f32_32x32x2_f32 - except that the stage-2
convolutions form each fp32 product from an EXACT 3-term bf16 decomposition of both operands (six bf16 MFMAs, dropped
cross terms <= 2^-24 of the product; error against a float64 oracle equal to the fp32 MFMA kernels',
tests/test_hip_kernels.py::test_conv2_split_is_fp32_accurate; a whole SGD round agrees with eager torch to 1e-6).
``--dtype fp32-mfma`` uses nothing but fp32 MFMAs; the default run ALSO measures that policy for a few rounds after its
timed region and reports it as ``config.fp32_mfma_only`` so both numbers come from the same process and box.
``--dtype bf16`` selects the labelled fast mode (operands rounded to bf16, fp32 accumulation).

``vs_baseline``: the reference publishes no numbers, so the ratio is taken against the
reference-equivalent eager PyTorch fp32 loop measured by THIS script (``--impl eager --save-baseline`` writes
profiles/eager_baseline.json, which is read back here); null when that file is absent or for other presets.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from p2pfl_amd.datasets.synthetic import reference_shard_sizes, synthetic_shards  # noqa: E402
from p2pfl_amd.models.specs import get_spec  # noqa: E402
from p2pfl_amd.runtime.federation import BatchedFederation  # noqa: E402
from p2pfl_amd.utils.topologymanager import TopologyManager  # noqa: E402

# written by `python bench.py --impl eager --save-baseline` on one MI355X: the
# reference-equivalent per-node eager PyTorch fp32 loop, client samples/sec on the headline config
BASELINE_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "eager_baseline.json")


def eager_baseline_samples_per_sec():
    try:
        with open(BASELINE_FILE) as f:
            rec = json.load(f)
        return float(rec["value"]) / max(int(rec.get("n_gpus", 1)), 1)
    except (OSError, ValueError, KeyError):
        return None


# the other BASELINE.json configurations (the default flags are #4 without encryption/churn = the headline)
PRESETS = {
    "cnn10-fully": dict(model="mnist-cnn", nodes_per_gpu=10, split_nodes=10, topology="fully"),
    "femnist50-ring": dict(model="femnist-cnn", nodes_per_gpu=50, split_nodes=50, topology="ring"),
    # config #4: AES-128 per-link encryption of every row that crosses a (virtual) device boundary + 10 % churn with the
    # reference's failure semantics (dead nodes are waited for AGGREGATION_TIMEOUT on the virtual clock, then evicted
    # NODE_TIMEOUT later; half of the victims die mid-round, after training). On ONE GPU the 100 nodes are split into
    # 2 virtual devices so that the exchange takes the inter-GPU path (gather -> AES -> staging -> AES^-1 -> reduce).
    "cnn100-encrypted-churn": dict(model="mnist-cnn", nodes_per_gpu=100, split_nodes=100, topology="random",
                                   encrypt=True, churn=0.1, virtual_devices=2),
    "mlp100-random": dict(model="mnist-mlp", nodes_per_gpu=100, split_nodes=100, topology="random"),   # extra: the MLP
    "femnist1000-star": dict(model="femnist-cnn", nodes_per_gpu=1000, split_nodes=1000, topology="star",
                             federation="CFL"),
}


def build_topology(kind: str, n: int, seed: int):
    tm = TopologyManager(n_nodes=n, b_symmetric=True, undirected_neighbor_num=max(n - 1, 2) if kind == "fully" else 3,
                         seed=seed)
    if kind == "fully":
        tm.generate_topology()
    elif kind == "ring":
        tm.generate_ring_topology()
    elif kind == "star":
        tm.generate_server_topology()
    else:   # random: ring + seeded random links (mean degree ~4)
        tm.generate_topology(extra_random_links=2.0)
    return tm.topology


def launch_ranks(n: int, argv) -> int:
    """``python bench.py --gpus N`` without an external launcher: start N fresh rank processes of this script (one per
    GPU; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment), relay their output - rank 0 prints the one JSON
    line - and return non-zero if any rank failed. Runs before this process has made a single GPU call, and the ranks
    are children (never an exec of a process that initialised the GPU)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), P2PFL_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(n, 1) // 2)))
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__), *argv], env=env))
    rc = 0
    deadline = None
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0:
                if rc == 0:
                    print(f"bench.py: rank {r} of {n} exited with code {code}", file=sys.stderr, flush=True)
                    deadline = time.monotonic() + 20.0     # its peers are stuck in a collective: give them a moment, then end them
                rc = rc or (code if code > 0 else 1)
        if deadline is not None and time.monotonic() > deadline:
            for r in pending:
                procs[r].kill()
            for r in pending:
                procs[r].wait()
            break
        time.sleep(0.05)
    return rc


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--nodes-per-gpu", type=int, default=100)
    ap.add_argument("--model", default="mnist-cnn")
    ap.add_argument("--topology", default="random", choices=["random", "ring", "fully", "star"])
    ap.add_argument("--federation", default="DFL")
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--batch-size", type=int, default=32)
    ap.add_argument("--impl", default="hip", choices=["hip", "eager"])
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "fp32-mfma", "bf16"],
                    help="precision of the matrix products of the HIP engine (fp32 = the reference's precision)")
    ap.add_argument("--no-reference-run", action="store_true",
                    help="skip the untimed-for-the-headline extra measurement of the fp32-mfma policy (default dtype only)")
    ap.add_argument("--save-baseline", action="store_true",
                    help="with --impl eager on the headline config: record the result as profiles/eager_baseline.json")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--encrypt", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="single-stream schedule (A/B against the two-stream pipeline)")
    ap.add_argument("--cpu", action="store_true", help="debug: run on the CPU with gloo")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="rehearsal of a multi-rank run on a ONE-GPU box: every rank uses device 0 and the ranks talk "
                         "gloo (RCCL cannot put two ranks on one device); exercises the plan / staging / FedAvg of the "
                         "N-rank scenario at full size, not the fabric - the output is labelled and is no benchmark")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--split-nodes", type=int, default=100,
                    help="data shards are sized as in an N-node split of MNIST (reference mnist.py:100-118)")
    ap.add_argument("--churn", type=float, default=0.0,
                    help="fraction of nodes killed one round after warm-up and revived two rounds later")
    ap.add_argument("--virtual-devices", type=int, default=1,
                    help="1-GPU runs: split the nodes into virtual devices whose exchange takes the inter-GPU path")
    ap.add_argument("--config", default=None, choices=sorted(PRESETS),
                    help="one of the other BASELINE.json configurations (overrides the flags it names)")
    args = ap.parse_args()
    if args.config:
        for k, v in PRESETS[args.config].items():
            setattr(args, k, v)

    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no external launcher: become the launcher (nothing in this process has touched the GPU yet)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; running (and reporting) "
              f"{world}", file=sys.stderr, flush=True)
    if args.cpu:
        device = torch.device("cpu")
        args.impl = "eager"
    else:
        n_dev = torch.cuda.device_count()        # counts devices without initialising the GPU
        if args.rehearse_on_one_gpu:
            local_rank = 0
        elif n_dev < max(world, local_rank + 1):
            print(f"bench.py: rank {rank}: {world} ranks were asked for but this box has {n_dev} GPU(s); one rank per "
                  f"GPU is required (use --rehearse-on-one-gpu for a labelled functional rehearsal over gloo, or --cpu)",
                  file=sys.stderr, flush=True)
            sys.exit(3)
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    host_pg = args.cpu or args.rehearse_on_one_gpu          # gloo: scalars for the final reduction stay on the host
    pg = None
    if world > 1 or os.environ.get("P2PFL_FORCE_PG"):      # P2PFL_FORCE_PG: 1-rank RCCL communicator (smoke test of the multi-rank path)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if host_pg else "nccl", rank=rank, world_size=world,
                                **({} if host_pg else {"device_id": device}))
        pg = dist.group.WORLD

    n_nodes = args.nodes_per_gpu * world if args.scaling == "weak" else args.nodes_per_gpu
    spec = get_spec(args.model)
    topo = build_topology(args.topology, n_nodes, args.seed)
    fed = BatchedFederation(spec, topo, federation=args.federation, device=device, engine=args.impl,
                            batch_size=args.batch_size, epochs=args.epochs, seed=args.seed, rank=rank,
                            world_size=world, process_group=pg, encrypt=args.encrypt,
                            use_graph=not args.no_graph, overlap=not args.no_overlap, precision=args.dtype,
                            instant_detection=args.churn <= 0, virtual_devices=args.virtual_devices)
    # shard sizes of the named 100-node split (reference mnist.py:100-118)
    n_tr, n_va, n_te = reference_shard_sizes(args.split_nodes)
    tx, ty, vx, vy, ex, ey = synthetic_shards(fed.n_local, n_tr, n_va, n_te, spec.num_classes, seed=args.seed,
                                              device=device, first_client=fed.local.start)
    fed.init_params()
    fed.set_local_data(tx, ty, vx, vy, ex, ey)

    def sync():
        if pg is not None:
            dist.barrier()
        if device.type == "cuda":
            torch.cuda.synchronize()

    if args.churn > 0:
        import random
        victims = random.Random(args.seed).sample(range(1, n_nodes), max(1, int(args.churn * n_nodes)))
        for n_v, v in enumerate(victims):       # every other victim dies in the middle of its round (trained, never delivers)
            fed.schedule_fault(args.warmup + 1, "kill", v, phase="mid" if n_v % 2 else "start")
            fed.schedule_fault(args.warmup + 3, "revive", v)
    fed.prepare()                          # graph capture / lazy loading / planning happen here even with --warmup 0
    for _ in range(args.warmup):
        fed.run_round(sync_timing=False)
    sync()
    t0 = time.perf_counter()
    samples = 0
    last = None
    enc_bytes = timeouts = rows = 0
    for _ in range(args.steps):
        last = fed.run_round(sync_timing=False)
        samples += last.train_samples
        enc_bytes += last.encrypted_bytes
        timeouts += last.timeouts
        rows += last.exchanged_rows
    sync()
    elapsed = time.perf_counter() - t0

    # The same scenario under the fp32-mfma policy (nothing but v_mfma_f32_32x32x2_f32), measured AFTER and outside the
    # timed region above: a reader who wants the all-fp32-instruction number gets it from the same process.
    ref_run = None
    n_local = fed.n_local
    events = [[round(t, 1), what, node] for t, what, node in fed.membership.events][:64]
    if (args.impl == "hip" and args.dtype == "fp32" and not args.no_reference_run and not args.cpu and world == 1
            and spec.name in ("mnist-cnn", "femnist-cnn")):
        n_ref = max(1, min(args.steps, 5))
        del fed
        torch.cuda.empty_cache()
        fed = BatchedFederation(spec, topo, federation=args.federation, device=device, engine=args.impl,
                                batch_size=args.batch_size, epochs=args.epochs, seed=args.seed, rank=rank,
                                world_size=world, process_group=pg, encrypt=args.encrypt,
                                use_graph=not args.no_graph, overlap=not args.no_overlap, precision="fp32-mfma",
                                instant_detection=True, virtual_devices=args.virtual_devices)
        fed.init_params()
        fed.set_local_data(tx, ty, vx, vy, ex, ey)
        fed.prepare()
        for _ in range(2):
            fed.run_round(sync_timing=False)
        sync()
        t1 = time.perf_counter()
        ref_samples = 0
        for _ in range(n_ref):
            ref_samples += fed.run_round(sync_timing=False).train_samples
        sync()
        ref_elapsed = time.perf_counter() - t1
        ref = torch.tensor([ref_elapsed, float(ref_samples)], dtype=torch.float64, device="cpu" if host_pg else device)
        if pg is not None:
            tm = ref[:1].clone()
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            sm = ref[1:].clone()
            dist.all_reduce(sm, op=dist.ReduceOp.SUM)
            ref_elapsed, ref_samples = float(tm.item()), float(sm.item())
        ref_run = {"steps": n_ref, "warmup": 2, "ms_per_step": 1000.0 * ref_elapsed / n_ref,
                   "value": ref_samples / ref_elapsed, "unit": "samples/s",
                   "note": "same scenario, every matrix product on v_mfma_f32_32x32x2_f32; measured after the timed region"}

    stats = torch.tensor([elapsed, float(samples)], dtype=torch.float64, device="cpu" if host_pg else device)
    if pg is not None:
        tmax = stats[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        ssum = stats[1:].clone()
        dist.all_reduce(ssum, op=dist.ReduceOp.SUM)
        elapsed, samples = float(tmax.item()), float(ssum.item())
    if rank == 0:
        sps = samples / elapsed
        steps_per_epoch = -(-n_tr // args.batch_size)
        base = (eager_baseline_samples_per_sec()
                if args.config is None and args.impl == "hip" and not args.rehearse_on_one_gpu else None)
        out = {
            "metric": "aggregate client samples/sec (federated rounds, MNIST CNN, 100 nodes per GPU)"
            if args.config is None and spec.name == "mnist-cnn" and args.nodes_per_gpu == 100
            else f"aggregate client samples/sec (federated rounds, {args.config or spec.name}, "
                 f"{args.nodes_per_gpu} nodes per GPU)",
            "value": sps,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": (sps / (base * world)) if base else None,
            "baseline_source": "profiles/eager_baseline.json (bench.py --impl eager --save-baseline, same box class)"
            if base else None,
            "dtype": args.dtype if args.impl == "hip" else "fp32",
            "data": "synthetic",
            "rounds_per_sec": args.steps / elapsed,
            "config": {
                "model": spec.name, "impl": args.impl, "nodes": n_nodes, "nodes_per_gpu": n_local,
                "topology": args.topology, "federation": args.federation, "epochs": args.epochs,
                "global_batch": args.batch_size * n_nodes, "batch_per_node": args.batch_size,
                "train_samples_per_node": n_tr, "optimizer_steps_per_round": steps_per_epoch * args.epochs,
                "seq_len": None, "parallelism": f"fdp{world}x{n_local}", "encrypt": args.encrypt,
                "rccl_ranks": dist.get_world_size() if pg is not None else 1,
                "backend": (dist.get_backend() if pg is not None else None),
                "churn": args.churn, "preset": args.config, "virtual_devices": args.virtual_devices,
                **({"rehearsal": "all ranks on ONE GPU over gloo: a functional rehearsal, not a scaling measurement"}
                   if args.rehearse_on_one_gpu else {}),
                "precision_policy": {
                    "fp32": "fp32 storage / accumulation / optimizer state; fp32 operands on v_mfma_f32_32x32x2_f32, stage-2 "
                            "convolutions via an exact 3-term bf16 split of both operands (6 bf16 MFMAs per product "
                            "chunk, dropped terms <= 2^-24; float64-oracle error equal to the fp32 MFMA kernels)",
                    "fp32-mfma": "fp32 operands on v_mfma_f32_32x32x2_f32 for every matrix product",
                    "bf16": "operands rounded to bf16, fp32 accumulation (labelled fast mode)",
                }.get(args.dtype) if args.impl == "hip" else "torch fp32 (eager)",
                "fp32_mfma_only": ref_run,
                "encrypted_bytes": enc_bytes, "exchanged_rows": rows, "aggregation_timeouts": timeouts,
                "membership_events": events,
                "last_test_accuracy_mean": None if last is None or last.test_accuracy is None
                else float(last.test_accuracy.mean()),
            },
        }
        print(json.dumps(out), flush=True)
        if args.save_baseline and args.impl == "eager" and args.config is None:
            os.makedirs(os.path.dirname(BASELINE_FILE), exist_ok=True)
            with open(BASELINE_FILE, "w") as f:
                json.dump(out, f, indent=1)
    if pg is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
