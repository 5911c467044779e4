"""The RCCL path on the one GPU of the test box: a ONE-rank "nccl" process group (backend "nccl" IS RCCL on ROCm)
drives every collective the batched federation uses - variable-split all_to_all_single (self-split), all_reduce,
broadcast - through :class:`DeviceExchange`, then a whole 1-rank federation round with the group attached.
(SURVEY 4.2 tier 5; multi-rank index math is covered by the 2-rank gloo tests in test_engine_gpu.py /
test_federation_cpu.py. `rocprofv3 --kernel-trace` of this file is checked into profiles/r2_rccl_1rank_kernel_trace.csv.)
"""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nccl_group():
    import torch.distributed as dist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist.group.WORLD
    torch.cuda.synchronize()
    dist.destroy_process_group()


def test_one_rank_rccl_collectives(nccl_group):
    import torch.distributed as dist
    from p2pfl_amd.parallel.collectives import DeviceExchange
    assert dist.get_backend(nccl_group) == "nccl"
    ex = DeviceExchange(0, 1, nccl_group, "cuda", encrypt=False)
    P = 4096
    params = torch.randn(6, P, device="cuda")
    staging = torch.zeros(3, P, device="cuda")
    # self-split all-to-all: rank 0 "sends" rows 0, 2, 5 to rank 0 and receives 3 rows from it
    got = ex.exchange_rows(params, [[0, 2, 5]], [3], staging)
    torch.cuda.synchronize()
    assert got == 3 and torch.equal(staging, params[[0, 2, 5]])
    assert ex.bytes_sent == 3 * P * 4
    # the encrypted variant: device AES on the way out and back in
    exe = DeviceExchange(0, 1, nccl_group, "cuda", encrypt=True, scenario="t")
    staging.zero_()
    exe.exchange_rows(params, [[1, 3, 4]], [3], staging)
    torch.cuda.synchronize()
    assert torch.equal(staging, params[[1, 3, 4]]) and exe.bytes_encrypted == 2 * 3 * P * 4
    # all-reduce of a partial sum, broadcast of the initial model
    part = torch.randn(1, P, device="cuda")
    ref = part.clone()
    out = ex.allreduce_partial(part)
    row = torch.randn(P, device="cuda")
    ref_row = row.clone()
    ex.broadcast(row, 0)
    ex.barrier()
    torch.cuda.synchronize()
    assert torch.equal(out, ref) and torch.equal(row, ref_row)


def test_one_rank_federation_round_over_rccl(nccl_group):
    """A federation with the process group attached: init diffusion = RCCL broadcast, exchange = RCCL all-to-all
    (zero-length splits: every row is local) - must equal the group-less run bit for bit."""
    from p2pfl_amd.datasets.synthetic import synthetic_shards
    from p2pfl_amd.models.specs import get_spec
    from p2pfl_amd.runtime.federation import BatchedFederation
    from p2pfl_amd.utils.topologymanager import TopologyManager
    res = []
    for pg in (None, nccl_group):
        tm = TopologyManager(n_nodes=6, b_symmetric=True, undirected_neighbor_num=3, seed=3)
        tm.generate_ring_topology()
        fed = BatchedFederation(get_spec("mnist-mlp"), tm.topology, device="cuda", engine="hip", epochs=1, seed=2,
                                process_group=pg)
        fed.init_params()
        fed.set_local_data(*synthetic_shards(6, 64, 16, 32, 10, seed=9, device="cuda"))
        for _ in range(2):
            fed.run_round()
        torch.cuda.synchronize()
        res.append(fed.engine.params.clone())
        fed.engine.close()
    assert torch.equal(res[0], res[1])


def test_host_side_gloo_group_beside_the_rccl_group_and_steady_state_allocations(nccl_group):
    """What a multi-rank job adds on the GPU box: (1) the host-side gloo group that the ranks use to agree on
    evictions / STOP / gateways (runtime/federation.py ``ensure_host_group``, runtime/host.py ``_agree``) is created
    BESIDE the RCCL group and reduces CPU tensors while RCCL owns the device; (2) the row exchange keeps its buffers:
    after the first round an (encrypted) exchange allocates no device memory."""
    import torch.distributed as dist
    from p2pfl_amd.parallel.collectives import DeviceExchange
    host = dist.new_group(backend="gloo")
    assert dist.get_backend(host) == "gloo" and dist.get_backend(nccl_group) == "nccl"
    flags = torch.tensor([1, 0, 1], dtype=torch.uint8)
    dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=host)
    dev = torch.ones(4, device="cuda")
    dist.all_reduce(dev, group=nccl_group)                      # ... and RCCL still works next to it
    torch.cuda.synchronize()
    assert flags.tolist() == [1, 0, 1] and dev.tolist() == [1.0] * 4
    P = 8192
    params = torch.randn(8, P, device="cuda")
    staging = torch.zeros(4, P, device="cuda")
    part = torch.randn(1, P, device="cuda")
    for encrypt in (False, True):
        ex = DeviceExchange(0, 1, nccl_group, "cuda", encrypt=encrypt, scenario="t")
        ex.exchange_rows(params, [[0, 3, 5, 6]], [4], staging)      # round 1: buffers, index tensors, round keys
        if not encrypt:
            ex.allreduce_partial(part)
        torch.cuda.synchronize()
        before = torch.cuda.memory_stats()["allocation.all.allocated"]
        for _ in range(3):
            ex.exchange_rows(params, [[0, 3, 5, 6]], [4], staging)
            if not encrypt:
                ex.allreduce_partial(part)
        torch.cuda.synchronize()
        assert torch.cuda.memory_stats()["allocation.all.allocated"] == before, encrypt
        assert torch.equal(staging, params[[0, 3, 5, 6]])
