"""Shape fuzzing of the client-batched kernels with hypothesis (SURVEY 4.2 tier 2): odd batch sizes,
feature counts that are not multiples of the 32/128 tile edges or of the float4 width, slab-strided views.
Every case is compared with the plain PyTorch fp32 op."""
import pytest
import torch

hypothesis = pytest.importorskip("hypothesis")
from hypothesis import given, settings, strategies as st  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"
FUZZ = settings(max_examples=12, deadline=None, derandomize=True)


def hip():
    from p2pfl_amd.ops import hip as h
    h.load()
    return h


def rel(a, b):
    return ((a - b).norm() / b.norm().clamp(min=1e-12)).item()


def strided(K, shape, pad):
    n = 1
    for s in shape:
        n *= s
    slab = torch.randn(K, n + pad, device=DEV)
    return slab[:, :n].unflatten(1, shape)


dims = st.tuples(st.integers(1, 4), st.integers(1, 40), st.integers(1, 700), st.integers(1, 200), st.sampled_from([0, 3, 64]))


@FUZZ
@given(dims, st.booleans())
def test_fuzz_linear_fwd_dgrad(d, flag):
    K, B, In, Out, pad = d
    h = hip()
    x = torch.randn(K, B, In, device=DEV)
    w = strided(K, (Out, In), pad) * 0.1
    b = strided(K, (Out,), pad)
    y = torch.empty(K, B, Out, device=DEV)
    h.linear_fwd(x, w, b, y, flag)
    ref = torch.baddbmm(b.unsqueeze(1), x, w.transpose(1, 2))
    ref = ref.relu() if flag else ref
    assert rel(y, ref) < 1.5e-2 or ref.norm() < 1e-3
    dy = torch.randn(K, B, Out, device=DEV)
    act = torch.randn(K, B, In, device=DEV) if flag else None
    dx = torch.full((K, B, In), float("nan"), device=DEV)
    h.linear_dgrad(dy, w, act, dx)
    refd = torch.bmm(dy, w)
    refd = refd * (act > 0) if flag else refd
    assert torch.isfinite(dx).all() and (rel(dx, refd) < 1.5e-2 or refd.norm() < 1e-3)


@FUZZ
@given(dims)
def test_fuzz_linear_wgrad_and_fused_adam(d):
    K, B, In, Out, pad = d
    h = hip()
    dy = torch.randn(K, B, Out, device=DEV)
    x = torch.randn(K, B, In, device=DEV)
    dw = strided(K, (Out, In), pad)
    db = strided(K, (Out,), pad)
    h.linear_wgrad(dy, x, dw, db)
    rw, rb = torch.bmm(dy.transpose(1, 2), x), dy.sum(1)
    assert rel(dw, rw) < 1.5e-2 and rel(db, rb) < 1.5e-2
    # fused wgrad + Adam == Adam on the (bf16-operand) gradient the unfused kernel produced
    P = Out * In + Out + pad
    slabs = [torch.randn(K, P, device=DEV) * 0.1 for _ in range(3)]
    slabs[1].zero_(); slabs[2].zero_()
    views = [(s[:, :Out * In].unflatten(1, (Out, In)), s[:, Out * In:Out * In + Out]) for s in slabs]
    w0, b0 = views[0][0].clone(), views[0][1].clone()
    h.linear_wgrad_adam(dy, x, views[0][0], views[1][0], views[2][0], views[0][1], views[1][1], views[2][1],
                        lr=1e-2, t=1)
    for p0, g, p1 in ((w0, dw, views[0][0]), (b0, db, views[0][1])):
        m = 0.1 * g
        v = 0.001 * g * g
        ref = p0 - 1e-2 * (m / 0.1) / ((v / 0.001).sqrt() + 1e-8)
        assert torch.allclose(p1, ref, atol=2e-4, rtol=1e-3)


@FUZZ
@given(st.integers(1, 5), st.integers(1, 40), st.integers(2, 70), st.integers(0, 39))
def test_fuzz_softmax_ce(K, B, C, cut):
    h = hip()
    n = B + 3
    logits = torch.randn(K, B, C, device=DEV) * 3
    labels = torch.randint(0, C, (K, n), device=DEV, dtype=torch.int32)
    sidx = torch.stack([torch.randperm(n, device=DEV)[:B] for _ in range(K)]).int()
    nv = min(cut, B) if cut else B
    n_valid = torch.full((K,), nv, dtype=torch.int32, device=DEV)
    dl = torch.zeros_like(logits)
    loss = torch.zeros(K, device=DEV)
    correct = torch.zeros(K, dtype=torch.int32, device=DEV)
    count = torch.zeros(K, dtype=torch.int32, device=DEV)
    conf = torch.zeros(K, C, C, dtype=torch.int32, device=DEV)
    h.softmax_ce(logits, labels, sidx, dl, loss, correct, count, conf, n_valid)
    y = torch.gather(labels.long(), 1, sidx.long())
    for k in range(K):
        lg = logits[k, :nv].clone().requires_grad_(True)
        ref = torch.nn.functional.cross_entropy(lg, y[k, :nv], reduction="sum")
        ref.backward()
        assert abs(loss[k].item() - ref.item()) < 1e-3 * max(1.0, abs(ref.item()))
        assert torch.allclose(dl[k, :nv], lg.grad / nv, atol=1e-5)
        assert (dl[k, nv:] == 0).all()
        pred = logits[k, :nv].argmax(1)
        assert correct[k].item() == int((pred == y[k, :nv]).sum()) and count[k].item() == nv
        assert conf[k].sum().item() == nv and conf[k].diag().sum().item() == correct[k].item()


@FUZZ
@given(st.integers(1, 12), st.integers(0, 9), st.integers(1, 1250), st.integers(0, 2 ** 31 - 1))
def test_fuzz_fedavg_groups(n_local, n_remote, P4, seed):
    from p2pfl_amd.runtime.aggregation import build_plan, fedavg_reference, run_plan
    P = 4 * P4          # slab rows are padded to a multiple of 64 floats (runtime/slab.py): float4 lanes
    gen = torch.Generator().manual_seed(seed)
    local = torch.randn(n_local, P, device=DEV)
    remote = torch.randn(n_remote, P, device=DEV) if n_remote else None
    pool = n_local + n_remote
    groups, free = [], list(range(n_local))
    while free and len(groups) < 6:
        k = int(torch.randint(1, pool + 1, (1,), generator=gen))
        srcs = torch.randperm(pool, generator=gen)[:k].tolist()
        nd = int(torch.randint(1, len(free) + 1, (1,), generator=gen))
        dsts, free = free[:nd], free[nd:]
        groups.append((srcs, torch.randint(1, 1000, (k,), generator=gen).tolist(), dsts))
    plan = build_plan(groups)
    out = torch.zeros(n_local, P, device=DEV)
    ref = torch.zeros(n_local, P, device=DEV)
    run_plan(local, remote, out, plan)
    fedavg_reference(local, remote, ref, plan)
    assert torch.allclose(out, ref, atol=1e-5, rtol=1e-5)


@FUZZ
@given(st.integers(1, 4), st.integers(1, 17500), st.integers(1, 6))
def test_fuzz_optimizers_match_torch(K, n4, steps):
    h = hip()
    n = 4 * n4          # optimizer ranges are 64-float aligned pieces of the slab (float4 lanes)
    p = torch.randn(K, n, device=DEV)
    ref = [torch.nn.Parameter(p[k].clone()) for k in range(K)]
    for kind in ("adam", "sgd"):
        opt = (torch.optim.Adam(ref, lr=1e-3) if kind == "adam" else torch.optim.SGD(ref, lr=1e-3, momentum=0.9))
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        for t in range(1, steps + 1):
            g = torch.randn(K, n, device=DEV)
            for k in range(K):
                ref[k].grad = g[k].clone()
            opt.step()
            if kind == "adam":
                h.adam_step(p, g, m, v, lr=1e-3, t=t)
            else:
                h.sgdm_step(p, g, m, lr=1e-3, momentum=0.9)
        assert torch.allclose(p, torch.stack([r.data for r in ref]), atol=2e-6, rtol=1e-5)
