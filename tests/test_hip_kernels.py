"""Numerics of every gfx950 kernel against a plain PyTorch fp32 reference of the same op.

Every matrix-core kernel exists per precision policy (``prec`` fixture): "fp32" (the default, the reference's
precision: v_mfma_f32_32x32x2_f32, with the stage-2 convolutions on an exact 3-term bf16 split) and "fp32-mfma"
(nothing but fp32 MFMAs) must both match the fp32 reference to round-off (max-normalised error <= 2e-5 dense,
<= 2e-4 convolutions: MIOpen's own fp32 algorithms differ by that much);
"bf16" rounds the operands to bf16 (fp32 accumulate) and is compared with a bf16-appropriate tolerance.
VALU kernels (CE, optimizers, FedAvg, AES) must match to fp32 round-off / exactly.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def hip():
    from p2pfl_amd.ops import hip as h
    h.load()
    return h


@pytest.fixture(params=["fp32-mfma", "fp32", "bf16"])
def prec(request):
    """Runs the test once per precision policy, as the default of every product wrapper."""
    from p2pfl_amd.ops import hip as h
    old = h.default_precision()
    h.set_default_precision(request.param)
    yield request.param
    h.set_default_precision(old)


TOL_DENSE = {"fp32-mfma": 2e-5, "fp32": 2e-5, "bf16": 2e-2}
TOL_CONV = {"fp32-mfma": 2e-4, "fp32": 2e-4, "bf16": 2e-2}


def rel_err(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp(min=1e-12)).item()


def slab_view(K, shape, pad=64):
    """A [K, *shape] view with a slab-like (larger, aligned) client stride."""
    n = 1
    for s in shape:
        n *= s
    stride = -(-n // 64) * 64 + pad
    slab = torch.randn(K, stride, device=DEV)
    return slab[:, :n].unflatten(1, shape)


def route(g, code, P):
    """Dense pre-pool gradient from pooled gradient + arg-max codes. g/code [..., C, P*P]."""
    lead = g.shape[:-1]
    gg = g.reshape(*lead, P, P)
    cc = code.reshape(*lead, P, P)
    out = torch.zeros(*lead, P, 2, P, 2, device=g.device)
    for q in range(4):
        out[..., :, q >> 1, :, q & 1] = torch.where(cc == q, gg, torch.zeros_like(gg))
    return out.reshape(*lead, 2 * P, 2 * P)


@pytest.mark.parametrize("K,B,In,Out,relu", [(3, 32, 784, 256, True), (2, 20, 128, 10, False),
                                             (2, 32, 3136, 2048, True), (1, 5, 36, 7, False),
                                             # skinny heads (split-reduction kernel): aligned, 2 col tiles, ragged
                                             (3, 32, 2048, 10, False), (2, 32, 2048, 62, False),
                                             (2, 45, 1001, 33, True), (1, 7, 515, 64, True),
                                             # reduction split over blocks (few clients): 8 / 4 / 2 splits
                                             (10, 32, 2048, 10, False), (50, 32, 3136, 62, False),
                                             (100, 32, 2048, 10, True), (3, 70, 3137, 62, True)])
def test_linear_fwd(K, B, In, Out, relu, prec):
    h = hip()
    x = torch.randn(K, B, In, device=DEV)
    w = slab_view(K, (Out, In)) * 0.05
    b = slab_view(K, (Out,))
    y = torch.empty(K, B, Out, device=DEV)
    h.linear_fwd(x, w, b, y, relu)
    ref = torch.baddbmm(b.unsqueeze(1), x, w.transpose(1, 2))
    if relu:
        ref = ref.relu()
    assert rel_err(y, ref) < TOL_DENSE[prec]


@pytest.mark.parametrize("K,B,In,Out,mask", [(3, 32, 256, 128, True), (2, 32, 2048, 10, True),
                                             (2, 28, 3136, 2048, False), (1, 7, 50, 6, False)])
def test_linear_dgrad(K, B, In, Out, mask, prec):
    h = hip()
    dy = torch.randn(K, B, Out, device=DEV)
    w = slab_view(K, (Out, In)) * 0.05
    act = torch.randn(K, B, In, device=DEV).relu() if mask else None
    dx = torch.empty(K, B, In, device=DEV)
    h.linear_dgrad(dy, w, act, dx)
    ref = torch.bmm(dy, w)
    if mask:
        ref = ref * (act > 0)
    assert rel_err(dx, ref) < TOL_DENSE[prec]


@pytest.mark.parametrize("K,B,In,Out", [(3, 32, 784, 256), (2, 32, 2048, 10), (2, 17, 3136, 70), (1, 40, 33, 5)])
def test_linear_wgrad(K, B, In, Out, prec):
    h = hip()
    dy = torch.randn(K, B, Out, device=DEV)
    x = torch.randn(K, B, In, device=DEV)
    dw = slab_view(K, (Out, In))
    db = slab_view(K, (Out,))
    h.linear_wgrad(dy, x, dw, db)
    assert rel_err(dw, torch.bmm(dy.transpose(1, 2), x)) < TOL_DENSE[prec]
    assert rel_err(db, dy.sum(1)) < TOL_DENSE[prec]


@pytest.mark.parametrize("K,B,In,Out", [(2, 32, 3136, 300), (1, 20, 130, 70)])
def test_linear_wgrad_adam_matches_unfused(K, B, In, Out, prec):
    h = hip()
    P = In * Out + 64 - (In * Out) % 64 + 512
    slabs = [torch.randn(K, P, device=DEV) * 0.05 for _ in range(3)]
    slabs[1].zero_(); slabs[2].zero_()
    ref = [s.clone() for s in slabs]
    grads = torch.zeros(K, P, device=DEV)
    n = In * Out
    view = lambda s: (s[:, :n].unflatten(1, (Out, In)), s[:, P - 512:P - 512 + Out])
    for t in range(1, 4):
        dy = torch.randn(K, B, Out, device=DEV)
        x = torch.randn(K, B, In, device=DEV)
        (w, b), (mw, mb), (vw, vb) = view(slabs[0]), view(slabs[1]), view(slabs[2])
        h.linear_wgrad_adam(dy, x, w, mw, vw, b, mb, vb, lr=1e-3, t=t)
        gw, gb = view(grads)
        h.linear_wgrad(dy, x, gw, gb)
        h.adam_step(ref[0], grads, ref[1], ref[2], lr=1e-3, t=t)
    for a, r in zip(slabs, ref):
        assert torch.allclose(a, r, atol=1e-6, rtol=1e-5)


@pytest.mark.parametrize("K,B,n,In,Out", [(3, 32, 100, 784, 256), (2, 20, 57, 130, 10), (2, 32, 64, 2048, 10)])
def test_linear_gather_of_batch_rows(K, B, n, In, Out, prec):
    """sidx: forward / weight gradient / fused update read the mini-batch rows straight from the data set."""
    h = hip()
    data = torch.randn(K, n, In, device=DEV)
    sidx = torch.stack([torch.randperm(n, device=DEV)[:B] for _ in range(K)]).int()
    xb = data[torch.arange(K, device=DEV)[:, None], sidx.long()].contiguous()
    w = slab_view(K, (Out, In)) * 0.05
    b = slab_view(K, (Out,)) * 0.1
    y1, y2 = torch.empty(K, B, Out, device=DEV), torch.empty(K, B, Out, device=DEV)
    h.linear_fwd(data, w, b, y1, True, sidx=sidx)
    h.linear_fwd(xb, w, b, y2, True)
    assert torch.equal(y1, y2)
    dy = torch.randn(K, B, Out, device=DEV)
    g1, g2 = (slab_view(K, (Out, In)), slab_view(K, (Out,))), (slab_view(K, (Out, In)), slab_view(K, (Out,)))
    h.linear_wgrad(dy, data, g1[0], g1[1], sidx=sidx)
    h.linear_wgrad(dy, xb, g2[0], g2[1])
    assert torch.equal(g1[0], g2[0]) and torch.equal(g1[1], g2[1])
    P = In * Out + 64 - (In * Out) % 64 + 512
    view = lambda s: (s[:, :In * Out].unflatten(1, (Out, In)), s[:, P - 512:P - 512 + Out])
    res = []
    for gather in (True, False):
        torch.manual_seed(1)
        slabs = [torch.randn(K, P, device=DEV) * 0.05, torch.zeros(K, P, device=DEV), torch.zeros(K, P, device=DEV)]
        (ww, wb), (mw, mb), (vw, vb) = view(slabs[0]), view(slabs[1]), view(slabs[2])
        h.linear_wgrad_adam(dy, data if gather else xb, ww, mw, vw, wb, mb, vb, lr=1e-3, t=1,
                            sidx=sidx if gather else None)
        res.append(slabs[0].clone())
    assert torch.equal(res[0], res[1])


@pytest.mark.parametrize("kind", ["adam", "sgd"])
def test_fused_update_fresh_step_ignores_stale_moments(kind, prec):
    """fresh=True (first step of a new optimizer) on garbage moments == the ordinary step on zeroed ones."""
    h = hip()
    K, B, In, Out = 2, 32, 260, 70
    n = In * Out
    P = n + 64 - n % 64 + 512
    view = lambda s: (s[:, :n].unflatten(1, (Out, In)), s[:, P - 512:P - 512 + Out])
    dy = torch.randn(K, B, Out, device=DEV)
    x = torch.randn(K, B, In, device=DEV)
    res = []
    for fresh in (False, True):
        torch.manual_seed(3)
        w = torch.randn(K, P, device=DEV) * 0.05
        m = torch.zeros(K, P, device=DEV) if not fresh else torch.full((K, P), float("nan"), device=DEV)
        v = m.clone()
        (ww, wb), (mw, mb), (vw, vb) = view(w), view(m), view(v)
        if kind == "adam":
            h.linear_wgrad_adam(dy, x, ww, mw, vw, wb, mb, vb, lr=1e-3, t=1, fresh=fresh)
        else:
            h.linear_wgrad_sgdm(dy, x, ww, mw, wb, mb, lr=1e-2, momentum=0.9, fresh=fresh)
        res.append([t.clone() for t in (ww, wb, mw, mb)] + ([vw.clone(), vb.clone()] if kind == "adam" else []))
    for a, b in zip(*res):
        assert torch.isfinite(b).all() and torch.equal(a, b)


@pytest.mark.parametrize("K,B,In,Out", [(2, 32, 3136, 62), (3, 20, 130, 70), (2, 32, 256, 128)])
def test_linear_wgrad_sgdm_matches_unfused(K, B, In, Out, prec):
    """Fused weight gradient + SGD-momentum (with weight decay) == wgrad kernel followed by sgdm_step."""
    h = hip()
    P = In * Out + 64 - (In * Out) % 64 + 512
    slabs = [torch.randn(K, P, device=DEV) * 0.05 for _ in range(2)]
    slabs[1].zero_()
    ref = [s.clone() for s in slabs]
    grads = torch.zeros(K, P, device=DEV)
    n = In * Out
    view = lambda s: (s[:, :n].unflatten(1, (Out, In)), s[:, P - 512:P - 512 + Out])
    for t in range(1, 4):
        dy = torch.randn(K, B, Out, device=DEV)
        x = torch.randn(K, B, In, device=DEV)
        (w, b), (mw, mb) = view(slabs[0]), view(slabs[1])
        h.linear_wgrad_sgdm(dy, x, w, mw, b, mb, lr=1e-2, momentum=0.9, weight_decay=1e-3)
        gw, gb = view(grads)
        h.linear_wgrad(dy, x, gw, gb)
        h.sgdm_step(ref[0], grads, ref[1], lr=1e-2, momentum=0.9, weight_decay=1e-3)
    for a, r in zip(slabs, ref):     # (weight decay also moves the unused slab columns of the reference)
        for va, vr in zip(view(a), view(r)):
            assert torch.allclose(va, vr, atol=1e-5 * float(vr.abs().max()), rtol=1e-5)


@pytest.mark.parametrize("ks,B,Fm", [(5, 32, 32), (7, 32, 32), (5, 7, 32), (7, 13, 20)])
def test_conv1_fwd(ks, B, Fm, prec):
    h = hip()
    K, n = 3, 40
    data = torch.rand(K, n, 784, device=DEV)
    sidx = torch.stack([torch.randperm(n, device=DEV)[:B] for _ in range(K)]).int()
    w = slab_view(K, (Fm, ks * ks)) * 0.2
    b = slab_view(K, (Fm,)) * 0.1
    a1 = torch.full((K, B, Fm, 196), float("nan"), device=DEV)
    code = torch.full((K, B, Fm, 196), 9, dtype=torch.uint8, device=DEV)
    h.conv1_fwd(data, sidx, w, b, a1, code, ks)
    assert torch.isfinite(a1).all() and (code <= 4).all()
    for k in range(K):
        xb = data[k][sidx[k].long()].view(B, 1, 28, 28)
        # bf16 mode: operands are rounded to bf16 (fp32 accumulate) -> compare with the same rounding first
        xr, wr = (xb.bfloat16().float(), w[k].bfloat16().float()) if prec == "bf16" else (xb, w[k])
        pre = F.conv2d(xr, wr.reshape(Fm, 1, ks, ks), b[k], padding=ks // 2)
        ref = F.max_pool2d(pre.relu(), 2, 2).reshape(B, Fm, 196)
        assert torch.allclose(a1[k], ref, atol=2e-4, rtol=2e-4) if prec == "bf16" else \
            torch.allclose(a1[k], ref, atol=2e-5, rtol=2e-5)
        full = F.max_pool2d(F.conv2d(xb, w[k].reshape(Fm, 1, ks, ks), b[k], padding=ks // 2).relu(), 2, 2)
        assert rel_err(a1[k], full.reshape(B, Fm, 196)) < TOL_CONV[prec]
        # the code must point at the arg-max (or 4 when the window is <= 0)
        win = pre.reshape(B, Fm, 14, 2, 14, 2).permute(0, 1, 2, 4, 3, 5).reshape(B, Fm, 196, 4)
        c = code[k].long()
        pos = c < 4
        picked = torch.gather(win, 3, c.clamp(max=3).unsqueeze(-1)).squeeze(-1)
        assert torch.allclose(picked[pos], ref[pos], atol=2e-4, rtol=2e-4)
        assert (ref[~pos].abs() < 2e-4).all()


@pytest.mark.parametrize("ks,B,Fm,K", [(5, 32, 32, 2), (7, 32, 32, 2), (7, 9, 20, 3), (5, 32, 32, 300)])
def test_conv1_wgrad(ks, B, Fm, K, prec):
    h = hip()
    n = 40
    data = torch.rand(K, n, 784, device=DEV)
    sidx = torch.stack([torch.randperm(n, device=DEV)[:B] for _ in range(K)]).int()
    g = torch.randn(K, B, Fm, 196, device=DEV)
    code = torch.randint(0, 5, (K, B, Fm, 196), device=DEV).to(torch.uint8)
    dw = slab_view(K, (Fm, ks * ks))
    db = slab_view(K, (Fm,))
    h.conv1_wgrad(g, code, data, sidx, dw, db, ks)
    dpre = route(g, code, 14)
    for k in range(K):
        xb = data[k][sidx[k].long()].view(B, 1, 28, 28)
        ref = torch.nn.grad.conv2d_weight(xb, (Fm, 1, ks, ks), dpre[k], padding=ks // 2).reshape(Fm, ks * ks)
        assert rel_err(dw[k], ref) < TOL_CONV[prec]
        assert rel_err(db[k], dpre[k].sum(dim=(0, 2, 3))) < TOL_CONV[prec]


@pytest.mark.parametrize("cores", [False, True])
@pytest.mark.parametrize("ks,B,K", [(5, 32, 2), (3, 32, 2), (5, 9, 3), (5, 32, 11), (3, 7, 9)])
def test_conv2_fwd(ks, B, K, prec, cores):
    h = hip()
    assert not cores or prec == "fp32-mfma"      # (other combinations are dropped at collection: tests/conftest.py)
    Cin, Cout = 32, 64
    a1 = torch.randn(K, B, Cin, 196, device=DEV).relu()
    w = slab_view(K, (Cout, Cin * ks * ks)) * 0.05
    b = slab_view(K, (Cout,)) * 0.1
    a2 = torch.empty(K, B, Cout, 49, device=DEV)
    code = torch.empty(K, B, Cout, 49, dtype=torch.uint8, device=DEV)
    h.conv2_fwd(a1, w, b, a2, code, ks, coresident=cores)
    for k in range(K):
        pre = F.conv2d(a1[k].view(B, Cin, 14, 14), w[k].reshape(Cout, Cin, ks, ks), b[k], padding=ks // 2)
        ref = F.max_pool2d(pre.relu(), 2, 2).reshape(B, Cout, 49)
        assert rel_err(a2[k], ref) < TOL_CONV[prec]
        assert ((a2[k] == 0) == (code[k] == 4)).all()


# K = 2..11: batch-split blocks + partial reduction; K = 60 / 130: one block per (client, tap part)
@pytest.mark.parametrize("cores", [False, True])
@pytest.mark.parametrize("ks,B,K", [(5, 32, 2), (3, 32, 2), (5, 9, 3), (5, 32, 11), (3, 7, 9), (5, 4, 60), (3, 3, 130),
                                    (5, 32, 100)])
def test_conv2_wgrad_dgrad(ks, B, K, prec, cores):
    h = hip()
    assert not cores or prec == "fp32-mfma"      # (other combinations are dropped at collection: tests/conftest.py)
    Cin, Cout = 32, 64
    a1 = torch.randn(K, B, Cin, 196, device=DEV).relu()
    w = slab_view(K, (Cout, Cin * ks * ks)) * 0.05
    g = torch.randn(K, B, Cout, 49, device=DEV)
    code = torch.randint(0, 5, (K, B, Cout, 49), device=DEV).to(torch.uint8)
    dw = slab_view(K, (Cout, Cin * ks * ks))
    db = slab_view(K, (Cout,))
    da1 = torch.empty(K, B, Cin, 196, device=DEV)
    h.conv2_wgrad(g, code, a1, dw, db, ks, coresident=cores)
    h.conv2_dgrad(g, code, w, da1, ks, coresident=cores)
    dpre = route(g, code, 7)   # [K,B,Cout,14,14]
    for k in (range(K) if K <= 16 else range(0, K, 13)):
        x = a1[k].view(B, Cin, 14, 14)
        wk = w[k].reshape(Cout, Cin, ks, ks)
        ref_w = torch.nn.grad.conv2d_weight(x, wk.shape, dpre[k], padding=ks // 2).reshape(Cout, -1)
        ref_x = torch.nn.grad.conv2d_input(x.shape, wk, dpre[k], padding=ks // 2).reshape(B, Cin, 196)
        assert rel_err(dw[k], ref_w) < TOL_CONV[prec]
        assert rel_err(db[k], dpre[k].sum(dim=(0, 2, 3))) < TOL_CONV[prec]
        assert rel_err(da1[k], ref_x) < TOL_CONV[prec]


def _conv_fuzz_case(K, B, ks, precision, F1):
    """Both conv stages, forward and backward, for one (clients, batch, kernel size) against torch."""
    h = hip()
    Cin, Cout = 32, 64
    tol = TOL_CONV[precision]
    ks1 = 5 if ks == 5 else 7               # MNIST CNN: 5x5 / 5x5, FEMNIST CNN: 7x7 / 3x3
    # stage 1 (C_in = 1) with an index gather like the engine's
    n = B + 3
    data = torch.randn(K, n, 784, device=DEV)
    sidx = torch.stack([torch.randperm(n, device=DEV)[:B] for _ in range(K)]).to(torch.int32)
    w1 = slab_view(K, (F1, ks1 * ks1)) * 0.2
    b1 = slab_view(K, (F1,)) * 0.1
    a1 = torch.empty(K, B, F1, 196, device=DEV)
    code1 = torch.empty(K, B, F1, 196, dtype=torch.uint8, device=DEV)
    h.conv1_fwd(data, sidx, w1, b1, a1, code1, ks1, prec=precision)
    g1 = torch.randn(K, B, F1, 196, device=DEV)
    dw1, db1 = slab_view(K, (F1, ks1 * ks1)), slab_view(K, (F1,))
    h.conv1_wgrad(g1, code1, data, sidx, dw1, db1, ks1, prec=precision)
    # stage 2
    x2 = torch.randn(K, B, Cin, 196, device=DEV).relu()
    w2 = slab_view(K, (Cout, Cin * ks * ks)) * 0.05
    b2 = slab_view(K, (Cout,)) * 0.1
    a2 = torch.empty(K, B, Cout, 49, device=DEV)
    code2 = torch.empty(K, B, Cout, 49, dtype=torch.uint8, device=DEV)
    h.conv2_fwd(x2, w2, b2, a2, code2, ks, prec=precision)
    g2 = torch.randn(K, B, Cout, 49, device=DEV)
    dw2, db2 = slab_view(K, (Cout, Cin * ks * ks)), slab_view(K, (Cout,))
    da1 = torch.empty(K, B, Cin, 196, device=DEV)
    h.conv2_wgrad(g2, code2, x2, dw2, db2, ks, prec=precision)
    h.conv2_dgrad(g2, code2, w2, da1, ks, prec=precision)
    dpre1, dpre2 = route(g1, code1, 14), route(g2, code2, 7)
    for k in sorted({0, K - 1, K // 2}):
        xb = data[k][sidx[k].long()].view(B, 1, 28, 28)
        ref = F.max_pool2d(F.conv2d(xb, w1[k].reshape(F1, 1, ks1, ks1), b1[k], padding=ks1 // 2).relu(), 2, 2)
        assert rel_err(a1[k], ref.reshape(B, F1, 196)) < tol
        ref = torch.nn.grad.conv2d_weight(xb, (F1, 1, ks1, ks1), dpre1[k], padding=ks1 // 2).reshape(F1, ks1 * ks1)
        assert rel_err(dw1[k], ref) < tol and rel_err(db1[k], dpre1[k].sum(dim=(0, 2, 3))) < tol
        x = x2[k].view(B, Cin, 14, 14)
        wk = w2[k].reshape(Cout, Cin, ks, ks)
        ref = F.max_pool2d(F.conv2d(x, wk, b2[k], padding=ks // 2).relu(), 2, 2).reshape(B, Cout, 49)
        assert rel_err(a2[k], ref) < tol
        assert rel_err(dw2[k], torch.nn.grad.conv2d_weight(x, wk.shape, dpre2[k], padding=ks // 2).reshape(Cout, -1)) < tol
        assert rel_err(db2[k], dpre2[k].sum(dim=(0, 2, 3))) < tol
        assert rel_err(da1[k], torch.nn.grad.conv2d_input(x.shape, wk, dpre2[k], padding=ks // 2).reshape(B, Cin, 196)) < tol


def test_conv_shapes_fuzz():
    """Property test over the launch geometry: client counts that are not multiples of the 8 XCDs, odd batch sizes
    (the forward kernels pair samples), batches smaller than the batch-split counts, both kernel sizes, every
    precision policy. One process, bounded examples; every launcher validates its shapes on the host first."""
    hypothesis = pytest.importorskip("hypothesis")
    from hypothesis import given, settings, strategies as st, HealthCheck

    @settings(max_examples=14, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
    @given(K=st.integers(1, 19), B=st.integers(1, 37), ks=st.sampled_from([3, 5]),
           precision=st.sampled_from(["fp32-mfma", "fp32", "bf16"]), F1=st.sampled_from([32]))
    def run(K, B, ks, precision, F1):
        _conv_fuzz_case(K, B, ks, precision, F1)

    run()
    for K, B, ks in ((9, 1, 5), (17, 3, 3), (1, 33, 5)):            # corners, always
        for precision in ("fp32-mfma", "fp32", "bf16"):
            _conv_fuzz_case(K, B, ks, precision, 32)


@pytest.mark.parametrize("ks", [5, 3])
def test_conv2_split_is_fp32_accurate(ks):
    """The three-term bf16 split (the stage-2 convolutions of the default "fp32" policy) against a float64 oracle: its error must be in the class of the fp32
    MFMA kernels' own rounding error (both accumulate in fp32), nowhere near bf16's."""
    h = hip()
    K, B, Cin, Cout = 2, 16, 32, 64
    a1 = torch.randn(K, B, Cin, 196, device=DEV).relu()
    w = slab_view(K, (Cout, Cin * ks * ks)) * 0.05
    b = slab_view(K, (Cout,)) * 0.1
    g = torch.randn(K, B, Cout, 49, device=DEV)
    err = {}
    for prec in ("fp32-mfma", "fp32", "bf16"):
        a2 = torch.empty(K, B, Cout, 49, device=DEV)
        code = torch.empty(K, B, Cout, 49, dtype=torch.uint8, device=DEV)
        dw = slab_view(K, (Cout, Cin * ks * ks))
        db = slab_view(K, (Cout,))
        da1 = torch.empty(K, B, Cin, 196, device=DEV)
        h.conv2_fwd(a1, w, b, a2, code, ks, prec=prec)
        h.conv2_wgrad(g, code, a1, dw, db, ks, prec=prec)
        h.conv2_dgrad(g, code, w, da1, ks, prec=prec)
        dpre = route(g, code, 7).double()
        e = [0.0, 0.0, 0.0, 0.0]
        for k in range(K):
            x = a1[k].view(B, Cin, 14, 14).double()
            wk = w[k].reshape(Cout, Cin, ks, ks).double()
            pre = F.conv2d(x, wk, b[k].double(), padding=ks // 2)
            ref = F.max_pool2d(pre.relu(), 2, 2).reshape(B, Cout, 49)
            # pooling picks may differ between precisions only on near-ties; compare the pooled values
            e[0] = max(e[0], rel_err(a2[k].double(), ref))
            e[1] = max(e[1], rel_err(dw[k].double(), torch.nn.grad.conv2d_weight(x, wk.shape, dpre[k], padding=ks // 2)
                                     .reshape(Cout, -1)))
            e[2] = max(e[2], rel_err(da1[k].double(), torch.nn.grad.conv2d_input(x.shape, wk, dpre[k], padding=ks // 2)
                                     .reshape(B, Cin, 196)))
            e[3] = max(e[3], rel_err(db[k].double(), dpre[k].sum(dim=(0, 2, 3))))
        err[prec] = e
    print("conv2 max rel err vs float64 [fwd, wgrad, dgrad, bias]:", err)
    for i in range(4):
        assert err["fp32"][i] < 4 * err["fp32-mfma"][i] + 2e-7, (i, err)
        assert err["fp32"][i] < 5e-6, (i, err)
    assert err["bf16"][0] > 100 * err["fp32"][0]


@pytest.mark.parametrize("C,B,nv", [(10, 32, 32), (62, 32, 28), (10, 100, 77), (62, 256, 200), (17, 40, 40), (64, 5, 3)])
def test_softmax_ce(C, B, nv):
    h = hip()
    K, n = 4, 200
    logits = torch.randn(K, B, C, device=DEV) * 3
    labels = torch.randint(0, C, (K, n), device=DEV).int()
    sidx = torch.stack([torch.randperm(n, device=DEV)[:B] for _ in range(K)]).int()
    dl = torch.empty_like(logits)
    loss = torch.zeros(K, device=DEV)
    corr = torch.zeros(K, dtype=torch.int32, device=DEV)
    cnt = torch.zeros(K, dtype=torch.int32, device=DEV)
    cm = torch.zeros(K, C, C, dtype=torch.int32, device=DEV)
    nvt = torch.full((K,), nv, dtype=torch.int32, device=DEV)
    h.softmax_ce(logits, labels, sidx, dl, loss, corr, cnt, cm, nvt)
    for k in range(K):
        y = labels[k][sidx[k].long()].long()[:nv]
        z = logits[k, :nv].clone().requires_grad_(True)
        ref = F.cross_entropy(z, y, reduction="sum")
        (ref / nv).backward()
        assert torch.allclose(loss[k], ref.detach(), rtol=1e-4, atol=1e-4)
        assert torch.allclose(dl[k, :nv], z.grad, atol=1e-6, rtol=1e-4)
        assert (dl[k, nv:] == 0).all()
        pred = logits[k, :nv].argmax(1)
        assert corr[k].item() == (pred == y).sum().item()
        assert cnt[k].item() == nv
        ref_cm = torch.zeros(C, C, dtype=torch.int32, device=DEV)
        ref_cm.view(-1).index_add_(0, y * C + pred, torch.ones(nv, dtype=torch.int32, device=DEV))
        assert (cm[k] == ref_cm).all()


@pytest.mark.parametrize("B,nv", [(32, 32), (32, 19), (100, 64)])
def test_sigmoid_bce_for_one_logit_models(B, nv):
    """C = 1: sigmoid + binary cross-entropy (the WADI MLP's loss, wadi/models/mlp.py: torch.sigmoid + BCELoss) with a
    2 x 2 confusion matrix; extreme logits stay finite (BCELoss clamps a sample's loss at 100)."""
    h = hip()
    K, n = 3, 150
    logits = torch.randn(K, B, 1, device=DEV) * 4
    logits[0, 0, 0], logits[0, 1, 0] = 300.0, -300.0
    labels = torch.randint(0, 2, (K, n), device=DEV).int()
    sidx = torch.stack([torch.randperm(n, device=DEV)[:B] for _ in range(K)]).int()
    labels[0, sidx[0, 0].long()], labels[0, sidx[0, 1].long()] = 0, 1          # both as wrong as a sample can be
    dl = torch.empty_like(logits)
    loss = torch.zeros(K, device=DEV)
    corr = torch.zeros(K, dtype=torch.int32, device=DEV)
    cnt = torch.zeros(K, dtype=torch.int32, device=DEV)
    cm = torch.zeros(K, 2, 2, dtype=torch.int32, device=DEV)
    h.softmax_ce(logits, labels, sidx, dl, loss, corr, cnt, cm, torch.full((K,), nv, dtype=torch.int32, device=DEV))
    for k in range(K):
        y = labels[k][sidx[k].long()].float()[:nv]
        z = logits[k, :nv, 0].clone().requires_grad_(True)
        ref = F.binary_cross_entropy(torch.sigmoid(z), y, reduction="sum")        # the reference's own formulation
        (F.binary_cross_entropy_with_logits(z, y, reduction="sum") / nv).backward()
        exact = F.binary_cross_entropy_with_logits(z.detach().double(), y.double(), reduction="none").clamp(max=100).sum()
        assert torch.isfinite(loss[k]) and torch.allclose(loss[k].double(), exact, rtol=1e-5)
        assert torch.allclose(loss[k], ref.detach(), rtol=1e-3)     # (sigmoid saturates in fp32: the reference form is the loose one)
        assert torch.allclose(dl[k, :nv, 0], z.grad, atol=1e-6, rtol=1e-4) and (dl[k, nv:] == 0).all()
        pred = (logits[k, :nv, 0] > 0).long()
        assert corr[k].item() == (pred == y.long()).sum().item() and cnt[k].item() == nv
        ref_cm = torch.zeros(4, dtype=torch.int32, device=DEV).index_add_(
            0, y.long() * 2 + pred, torch.ones(nv, dtype=torch.int32, device=DEV)).view(2, 2)
        assert (cm[k] == ref_cm).all()


@pytest.mark.parametrize("B,nv,D,gather", [(32, 32, 17, True), (32, 21, 17, True), (96, 96, 17, False), (8, 5, 40, True)])
def test_mse_against_the_input_batch(B, nv, D, gather):
    """Reconstruction loss of the auto-encoders (syscall/models/autoencoder.py: MSELoss(decoder output, input))."""
    h = hip()
    K, n = 3, 120
    data = torch.randn(K, n, D, device=DEV)
    out = torch.randn(K, B, D, device=DEV)
    sidx = torch.stack([torch.randperm(n, device=DEV)[:B] for _ in range(K)]).int() if gather else None
    dout = torch.full_like(out, 7.0)
    loss = torch.zeros(K, device=DEV)
    cnt = torch.zeros(K, dtype=torch.int32, device=DEV)
    h.mse_input(out, data, sidx, dout, loss, cnt, torch.full((K,), nv, dtype=torch.int32, device=DEV))
    h.mse_input(out, data, sidx, None, loss, cnt, torch.full((K,), nv, dtype=torch.int32, device=DEV))   # evaluation form
    for k in range(K):
        x = (data[k][sidx[k].long()] if gather else data[k, :B])[:nv]
        o = out[k, :nv].clone().requires_grad_(True)
        ref = F.mse_loss(o, x)
        ref.backward()
        assert torch.allclose(loss[k], 2 * ref.detach() * nv, rtol=1e-5) and cnt[k].item() == 2 * nv
        assert torch.allclose(dout[k, :nv], o.grad, atol=1e-7, rtol=1e-5) and (dout[k, nv:] == 0).all()


def test_adam_two_ranges_in_one_launch_match_torch_and_leave_the_gap_alone():
    h = hip()
    K, P = 5, 4096
    ranges = [(64, 1088), (3072, 4032)]
    p = torch.randn(K, P, device=DEV)
    m, v = torch.zeros(K, P, device=DEV), torch.zeros(K, P, device=DEV)
    p0 = p.clone()
    cols = torch.cat([torch.arange(a, b) for a, b in ranges]).to(DEV)
    ref_p = [torch.nn.Parameter(p[k, cols].clone()) for k in range(K)]
    opt = torch.optim.Adam(ref_p, lr=1e-3)
    bc = torch.zeros(2, device=DEV)
    for t in range(1, 6):
        g = torch.randn(K, P, device=DEV)
        if t % 2:
            h.adam_step_ranges(p, g, m, v, ranges, lr=1e-3, t=t)
        else:
            bc.copy_(torch.tensor([1 / (1 - 0.9 ** t), 1 / (1 - 0.999 ** t) ** 0.5]))
            h.adam_step_ranges(p, g, m, v, ranges, lr=1e-3, t=1, bc=bc)
        for k in range(K):
            ref_p[k].grad = g[k, cols].clone()
        opt.step()
    for k in range(K):
        assert torch.allclose(p[k, cols], ref_p[k].detach(), atol=1e-6, rtol=1e-5)
    rest = torch.ones(P, dtype=torch.bool, device=DEV)
    rest[cols] = False
    assert torch.equal(p[:, rest], p0[:, rest]) and not m[:, rest].any() and not v[:, rest].any()
    with pytest.raises(Exception):
        h.adam_step_ranges(p, g, m, v, [(64, 1088), (1000, 2000)], lr=1e-3, t=1)      # overlapping ranges
    # the same on a client slice of the slabs (what a client group of the engine passes)
    q = p.clone()
    h.adam_step_ranges(q[2:4], g[2:4], m[2:4], v[2:4], ranges, lr=1e-3, t=9)
    assert torch.equal(q[:2], p[:2]) and torch.equal(q[4:], p[4:]) and not torch.equal(q[2:4, cols], p[2:4, cols])


def test_adam_and_sgd_trajectories():
    h = hip()
    K, n = 3, 1000
    stride = 1024
    p = torch.randn(K, stride, device=DEV)[:, :n]
    m = torch.zeros(K, stride, device=DEV)[:, :n]
    v = torch.zeros(K, stride, device=DEV)[:, :n]
    ref_p = [torch.nn.Parameter(p[k].clone()) for k in range(K)]
    opt = torch.optim.Adam(ref_p, lr=1e-3)
    bc = torch.zeros(2, device=DEV)
    for t in range(1, 8):
        g = torch.randn(K, stride, device=DEV)[:, :n]
        if t % 2:
            h.adam_step(p, g, m, v, lr=1e-3, t=t)
        else:   # run-time bias corrections (graph-replay path)
            bc.copy_(torch.tensor([1 / (1 - 0.9 ** t), 1 / (1 - 0.999 ** t) ** 0.5]))
            h.adam_step(p, g, m, v, lr=1e-3, t=1, bc=bc)
        for k in range(K):
            ref_p[k].grad = g[k].clone()
        opt.step()
    for k in range(K):
        assert torch.allclose(p[k], ref_p[k].detach(), atol=1e-6, rtol=1e-5)

    p = torch.randn(K, stride, device=DEV)[:, :n]
    buf = torch.zeros(K, stride, device=DEV)[:, :n]
    ref_p = [torch.nn.Parameter(p[k].clone()) for k in range(K)]
    opt = torch.optim.SGD(ref_p, lr=1e-2, momentum=0.9)
    for t in range(5):
        g = torch.randn(K, stride, device=DEV)[:, :n]
        h.sgdm_step(p, g, buf, lr=1e-2, momentum=0.9)
        for k in range(K):
            ref_p[k].grad = g[k].clone()
        opt.step()
    for k in range(K):
        assert torch.allclose(p[k], ref_p[k].detach(), atol=1e-6, rtol=1e-5)


@pytest.mark.parametrize("n_local,n_remote,P", [(8, 0, 4096), (5, 3, 1000 * 4), (1000, 0, 256)])
def test_fedavg_groups(n_local, n_remote, P):
    from p2pfl_amd.runtime.aggregation import build_plan, fedavg_reference, run_plan
    local = torch.randn(n_local, P, device=DEV)
    remote = torch.randn(n_remote, P, device=DEV) if n_remote else None
    pool_n = n_local + n_remote
    gen = torch.Generator().manual_seed(0)
    groups = []
    # star-like fan-in over everything -> all rows; then a few random subsets
    groups.append((list(range(pool_n)), torch.randint(1, 600, (pool_n,), generator=gen).tolist(), [0]))
    for d in range(1, min(n_local, 6)):
        k = int(torch.randint(1, min(pool_n, 7) + 1, (1,), generator=gen))
        srcs = torch.randperm(pool_n, generator=gen)[:k].tolist()
        groups.append((srcs, torch.randint(1, 600, (k,), generator=gen).tolist(), [d]))
    plan = build_plan(groups)
    out = torch.zeros(n_local, P, device=DEV)
    ref = torch.zeros(n_local, P, device=DEV)
    run_plan(local, remote, out, plan)
    fedavg_reference(local, remote, ref, plan)
    assert torch.allclose(out, ref, atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("n_local,n_remote,P", [(5, 0, 1024), (12, 7, 4100), (100, 0, 50_000), (100, 100, 8192), (300, 40, 2052), (460, 40, 1024), (600, 0, 1024)])
def test_fedavg_inplace_matches_two_slab_reduce(n_local, n_remote, P):
    """In-place reduce (every source row read once into LDS, destinations overwritten) == the reference into a second
    slab; rows no group writes keep their content; more rows than the LDS pool holds -> declined, nothing touched."""
    from p2pfl_amd.runtime.aggregation import build_plan, fedavg_reference, inplace_supported, run_plan_inplace
    P = (P + 63) // 64 * 64
    gen = torch.Generator().manual_seed(1)
    local = torch.randn(n_local, P, generator=gen).to(DEV)
    remote = torch.randn(max(n_remote, 1) + 3, P, generator=gen).to(DEV)    # staging may be larger than what is used
    pool_n = n_local + n_remote
    groups = []
    dsts = torch.randperm(n_local, generator=gen).tolist()
    n_written = max(1, n_local - 2)                                         # the last two rows stay unwritten
    i = 0
    while i < n_written:
        k = int(torch.randint(1, min(pool_n, 7) + 1, (1,), generator=gen))
        nd = int(torch.randint(1, 4, (1,), generator=gen))
        srcs = torch.randperm(pool_n, generator=gen)[:k].tolist()
        groups.append((srcs, torch.randint(1, 600, (k,), generator=gen).tolist(), dsts[i:min(i + nd, n_written)]))
        i += nd
    plan = build_plan(groups)
    ref = local.clone()
    fedavg_reference(local, remote[:n_remote] if n_remote else None, ref, plan)
    before = local.clone()
    ok = run_plan_inplace(local, remote, plan, n_remote)
    assert ok == inplace_supported(before, n_remote)
    if pool_n * 32 * 4 > 64 * 1024:
        assert not ok and torch.equal(local, before)
        return
    assert ok
    assert torch.allclose(local, ref, atol=1e-5, rtol=1e-5)
    untouched = dsts[n_written:]
    assert torch.equal(local[untouched], before[untouched])


def test_fedavg_single_group_reduces_in_place_whatever_the_row_count():
    """A star / fully connected federation is ONE group: in place with the plain kernel, for 1000 rows too."""
    from p2pfl_amd.runtime.aggregation import build_plan, fedavg_reference, run_plan_inplace
    gen = torch.Generator().manual_seed(3)
    slab = torch.randn(1000, 512, generator=gen).to(DEV)
    w = torch.randint(1, 600, (1000,), generator=gen).tolist()
    plan = build_plan([(list(range(1000)), w, list(range(1, 1000)))])     # row 0 (say: a dead node) keeps its model
    ref = slab.clone()
    fedavg_reference(slab, None, ref, plan)
    row0 = slab[0].clone()
    assert run_plan_inplace(slab, None, plan)
    assert torch.allclose(slab, ref, atol=1e-5, rtol=1e-5) and torch.equal(slab[0], row0)


def test_aes128_ecb_known_answer_and_roundtrip():
    h = hip()
    from p2pfl_amd.utils.aes import expand_key, ecb_host
    key = bytes(range(16))
    pt = bytes.fromhex("00112233445566778899aabbccddeeff")
    expect = bytes.fromhex("69c4e0d86a7b0430d8cdb78070b4c55a")   # FIPS-197 C.1
    rk = torch.tensor(list(expand_key(key)), dtype=torch.uint8, device=DEV)
    src = torch.tensor(list(pt) * 64, dtype=torch.uint8, device=DEV)
    enc = torch.empty_like(src)
    h.aes128_ecb(src, enc, rk, decrypt=False)
    assert bytes(enc[:16].cpu().tolist()) == expect
    assert bytes(enc.cpu().tolist()) == ecb_host(key, bytes(src.cpu().tolist()), decrypt=False)
    big = torch.randint(0, 256, (1 << 20,), dtype=torch.uint8, device=DEV)
    e2 = torch.empty_like(big)
    d2 = torch.empty_like(big)
    h.aes128_ecb(big, e2, rk, decrypt=False)
    h.aes128_ecb(e2, d2, rk, decrypt=True)
    assert (d2 == big).all()
    assert bytes(e2[:4096].cpu().tolist()) == ecb_host(key, bytes(big[:4096].cpu().tolist()), decrypt=False)


@pytest.mark.parametrize("K,B", [(1, 32), (10, 32), (3, 17)])
def test_linear_split_reduction_for_small_shards(K, B):
    """Few clients x a long reduction (MNIST-CNN l1 for a 10-client shard: 160 forward / 250 data-gradient tiles on 256
    CUs): the reduction is split over several workgroups and summed in range order by a finishing pass that applies
    bias + ReLU / the ReLU mask. Same numbers as the unsplit kernels' oracle, bitwise reproducible."""
    from p2pfl_amd.ops import hip
    lib = hip.load()
    In, Out = 3136, 2048
    assert lib.p2pfl_linear_fwd_scratch(K, B, In, Out) > 0 and lib.p2pfl_linear_dgrad_scratch(K, B, In, Out) > 0
    assert lib.p2pfl_linear_fwd_scratch(100, B, In, Out) == 0      # (100 clients fill the chip: no split)
    torch.manual_seed(2)
    x = torch.randn(K, B, In, device="cuda")
    w = slab_view(K, (Out, In)) * 0.03
    b = slab_view(K, (Out,))
    y = torch.full((K, B, Out), float("nan"), device="cuda")
    hip.linear_fwd(x, w, b, y, True, prec="fp32")
    ref = (torch.einsum("kbi,koi->kbo", x.double(), w.double()) + b.double()[:, None]).relu()
    assert rel_err(y.double(), ref) < 2e-6
    y2 = torch.empty_like(y)
    hip.linear_fwd(x, w, b, y2, True, prec="fp32")
    assert torch.equal(y, y2)
    dy = torch.randn(K, B, Out, device="cuda")
    act = torch.randn(K, B, In, device="cuda")
    dx = torch.full((K, B, In), float("nan"), device="cuda")
    hip.linear_dgrad(dy, w, act, dx, prec="fp32")
    refd = torch.einsum("kbo,koi->kbi", dy.double(), w.double()) * (act > 0)
    assert rel_err(dx.double(), refd) < 2e-6 and ((dx == 0) | (act > 0)).all()
    dx2 = torch.empty_like(dx)
    hip.linear_dgrad(dy, w, act, dx2, prec="fp32")
    assert torch.equal(dx, dx2)
    hip.linear_dgrad(dy, w, None, dx2, prec="fp32")
    assert rel_err(dx2.double(), torch.einsum("kbo,koi->kbi", dy.double(), w.double())) < 2e-6


@pytest.mark.parametrize("deep", [True, False])
@pytest.mark.parametrize("In,Out", [(1024, 96), (1100, 70)])
def test_linear_long_reduction_variants(In, Out, deep, monkeypatch, prec):
    """In / Out >= 1024 dispatch to the 64-deep k-step kernels, or to the 128-deep ones when the grid has at
    most one block per CU (as here; P2PFL_NO_BK128 turns that off) - incl. a ragged last step."""
    from p2pfl_amd.ops import hip
    if not deep:
        monkeypatch.setenv("P2PFL_NO_BK128", "1")
    torch.manual_seed(1)
    K, B = 2, 32
    x = torch.randn(K, B, In, device="cuda")
    w = torch.randn(K, Out, In, device="cuda") * 0.05
    b = torch.randn(K, Out, device="cuda")
    y = torch.empty(K, B, Out, device="cuda")
    hip.linear_fwd(x, w, b, y, False)
    ref = torch.einsum("kbi,koi->kbo", x, w) + b[:, None]
    assert ((y - ref).norm() / ref.norm()).item() < (6e-3 if prec == "bf16" else 1e-5)
    # data gradient with the roles of In / Out swapped: the reduction runs over `Out2 = In`
    dy = torch.randn(K, B, In, device="cuda")
    w2 = torch.randn(K, In, Out, device="cuda") * 0.05
    dx = torch.empty(K, B, Out, device="cuda")
    hip.linear_dgrad(dy, w2, None, dx)
    refd = torch.einsum("kbo,koi->kbi", dy, w2)
    assert ((dx - refd).norm() / refd.norm()).item() < (6e-3 if prec == "bf16" else 1e-5)


@pytest.mark.parametrize("K,B,In,Out,wd", [(2, 32, 3136, 300, 0.0), (3, 20, 130, 70, 1e-2), (1, 32, 784, 256, 0.0)])
def test_linear_wgrad_adam_against_torch_optim(K, B, In, Out, wd):
    """fp32 fused weight gradient + Adam, three steps, directly against torch.optim.Adam fed with dy.T @ x
    (not against this repo's own unfused kernels)."""
    h = hip()
    torch.manual_seed(5)
    n = In * Out
    P = n + 64 - n % 64 + 512
    slabs = [torch.randn(K, P, device=DEV) * 0.05, torch.zeros(K, P, device=DEV), torch.zeros(K, P, device=DEV)]
    view = lambda s: (s[:, :n].unflatten(1, (Out, In)), s[:, P - 512:P - 512 + Out])
    (w, b), (mw, mb), (vw, vb) = view(slabs[0]), view(slabs[1]), view(slabs[2])
    ref_w = [torch.nn.Parameter(w[k].clone()) for k in range(K)]
    ref_b = [torch.nn.Parameter(b[k].clone()) for k in range(K)]
    opt = torch.optim.Adam(ref_w + ref_b, lr=1e-3, weight_decay=wd)
    for t in range(1, 4):
        dy = torch.randn(K, B, Out, device=DEV) * 0.1
        x = torch.randn(K, B, In, device=DEV)
        h.linear_wgrad_adam(dy, x, w, mw, vw, b, mb, vb, lr=1e-3, weight_decay=wd, t=t, fresh=(t == 1), prec="fp32-mfma")
        for k in range(K):
            ref_w[k].grad = dy[k].T @ x[k]
            ref_b[k].grad = dy[k].sum(0)
        opt.step()
    for k in range(K):
        # An Adam step is lr * m / (sqrt(v) + eps): gradient round-off (1e-7 relative to a typical gradient) moves it
        # by ~1e-7 * lr - except for the one-in-a-million weight whose gradient is itself round-off-sized, where the
        # normalisation turns that into a step of up to lr per step. Everything else must agree to 2e-6.
        for got, want in ((w[k], ref_w[k].detach()), (b[k], ref_b[k].detach())):
            diff = (got - want).abs()
            assert (diff > 2e-6 + 1e-5 * want.abs()).float().mean().item() < 1e-5
            assert diff.max().item() <= 3 * 2e-3


@pytest.mark.parametrize("precision", ["fp32-mfma", "fp32"])
@pytest.mark.parametrize("K,B,In,Out,wd", [(3, 32, 3136, 2048, 0.0), (2, 20, 132, 70, 1e-2), (1, 32, 784, 300, 0.0),
                                            (11, 7, 3136, 256, 0.0)])
def test_linear_wgrad_adam_dgrad_fused(K, B, In, Out, wd, precision, monkeypatch):
    """The fused update that also emits the data gradient: weights / moments BITWISE equal to the plain fused update
    (same products, same order), dx = dy @ w_before against torch at fp32 tolerance; three steps (fresh first)."""
    h = hip()
    # few clients: the launcher declines (too few column strips to fill the chip) unless forced
    dx0 = torch.empty(K, B, In, device=DEV)
    z = lambda *s: torch.zeros(*s, device=DEV)
    assert not h.linear_wgrad_adam_dgrad(z(K, B, Out), z(K, B, In), z(K, Out, In), z(K, Out, In), z(K, Out, In), z(K, Out),
                                         z(K, Out), z(K, Out), dx0, lr=1e-3, t=1, prec=precision)
    monkeypatch.setenv("P2PFL_FUSE_DGRAD_FORCE", "1")
    torch.manual_seed(7)
    n = In * Out
    P = n + 64 - n % 64 + 4096
    base = [torch.randn(K, P, device=DEV) * 0.05, torch.zeros(K, P, device=DEV), torch.zeros(K, P, device=DEV)]
    slabs_a = [t.clone() for t in base]
    slabs_b = [t.clone() for t in base]
    view = lambda s: (s[:, :n].unflatten(1, (Out, In)), s[:, P - 4096:P - 4096 + Out])
    A = [view(t) for t in slabs_a]
    Bv = [view(t) for t in slabs_b]
    n_ids = 40
    for t in range(1, 4):
        dy = torch.randn(K, B, Out, device=DEV) * 0.1
        x = torch.randn(K, B, In, device=DEV)
        dx = torch.full((K, B, In), float("nan"), device=DEV)
        w_before = A[0][0].clone()
        took = h.linear_wgrad_adam_dgrad(dy, x, A[0][0], A[1][0], A[2][0], A[0][1], A[1][1], A[2][1], dx, lr=1e-3,
                                         weight_decay=wd, t=t, fresh=(t == 1), prec=precision)
        assert took
        h.linear_wgrad_adam(dy, x, Bv[0][0], Bv[1][0], Bv[2][0], Bv[0][1], Bv[1][1], Bv[2][1], lr=1e-3, weight_decay=wd,
                            t=t, fresh=(t == 1), prec=precision)
        for sa, sb in zip(slabs_a, slabs_b):
            assert torch.equal(sa, sb)
        ref = torch.einsum("kbo,koi->kbi", dy, w_before)
        assert torch.isfinite(dx).all()
        assert rel_err(dx, ref) < TOL_DENSE[precision]
    # bf16 and unsupported shapes decline (the caller then runs the two kernels)
    dx = torch.empty(K, B, In, device=DEV)
    assert not h.linear_wgrad_adam_dgrad(dy, x, A[0][0], A[1][0], A[2][0], A[0][1], A[1][1], A[2][1], dx, lr=1e-3, t=4,
                                         prec="bf16")


@pytest.mark.parametrize("K,B,In,Out,wd", [(2, 32, 3136, 62, 0.0), (3, 20, 130, 70, 1e-3)])
def test_linear_wgrad_sgdm_against_torch_optim(K, B, In, Out, wd):
    """fp32 fused weight gradient + SGD-momentum directly against torch.optim.SGD on dy.T @ x."""
    h = hip()
    torch.manual_seed(6)
    n = In * Out
    P = n + 64 - n % 64 + 512
    slabs = [torch.randn(K, P, device=DEV) * 0.05, torch.zeros(K, P, device=DEV)]
    view = lambda s: (s[:, :n].unflatten(1, (Out, In)), s[:, P - 512:P - 512 + Out])
    (w, b), (mw, mb) = view(slabs[0]), view(slabs[1])
    ref_w = [torch.nn.Parameter(w[k].clone()) for k in range(K)]
    ref_b = [torch.nn.Parameter(b[k].clone()) for k in range(K)]
    opt = torch.optim.SGD(ref_w + ref_b, lr=1e-2, momentum=0.9, weight_decay=wd)
    for t in range(1, 4):
        dy = torch.randn(K, B, Out, device=DEV) * 0.1
        x = torch.randn(K, B, In, device=DEV)
        h.linear_wgrad_sgdm(dy, x, w, mw, b, mb, lr=1e-2, momentum=0.9, weight_decay=wd, fresh=(t == 1), prec="fp32-mfma")
        for k in range(K):
            ref_w[k].grad = dy[k].T @ x[k]
            ref_b[k].grad = dy[k].sum(0)
        opt.step()
    for k in range(K):
        assert torch.allclose(w[k], ref_w[k].detach(), atol=1e-6, rtol=1e-5)
        assert torch.allclose(b[k], ref_b[k].detach(), atol=1e-6, rtol=1e-5)


# ------------------------------------------------------------------------------------------------------------------
# round 3: optimizer steps folded into the gradient-producing kernels, the one-kernel classification head
# ------------------------------------------------------------------------------------------------------------------
def _slabs(K, P, n=3):
    """n [K, P + pad] slabs with a common row stride (parameters, first / second moment)."""
    stride = -(-P // 64) * 64 + 64
    return [torch.randn(K, stride, device=DEV) * 0.1 for _ in range(n)]


@pytest.mark.parametrize("K,B,nv,In,C,mask", [(3, 32, 32, 2048, 10, True), (2, 32, 12, 128, 10, True),
                                              (2, 20, 20, 3136, 16, False), (5, 7, 5, 260, 3, True),
                                              (4, 32, 12, 2048, 10, True), (2, 9, 4, 1100, 5, False)])
@pytest.mark.parametrize("kind", [None, "adam", "sgd"])
@pytest.mark.parametrize("form", ["one-kernel", "sliced"])
def test_linear_head_matches_torch(K, B, nv, In, C, mask, kind, form):
    """csrc/head.hip against autograd: logits, loss / accuracy / confusion counters, data gradient with the ReLU mask,
    and either the parameter gradients or one step of torch.optim on (W, b) - the one-kernel form and the two-kernel
    form over column slices (wide heads, with the caller's scratch)."""
    h = hip()
    assert h.linear_head_accepts(B, In, C)
    scratch = h.linear_head_scratch(K, B, In, C, DEV) if form == "sliced" else None
    assert (scratch is not None) == (form == "sliced" and In >= 1024)
    if form == "sliced" and scratch is None:
        return          # narrow heads have the one form only (covered by the other parameter)
    n = 50
    P = C * In
    off_b = -(-P // 64) * 64
    p, m, v = _slabs(K, off_b + 64)
    m.abs_(); v.abs_()
    W = lambda s: s[:, :P].unflatten(1, (C, In))
    Bv = lambda s: s[:, off_b:off_b + C]
    x = torch.randn(K, B, In, device=DEV)
    if mask:
        x = x.relu()
    labels = torch.randint(0, C, (K, n), device=DEV).int()
    sidx = torch.stack([torch.randperm(n, device=DEV)[:B] for _ in range(K)]).int()
    nvt = torch.full((K,), nv, dtype=torch.int32, device=DEV)
    loss_sum = torch.zeros(K, device=DEV)
    correct = torch.zeros(K, dtype=torch.int32, device=DEV)
    count = torch.zeros(K, dtype=torch.int32, device=DEV)
    conf = torch.zeros(K, C, C, dtype=torch.int32, device=DEV)
    dx = torch.full((K, B, In), 7.0, device=DEV)
    logits = torch.zeros(K, B, C, device=DEV)
    p0, m0, v0 = p.clone(), m.clone(), v.clone()
    hyper = dict(lr=1e-2, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-3)
    t = 3
    if kind is None:
        g = torch.zeros_like(p)
        h.linear_head(x, W(p), Bv(p), labels, sidx, loss_sum, correct, count, conf, nvt, dx=dx, dw=W(g), db=Bv(g),
                      logits=logits, relu_mask=mask, scratch=scratch)
    else:
        opt = h.fold_opt(kind, W(p), W(m), W(v), Bv(p), Bv(m), Bv(v), momentum=0.9, t=t, **hyper)
        h.linear_head(x, W(p), Bv(p), labels, sidx, loss_sum, correct, count, conf, nvt, dx=dx, logits=logits,
                      relu_mask=mask, opt=opt, scratch=scratch)
    torch.cuda.synchronize()
    for k in range(K):
        w_ref = W(p0)[k].clone().requires_grad_(True)
        b_ref = Bv(p0)[k].clone().requires_grad_(True)
        x_ref = x[k].clone().requires_grad_(True)
        z = x_ref @ w_ref.T + b_ref
        y = labels[k][sidx[k].long()].long()
        loss = F.cross_entropy(z[:nv], y[:nv], reduction="sum")
        (loss / nv).backward()
        assert rel_err(logits[k][:nv], z[:nv].detach()) < 2e-5
        assert abs(loss_sum[k].item() - loss.item()) < 1e-4 * max(1.0, abs(loss.item()))
        pred = z[:nv].argmax(1)
        assert correct[k].item() == int((pred == y[:nv]).sum()) and count[k].item() == nv
        ref_conf = torch.zeros(C, C, dtype=torch.int32, device=DEV)
        ref_conf.view(-1).index_add_(0, y[:nv] * C + pred, torch.ones(nv, dtype=torch.int32, device=DEV))
        assert torch.equal(conf[k], ref_conf)
        gx = x_ref.grad * (x[k] > 0) if mask else x_ref.grad
        assert rel_err(dx[k], gx) < 2e-5
        if kind is None:
            assert rel_err(W(g)[k], w_ref.grad) < 2e-5 and rel_err(Bv(g)[k], b_ref.grad) < 2e-5
            assert torch.equal(p, p0)
            continue
        # one step of the real optimizer from the same state
        pw, pb = W(p0)[k].clone().requires_grad_(True), Bv(p0)[k].clone().requires_grad_(True)
        pw.grad, pb.grad = w_ref.grad.clone(), b_ref.grad.clone()
        if kind == "adam":
            o = torch.optim.Adam([pw, pb], lr=hyper["lr"], betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-3)
            for q, mm, vv in ((pw, W(m0)[k], W(v0)[k]), (pb, Bv(m0)[k], Bv(v0)[k])):
                o.state[q] = {"step": torch.tensor(float(t - 1)), "exp_avg": mm.clone(), "exp_avg_sq": vv.clone()}
        else:
            o = torch.optim.SGD([pw, pb], lr=hyper["lr"], momentum=0.9, weight_decay=1e-3)
            for q, mm in ((pw, W(m0)[k]), (pb, Bv(m0)[k])):
                o.state[q] = {"momentum_buffer": mm.clone()}
        o.step()
        # (+ 2e-4 of the step itself: where sqrt(v) is tiny a gradient that is right to 2e-5 moves the parameter visibly)
        for ours, ref, start in ((W(p)[k], pw.detach(), W(p0)[k]), (Bv(p)[k], pb.detach(), Bv(p0)[k])):
            assert ((ours - ref).abs() <= 1e-6 + 1e-5 * ref.abs() + 2e-4 * (ref - start).abs()).all()
        st = o.state[pw]
        assert rel_err(W(m)[k], st["exp_avg" if kind == "adam" else "momentum_buffer"]) < 2e-5
        if kind == "adam":
            assert rel_err(W(v)[k], st["exp_avg_sq"]) < 2e-5
    # nothing outside the layer's columns moved
    assert torch.equal(p[:, P:off_b], p0[:, P:off_b]) and torch.equal(p[:, off_b + C:], p0[:, off_b + C:])


def test_sliced_head_bits_do_not_depend_on_the_shard_size():
    """The column slices of the two-kernel head are a function of the layer width only: clients 0..2 of a 7-client launch
    end with the very bits a 3-client launch gives them (client groups of one round run shards of different sizes)."""
    h = hip()
    B, In, C, n = 32, 2048, 10, 40
    P = C * In
    off_b = -(-P // 64) * 64
    torch.manual_seed(5)
    base = _slabs(7, off_b + 64)
    base[1].abs_(); base[2].abs_()
    x = torch.randn(7, B, In, device=DEV).relu()
    labels = torch.randint(0, C, (7, n), device=DEV).int()
    W = lambda s: s[:, :P].unflatten(1, (C, In))
    Bv = lambda s: s[:, off_b:off_b + C]
    out = {}
    for K in (7, 3):
        p, m, v = (t[:K].clone() for t in base)
        dx = torch.zeros(K, B, In, device=DEV)
        stats = (torch.zeros(K, device=DEV), torch.zeros(K, dtype=torch.int32, device=DEV),
                 torch.zeros(K, dtype=torch.int32, device=DEV))
        opt = h.fold_opt("adam", W(p), W(m), W(v), Bv(p), Bv(m), Bv(v), lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8,
                         weight_decay=0.0, momentum=0.0, t=2)
        h.linear_head(x[:K], W(p), Bv(p), labels[:K], None, *stats, None, None, dx=dx, relu_mask=True, opt=opt,
                      scratch=h.linear_head_scratch(K, B, In, C, DEV))
        torch.cuda.synchronize()
        out[K] = (p, m, v, dx, stats[0])
    for a, b in zip(out[7], out[3]):
        assert torch.equal(a[:3], b)


@pytest.mark.parametrize("kind", ["adam", "sgd"])
@pytest.mark.parametrize("ks,B,K", [(5, 32, 3), (7, 9, 2), (5, 32, 100)])
def test_conv1_wgrad_folded_optimizer_matches_unfused(ks, B, K, kind, prec):
    """The finishing pass of the stage-1 weight gradient applies the optimizer step itself: identical (bitwise) to the
    gradient store followed by the element-wise optimizer kernel, and the gradient pass is bitwise reproducible."""
    h = hip()
    Fm, n = 32, 40
    T = ks * ks
    off_b = -(-Fm * T // 64) * 64
    p, m, v = _slabs(K, off_b + 64)
    v.abs_()
    W = lambda s: s[:, :Fm * T].unflatten(1, (Fm, T))
    Bv = lambda s: s[:, off_b:off_b + Fm]
    data = torch.rand(K, n, 784, device=DEV)
    sidx = torch.stack([torch.randperm(n, device=DEV)[:B] for _ in range(K)]).int()
    g = torch.randn(K, B, Fm, 196, device=DEV)
    code = torch.randint(0, 5, (K, B, Fm, 196), device=DEV).to(torch.uint8)
    grads = torch.zeros_like(p)
    h.conv1_wgrad(g, code, data, sidx, W(grads), Bv(grads), ks)
    again = torch.zeros_like(p)
    h.conv1_wgrad(g, code, data, sidx, W(again), Bv(again), ks)
    assert torch.equal(grads, again)                         # fixed-order reduction: no atomics any more
    p1, m1, v1 = p.clone(), m.clone(), v.clone()
    hyper = dict(lr=1e-2, weight_decay=1e-3)
    if kind == "adam":
        h.adam_step(p1, grads, m1, v1, t=4, **hyper)
    else:
        h.sgdm_step(p1, grads, m1, momentum=0.9, **hyper)
    opt = h.fold_opt(kind, W(p), W(m), W(v), Bv(p), Bv(m), Bv(v), momentum=0.9, t=4, **hyper)
    h.conv1_wgrad(g, code, data, sidx, None, None, ks, opt=opt)
    cols = torch.zeros(p.shape[1], dtype=torch.bool, device=DEV)
    cols[:Fm * T] = True
    cols[off_b:off_b + Fm] = True
    # (same arithmetic; the compiler contracts multiply-adds differently in the two kernels: last-bit differences)
    assert rel_err(p[:, cols], p1[:, cols]) < 2e-6 and rel_err(m[:, cols], m1[:, cols]) < 2e-6
    if kind == "adam":
        assert rel_err(v[:, cols], v1[:, cols]) < 2e-6


@pytest.mark.parametrize("kind", ["adam", "sgd"])
@pytest.mark.parametrize("ks,B,K", [(5, 32, 3), (3, 7, 9), (5, 4, 60), (5, 32, 100)])
def test_conv2_wgrad_folded_optimizer_matches_unfused(ks, B, K, kind, prec):
    h = hip()
    Cin, Cout = 32, 64
    T = ks * ks
    P = Cout * Cin * T
    off_b = -(-P // 64) * 64
    p, m, v = _slabs(K, off_b + 64)
    v.abs_()
    W = lambda s: s[:, :P].unflatten(1, (Cout, Cin * T))
    Bv = lambda s: s[:, off_b:off_b + Cout]
    a1 = torch.randn(K, B, Cin, 196, device=DEV).relu()
    g = torch.randn(K, B, Cout, 49, device=DEV)
    code = torch.randint(0, 5, (K, B, Cout, 49), device=DEV).to(torch.uint8)
    grads = torch.zeros_like(p)
    h.conv2_wgrad(g, code, a1, W(grads), Bv(grads), ks)
    p1, m1, v1 = p.clone(), m.clone(), v.clone()
    p_before = p.clone()
    hyper = dict(lr=1e-2, weight_decay=1e-3)
    if kind == "adam":
        h.adam_step(p1, grads, m1, v1, t=2, **hyper)
    else:
        h.sgdm_step(p1, grads, m1, momentum=0.9, **hyper)
    opt = h.fold_opt(kind, W(p), W(m), W(v), Bv(p), Bv(m), Bv(v), momentum=0.9, t=2, **hyper)
    h.conv2_wgrad(g, code, a1, None, None, ks, opt=opt)
    cols = torch.zeros(p.shape[1], dtype=torch.bool, device=DEV)
    cols[:P] = True
    cols[off_b:off_b + Cout] = True
    # (same arithmetic; the compiler contracts multiply-adds differently in the two kernels: last-bit differences)
    assert rel_err(p[:, cols], p1[:, cols]) < 2e-6 and rel_err(m[:, cols], m1[:, cols]) < 2e-6
    if kind == "adam":
        assert rel_err(v[:, cols], v1[:, cols]) < 2e-6
    assert torch.equal(p[:, ~cols], p_before[:, ~cols])          # nothing outside the layer's columns moved
