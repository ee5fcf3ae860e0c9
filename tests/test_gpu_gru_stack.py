"""The layer wavefront over a ConvGRU stack (dvd_convgru_stack_*, gru.hip; functional.ConvGRUStack) against the layer-by-layer
path (dvd_convgru_layer_*, functional.ConvGRULayer) it replaces in bf16 mode (Module/ConvGRU.py:57-133, Generator.py:87-97):

  * with the per-layer path's split-K factors (`layer_policy` = 1) the two paths execute the same arithmetic per output element
    -- per-step instead of batched x-part / backward-data convolutions, grouped instead of single launches, other tile shapes --
    so every state of every layer and the input gradient must be BIT-EQUAL, the weight gradients equal up to the fp32 atomics of
    the narrow test layers;
  * with the production policy (split-K factors chosen per grouped launch) only fp32 summation order changes.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"

CASES = [   # T, B, S, cin, hidden sizes, kernel sizes, shared input (first ConvGRU of the generator), supplied initial states, split-K cap
    (5, 8, 4, 16, [16, 32, 16], [3, 5, 3], False, None, 1),
    (5, 8, 4, 32, [64, 64, 64], [3, 5, 3], True, (True, False, True), 2),
    (5, 4, 8, 16, [16, 32, 16], [3, 5, 3], False, None, 1),
    (4, 4, 8, 32, [64, 128, 64], [3, 5, 3], False, (False, True, True), 2),
    (4, 2, 16, 16, [16, 32, 16], [3, 5, 5], False, None, 1),
    (3, 3, 16, 16, [64, 64, 64], [3, 5, 3], False, (True, True, True), 2),
    (3, 2, 32, 8, [8, 16, 8], [3, 5, 5], False, None, 1),
    (3, 1, 32, 8, [16, 8], [5, 3], False, None, 1),            # two layers
    (2, 2, 16, 8, [16], [5], False, None, 1),                  # one layer: nothing to overlap, same entry point
]


def _build(case):
    from dvd_gan_amd.gen_net import ConvGRU
    T, B, S, cin, hids, ks, shared, h0, _ = case
    torch.manual_seed(5)
    gru = ConvGRU(cin, hids, ks, len(hids))
    for p in gru.parameters():
        if p.dim() == 1:
            p.data.normal_(0, 0.1)
    x = torch.randn((B if shared else T * B), cin, S, S)
    gys = [torch.randn(T * B, h, S, S) for h in hids]
    h0s = None if h0 is None else [torch.randn(B, h, S, S) * 0.5 if on else None for h, on in zip(hids, h0)]
    return gru.to(DEV), x, gys, h0s


def _run(gru, case, x, gys, h0s, all_layers):
    from dvd_gan_amd import functional as Fn
    T, B, S, cin, hids, ks, shared = case[:7]
    for p in gru.parameters():
        p.grad = None
    xg = x.to(DEV).requires_grad_(True)
    hg = None if h0s is None else [None if h is None else h.to(DEV).requires_grad_(True) for h in h0s]
    hcl = None if hg is None else [None if h is None else Fn.ToChannelsLast.apply(h, torch.bfloat16, None) for h in hg]
    outs = gru.run(Fn.ToChannelsLast.apply(xg, torch.bfloat16, None), T, shared, hcl)
    ys = [Fn.FromChannelsLast.apply(o, h, None) for o, h in zip(outs, hids)]
    loss = sum((y * gy.to(DEV)).sum() for y, gy in zip(ys, gys)) if all_layers else (ys[-1] * gys[-1].to(DEV)).sum()
    loss.backward()
    torch.cuda.synchronize()
    res = {"ys": [y.detach().clone() for y in ys], "dx": xg.grad.clone(),
           "dh0": [] if hg is None else [h.grad.clone() for h in hg if h is not None],
           "dw": [p.grad.clone() for p in gru.parameters()]}
    return res


def _rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"T{c[0]}B{c[1]}S{c[2]}h{'-'.join(map(str, c[4]))}{'s' if c[6] else ''}{'i' if c[7] else ''}")
def test_wavefront_equals_layer_by_layer_bitwise(case, monkeypatch):
    from dvd_gan_amd import functional as Fn
    from dvd_gan_amd import lib as L
    gru, x, gys, h0s = _build(case)
    monkeypatch.setattr(Fn, "GRU_COMBINE_MAX", 8)               # layer path: every split factor combined in-launch, like the stack's
    monkeypatch.setattr(Fn, "GRU_STACK_LAYER_POLICY", 1)
    monkeypatch.setattr(Fn, "GRU_NS_CAP", case[8])               # ... and the same factors (at most the channel chunks of these narrow layers)
    monkeypatch.setattr(Fn, "GRU_STACK", True)
    assert Fn.ConvGRUStack.usable(torch.empty(1, case[2], case[2], 8, dtype=torch.bfloat16), gru.cells)
    new = _run(gru, case, x, gys, h0s, all_layers=False)
    assert int(L.gru_tickets(torch.device(DEV, torch.cuda.current_device())).abs().sum()) == 0
    again = _run(gru, case, x, gys, h0s, all_layers=False)
    monkeypatch.setattr(Fn, "GRU_STACK", False)
    old = _run(gru, case, x, gys, h0s, all_layers=False)
    for l, (a, b, c) in enumerate(zip(new["ys"], old["ys"], again["ys"])):
        assert torch.equal(a, b), f"layer {l}: states differ (rel {_rel(a, b):.3e})"
        assert torch.equal(a, c), f"layer {l}: not reproducible"
    assert torch.equal(new["dx"], old["dx"]), f"dx differs (rel {_rel(new['dx'], old['dx']):.3e})"
    for a, b in zip(new["dh0"], old["dh0"]):
        assert torch.equal(a, b)
    for (k, _), a, b in zip(gru.named_parameters(), new["dw"], old["dw"]):
        assert _rel(a, b) < 1e-5, (k, _rel(a, b))               # same kernels on bit-equal operands; fp32 atomics in the narrow layers


WIDE = [    # layers wide enough for the production split-K policy to split members of a grouped launch (x-part convolutions included)
    (3, 2, 16, 32, [64, 128, 64], [3, 5, 5], False, None, 0),
    (3, 1, 32, 32, [64, 128, 64], [3, 5, 5], False, (True, False, True), 0),
    (3, 4, 8, 64, [128, 256, 128], [3, 5, 3], False, None, 0),
    (3, 8, 4, 64, [128, 256, 128], [3, 5, 3], True, None, 0),
]


@pytest.mark.parametrize("case", CASES[:7:2] + WIDE, ids=lambda c: f"T{c[0]}B{c[1]}S{c[2]}h{c[4][0]}")
def test_wavefront_production_policy_and_outer_gradients(case, monkeypatch):
    """Split-K factors chosen per grouped launch, and a loss on EVERY layer's states (the gradient from outside the stack rides
    in the epilogue of the x-part backward-data convolution, in fp32, where autograd adds two bf16 tensors)."""
    from dvd_gan_amd import functional as Fn
    gru, x, gys, h0s = _build(case)
    monkeypatch.setattr(Fn, "GRU_STACK", True)
    new = _run(gru, case, x, gys, h0s, all_layers=True)
    monkeypatch.setattr(Fn, "GRU_STACK", False)
    old = _run(gru, case, x, gys, h0s, all_layers=True)
    for a, b in zip(new["ys"], old["ys"]):
        assert _rel(a, b) < 4e-3
    assert _rel(new["dx"], old["dx"]) < 1e-2
    for a, b in zip(new["dh0"], old["dh0"]):
        assert _rel(a, b) < 1e-2
    for (k, _), a, b in zip(gru.named_parameters(), new["dw"], old["dw"]):
        assert _rel(a, b) < 1e-2, (k, _rel(a, b))


def test_wavefront_inference_form(monkeypatch):
    """Under torch.no_grad() (the sampling path, trainer.py:323-334) nothing is kept for a backward pass: same states."""
    from dvd_gan_amd import functional as Fn
    case = CASES[4]
    gru, x, gys, h0s = _build(case)
    T, B, S, cin, hids, ks, shared = case[:7]
    monkeypatch.setattr(Fn, "GRU_STACK", True)
    xc = Fn.ToChannelsLast.apply(x.to(DEV), torch.bfloat16, None)
    a = gru.run(xc, T, shared)
    with torch.no_grad():
        b = gru.run(xc, T, shared)
    for u, v in zip(a, b):
        assert torch.equal(u, v)


# ------------------------------------------------------------------ the schedule the benchmark ships (B = 64, real widths)
# Generator.py:43-55 at ch = 32: three ConvGRUs of 256 / 512 / 256 channels (3, 5, 3 taps) on 4, 8 and 16 pixel frames, one of
# 128 / 256 / 128 (3, 5, 5 taps) on 32 pixel frames; the first one reads one shared input for every step (Generator.py:87-97).
# plan_splits picks the per-member split-K factors from the tile counts, i.e. from B: these are the plans bench.py executes.
PROD = [   # T, B, S, cin, hidden sizes, kernel sizes, shared input, supplied initial states, split-K cap (0 = production policy)
    (6, 64, 4, 256, [256, 512, 256], [3, 5, 3], True, None, 0),
    (48, 64, 4, 256, [256, 512, 256], [3, 5, 3], True, None, 0),
    (6, 64, 8, 256, [256, 512, 256], [3, 5, 3], False, None, 0),
    (48, 64, 8, 256, [256, 512, 256], [3, 5, 3], False, None, 0),
    (6, 64, 16, 256, [256, 512, 256], [3, 5, 3], False, None, 0),
    (6, 64, 32, 128, [128, 256, 128], [3, 5, 5], False, None, 0),
    (6, 64, 8, 256, [256, 512, 256], [3, 5, 3], False, (True, True, True), 0),      # configs[4]: states carried in
]


def _stack_ws():
    import ctypes as C
    from dvd_gan_amd import lib as L
    out = (C.c_longlong * 2)()
    L.lib().dvd_debug_stack_ws(out, 1)
    return int(out[0]), int(out[1])


@pytest.mark.parametrize("case", PROD, ids=lambda c: f"T{c[0]}B{c[1]}S{c[2]}h{c[4][0]}{'s' if c[6] else ''}{'i' if c[7] else ''}")
def test_production_schedule_at_benchmark_size(case, monkeypatch):
    """The four generator ConvGRUs at their real widths and B = 64:
      * the grouped launches with the per-layer path's split-K factors are BIT-EQUAL to the layer-by-layer path at this size too
        (states, input gradient, initial-state gradients; weight gradients of these wide layers have no atomics: equal as well);
      * the production plans (plan_splits: factors chosen from the tile counts, i.e. from B) against the layer-by-layer path at the
        WIDE cases' bounds -- where 48 steps of bf16 BPTT amplify a change of fp32 summation order beyond them (measured: the layer
        path against ITSELF with other split-K factors moves `cells.2.update_gate.weight` by 2.4e-2 at T = 48, S = 4), the bound
        of a tensor is 1.5 x that yardstick;
      * every split-K ticket is back at zero, and the slab workspace the sizing query asked for is EXACTLY the largest slab cursor
        a launched group used (the round-5 slab overrun faulted only in the full step)."""
    from dvd_gan_amd import functional as Fn
    from dvd_gan_amd import lib as L
    gru, x, gys, h0s = _build(case)
    dev = torch.device(DEV, torch.cuda.current_device())
    monkeypatch.setattr(Fn, "GRU_STACK", True)
    L.reset_gru_tickets()
    _stack_ws()
    new = _run(gru, case, x, gys, h0s, all_layers=False)
    sized, used = _stack_ws()
    assert int(L.gru_tickets(dev).abs().sum()) == 0
    assert sized > 0, "the production plan of a benchmark-sized stack splits no member at all?"
    assert used == sized, (used, sized)
    # the same grouped kernels with the layer path's factors: bit-equal
    monkeypatch.setattr(Fn, "GRU_COMBINE_MAX", 8)
    monkeypatch.setattr(Fn, "GRU_STACK_LAYER_POLICY", 1)
    pol = _run(gru, case, x, gys, h0s, all_layers=False)
    assert int(L.gru_tickets(dev).abs().sum()) == 0
    monkeypatch.setattr(Fn, "GRU_STACK", False)
    old = _run(gru, case, x, gys, h0s, all_layers=False)
    for l, (a, b) in enumerate(zip(pol["ys"], old["ys"])):
        assert torch.equal(a, b), f"layer {l}: states differ (rel {_rel(a, b):.3e})"
    assert torch.equal(pol["dx"], old["dx"])
    for a, b in zip(pol["dh0"], old["dh0"]):
        assert torch.equal(a, b)
    for (k, _), a, b in zip(gru.named_parameters(), pol["dw"], old["dw"]):
        assert _rel(a, b) < 1e-6, (k, _rel(a, b))
    # yardstick: the layer path with other split-K factors (a different, equally valid fp32 summation order)
    monkeypatch.setattr(Fn, "GRU_NS_CAP", 1)
    alt = _run(gru, case, x, gys, h0s, all_layers=False)
    yard = lambda key, i=None: _rel(alt[key], old[key]) if i is None else _rel(alt[key][i], old[key][i])
    for l, (a, b) in enumerate(zip(new["ys"], old["ys"])):
        assert torch.isfinite(a).all()
        assert _rel(a, b) < max(4e-3, 1.5 * yard("ys", l)), (l, _rel(a, b), yard("ys", l))
    assert _rel(new["dx"], old["dx"]) < max(1e-2, 1.5 * yard("dx"))
    for i, (a, b) in enumerate(zip(new["dh0"], old["dh0"])):
        assert _rel(a, b) < max(1e-2, 1.5 * yard("dh0", i))
    for i, ((k, _), a, b) in enumerate(zip(gru.named_parameters(), new["dw"], old["dw"])):
        assert _rel(a, b) < max(1e-2, 1.5 * yard("dw", i)), (k, _rel(a, b), yard("dw", i))


@pytest.mark.parametrize("S,shared", [(4, True), (8, False)], ids=["gru0", "gru1"])
def test_benchmark_sized_stack_against_the_oracle(S, shared, monkeypatch):
    """gru0 (4 x 4 frames, shared input) and gru1 (8 x 8) of the generator at ch = 32 and B = 64, T = 4: forward and BPTT of the
    wavefront under the production plans against the CPU restatement of ConvGRU.py:29-54, 104-133 (oracle.convgru), at the
    bf16 module bounds of tests/test_gpu_fullwidth.py (outputs 2e-2, input gradient 5e-2, parameter gradients cosine 0.999 / 5e-2)."""
    from oracle import dvdgan_cpu as O
    from dvd_gan_amd import functional as Fn
    T, B, cin, hids, ks = 4, 64, 256, [256, 512, 256], [3, 5, 3]
    case = (T, B, S, cin, hids, ks, shared, None, 0)
    gru, x, gys, _ = _build(case)
    monkeypatch.setattr(Fn, "GRU_STACK", True)
    got = _run(gru, case, x, gys, None, all_layers=False)
    # ---- the oracle on the same weights / inputs (fp32, CPU)
    torch.set_num_threads(min(16, __import__("os").cpu_count() or 1))
    sd = {k: v.detach().float().cpu() for k, v in gru.state_dict().items()}
    for p in sd.values():
        p.requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    hidden, ys = None, []
    for t in range(T):
        hidden = O.convgru(sd, "", xr if shared else xr[t * B:(t + 1) * B], hidden, 3)
        ys.append(hidden[-1])
    y = torch.cat(ys, 0)
    (y * gys[-1]).sum().backward()
    assert _rel(got["ys"][-1].cpu(), y.detach()) < 2e-2
    assert _rel(got["dx"].cpu(), xr.grad) < 5e-2
    scale = max(float(p.grad.abs().max()) for p in sd.values())
    for (k, _), g in zip(gru.named_parameters(), got["dw"]):
        ref = sd[k].grad
        if float(ref.abs().max()) < 1e-4 * scale:
            continue
        cos = float((g.cpu().double() * ref.double()).sum() / (g.cpu().double().norm() * ref.double().norm()))
        assert cos > 0.999 and _rel(g.cpu(), ref) < 5e-2, (k, cos, _rel(g.cpu(), ref))
