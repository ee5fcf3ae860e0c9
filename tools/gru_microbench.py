"""Times the SERIAL part of each ConvGRU layer of config C2 (B=64, T=48): the in-library time loops
dvd_convgru_layer_forward / _backward (recurrent convolutions + gate math), without the batched x-path / weight-gradient
launches around them.  HIP events on the launch stream, median of `iters` runs.
usage: python tools/gru_microbench.py [iters] [layer indices, e.g. 0,1,2] [B] [T]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvd_gan_amd import kern as K
from dvd_gan_amd import lib as L

# (name, S, cin, hidden, k) of the 12 ConvGRU layers at ch=32 (Generator.py:39,43,47,51)
LAYERS = [("gru0.l0", 4, 256, 256, 3), ("gru0.l1", 4, 256, 512, 5), ("gru0.l2", 4, 512, 256, 3),
          ("gru1.l0", 8, 256, 256, 3), ("gru1.l1", 8, 256, 512, 5), ("gru1.l2", 8, 512, 256, 3),
          ("gru2.l0", 16, 256, 256, 3), ("gru2.l1", 16, 256, 512, 5), ("gru2.l2", 16, 512, 256, 3),
          ("gru3.l0", 32, 128, 128, 3), ("gru3.l1", 32, 128, 256, 5), ("gru3.l2", 32, 256, 128, 5)]


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def setup(name, S, cin, hid, k, B, T, dev, dt, lib):
    M = B * S * S
    gx = (torch.randn(T, M, 3 * hid, device=dev) * 0.5).to(dt)
    pur = K.PackedConv(dt, 2 * hid, hid, (k, k), dev).fill(torch.randn(2 * hid, hid, k, k, device=dev) * (0.5 / (hid * k * k) ** 0.5))
    po = K.PackedConv(dt, hid, hid, (k, k), dev).fill(torch.randn(hid, hid, k, k, device=dev) * (0.5 / (hid * k * k) ** 0.5))
    keep = [gx, pur, po]
    mk = lambda: torch.empty(T, M, hid, dtype=dt, device=dev)
    bufs = [mk() for _ in range(5)]
    h32 = torch.empty(2, M, hid, dtype=torch.float32, device=dev)
    ntaps = k * k
    ns = [lib.dvd_conv_pick_nsplit(L.BF16, C.c_longlong(M), co, ci, ntaps) for co, ci in ((2 * hid, hid), (hid, hid), (hid, 2 * hid))]
    lib.dvd_convgru_ws_floats.restype = C.c_longlong
    ws = torch.empty(lib.dvd_convgru_ws_floats(L.BF16, B, S, S, hid, k), dtype=torch.float32, device=dev)
    tickets = torch.zeros(L.GRU_TICKETS, dtype=torch.int32, device=dev)    # per layer: `pairs` / `halves` run layers on two streams
    d = L.GruDesc()
    d.dtype, d.T, d.B, d.H, d.W, d.hidden, d.k = L.BF16, T, B, S, S, hid, k
    d.gx_stride = M * 3 * hid
    d.gx, d.w_ur, d.w_o = gx.data_ptr(), pur.wf.data_ptr(), po.wf.data_ptr()
    d.wd_ur, d.wd_o = pur.wd.data_ptr(), po.wd.data_ptr()
    if os.environ.get("GB", "1") != "0":       # fragment-major weight images, as functional.ConvGRULayer supplies them
        d.w_ur_q, d.w_o_q = pur.fragment_major("wf").data_ptr(), po.fragment_major("wf").data_ptr()
        d.wd_ur_q, d.wd_o_q = pur.fragment_major("wd").data_ptr(), po.fragment_major("wd").data_ptr()
    d.h_all, d.u_all, d.r_all, d.o_all, d.hr_all = (t.data_ptr() for t in bufs)
    d.h32, d.ws = h32.data_ptr(), ws.data_ptr()
    if os.environ.get("DVD_GRU_TICKETS", "1") != "0":
        d.tickets = tickets.data_ptr()
    dh = (torch.randn(T, M, hid, device=dev) * 0.1).to(dt)
    dg = torch.empty(T, M, 3 * hid, dtype=dt, device=dev)
    carry = torch.empty(M, hid, dtype=torch.float32, device=dev)
    d.dh_out, d.dg, d.carry = dh.data_ptr(), dg.data_ptr(), carry.data_ptr()
    keep += bufs + [h32, ws, dh, dg, carry, tickets]
    return d, keep


def pairs(iters, B, T):
    """Two layers' time loops on two streams at once vs one after the other (how much a layer wavefront could hide)."""
    dev, dt = "cuda", torch.bfloat16
    lib = L.lib()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for a, b in ((0, 1), (3, 4), (6, 7), (9, 10), (1, 2), (4, 5)):
        da, ka = setup(*LAYERS[a], B, T, dev, dt, lib)
        db, kb = setup(*LAYERS[b], B, T, dev, dt, lib)

        def run(fn_name, concurrent):
            fn = getattr(lib, fn_name)
            cur = torch.cuda.current_stream()
            if not concurrent:
                st = C.c_void_p(cur.cuda_stream)
                L.check(fn(C.byref(da), st)); L.check(fn(C.byref(db), st))
                return
            s1.wait_stream(cur); s2.wait_stream(cur)
            L.check(fn(C.byref(da), C.c_void_p(s1.cuda_stream)))
            L.check(fn(C.byref(db), C.c_void_p(s2.cuda_stream)))
            cur.wait_stream(s1); cur.wait_stream(s2)
        for fn_name in ("dvd_convgru_layer_forward", "dvd_convgru_layer_backward"):
            ts = timed(lambda: run(fn_name, False), iters)
            tc = timed(lambda: run(fn_name, True), iters)
            print(f"{LAYERS[a][0]} + {LAYERS[b][0]} {fn_name[19:]:9s}: sequential {ts:7.2f} ms   concurrent {tc:7.2f} ms   ({100 * (1 - tc / ts):4.1f} % saved)", flush=True)


def triples(iters, B, T):
    """The three layers of each ConvGRU as three INDEPENDENT recurrences (own random gx) on three streams vs one after the other:
    an upper bound of what per-layer streams (instead of grouped launches) could overlap."""
    dev, dt = "cuda", torch.bfloat16
    lib = L.lib()
    streams = [torch.cuda.Stream() for _ in range(3)]
    for gi in range(4):
        ds = [setup(*LAYERS[3 * gi + l], B, T, dev, dt, lib) for l in range(3)]

        def run(fn_name, concurrent):
            fn = getattr(lib, fn_name)
            cur = torch.cuda.current_stream()
            if not concurrent:
                for d_, _ in ds:
                    L.check(fn(C.byref(d_), C.c_void_p(cur.cuda_stream)))
                return
            for st_ in streams:
                st_.wait_stream(cur)
            for (d_, _), st_ in zip(ds, streams):
                L.check(fn(C.byref(d_), C.c_void_p(st_.cuda_stream)))
            for st_ in streams:
                cur.wait_stream(st_)
        for fn_name in ("dvd_convgru_layer_forward", "dvd_convgru_layer_backward"):
            ts = timed(lambda: run(fn_name, False), iters)
            tc = timed(lambda: run(fn_name, True), iters)
            print(f"gru{gi} {fn_name[19:]:9s}: three layers sequential {ts:7.2f} ms   on three streams {tc:7.2f} ms   ({100 * (1 - tc / ts):4.1f} % saved)", flush=True)


def halves(iters, B, T):
    """One layer at batch B on one stream vs the two halves of the batch (independent recurrences) on two streams."""
    dev, dt = "cuda", torch.bfloat16
    lib = L.lib()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for i in range(len(LAYERS)):
        full, kf = setup(*LAYERS[i], B, T, dev, dt, lib)
        ha, ka = setup(*LAYERS[i], B // 2, T, dev, dt, lib)
        hb, kb = setup(*LAYERS[i], B // 2, T, dev, dt, lib)
        for fn_name in ("dvd_convgru_layer_forward", "dvd_convgru_layer_backward"):
            fn = getattr(lib, fn_name)

            def one():
                L.check(fn(C.byref(full), C.c_void_p(torch.cuda.current_stream().cuda_stream)))

            def two():
                cur = torch.cuda.current_stream()
                s1.wait_stream(cur); s2.wait_stream(cur)
                L.check(fn(C.byref(ha), C.c_void_p(s1.cuda_stream)))
                L.check(fn(C.byref(hb), C.c_void_p(s2.cuda_stream)))
                cur.wait_stream(s1); cur.wait_stream(s2)
            t1, t2 = timed(one, iters), timed(two, iters)
            print(f"{LAYERS[i][0]} {fn_name[19:]:9s}: whole batch {t1:7.2f} ms   two halves on two streams {t2:7.2f} ms   ({100 * (1 - t2 / t1):5.1f} % saved)", flush=True)


def parts(iters, B, T, n, sel):
    """One layer at batch B on one stream vs n equal parts of the batch (independent recurrences) on n streams."""
    dev, dt = "cuda", torch.bfloat16
    lib = L.lib()
    streams = [torch.cuda.Stream() for _ in range(n)]
    for i in sel:
        full, kf = setup(*LAYERS[i], B, T, dev, dt, lib)
        ps = [setup(*LAYERS[i], B // n, T, dev, dt, lib) for _ in range(n)]
        for fn_name in ("dvd_convgru_layer_forward", "dvd_convgru_layer_backward"):
            fn = getattr(lib, fn_name)

            def one():
                L.check(fn(C.byref(full), C.c_void_p(torch.cuda.current_stream().cuda_stream)))

            def many():
                cur = torch.cuda.current_stream()
                for st_ in streams:
                    st_.wait_stream(cur)
                for (d_, _), st_ in zip(ps, streams):
                    L.check(fn(C.byref(d_), C.c_void_p(st_.cuda_stream)))
                for st_ in streams:
                    cur.wait_stream(st_)
            t1, t2 = timed(one, iters), timed(many, iters)
            print(f"{LAYERS[i][0]} {fn_name[19:]:9s}: whole batch {t1:7.2f} ms   {n} parts on {n} streams {t2:7.2f} ms   ({100 * (1 - t2 / t1):5.1f} % saved)", flush=True)


def stack(iters, B, T, sel, run=0, policy=0):
    """Whole ConvGRUs: the layer wavefront (dvd_convgru_stack_*) against the layer-by-layer path INCLUDING what the wavefront
    absorbs -- the batched x-part convolutions of layers 1, 2 (forward) and their batched backward-data convolutions."""
    dev, dt = "cuda", torch.bfloat16
    lib = L.lib()
    lib.dvd_convgru_stack_ws_floats.restype = C.c_longlong
    tot = [0.0, 0.0, 0.0, 0.0]
    for gi in sel:
        lays = LAYERS[3 * gi:3 * gi + 3]
        S = lays[0][1]
        M = B * S * S
        descs, keeps, pxs, gxs, dhm = [], [], [], [], []
        for (name, S_, cin, hid, k) in lays:
            d, keep = setup(name, S_, cin, hid, k, B, T, dev, dt, lib)
            px = K.PackedConv(dt, 3 * hid, cin, (k, k), dev).fill(torch.randn(3 * hid, cin, k, k, device=dev) * (0.5 / (cin * k * k) ** 0.5))
            descs.append(d); keeps.append(keep); pxs.append(px)
            gxs.append(keep[0]); dhm.append(torch.empty(T * B, S, S, cin, dtype=dt, device=dev))
        bias = [torch.zeros(3 * l[3], device=dev) for l in lays]
        sd = L.GruStackDesc()
        sd.n_layers, sd.layer_policy, sd.run = 3, policy, run
        for l in range(3):
            C.memmove(C.byref(sd.layer[l]), C.byref(descs[l]), C.sizeof(L.GruDesc))
            if l:
                sd.cin[l] = lays[l][2]
                sd.wx[l], sd.wx_q[l], sd.bx[l] = pxs[l].wf.data_ptr(), pxs[l].fragment_major("wf").data_ptr(), bias[l].data_ptr()
                sd.wdx[l], sd.wdx_q[l] = pxs[l].wd.data_ptr(), pxs[l].fragment_major("wd").data_ptr()
                sd.dh_mid[l] = dhm[l].data_ptr()
        sd.layer[0].dh_out = None
        sd.layer[1].dh_out = None
        ws = torch.empty(lib.dvd_convgru_stack_ws_floats(C.byref(sd)), dtype=torch.float32, device=dev)
        sd.ws = ws.data_ptr()
        assert lib.dvd_convgru_stack_ok(C.byref(sd), 0) and lib.dvd_convgru_stack_ok(C.byref(sd), 1)
        st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
        hs = [keeps[l][3].view(T * B, S, S, lays[l][3]) for l in range(3)]           # h_all of every layer
        dgs = [keeps[l][-3].view(T * B, S, S, 3 * lays[l][3]) for l in range(3)]

        def layer_fwd():
            for l in range(3):
                if l:
                    K.conv_forward(hs[l - 1], pxs[l].wf, (lays[l][4],) * 2, 3 * lays[l][3], bias=bias[l], out=gxs[l].view(T * B, S, S, -1),
                                   wq=lambda: pxs[l].fragment_major("wf"))
                L.check(lib.dvd_convgru_layer_forward(C.byref(descs[l]), st()))

        def layer_bwd():
            for l in (2, 1, 0):
                L.check(lib.dvd_convgru_layer_backward(C.byref(descs[l]), st()))
                if l:
                    K.conv_forward(dgs[l], pxs[l].wd, (lays[l][4],) * 2, pxs[l].cip, out=dhm[l], wq=lambda: pxs[l].fragment_major("wd"))
        t_lf = timed(layer_fwd, iters)
        t_sf = timed(lambda: L.check(lib.dvd_convgru_stack_forward(C.byref(sd), st())), iters)
        t_lb = timed(layer_bwd, iters)
        t_sb = timed(lambda: L.check(lib.dvd_convgru_stack_backward(C.byref(sd), st())), iters)
        for i, v in enumerate((t_lf, t_sf, t_lb, t_sb)):
            tot[i] += v
        print(f"gru{gi} S={S:2d}: forward layers {t_lf:7.2f} ms  wavefront {t_sf:7.2f} ms ({100 * (1 - t_sf / t_lf):5.1f} % saved)   "
              f"backward layers {t_lb:7.2f} ms  wavefront {t_sb:7.2f} ms ({100 * (1 - t_sb / t_lb):5.1f} % saved)", flush=True)
    print(f"total: forward {tot[0]:.2f} -> {tot[1]:.2f} ms   backward {tot[2]:.2f} -> {tot[3]:.2f} ms   sum {tot[0] + tot[2]:.2f} -> {tot[1] + tot[3]:.2f} ms")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "stack":
        sel = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 and sys.argv[2] else range(4)
        run = int(sys.argv[3]) if len(sys.argv) > 3 else 0
        policy = int(sys.argv[4]) if len(sys.argv) > 4 else 0
        return stack(3, 64, 48, sel, run, policy)
    if len(sys.argv) > 1 and sys.argv[1] == "parts":
        n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
        sel = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else range(len(LAYERS))
        return parts(3, 64, 48, n, sel)
    if len(sys.argv) > 1 and sys.argv[1] == "triples":
        return triples(3, 64, 48)
    if len(sys.argv) > 1 and sys.argv[1] == "pairs":
        return pairs(3, 64, 48)
    if len(sys.argv) > 1 and sys.argv[1] == "halves":
        return halves(3, 64, 48)
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    sel = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 and sys.argv[2] else range(len(LAYERS))
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    T = int(sys.argv[4]) if len(sys.argv) > 4 else 48
    dev, dt = "cuda", torch.bfloat16
    lib = L.lib()
    tot_f = tot_b = 0.0
    for name, S, cin, hid, k in [LAYERS[i] for i in sel]:
        M, ntaps = B * S * S, k * k
        d, keep = setup(name, S, cin, hid, k, B, T, dev, dt, lib)
        ns = [lib.dvd_conv_pick_nsplit(L.BF16, C.c_longlong(M), co, ci, ntaps) for co, ci in ((2 * hid, hid), (hid, hid), (hid, 2 * hid))]
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        tf = timed(lambda: L.check(lib.dvd_convgru_layer_forward(C.byref(d), st)), iters)
        tb = timed(lambda: L.check(lib.dvd_convgru_layer_backward(C.byref(d), st)), iters)
        fl_f = 2.0 * (T - 1) * M * hid * 3 * hid * ntaps
        fl_b = fl_f
        tot_f += tf; tot_b += tb
        print(f"{name} S={S:2d} h={hid:3d} k={k} ns(ur,o,dur)={ns}: fwd {tf:7.2f} ms ({fl_f / tf / 1e9:6.0f} TF/s, {tf / T * 1e3:6.1f} us/step)"
              f"   bwd {tb:7.2f} ms ({fl_b / tb / 1e9:6.0f} TF/s, {tb / T * 1e3:6.1f} us/step)", flush=True)
    print(f"total fwd {tot_f:.2f} ms  bwd {tot_b:.2f} ms  sum {tot_f + tot_b:.2f} ms")


if __name__ == "__main__":
    main()
