#!/usr/bin/env python3
"""bench.py -- clips/s for one full G + D_s + D_t training step (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W            (N > 1 without a launcher: re-executes itself under
                                                               torch.distributed.run, one rank per GPU, RCCL;
                                                               defaults: N = 1, K = 10, W = 3 -- about a minute incl. cpu_baseline)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1], per GPU): UCF-101-shaped synthetic clips, T=48, 64x64,
101 classes, ch=32, z_dim=120, k_sample=8, hinge, Adam(5e-5, (0, 0.9)), batch 64 per GPU, bf16
MFMA operands with fp32 accumulation.  Weak scaling: every rank runs 64 clips, gradients are
averaged with RCCL; `value` = N*64 / (max over ranks of the step time).
Other shapes: --size 128 --n-class 600 (configs[3]; the batch defaults to the largest of 64 / 48 / 32 / 16 that fits),
--frames 12 --size 128 --state-carry (configs[4]: initial ConvGRU states supplied and differentiated).

Extra objects on the JSON line:
  roofline      dominant kernel family (conv_halo / conv_igemm bf16: forward + backward-data launches) --
                algorithmic FLOPs of its launches / their summed durations, measured with HIP
                events recorded on the launch stream around every launch of ONE EXTRA step run after the timed region
                (not part of `value`; --no-kernel-prof switches it off); also the whole-step figure (F = 8148.5 GFLOP/clip).
                `traffic`: HBM bytes per launch from the committed rocprofv3 PMC passes of THIS shape, else null.
  cpu_baseline  the CPU oracle (oracle/dvdgan_cpu.py, a port of the reference step pinned on reference fixtures)
                timed on the host: 1 warm-up + 2 timed steps, same shape, B=2, 16 threads (host core count alongside).
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

# the host driver only supports dmabuf IPC: without this RCCL's cross-process buffer sharing fails (hipIpcGetMemHandle)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# SURVEY.md section 8(d) / BASELINE.md section 3: algorithmic GFLOP per clip per step, keyed (ch, frames, size)
F_GFLOP_PER_CLIP = {(32, 48, 64): 8148.5, (32, 48, 128): 32608.6, (32, 12, 128): 8530.6}
PEAK_F32_MFMA_TFLOPS = 157.3     # f32-input MFMA = vector rate (MI355X_MICROARCH.md, matrix cores table); exact mode
PEAK_BF16_TFLOPS = 2500.0                       # MI355X_MICROARCH.md: dense bf16 MFMA


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="clips per GPU (default 64; at --size 128 the largest that fits)")
    ap.add_argument("--ch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=48)
    ap.add_argument("--k-sample", type=int, default=8)
    ap.add_argument("--n-class", type=int, default=101)
    ap.add_argument("--size", type=int, default=64, choices=[64, 128],
                    help="frame size; 128 = the Kinetics-600-shaped clips of BASELINE configs[3] (use --n-class 600)")
    ap.add_argument("--state-carry", action="store_true",
                    help="supply (and differentiate) initial ConvGRU states: the frame-conditional variant of BASELINE configs[4]")
    ap.add_argument("--g-attn", default="none", choices=["none", "self", "sep", "both"],
                    help="switch on the generator's optional attention blocks (Attention.py:114-185 SelfAttention over the (T, ld, ld) "
                         "latent clip after the first ConvGRU / :8-111 SeparableAttn after module 8); off in the reference and in BASELINE's configs")
    ap.add_argument("--dp-mode", default="replica", choices=["replica", "global"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--force-exchange", action="store_true",
                    help="N=1 only: create a ONE-rank RCCL group and run the whole data-parallel exchange (all-reduces on the exchange "
                         "stream, bucket hooks, side-stream fences) -- prices the exchange's fences / host work at this batch size")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-prof", action="store_true")
    return ap.parse_args()


def respawn_if_needed(a):
    """`python bench.py --gpus N` with no launcher: run N ranks of this script under torch.distributed.run."""
    if a.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    if torch.cuda.device_count() < a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus}: only {torch.cuda.device_count()} GPU(s) visible")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


def hbm_traffic(a, batch):
    """HBM bytes per launch of the dominant kernel family from the committed PMC passes (FETCH_SIZE / WRITE_SIZE collected
    with rocprofv3 in separate runs and corrected as MI355X_MICROARCH.md prescribes) -- only when they were taken on the
    shape being run; PMC collection cannot run inside the timed benchmark."""
    best, src = None, None
    for name in sorted(os.listdir(os.path.join(ROOT, "profiles"))):
        if not (name.endswith("hbm_traffic.json")):
            continue
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)
            shape = d.get("shape", {"size": 64, "frames": 48, "batch": 64, "ch": 32, "dtype": "bf16"})
            if shape == {"size": a.size, "frames": a.frames, "batch": batch, "ch": a.ch, "dtype": a.dtype}:
                best = d["kernels"]["conv_igemm"]["hbm_bytes_per_launch_corrected"]      # latest round wins (sorted names)
                src = "profiles/" + name
        except Exception:
            pass
    return best, src


def cpu_baseline(a):
    """Reference-equivalent steps at B=2 on the host cores (oracle = checker, timed as baseline): 1 warm-up + 2 timed."""
    from oracle import dvdgan_cpu as O
    from dvd_gan_amd.gen_net import Generator
    from dvd_gan_amd.disc_nets import SpatialDiscriminator, TemporalDiscriminator
    # measured on the 256-core GPU host: this step takes 12.5 s on 16 threads, 19.8 s on 32, 45 s on 64
    # (thousands of small ops: more threads only add synchronisation) -> 16 threads is the fastest setting
    host = os.cpu_count() or 1
    cores = min(16, host)
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    B = 2
    sds = []
    for net in (Generator(120, a.size // 16, a.n_class, a.ch, a.frames), SpatialDiscriminator(a.ch, a.n_class),
                TemporalDiscriminator(a.ch, a.n_class)):
        sds.append(O.make_state({k: v.detach().clone() for k, v in net.state_dict().items()}))
    st = O.TrainState(*sds, ch=a.ch, n_frames=a.frames, k_sample=a.k_sample, n_class=a.n_class, latent_dim=a.size // 16)
    real = torch.rand(B, 3, a.frames, a.size, a.size) * 2 - 1
    labels = torch.randint(0, a.n_class, (B,))
    times = []
    for i in range(3):
        t0 = time.time()
        O.train_step(st, real, labels, torch.randn(B, 120), torch.randint(0, a.n_class, (B,)),
                     torch.randperm(a.frames), torch.randperm(a.frames))
        times.append(time.time() - t0)
    dt = sum(times[1:]) / 2
    return {"value": round(B / dt, 5), "unit": "clips/s", "cores": cores, "host_cores": host, "kind": "port",
            "sample": f"1 warm-up + 2 timed full G+Ds+Dt steps, B={B}, T={a.frames}, {a.size}x{a.size}, ch={a.ch} "
                      f"(fp32 torch CPU ops), {dt:.1f} s per step on {cores} of {host} host cores"}


def carried_states(a, batch, dev):
    """Random initial states for every layer of the four ConvGRUs (configs[4]); they require grad, so d/dh0 is computed."""
    c8, c4, ld = 8 * a.ch, 4 * a.ch, a.size // 16
    sizes = [(c8, ld), (c8, 2 * ld), (c8, 4 * ld), (c4, 8 * ld)]
    return [[torch.randn(batch, h, s, s, device=dev, requires_grad=True) for h in (c, 2 * c, c)] for c, s in sizes]


def main():
    a = parse()
    if a.force_exchange:
        os.environ["DVD_FORCE_EXCHANGE"] = "1"
    respawn_if_needed(a)
    from dvd_gan_amd import dist as D
    from dvd_gan_amd import lib as L
    from dvd_gan_amd.train_step import Trainer
    rank, world, dev = D.init_from_env()
    if world != a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus} launched with WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    import torch.distributed as dist
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    candidates = [a.batch] if a.batch else ([64] if a.size == 64 else [64, 48, 32, 16])
    lib = L.lib()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1 and len(candidates) > 1:
        # Data parallel: a rank that ran out of memory inside a step would leave its peers stuck in that step's gradient
        # all-reduces, so the ranks agree on the batch size BEFORE any collective step -- from an estimate of the step's peak
        # memory (measured on one GPU: 174.7 GB at 64 clips of 48 x 128 x 128, ch=32; activations dominate, linear in clips,
        # frames, pixels and channels) against the free memory each rank reports, minimum over the ranks.
        per_clip = 174.7 / 64 * (a.frames / 48) * (a.size / 128) ** 2 * (a.ch / 32) * 2 ** 30 * 1.05
        free = torch.cuda.mem_get_info(dev)[0]
        fit = [b for b in candidates if b * per_clip <= free] or [candidates[-1]]
        pick = torch.tensor([float(fit[0])], device=dev)
        dist.all_reduce(pick, op=dist.ReduceOp.MIN)
        candidates = [int(pick)]
    tr = None
    for batch in candidates:
        cfg = argparse.Namespace(adv_loss="hinge", z_dim=120, g_chn=a.ch, ds_chn=a.ch, dt_chn=a.ch, n_frames=a.frames,
                                 lr_schr="const", total_epoch=1, d_iters=1, batch_size=batch, g_lr=5e-5, d_lr=5e-5,
                                 beta1=0.0, beta2=0.9, n_class=a.n_class, k_sample=a.k_sample,
                                 g_self_attn=a.g_attn in ("self", "both"), g_sep_attn=a.g_attn in ("sep", "both"))
        try:
            torch.manual_seed(0)                               # (the Trainer broadcasts rank 0's model in any case)
            tr = Trainer([], cfg, device=dev, compute_dtype=dtype, latent_dim=a.size // 16, dp_mode=a.dp_mode)
            tr.G.train(); tr.D_s.train(); tr.D_t.train()
            gen = torch.Generator().manual_seed(1)
            gB = batch * world
            real = D.shard(torch.rand(gB, 3, a.frames, a.size, a.size, generator=gen) * 2 - 1, rank, world).to(dev)
            labels = D.shard(torch.randint(0, a.n_class, (gB,), generator=gen), rank, world).to(dev)
            torch.manual_seed(100 + rank)                      # per-rank z / labels; frame ids come from the shared generator
            hidden = carried_states(a, batch, dev) if a.state_carry else None
            tr.register_label_buffer(labels)                   # one device buffer for every step: range-checked once
            for _ in range(a.warmup):
                tr.train_step(real, labels, hidden=hidden)
            ok = torch.ones(1, device=dev)
        except torch.cuda.OutOfMemoryError:
            if batch == candidates[-1]:
                raise
            ok = torch.zeros(1, device=dev)
        if float(ok) > 0:
            break
        tr = real = labels = hidden = None
        torch.cuda.empty_cache()
    sync()
    t0 = time.perf_counter()
    from dvd_gan_amd import functional as Fn
    for i in range(a.steps):
        losses = tr.train_step(real, labels, hidden=hidden)
    sync()
    dt = time.perf_counter() - t0               # exactly K steps in the execution mode training runs in
    lossv = [float(v.detach()) for v in losses]
    if not a.no_kernel_prof:
        # One EXTRA step after the timed region (not part of `value`): HIP events around every conv launch, recorded on the
        # launch stream, with the weight-gradient kernels on the launch stream as well instead of beside it -- a kernel's
        # duration is taken while it has the GPU to itself.
        lib.dvd_prof_enable(1)
        Fn.serialize_weight_grads(True)
        tr.train_step(real, labels, hidden=hidden)
        sync()
        lib.dvd_prof_enable(0)
        Fn.serialize_weight_grads(False)
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    ms = dt / a.steps * 1e3
    value = gB * a.steps / dt

    roof = None
    if not a.no_kernel_prof:
        lib.dvd_prof_report_variants.restype = C.c_longlong
        res, by_kernel = {}, {}
        # bf16: the halo-staged kernels read their weights from L2 (conv_halo_gb_kernel; conv_halo_gbs_kernel on 4x4 / 8x8 frames);
        # exact mode runs the LDS-staged conv_halo_kernel under the same variant numbers
        hk = "conv_halo_gb_kernel" if a.dtype == "bf16" else "conv_halo_kernel"
        vnames = {0: {1: hk + "<%s, 256x128 tile>" % a.dtype, 2: hk + "<%s, 128x128 tile>" % a.dtype,
                      3: hk + "<%s, 256x64 tile>" % a.dtype, 4: "conv_igemm_kernel<%s, 128x128 tile>" % a.dtype,
                      5: "conv_igemm_kernel<%s, 256x128 tile>" % a.dtype, 6: "conv_igemm_kernel<bf16, 256x256 tile, 8 waves>",
                      7: "conv_halo_gbs_kernel<bf16, 256x128 tile, 8x8 frames>",
                      8: "conv_halo_gbs_kernel<bf16, 128x128 tile, 4x4 / 8x8 frames>", 9: "conv_thin_in_kernel<bf16, 3 -> 64 channels>",
                      10: "conv_group_gb_kernel<bf16, 256x128 tiles, grouped ConvGRU wavefront launches>",
                      11: "conv_group_gbs_kernel<bf16, grouped ConvGRU wavefront launches on 4x4 / 8x8 frames>"},
                  1: {1: "conv_wgrad_row_kernel<8-wave filter-row tiles>", 2: "conv_wgrad_kernel", 3: "wgrad_thin_kernel",
                      4: "conv_wgrad_row4_kernel<one wave per SIMD: 3 taps 256x128 / 128x128, 5 taps 256x64 / 128x128>"}}
        for kind, name in ((0, "conv_igemm"), (1, "conv_wgrad")):
            NV = 12
            nn, tms, fl = (C.c_longlong * NV)(), (C.c_double * NV)(), (C.c_double * NV)()
            lib.dvd_prof_report_variants(kind, NV, nn, tms, fl)          # totals of the instrumented step, per kernel variant
            mk = lambda v: {"launches": int(nn[v]), "ms": tms[v],
                            "tflops": (fl[v] / (tms[v] * 1e-3) / 1e12) if tms[v] else 0.0,
                            "avg_us": tms[v] * 1e3 / max(nn[v], 1), "gflop_per_launch": fl[v] / max(nn[v], 1) / 1e9}
            res[name] = mk(0)
            for v, vn in vnames[kind].items():
                if nn[v]:
                    e = mk(v)
                    by_kernel[vn] = {"launches": e["launches"], "ms": round(e["ms"], 1), "avg_us": round(e["avg_us"], 1),
                                     "achieved": round(e["tflops"], 1)}
        F = F_GFLOP_PER_CLIP.get((a.ch, a.frames, a.size))
        traffic, traffic_src = hbm_traffic(a, batch)     # a committed PMC collection (tools/collect_evidence.sh), not this run
        dom = res["conv_igemm"]
        peak = PEAK_BF16_TFLOPS if a.dtype == "bf16" else PEAK_F32_MFMA_TFLOPS
        roof = {"bound": "mfma", "kernel": "conv_halo_gb_kernel + conv_group_gb_kernel + conv_halo_gbs_kernel + conv_group_gbs_kernel + conv_igemm_kernel <bf16> (forward + backward-data convolutions)" if a.dtype == "bf16" else "conv_halo_kernel + conv_igemm_kernel <f32>",
                "achieved": round(dom["tflops"], 1), "peak": peak, "unit": "TFLOP/s",
                "frac": round(dom["tflops"] / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                "launches_per_step": dom["launches"], "avg_launch_us": round(dom["avg_us"], 1),
                "gflop_per_launch": round(dom["gflop_per_launch"], 2), "kernel_ms_per_step": round(dom["ms"], 1),
                "wgrad": {k: round(v, 2) if isinstance(v, float) else v for k, v in res["conv_wgrad"].items()},
                "kernels": {k: dict(v, frac=round(v["achieved"] / peak, 4)) for k, v in by_kernel.items()},
                "timing": "HIP events on the launch stream around each launch of one extra step after the timed region, kernels serialised (no concurrent weight-gradient stream) in that step",
                "step_achieved": round(value / world * F / 1e3, 1) if F else None,
                "step_frac": round(value / world * F / 1e3 / peak, 4) if F else None}
    if rank == 0:
        cfg_id = 1 if a.size == 64 else (4 if a.state_carry else 3)
        # configs[1] / [2]: UCF-101-shaped; configs[3]: Kinetics-600-shaped; configs[4]: the frame-conditional video-prediction
        # variant (BASELINE names no dataset for it)
        kind = "UCF-101" if a.size == 64 else ("frame-conditional video-prediction" if a.state_carry else "Kinetics-600")
        out = {"metric": "clips/sec per G+Ds+Dt step, 48x64x64 UCF-101 synth" if (a.size == 64 and a.frames == 48)
               else f"clips/sec per G+Ds+Dt step, {a.frames}x{a.size}x{a.size} {kind}-shaped synth",
               "value": round(value, 3), "unit": "clips/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 2),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
               "config": {"workload": f"{kind}-shaped {a.n_class}-class {a.frames}x{a.size}x{a.size} clips, G+Ds+Dt hinge step, ch={a.ch}, "
                                      f"k_sample={a.k_sample}, batch {batch}/GPU"
                                      + (", initial ConvGRU states supplied and differentiated" if a.state_carry else "")
                                      + (f", generator attention blocks ON ({a.g_attn}; not part of BASELINE's configs)" if a.g_attn != "none" else "")
                                      + f" (BASELINE configs[{cfg_id}])",
                          "global_batch": gB, "parallelism": f"dp{world}" + ("" if world == 1 else f" ({a.dp_mode} batch norm)")
                          + (f" + forced one-rank RCCL exchange ({a.dp_mode} batch norm)" if a.force_exchange else "")},
               "losses": [round(v, 4) for v in lossv],
               "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}
        if roof:
            out["roofline"] = roof
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
