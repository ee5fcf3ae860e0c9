#!/usr/bin/env python3
"""bench.py -- clips/s for one full G + D_s + D_t training step (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1], per GPU): UCF-101-shaped synthetic clips, T=48, 64x64,
101 classes, ch=32, z_dim=120, k_sample=8, hinge, Adam(5e-5, (0, 0.9)), batch 64 per GPU, bf16
MFMA operands with fp32 accumulation.  Weak scaling: every rank runs 64 clips, gradients are
averaged with RCCL; `value` = N*64 / (max over ranks of the step time).

Extra objects on the JSON line:
  roofline      dominant kernel family (conv_halo / conv_igemm bf16: forward + backward-data launches) --
                algorithmic FLOPs of its launches / their summed durations, measured with HIP
                events recorded on the launch stream around every launch of the last timed step
                (--no-kernel-prof switches them off); also the whole-step figure (F = 8148.5 GFLOP/clip).
  cpu_baseline  the CPU oracle (oracle/dvdgan_cpu.py, a validated port of the reference step)
                timed on 16 host threads (the fastest setting measured) on a bounded sample: ONE step, same shape, B=2.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

# the host driver only supports dmabuf IPC: without this RCCL's cross-process buffer sharing fails (hipIpcGetMemHandle)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F_GFLOP_PER_CLIP = {(32, 48, 64): 8148.5, (32, 48, 128): 32608.6}       # SURVEY.md section 8(d) / BASELINE.md section 3
PEAK_F32_MFMA_TFLOPS = 157.3     # f32-input MFMA = vector rate (MI355X_MICROARCH.md, matrix cores table); exact mode
PEAK_BF16_TFLOPS = 2500.0                       # MI355X_MICROARCH.md: dense bf16 MFMA


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64, help="clips per GPU")
    ap.add_argument("--ch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=48)
    ap.add_argument("--k-sample", type=int, default=8)
    ap.add_argument("--n-class", type=int, default=101)
    ap.add_argument("--size", type=int, default=64, choices=[64, 128],
                    help="frame size; 128 = the Kinetics-600-shaped clips of BASELINE configs[3] (use --n-class 600 and a --batch that fits)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-prof", action="store_true")
    return ap.parse_args()


def hbm_traffic(kernel):
    """HBM bytes per launch of `kernel`, from the committed PMC passes (FETCH_SIZE / WRITE_SIZE collected with
    rocprofv3 in separate runs and corrected as MI355X_MICROARCH.md prescribes; see profiles/r01_hbm_traffic.json).
    PMC collection cannot run inside the timed benchmark, so this is a recorded measurement of the same command."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")) as f:
            return json.load(f)["kernels"][kernel]["hbm_bytes_per_launch_corrected"]
    except Exception:
        return None


def cpu_baseline(a):
    """One reference-equivalent step at B=1 on the host cores (oracle = checker, timed as baseline)."""
    from oracle import dvdgan_cpu as O
    from dvd_gan_amd.gen_net import Generator
    from dvd_gan_amd.disc_nets import SpatialDiscriminator, TemporalDiscriminator
    # measured on the 256-core GPU host: this step takes 12.5 s on 16 threads, 19.8 s on 32, 45 s on 64
    # (thousands of small ops: more threads only add synchronisation) -> 16 threads is the fastest setting
    cores = min(16, os.cpu_count() or 1)
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
    t0 = time.time()
    O.train_step(st, real, labels, torch.randn(B, 120), torch.randint(0, a.n_class, (B,)),
                 torch.randperm(a.frames), torch.randperm(a.frames))
    dt = time.time() - t0
    return {"value": round(B / dt, 5), "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": f"1 full G+Ds+Dt step (no warm-up), B={B}, T={a.frames}, {a.size}x{a.size}, ch={a.ch} (fp32 torch CPU ops), {dt:.1f} s"}


def main():
    a = parse()
    from dvd_gan_amd import dist as D
    from dvd_gan_amd import lib as L
    from dvd_gan_amd.train_step import Trainer
    rank, world, dev = D.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    import torch.distributed as dist
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    cfg = argparse.Namespace(adv_loss="hinge", z_dim=120, g_chn=a.ch, ds_chn=a.ch, dt_chn=a.ch, n_frames=a.frames,
                             lr_schr="const", total_epoch=1, d_iters=1, batch_size=a.batch, g_lr=5e-5, d_lr=5e-5,
                             beta1=0.0, beta2=0.9, n_class=a.n_class, k_sample=a.k_sample)
    torch.manual_seed(0)                                   # identical initial weights on every rank
    tr = Trainer([], cfg, device=dev, compute_dtype=dtype, latent_dim=a.size // 16)
    tr.G.train(); tr.D_s.train(); tr.D_t.train()
    gen = torch.Generator().manual_seed(1)
    gB = a.batch * world
    real = D.shard(torch.rand(gB, 3, a.frames, a.size, a.size, generator=gen) * 2 - 1, rank, world).to(dev)
    labels = D.shard(torch.randint(0, a.n_class, (gB,), generator=gen), rank, world).to(dev)
    torch.manual_seed(100 + rank)                          # per-rank z / labels, shared frame ids do not matter here

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        tr.train_step(real, labels)
    sync()
    lib = L.lib()
    t0 = time.perf_counter()
    for i in range(a.steps):
        if i == a.steps - 1 and not a.no_kernel_prof:
            lib.dvd_prof_enable(1)      # HIP events around every conv launch of the LAST timed step, recorded on the
        losses = tr.train_step(real, labels)    # launch stream (5.4k event pairs cost ~2 % of a step, so not on all K)
    sync()
    dt = time.perf_counter() - t0
    lib.dvd_prof_enable(0)
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    ms = dt / a.steps * 1e3
    value = gB * a.steps / dt
    lossv = [float(v.detach()) for v in losses]

    roof = None
    if not a.no_kernel_prof:
        lib.dvd_prof_report.restype = C.c_longlong
        res = {}
        for kind, name in ((0, "conv_igemm"), (1, "conv_wgrad")):
            tms, fl = C.c_double(), C.c_double()
            n = lib.dvd_prof_report(kind, C.byref(tms), C.byref(fl))     # totals of the instrumented (last timed) step
            res[name] = {"launches": int(n), "ms": tms.value,
                         "tflops": (fl.value / (tms.value * 1e-3) / 1e12) if tms.value else 0.0,
                         "avg_us": tms.value * 1e3 / max(n, 1), "gflop_per_launch": fl.value / max(n, 1) / 1e9}
        F = F_GFLOP_PER_CLIP.get((a.ch, a.frames, a.size))
        dom = res["conv_igemm"]
        peak = PEAK_BF16_TFLOPS if a.dtype == "bf16" else PEAK_F32_MFMA_TFLOPS
        roof = {"bound": "mfma", "kernel": "conv_halo_kernel + conv_igemm_kernel <bf16> (forward + backward-data convolutions)" if a.dtype == "bf16" else "conv_halo_kernel + conv_igemm_kernel <f32>",
                "achieved": round(dom["tflops"], 1), "peak": peak, "unit": "TFLOP/s",
                "frac": round(dom["tflops"] / peak, 4), "traffic": hbm_traffic("conv_igemm"),
                "launches_per_step": dom["launches"], "avg_launch_us": round(dom["avg_us"], 1),
                "gflop_per_launch": round(dom["gflop_per_launch"], 2), "kernel_ms_per_step": round(dom["ms"], 1),
                "wgrad": {k: round(v, 2) if isinstance(v, float) else v for k, v in res["conv_wgrad"].items()},
                "step_achieved": round(value / world * F / 1e3, 1) if F else None,
                "step_frac": round(value / world * F / 1e3 / peak, 4) if F else None}
    if rank == 0:
        out = {"metric": "clips/sec per G+Ds+Dt step, 48x64x64 UCF-101 synth" if a.size == 64 else f"clips/sec per G+Ds+Dt step, {a.frames}x{a.size}x{a.size} Kinetics-600-shaped synth", "value": round(value, 3), "unit": "clips/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 2),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
               "config": {"workload": f"{'UCF-101' if a.size == 64 else 'Kinetics-600'}-shaped {a.n_class}-class {a.frames}x{a.size}x{a.size} clips, G+Ds+Dt hinge step, ch={a.ch}, "
                                      f"k_sample={a.k_sample}, batch {a.batch}/GPU (BASELINE configs[{1 if a.size == 64 else 3}])",
                          "global_batch": gB, "parallelism": f"dp{world}"},
               "losses": [round(v, 4) for v in lossv],
               "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}
        if roof:
            out["roofline"] = roof
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
