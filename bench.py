#!/usr/bin/env python3
"""bench.py -- Newton-step throughput of the batched FOM / ROM hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config fom|pod_galerkin|pod_lspg|quadratic|ann|decoder_bf16]

Default = BASELINE.json configs[1], the configuration the metric is quoted on: B = 1024 (mu1, mu2) samples per GPU,
N = 1024 nodes, fp64, 500 implicit-Euler steps at dt = 0.025 (dt = 0.05 diverges at N = 1024 in the reference itself,
BASELINE.md section 2), u0 = 1, E = 0, mu ~ U[4.25,5.5] x U[0.015,0.03], rng seed 20251121.  The other --config values
are BASELINE.json configs[2..4]: each prints the same driver-contract line with its own roofline.
One bench "step" = one full pass of the workload (all time steps, every inner iteration of every sample of this rank).
Unit: the sample-Newton-step, one (sample, inner iteration) pair = one assembly + one solve (+ projection for the ROMs).

Multi-GPU: `python bench.py --gpus N` starts its own N rank processes (rank i -> device i, RCCL) BEFORE the parent makes
any GPU call; the driver's `python -m torch.distributed.run ... bench.py --gpus N` form (RANK in the environment) runs
the rank body directly.  The mu-sweep shards with no data-path collective (weak scaling, fixed work per GPU); barrier +
max-over-ranks timing; value = all ranks' units / that time.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(REPO, "1d-burgers-equation-roms_amd")
for _p in (REPO, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

SEED = 20251121
# Peaks.  HBM and the MFMA table: /opt/skills/guides/MI355X_MICROARCH.md.  fp64: that guide lists no fp64 figure;
# 78.6 TFLOP/s is the vendor's MI355X datasheet value for fp64 vector AND matrix, the denominator SURVEY.md 8d names
# (256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz = 78.6e12).
HBM_PEAK_GBS = 8000.0
FP64_PEAK_TF = 78.6
# measured on this part (tools/mfma_bench.hip): back-to-back v_mfma_f64_4x4x4_4b_f64 = 62-66 TFLOP/s chip-wide
# (16.5 cycles per 512-flop instruction), v_mfma_f64_16x16x4_f64 = 33 TFLOP/s.  The ROM reduce kernel issues the former.
FP64_MFMA_4X4X4_MEASURED_TF = 62.0
# FOM algorithmic work per mesh row and Picard iteration ("~60 fp64 flop/row incl. 2-3 divisions":
# assembly 26 + sequential Thomas 9 + update/norms 5 + divisions priced as rcp + Newton step): kernel-independent.
FOM_ALG_FLOP_PER_ROW = 60.0
# what fom_fused_kernel<16> actually executes per lane and iteration, read off the ISA (tools/asm_stats.py); the parallel
# solver (Wang partition + PCR) costs more flops than Thomas, which is why executed > algorithmic.
FOM_EXEC_FLOP_PER_LANE_ITER = {16: 1033}

CONFIGS = ("fom", "pod_galerkin", "pod_lspg", "quadratic", "ann", "decoder_bf16", "pod_r96_galerkin", "pod_r96_lspg")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed passes (default 10 for fom, 3 for the ROM configs)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", choices=CONFIGS, default="fom")
    ap.add_argument("--batch", type=int, default=None, help="samples per GPU (default: the config's own)")
    ap.add_argument("--n", type=int, default=None, help="mesh nodes (default 1024 for fom, 512 for the ROMs)")
    ap.add_argument("--time-steps", type=int, default=500)
    ap.add_argument("--dt", type=float, default=None)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of each CPU-baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak (the driver's contract): --batch samples PER GPU; strong: --batch is the GLOBAL sweep, split "
                         "evenly over the ranks (each rank's share may then take the low-latency workgroup-per-sample kernel)")
    ap.add_argument("--other-configs", choices=("auto", "full", "small", "none"), default="auto",
                    help="1-GPU fom run: append short runs of BASELINE configs[2..4] as other_configs.  auto = full for the "
                         "driver's default command line, none once a workload flag is given; small = 96 samples x 12 steps "
                         "(the contract test)")
    ap.add_argument("--other-steps", type=int, default=3, help="timed passes of each other_configs entry")
    args = ap.parse_args(argv)
    # the driver's own line (no workload flag given): configs[1] as the headline + one short run of every other config
    args.default_workload = (args.config == "fom" and args.batch is None and args.n is None and args.dt is None
                             and args.time_steps == 500)
    if args.other_configs == "auto":
        args.other_configs = "full" if args.default_workload else "none"
    rom = args.config != "fom"
    if args.steps is None:
        args.steps = 3 if rom else 10
    if args.warmup is None:
        args.warmup = 1 if rom else 2
    if args.n is None:
        args.n = 512 if rom else 1024
    if args.dt is None:
        args.dt = 0.05 if rom else 0.025
    if args.batch is None:
        args.batch = {"fom": 1024, "pod_galerkin": 4096, "pod_lspg": 4096, "quadratic": 1024, "ann": 2048,
                      "decoder_bf16": 2048, "pod_r96_galerkin": 1024, "pod_r96_lspg": 1024}[args.config]
    return args


# ------------------------------------------------------------------------------------------ rank launcher
def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """Parent of a self-launched multi-rank run.  Makes NO GPU call (torch.cuda.device_count() does not initialise
    the device): it starts one child per rank, relays rank 0's JSON line, and fails if any child fails."""
    import torch
    try:
        ndev = torch.cuda.device_count()
    except Exception:
        ndev = 0
    env = dict(os.environ, WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    # fewer devices than ranks (a rehearsal on a 1-GPU box): RCCL refuses two ranks per device -> gloo for the two
    # scalar reductions and the all-gather; ranks share devices round-robin
    env.setdefault("BG_DIST_BACKEND", "nccl" if ndev >= args.gpus else "gloo")
    procs = []
    for r in range(args.gpus):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=e,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=(r == 0)))
    out0 = None
    failed = None
    pending = set(range(args.gpus))
    while pending and failed is None:
        for r in sorted(pending):
            p = procs[r]
            if r == 0 and out0 is None and p.poll() is not None:
                out0 = p.stdout.read()
            if p.poll() is not None:
                pending.discard(r)
                if p.returncode != 0:
                    failed = (r, p.returncode)
        time.sleep(0.05)
    if failed is not None:
        for p in procs:                       # exact PIDs we started, never a pattern
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
        print(f"bench.py: rank {failed[0]} exited with code {failed[1]}", file=sys.stderr)
        return 1
    if out0 is None:
        out0 = procs[0].stdout.read()
    sys.stdout.write(out0)
    sys.stdout.flush()
    return 0


# ------------------------------------------------------------------------------------------ workloads
def mu_shard(total_per_rank, world, rank):
    """(mu1, mu2) of this rank: rank r owns samples [r*B, (r+1)*B) of the global sweep."""
    import numpy as np
    rng = np.random.default_rng(SEED)
    total = total_per_rank * world
    mu1 = rng.uniform(4.25, 5.5, total)
    mu2 = rng.uniform(0.015, 0.03, total)
    sl = slice(rank * total_per_rank, (rank + 1) * total_per_rank)
    return mu1[sl], mu2[sl]


def golden(name):
    import numpy as np
    return np.load(os.path.join(REPO, "tests", "golden", name))     # data fixtures (weights, bases), not code


def ann_model(g):
    import torch
    import torch.nn as nn
    dims = [5, 32, 64, 128, 256, 256, 91]                               # POD-ANN/pod_ann.py:38-56
    layers = []
    for i in range(6):
        lin = nn.Linear(dims[i], dims[i + 1])
        with torch.no_grad():
            lin.weight.copy_(torch.from_numpy(g[f"W{i}"])); lin.bias.copy_(torch.from_numpy(g[f"b{i}"]))
        layers.append(lin)
        if i < 5:
            layers.append(nn.ELU())
    return nn.Sequential(*layers).eval()


def decoder_model(g):
    import torch
    import torch.nn as nn
    dims = [3, 32, 64, 128, 160]                                        # Non-Instrusive/train_pod_ann.py:20,110-121
    layers = []
    for i, key in enumerate((0, 2, 4, 6)):
        lin = nn.Linear(dims[i], dims[i + 1])
        with torch.no_grad():
            lin.weight.copy_(torch.from_numpy(g[f"{key}_weight"])); lin.bias.copy_(torch.from_numpy(g[f"{key}_bias"]))
        layers.append(lin)
        if i < 3:
            layers.append(nn.ELU())
    return nn.Sequential(*layers).eval()


class Workload:
    """One BASELINE config on one rank: device-resident inputs, one_pass(), unit count, roofline, description."""
    unit = "sample-Newton-steps/s"
    dtype = "f64"

    def __init__(self, args, rank, world, dev):
        import numpy as np
        self.args, self.rank, self.world, self.dev = args, rank, world, dev
        if args.scaling == "strong" and not getattr(args, "_split", False):
            if args.batch % world:
                raise SystemExit(f"--scaling strong: --batch {args.batch} does not split evenly over {world} ranks")
            args.global_batch, args.batch, args._split = args.batch, args.batch // world, True
        self.mu1, self.mu2 = mu_shard(args.batch, world, rank)
        self.X = np.linspace(0.0, 100.0, args.n)
        self.extra = {}

    def build_training_bases(self, n_pod=None, n_quad=None):
        """This framework's own training sweep (3 x 3 grid of FEM/paper_training_stage.py:8-10, 500 steps at dt 0.05)
        -> snapshots -> device SVD / quadratic-manifold fit: the bases of configs[2] and configs[3]."""
        import numpy as np
        from burgers_hip import fom, pod
        m1, m2 = np.meshgrid(np.linspace(4.25, 5.5, 3), np.linspace(0.015, 0.03, 3), indexing="ij")
        res = fom.fom_run(self.X, np.ones(self.args.n), m1.ravel(), m2.ravel(), 0.05, 500, device=self.dev)
        S = pod.snapshot_matrix(res.hist).contiguous()
        out = []
        if n_pod:
            out.append(pod.pod_basis(S, n_modes=n_pod)[0].contiguous())
        if n_quad:
            Phi, H, _ = pod.build_quadratic_manifold(S, n_quad, alpha=1e-2)
            out += [Phi.contiguous(), H.contiguous()]
        return out

    def units(self, res):
        return int(res.iters.sum().item())

    def status(self, res):
        from burgers_hip import lib
        return {"nonfinite_samples": int((res.flags & lib.BG_FLAG_NONFINITE).ne(0).sum().item()),
                "hit_cap_samples": int((res.flags & lib.BG_FLAG_HIT_CAP).ne(0).sum().item())}


class FomWorkload(Workload):
    kind = "fom"

    def __init__(self, *a):
        super().__init__(*a)
        import torch
        from burgers_hip import fom
        args, dev = self.args, self.dev
        self.Xd = torch.as_tensor(self.X, device=dev)
        self.u0 = torch.ones((args.batch, args.n), dtype=torch.float64, device=dev)
        self.mu1d, self.mu2d = torch.as_tensor(self.mu1, device=dev), torch.as_tensor(self.mu2, device=dev)
        self.out = fom.FomResult(torch.empty((args.batch, args.time_steps + 1, args.n), dtype=torch.float64, device=dev),
                                 torch.empty((args.batch, args.time_steps), dtype=torch.int32, device=dev),
                                 torch.empty((args.batch,), dtype=torch.int32, device=dev))

    def one_pass(self):
        from burgers_hip import fom
        a = self.args
        return fom.fom_run(self.Xd, self.u0, self.mu1d, self.mu2d, a.dt, a.time_steps, device=self.dev, out=self.out,
                           validate_mesh=False)

    def describe(self):
        a = self.args
        return ("batched Newton-steps/sec over mu-sweep (sample-Newton-steps/s, FOM N=%d)" % a.n,
                "configs[1]: batched FOM, %d mu-samples/GPU x N=%d, fp64, %d implicit-Euler steps, dt=%g"
                % (a.batch, a.n, a.time_steps, a.dt))

    def roofline(self, units, kernel_s):
        a = self.args
        alg_tf = units * FOM_ALG_FLOP_PER_ROW * a.n / kernel_s / 1e12
        stream_gbs = units * 24.0 * a.n / kernel_s / 1e9
        r = {"bound": "fp64_valu", "achieved": alg_tf, "peak": FP64_PEAK_TF, "unit": "TFLOP/s", "frac": alg_tf / FP64_PEAK_TF,
             "kernel": "fom_fused_kernel", "kernel_ms_avg": kernel_s * 1e3,
             "algorithmic_flop_per_row_iteration": FOM_ALG_FLOP_PER_ROW,
             "algorithmic_flops_per_launch": units * FOM_ALG_FLOP_PER_ROW * a.n,
             "note": "the state stays in registers for the whole time loop, so the binding resource is fp64 VALU issue, not "
                     "HBM; achieved = SURVEY 8d's algorithmic 60 flop per row and iteration (sequential-Thomas count) x "
                     "rows x iterations / kernel time",
             # SURVEY 8d's streaming model (one HBM round trip of the state per Picard iteration), kept as a labelled
             # secondary: a fused kernel does not make those round trips, so this is NOT a fraction of a physical limit
             "streaming_model": {"equivalent_GBps": stream_gbs, "hbm_peak_GBps": HBM_PEAK_GBS,
                                 "ratio_to_hbm_peak": stream_gbs / HBM_PEAK_GBS, "bytes_per_sample_step": 24 * a.n}}
        rows = -(-a.n // 64)
        if rows in FOM_EXEC_FLOP_PER_LANE_ITER and a.n == 64 * rows:
            ex = units * FOM_EXEC_FLOP_PER_LANE_ITER[rows] * 64 / kernel_s / 1e12
            r["executed"] = {"TFLOP/s": ex, "frac_of_peak": ex / FP64_PEAK_TF,
                             "flop_per_lane_iteration_from_isa": FOM_EXEC_FLOP_PER_LANE_ITER[rows]}
        tr = measured_traffic(self.kind, a)
        r["traffic"] = tr["bytes"] if tr else None
        if tr:
            r["traffic_source"] = tr["source"]
        return r


class RomWorkload(Workload):
    """Common part of the projection-ROM configs: MFMA roofline with SURVEY 8d's algorithmic flop counts."""

    def roofline(self, units, kernel_s):
        tf = units * self.flops_per_step / kernel_s / 1e12
        tr = measured_traffic(self.kind, self.args)
        return {"bound": "mfma", "achieved": tf, "peak": FP64_PEAK_TF, "unit": "TFLOP/s", "frac": tf / FP64_PEAK_TF,
                "traffic": tr["bytes"] if tr else None, **({"traffic_source": tr["source"]} if tr else {}),
                "kernel": self.kernel_name, "pass_ms_avg": kernel_s * 1e3,
                "algorithmic_flops_per_step": self.flops_per_step,
                "measured_ceiling_of_the_instruction_used": {"instruction": "v_mfma_f64_4x4x4_4b_f64",
                                                             "TFLOP/s": FP64_MFMA_4X4X4_MEASURED_TF,
                                                             "frac": tf / FP64_MFMA_4X4X4_MEASURED_TF}}


class PodWorkload(RomWorkload):
    kernel_name = "rom_fused_kernel (bg_rom_run: one launch per pass, the whole time loop of a sample per workgroup)"

    def __init__(self, *a):
        super().__init__(*a)
        import torch
        self.kind = self.args.config
        self.proj = "Galerkin" if self.kind.endswith("galerkin") else "LSPG"
        self.wide = self.kind.startswith("pod_r96")
        if self.wide:
            # beyond BASELINE's configs: the thesis' tol 1e-04 basis, POD/modes/U_modes_tol_1e-04.npy (512 x 96),
            # the committed reference artefact itself (data fixture tests/golden/committed_pod_r96.npz)
            self.r = 96
            self.Phi = torch.as_tensor(golden("committed_pod_r96.npz")["Phi"], device=self.dev).contiguous()
            self.kernel_name = ("rom_wide_kernel (bg_rom_run_wide: one launch per pass, the whole time loop of a sample per workgroup; "
                                "basis streamed through LDS)")
        else:
            self.r = 40
            (self.Phi,) = self.build_training_bases(n_pod=self.r)
        N, r = self.args.n, self.r
        self.flops_per_step = 2 * N * r * r + 11 * N * r + (2 * r ** 3) / 3
        self.u0 = torch.ones((self.args.batch, N), dtype=torch.float64, device=self.dev)
        self.mu1d, self.mu2d = torch.as_tensor(self.mu1, device=self.dev), torch.as_tensor(self.mu2, device=self.dev)

    def one_pass(self):
        from burgers_hip import rom
        a = self.args
        return rom.pod_prom_run(self.X, self.u0, self.mu1d, self.mu2d, a.dt, a.time_steps, self.Phi, projection=self.proj,
                                device=self.dev)

    def describe(self):
        a = self.args
        tag = "beyond BASELINE (thesis basis U_modes_tol_1e-04)" if self.wide else "configs[2]"
        return ("batched Newton-steps/sec over mu-sweep (sample-Newton-steps/s, POD-%s ROM r=%d)" % (self.proj, self.r),
                "%s: POD-%s ROM r=%d, %d-sample batch/GPU, N=%d, fp64, %d steps, dt=%g, V^T J V on fp64 MFMA"
                % (tag, self.proj, self.r, a.batch, a.n, a.time_steps, a.dt))


class QuadWorkload(RomWorkload):
    kind = "quadratic"
    kernel_name = ("quad_fused_kernel (bg_quad_rom_run: one launch per pass, the whole time loop of four samples per "
                   "workgroup; tangent, decode, projection and solve fused)")

    def __init__(self, *a):
        super().__init__(*a)
        import torch
        self.r = 40
        self.Phi, self.H = self.build_training_bases(n_quad=self.r)
        N, r = self.args.n, self.r
        k = r * (r + 1) // 2
        self.k = k
        self.flops_per_step = 2 * N * (r + k) + 4 * N * k + 2 * N * r * r + 11 * N * r + (2 * r ** 3) / 3
        self.u0 = torch.ones((self.args.batch, N), dtype=torch.float64, device=self.dev)
        self.mu1d, self.mu2d = torch.as_tensor(self.mu1, device=self.dev), torch.as_tensor(self.mu2, device=self.dev)
        from burgers_hip import rom
        self.plan = rom.QuadFusedPlan(self.Phi, self.H, self.dev)      # operand copies of (Phi, H): once per basis

    def one_pass(self):
        from burgers_hip import rom
        a = self.args
        return rom.quadratic_run(self.X, self.u0, self.mu1d, self.mu2d, a.dt, a.time_steps, self.Phi, self.H,
                                 projection="LSPG", device=self.dev, plan=self.plan)

    def describe(self):
        a = self.args
        return ("batched Newton-steps/sec over mu-sweep (sample-Newton-steps/s, quadratic-manifold ROM r=%d)" % self.r,
                "configs[3]: quadratic-manifold LSPG ROM r=%d (k=%d), batched q(x)q decode, %d samples/GPU, N=%d, fp64, "
                "%d steps, dt=%g" % (self.r, self.k, a.batch, a.n, a.time_steps, a.dt))


class AnnWorkload(RomWorkload):
    kind = "ann"
    kernel_name = "rom_ann_fused_kernel (bg_ann_rom_run: one launch per pass, the whole time loop of a sample per workgroup)"
    dtype = "f64 (MLP closure f32, as the reference)"

    def __init__(self, *a):
        super().__init__(*a)
        import torch
        self.g = golden("ann_n5.npz")
        self.model = ann_model(self.g)
        N, n, nb = self.args.n, 5, 91
        self.flops_per_step = 2 * 132000 * (1 + n) + 2 * N * (n + nb) * (1 + n) + 2 * N * n * n + 11 * N * n
        self.u0 = torch.ones((self.args.batch, N), dtype=torch.float64, device=self.dev)
        self.mu1d, self.mu2d = torch.as_tensor(self.mu1, device=self.dev), torch.as_tensor(self.mu2, device=self.dev)
        # everything that does not depend on the batch is built once, outside the timed passes: the closure as
        # bg_ann_rom_run wants it (recognised-MLP probe, padded weight uploads) and the device-resident bases
        from burgers_hip import rom
        self.model = self.model.to(device=self.dev, dtype=torch.float32).eval()
        self.Up = torch.as_tensor(self.g["U_p"], device=self.dev)
        self.Us = torch.as_tensor(self.g["U_s"], device=self.dev)
        self.plan = rom._ann_fused_plan(self.model, n, nb, N, torch.float32, self.dev)
        assert self.plan is not None, "the committed closure must take the device-side loop (bg_ann_rom_run)"

    def one_pass(self):
        from burgers_hip import rom
        a = self.args
        return rom.pod_ann_run_fused(self.X, self.u0, self.mu1d, self.mu2d, a.dt, a.time_steps, self.Up, self.Us,
                                     self.model, rom.PROJ["lspg"], tol=1e-6, max_it=50, device=self.dev, plan=self.plan)

    def roofline(self, units, kernel_s):
        r = super().roofline(units, kernel_s)
        N, n, nb = self.args.n, 5, 91
        f32 = 2 * 132000 * (1 + n)
        r["note"] = ("mixed precision: %.0f %% of the algorithmic flops are the float32 closure (value + input-Jacobian, packed "
                     "VALU FMAs), the rest fp64 (decode + tangent sweep on the VALU, projection on v_mfma_f64_4x4x4); priced "
                     "against the fp64 peak like the other ROM configs" % (100.0 * f32 / self.flops_per_step))
        return r

    def describe(self):
        a = self.args
        return ("batched Newton-steps/sec over mu-sweep (sample-Newton-steps/s, intrusive POD-ANN ROM n=5, nbar=91)",
                "configs[4] (intrusive form): POD-ANN PROM, committed 5->32->64->128->256->256->91 ELU closure in f32, "
                "%d samples/GPU, N=%d, %d steps, dt=%g" % (a.batch, a.n, a.time_steps, a.dt))


class DecoderWorkload(Workload):
    """configs[4], decoder-only form (Non-Instrusive/predict_pod_ann.py:73-80): no Newton loop; one MLP evaluation and
    one dense contraction per (sample, time) column, bf16 weights/activations with fp32 accumulate."""
    kind = "decoder_bf16"
    unit = "decoded snapshot columns/s"
    dtype = "bf16"

    def __init__(self, *a):
        super().__init__(*a)
        import copy
        import torch
        self.g = golden("nonintrusive_decoder.npz")
        self.model = decoder_model(self.g)
        from burgers_hip import decoder
        self.Nt = self.args.time_steps + 1
        self.dec = decoder.GridDecoder(self.Nt, self.g["U_modes"], copy.deepcopy(self.model), self.g["mean"], self.g["std"],
                                       dtype=torch.bfloat16, device=self.dev)
        assert self.dec.plan is not None, "the committed decoder must take the one-kernel form (bg_decode_mlp_bf16)"
        self.mu1d, self.mu2d = torch.as_tensor(self.mu1, device=self.dev), torch.as_tensor(self.mu2, device=self.dev)
        self.chunk = 1024                                  # samples per contraction: the fp64 result block is 2.1 GB

    def one_pass(self):
        last = None
        for lo in range(0, self.args.batch, self.chunk):
            last = self.dec.predict(self.mu1d[lo:lo + self.chunk], self.mu2d[lo:lo + self.chunk])
        return last

    def units(self, res):
        return self.args.batch * self.Nt

    def status(self, res):
        import torch
        return {"nonfinite_samples": int((~torch.isfinite(res)).any(dim=2).any(dim=1).sum().item())}

    def describe(self):
        a = self.args
        return ("decoded snapshot columns/sec over mu-sweep (non-intrusive POD-ANN decoder, bf16)",
                "configs[4] (decoder-only form): 3->32->64->128->160 MLP + U_modes (512,160) contraction, bf16 weights/"
                "activations, fp32 accumulate, %d samples/GPU x %d time levels" % (a.batch, self.Nt))

    def roofline(self, units, kernel_s):
        per_col = 512 * 8 + 160 * 2                        # the fp64 snapshot column written + its bf16 coefficients read
        gbs = units * per_col / kernel_s / 1e9
        tr = measured_traffic(self.kind, self.args)         # per LAUNCH = per chunk of samples; a pass is batch / chunk launches
        nlaunch = -(-self.args.batch // self.chunk)
        return {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                "traffic": tr["bytes"] * nlaunch if tr else None,
                **({"traffic_source": tr["source"] + f"; x {nlaunch} launches per pass"} if tr else {}), "kernel": "decode_mlp_kernel<10, 8> (bg_decode_mlp_bf16: the bf16 MLP evaluated per 128-column workgroup "
                                           "inside the contraction kernel -- bitwise the PyTorch-ROCm bf16 module's coefficients -- then the "
                                           "bf16 MFMA contraction, float64 result written once; no activation or coefficient crosses HBM)",
                "pass_ms_avg": kernel_s * 1e3, "algorithmic_bytes_per_column": per_col}


WORKLOADS = {"fom": FomWorkload, "pod_galerkin": PodWorkload, "pod_lspg": PodWorkload, "quadratic": QuadWorkload,
             "ann": AnnWorkload, "decoder_bf16": DecoderWorkload, "pod_r96_galerkin": PodWorkload, "pod_r96_lspg": PodWorkload}


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_leg(w, res, timed):
    """The CPU-reference leg -- the ONLY place bench.py touches oracle/ (test infrastructure: the NumPy/SciPy and C
    restatements of the reference algorithm), outside the timed region.  Always: parity of a few samples of this
    rank's result against it (BASELINE.json's "rel-L2 vs CPU ref").  With ``timed`` (N=1 only): the same code on the
    host cores over a bounded sample of the workload = ``cpu_baseline`` (1-core sequential restatement,
    plus all host cores; core count and CPU model stated)."""
    import numpy as np
    import torch
    from oracle import burgers_ref as br, burgers_ref_c as bc
    a = w.args
    N, dt = a.n, a.dt
    ones = np.ones(N)
    nsub = min(4, a.batch)
    idx = np.unique(np.linspace(0, a.batch - 1, nsub).astype(int))
    par = {}
    budget = a.cpu_seconds

    def numpy_leg(run_one, what, steps):
        """Per-sample sequential restatement on ONE core, like the reference's own loop; stops at the time budget."""
        t0 = time.perf_counter(); its = 0; ns = 0
        for b in range(a.batch):
            its += int(run_one(b, steps).sum()); ns += 1
            if time.perf_counter() - t0 > budget:
                break
        t = time.perf_counter() - t0
        return {"value": its / t, "unit": "sample-Newton-steps/s", "cores": 1, "kind": "port",
                "sample": f"{ns} samples x first {steps} time steps of the same workload, {what}, sequential, {t:.1f} s"}

    if w.kind == "fom":
        ho, ito = bc.fom_run(w.X, ones, w.mu1[idx], w.mu2[idx], dt, a.time_steps)
        it_dev = torch.as_tensor(idx, device=res.hist.device)
        hg, ig = res.hist[it_dev].cpu().numpy(), res.iters[it_dev].cpu().numpy()
        par = {"rel_l2_vs_cpu_ref": float(np.linalg.norm(hg - ho) / np.linalg.norm(ho)),
               "iters_match_cpu_ref": bool(np.array_equal(ig, ito)),
               "parity_sample": f"{len(idx)} full trajectories vs the C restatement (oracle/burgers_ref_c.c)"}
        if not timed:
            return par, None
        steps = min(a.time_steps, 50)
        one = numpy_leg(lambda b, s: br.fom_burgers(w.X, dt, s, ones, w.mu1[b], 0.0, w.mu2[b], return_iters=True)[1],
                        "NumPy/SciPy restatement oracle/burgers_ref.py", steps)
        threads = bc.max_threads()
        bc.fom_run(w.X, ones, w.mu1[:2], w.mu2[:2], dt, 2)                       # warm the thread pool
        legs = [one]
        for nthr in (1, threads):
            nb, st = min(a.batch, max(nthr * 8, 16)), a.time_steps
            # scale the sample to the budget: about 1e4 sample-steps/s per core for this code at N = 1024
            while nb * st * 17 / (1.0e4 * (1024.0 / N) * nthr) > budget and st > 10:
                st //= 2
            t0 = time.perf_counter()
            _, it_cpu = bc.fom_run(w.X, ones, w.mu1[:nb], w.mu2[:nb], dt, st, nthreads=nthr)
            t = time.perf_counter() - t0
            legs.append({"value": float(it_cpu.sum() / t), "unit": "sample-Newton-steps/s", "cores": int(nthr), "kind": "port",
                         "sample": f"{nb} samples x first {st} time steps of the same workload, C restatement + OpenMP "
                                   f"(oracle/burgers_ref_c.c), {t:.1f} s"})
        cpu = dict(legs[-1])
        cpu["legs"] = legs
    elif w.kind in ("pod_galerkin", "pod_lspg", "quadratic", "ann", "pod_r96_galerkin", "pod_r96_lspg"):
        steps = min(a.time_steps, 10)
        if w.kind == "quadratic":
            Phi, H = w.Phi.cpu().numpy(), w.H.cpu().numpy()
            run = lambda b, s: br.pod_quadratic_manifold(w.X, dt, s, ones, w.mu1[b], 0.0, w.mu2[b], Phi, H, projection="LSPG",
                                                         return_iters=True)
            tol = 1e-9
        elif w.kind == "ann":
            g = w.g
            Ws = [g[f"W{i}"] for i in range(6)]; bs = [g[f"b{i}"] for i in range(6)]
            run = lambda b, s: br.pod_ann_prom(w.X, dt, s, ones, w.mu1[b], 0.0, w.mu2[b], g["U_p"], g["U_s"], Ws, bs,
                                               return_iters=True)
            tol = 5e-6                                     # the closure is evaluated in float32 on both sides
        else:
            Phi = w.Phi.cpu().numpy()
            run = lambda b, s: br.pod_prom_burgers(w.X, dt, s, ones, w.mu1[b], 0.0, w.mu2[b], Phi, projection=w.proj,
                                                   return_iters=True)
            tol = 1e-10
        worst, same = 0.0, True
        for b in idx:
            U, ito = run(b, steps)
            hg = res.hist[b, :steps + 1].cpu().numpy().T
            worst = max(worst, float(np.linalg.norm(hg - U) / np.linalg.norm(U)))
            same = same and bool(np.array_equal(res.iters[b, :steps].cpu().numpy(), ito))
        par = {"rel_l2_vs_cpu_ref": worst, "iters_match_cpu_ref": same, "parity_tolerance": tol,
               "parity_sample": f"{len(idx)} samples x first {steps} time steps vs the NumPy restatement (oracle/burgers_ref.py)"}
        if not timed:
            return par, None
        cpu = numpy_leg(lambda b, s: run(b, s)[1], "NumPy/SciPy restatement oracle/burgers_ref.py", min(a.time_steps, 25))
    else:                                                   # decoder: bf16 tier measured against the fp32 reference path
        from burgers_hip import decoder
        g = w.g
        Ws = [g[f"{k}_weight"] for k in (0, 2, 4, 6)]; bs = [g[f"{k}_bias"] for k in (0, 2, 4, 6)]
        lo = (a.batch - 1) // w.chunk * w.chunk             # `res` is the last chunk of the pass
        worst = 0.0
        t0 = time.perf_counter()
        for j in range(min(3, res.shape[0])):
            Uo = br.predict_on_fom_grid(w.mu1[lo + j], w.mu2[lo + j], w.Nt, g["U_modes"], Ws, bs, g["mean"], g["std"])
            worst = max(worst, float(np.linalg.norm(res[j].cpu().numpy() - Uo) / np.linalg.norm(Uo)))
        t = time.perf_counter() - t0
        par = {"rel_l2_vs_cpu_ref": worst, "parity_tolerance": 2e-2,
               "parity_sample": "3 samples of the bf16 result vs the float32 NumPy restatement (bf16 keeps 8 significant bits)"}
        if not timed:
            return par, None
        cpu = {"value": 3 * w.Nt / t, "unit": w.unit, "cores": 1, "kind": "port",
               "sample": f"3 samples x {w.Nt} time levels, NumPy restatement oracle/burgers_ref.py (float32 MLP), {t:.1f} s"}
    cpu["cpu_model"] = cpu_model()
    cpu["host_cores"] = os.cpu_count()
    cpu["reference_as_written"] = ("Python element loops, 1 core, survey container (BASELINE.md section 2): FOM N=1024 about 8, "
                                   "POD r=40 16, quadratic n=21 14, POD-ANN 12.5 Newton-steps/s")
    return par, cpu


def measured_traffic(kind, args):
    """HBM bytes per launch from the committed rocprofv3 --pmc summary (separate FETCH_SIZE / WRITE_SIZE passes,
    FETCH doubled per MI355X_MICROARCH section HBM) -- only when it was collected on THIS configuration."""
    if kind != "fom":                                      # ROM configs: profiles/rom_pmc_summary.json, full bench sizes only
        default = parse_args(["--config", kind])
        if (args.batch, args.n, args.time_steps, args.dt) != (default.batch, default.n, default.time_steps, default.dt):
            return None
        try:
            with open(os.path.join(REPO, "profiles", "rom_pmc_summary.json")) as f:
                d = json.load(f)
            rec = d["configs"][kind]
        except Exception:
            return None
        return {"bytes": rec["hbm_bytes_per_launch"],
                "source": "profiles/rom_pmc_summary.json (static: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same "
                          "configuration, round %s: %.3g B fetched + %.3g B written by %s; not re-measured in this run)"
                          % (d.get("round", "?"), rec["fetch_bytes_per_launch"], rec["write_bytes_per_launch"], rec["kernel"][:40])}
    path = os.path.join(REPO, "profiles", "fom_pmc_summary.json")
    try:
        with open(path) as f:
            d = json.load(f)
    except Exception:
        return None
    cfg = d.get("config", {"config": "fom", "batch": 1024, "n": 1024, "time_steps": 500, "dt": 0.025})
    if kind != cfg.get("config") or (args.batch, args.n, args.time_steps, args.dt) != (
            cfg.get("batch"), cfg.get("n"), cfg.get("time_steps"), cfg.get("dt")):
        return None
    return {"bytes": d.get("hbm_bytes_per_launch"),
            "source": "profiles/fom_pmc_summary.json (static: rocprofv3 --pmc passes of this same command, round %s; not "
                      "re-measured in this run)" % d.get("round", "?")}


# ------------------------------------------------------------------------------------------ rank body
def allgather_svd_ms(w, res, dist_mod, backend):
    """The one exchange of the multi-GPU design, timed outside the throughput region: all-gather of a per-rank block
    of snapshots (POD/pod.py:80-84 stacks the same blocks from .npy files) + the POD basis of the gathered block."""
    import torch
    from burgers_hip import dist as bdist, pod
    ns = min(8, res.hist.shape[0])
    block = res.hist[:ns, ::5].contiguous()                         # 8 samples x every 5th time level per rank
    if backend != "nccl":
        block = block.cpu()
    torch.cuda.synchronize(w.dev)
    dist_mod.barrier()
    t0 = time.perf_counter()
    allb = bdist.all_gather_blocks(block, ns * w.world)
    if allb.device.type == "cuda":
        torch.cuda.synchronize(w.dev)
    t1 = time.perf_counter()
    S = pod.snapshot_matrix(allb.to(w.dev)).contiguous()
    U, s, _ = pod.pod_basis(S, epsilon_squared=1e-6)
    torch.cuda.synchronize(w.dev)
    t2 = time.perf_counter()
    return {"allgather_ms": (t1 - t0) * 1e3, "svd_ms": (t2 - t1) * 1e3, "allgather_svd_ms": (t2 - t0) * 1e3,
            "gathered_bytes": int(allb.numel() * 8), "snapshots": int(S.shape[1]), "modes_at_1e-6": int(U.shape[1]),
            "backend": backend}


def timed_passes(w, warmup, steps, barrier):
    """``warmup`` untimed passes, then exactly ``steps`` passes bracketed by barrier + synchronize on both sides.
    Returns (last result, wall seconds of the timed region, per-pass ms by HIP events on the launch stream)."""
    import torch
    res = None
    for _ in range(warmup):
        res = w.one_pass()
    barrier()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for e0, e1 in evs:
        e0.record()                      # our kernels are launched on torch's current stream (lib.stream_ptr)
        res = w.one_pass()
        e1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    return res, elapsed, [e0.elapsed_time(e1) for e0, e1 in evs]


def other_config_entries(args, dev, barrier):
    """The default 1-GPU run also measures BASELINE configs[2..4]: each --config workload at its own
    full size for a few passes, same timing protocol as the headline, parity of a sample subset against the CPU
    restatement outside the timed region.  One compact entry each; the full line of a config is `--config NAME`."""
    import gc
    import numpy as np
    import torch
    out = []
    for cfg in CONFIGS[1:]:
        t_all = time.perf_counter()
        try:
            small = ["--batch", "96", "--time-steps", "12"] if args.other_configs == "small" else []
            a2 = parse_args(["--config", cfg, "--warmup", "1", "--steps", str(args.other_steps)] + small)
            w = WORKLOADS[cfg](a2, 0, 1, dev)
            res, elapsed, pass_ms = timed_passes(w, a2.warmup, a2.steps, barrier)
            units = w.units(res)
            par, _ = cpu_leg(w, res, timed=False)
            roof = w.roofline(units, float(np.mean(pass_ms)) * 1e-3)
            entry = {"config": {"workload": w.describe()[1], "name": cfg, "global_batch": a2.batch, "units_per_pass": units},
                     "value": units * a2.steps / elapsed, "unit": w.unit, "ms_per_step": elapsed / a2.steps * 1e3,
                     "steps": a2.steps, "warmup": a2.warmup, "dtype": w.dtype,
                     "roofline": {k: roof.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel")}}
            entry.update(par)
            entry.update(w.status(res))
        except Exception as e:                               # the headline line must survive a failing side run
            entry = {"config": {"name": cfg}, "error": f"{type(e).__name__}: {e}"}
        w = res = None
        gc.collect()
        torch.cuda.empty_cache()
        entry["wall_s_incl_setup_and_parity"] = time.perf_counter() - t_all
        out.append(entry)
        print(f"bench.py: other_configs[{cfg}] done in {entry['wall_s_incl_setup_and_parity']:.1f} s", file=sys.stderr)
    return out


def run_rank(args):
    # Exactly ONE line may reach stdout, and libraries below us write there too (gloo announces its connections on fd 1):
    # everything printed while the rank runs goes to stderr, the JSON line alone to the real stdout.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        return _run_rank(args, real_stdout)
    finally:
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        os.close(real_stdout)


def _run_rank(args, real_stdout):
    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        return 2
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback for the product path)"
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", local_rank % ndev)
    torch.cuda.set_device(dev)
    dist = None
    backend = os.environ.get("BG_DIST_BACKEND", "nccl" if ndev >= world else "gloo")
    red_dev = dev if backend == "nccl" else torch.device("cpu")
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from burgers_hip import lib
    lib.load()
    w = WORKLOADS[args.config](args, rank, world, dev)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    res, elapsed, pass_ms = timed_passes(w, args.warmup, args.steps, barrier)

    units = w.units(res)                                     # units in one pass, this rank
    el = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    tot = torch.tensor([float(units)], dtype=torch.float64, device=red_dev)
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    elapsed, total_units = float(el.item()), float(tot.item())
    gather = None
    if dist is not None and args.config == "fom":
        gather = allgather_svd_ms(w, res, dist, backend)

    others = None
    if rank == 0 and args.gpus == 1 and args.config == "fom" and args.other_configs != "none":
        # before the CPU legs: half a minute of host-only work lets the GPU's clocks fall, and the first config after it
        # measured 15 % low
        others = other_config_entries(args, dev, barrier)
    if rank == 0:
        value = total_units * args.steps / elapsed
        metric, workload = w.describe()
        par, cpu = cpu_leg(w, res, timed=(args.gpus == 1 and not args.no_cpu_baseline))
        line = {"metric": metric, "value": value, "unit": w.unit, "n_gpus": args.gpus, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
                "scaling": args.scaling, "vs_baseline": None, "dtype": w.dtype, "data": "synthetic",
                "config": {"workload": workload, "global_batch": args.batch * args.gpus,
                           "parallelism": "mu-shard x%d, no data-path collective" % args.gpus,
                           "units_per_pass": total_units, "seed": SEED},
                "batched_steps_per_s": value / (args.batch * args.gpus)}
        line.update(par)
        line.update(w.status(res))
        line["roofline"] = w.roofline(units, float(np.mean(pass_ms)) * 1e-3)
        if cpu is not None:
            line["cpu_baseline"] = cpu
        if gather is not None:
            line["allgather_svd"] = gather
            line["allgather_svd_ms"] = gather["allgather_svd_ms"]
        if others is not None:
            line["other_configs"] = others
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "RANK" not in os.environ:
        return launch_ranks(args, argv)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
