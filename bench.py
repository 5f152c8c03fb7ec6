#!/usr/bin/env python
"""Headline benchmark: ADMM iterations/sec (and PSNR) on the BASELINE.json config-2 workload --
batch-8 3x1024x1024 RGB deconvolution, sum_squares(conv(x,psf)-b) + norm1(grad_H x) + norm1(grad_W x),
FFT-diagonalised x-update + L1-TV prox -- through the drop-in `dprox` API on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is ONE ADMM iteration over one per-GPU batch of 8 images, inputs resident in HBM.  With N
GPUs every rank solves its own independent batch of 8 (weak scaling, no data-path collective: the
images of a batch never interact); `value` = N*K / max-over-ranks time.

The JSON line also carries
  roofline     : dominant kernel's algorithmic bytes per launch / its average duration measured with
                 HIP events on the launch stream (second pass of K iterations with the library's
                 per-kernel timers on), against 8 TB/s HBM3E;
  roofline_iteration : the whole iteration against HBM peak on the 36 B/element the two-kernel schedule
                 really moves (`frac`); SURVEY 8(d)'s 64 B/element un-fused accounting is kept as a labelled
                 comparison only (`survey_accounting_frac`);
  cpu_baseline : the reference-schedule CPU oracle (oracle/, a port of the reference's PyTorch-CPU
                 path pinned on its golden vectors) timed on this box's host cores on a bounded sample;
  parity_rel_l2: rel-L2 between the GPU iterate and that same oracle run (same images, same iteration count);
  headline_protocol / steady_state / value_after_idle_gpu : which leg `value` is (W warm-up + K timed steps directly behind the
                 200-step steady-state leg), the 200-step figure, and the same K-step region on a GPU that idled first;
  cold_solve   : wall clock of solve(max_iter=50) on a freshly compiled solver (nothing cached) against a warm one and against 50
                 steady iterations, with the per-kernel times of everything a cold solve launches besides its iterations;
  psnr_db_detail : the quality leg on a detail-scaled synthetic (the SURVEY generator barely degrades under the blur at 1024^2);
  configs      : short timed runs of BASELINE.json's other configurations (1, 3, 4, 5) through the same API
                 with their own roofline figures (N=1 only, `--no-extra-configs` skips them).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "delta-prox_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK = 8.0e12          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
HBM_COPY = 6.29e12
B, C, H, W = 8, 3, 1024, 1024
RHO, LAM = 0.1, 0.005
# algorithmic HBM bytes per element moved by each kernel of the iteration (fp32, N = B*C*H*W elements);
# DESIGN.md section "kernels" derives them.  SURVEY 8(d) whole-iteration figure: 64 B/element.
KERNEL_BYTES_PER_ELEM = {
    "k_rhs": 24.0,            # reads v0,v1,u0,u1,K^T b ; writes rhs
    "k_rows_r2c": 8.0,        # reads rhs ; writes half spectrum
    "k_cols": 8.0,            # reads + writes half spectrum
    "k_rows_c2r": 8.0,        # reads half spectrum ; writes x
    "k_zupdate": 28.0,        # reads x,u0,u1 ; writes v0,v1,u0,u1
    "k_iter_rows_seq": 24.0,  # streaming fused rows (wave per band): same algorithmic traffic as k_iter_rows
    "k_iter_rows": 24.0,      # fused rows: reads spectrum, u0, u1 ; writes u0, u1, spectrum (x, v stay on chip)
    "k_cols_p2": 12.0,        # column solve: reads spectrum + data spectrum ; writes spectrum (denominators are L2-resident)
    "k_rows_r2c_p2": 8.0,
    "k_rows_c2r_p2": 8.0,
}
ITER_BYTES_PER_ELEM = 64.0          # 16 fp32 passes per element for the un-fused 5-kernel schedule
DESIGN_BYTES_PER_ELEM = 36.0        # what the two-kernel iteration actually has to move (k_cols_p2 12 + k_iter_rows 24)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true")
    ap.add_argument("--pmc", dest="pmc", action="store_true", default=None, help="measure roofline.traffic in this run: two rocprofv3 --pmc passes "
                    "(FETCH_SIZE, WRITE_SIZE; --kernel-trace only) of this same command as subprocesses (default at N = 1)")
    ap.add_argument("--no-pmc", dest="pmc", action="store_false", help="roofline.traffic from the committed profiles/ figure instead (labelled)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)      # (the counter passes' own invocation: the timed workload only)
    ap.add_argument("--chains", type=int, default=0, help="sub-batch chains of the two-kernel iteration (0 = the library's own choice, 1 = off: "
                    "every launch has the GPU to itself -- the setting the committed rocprofv3 kernel statistics are taken with)")
    return ap.parse_args()


def make_problem(dp, synthetic, rank, device):
    rng = np.random.RandomState(2023 + rank)
    gt = synthetic.synth(rng, B, C, H, W)
    psf = synthetic.point_spread_function(15, 5.0)
    gt_d = torch.from_numpy(gt).to(device)
    blur = dp.conv(dp.Variable(), psf).to(device)          # blur through the backend's own conv
    noise = torch.from_numpy((rng.randn(B, C, H, W) * (2.0 / 255.0)).astype(np.float32)).to(device)
    b = (blur.forward(gt_d) + noise).contiguous()
    x = dp.Variable()
    fns = dp.sum_squares(dp.conv(x, psf) - b) + dp.norm1(dp.grad(x, dim=0)) + dp.norm1(dp.grad(x, dim=1))
    solver = dp.compile(fns, method="admm", device=device)
    return solver, x, b, gt_d, psf


def psnr_per_image(out, gt):
    mse = ((out - gt) ** 2).reshape(out.shape[0], -1).mean(dim=1)
    return (10.0 * torch.log10(1.0 / mse)).tolist()


def timing_report(be):
    buf = ctypes.create_string_buffer(1 << 16)
    be.lib().call("dpx_timing_report", buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        name, cnt, tot = line.split()
        out[name] = (int(cnt), float(tot))
    return out


SETUP_NOT = ("k_cols_p2", "k_iter_rows", "k_iter_rows_seq")


def setup_profile(dp, be, b, psf, device):
    """per-kernel times (HIP events on the dispatch packets) of everything a COLD solve launches besides its iterations: a freshly
    compiled solver (no OTF / denominator table, no data spectrum cached) runs ONE iteration with the timers on"""
    x = dp.Variable()
    fns = dp.sum_squares(dp.conv(x, psf) - b) + dp.norm1(dp.grad(x, dim=0)) + dp.norm1(dp.grad(x, dim=1))
    s = dp.compile(fns, method="admm", device=device)
    torch.cuda.synchronize()
    be.lib().call("dpx_timing_enable", 1)
    timing_report(be)
    s.solve(x0=b, rhos=RHO, lams=LAM, max_iter=1)
    torch.cuda.synchronize()
    rep = timing_report(be)
    be.lib().call("dpx_timing_enable", 0)
    ks = {k: {"launches": c, "total_us": 1e3 * t} for k, (c, t) in rep.items() if k not in SETUP_NOT}
    tables = sum(v["total_us"] for k, v in ks.items() if k.startswith(("k_psf2otf", "k_rows_r2c_f64", "k_cols_fwd_f64", "k_twiddle", "k_denominator")))
    return {"kernels": ks, "setup_kernels_ms": sum(v["total_us"] for v in ks.values()) / 1e3,
            "tables_and_data_spectrum_ms": tables / 1e3,
            "note": "kernels of a cold solve other than the iteration's two: OTF / |OTF|^2 / denominator tables (k_psf2otf*, "
                    "k_denominator_pack), the fp64 data spectrum (k_twiddle_table_f64, k_rows_r2c_f64, k_cols_fwd_f64), initialize() "
                    "(v = K x0: k_grad, zeros) and the seed pass (k_seed_rows); torch's own fill / copy kernels are not in this list"}


def pmc_traffic(kernel, timeout_s=150):
    """HBM traffic per launch of `kernel`, measured NOW: two rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE need separate passes:
    3 + 2 of the 4 TCC slots; --pmc with --kernel-trace only, as the box's rules ask) of this same bench command -- one chain, no
    companion legs, 10 steps -- as subprocesses; per launch: FETCH_SIZE [KiB] x 1024 x 2 (gfx950: wide coalesced reads are tallied at
    half their bytes, MI355X_MICROARCH.md "HBM") + WRITE_SIZE [KiB] x 1024.  None (with the reason) if rocprofv3 is missing / fails."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    vals = {}
    tmp = tempfile.mkdtemp(prefix="dpx_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    env.pop("DPX_CHAINS", None)
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            cmd = [exe, "--pmc", ctr, "--kernel-trace", "-d", d, "-o", "p", "--output-format", "csv", "--", sys.executable, os.path.abspath(__file__),
                   "--pmc-child", "--no-pmc", "--no-cpu-baseline", "--no-extra-configs", "--steps", "10", "--warmup", "2", "--chains", "1"]
            pr = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, start_new_session=True)
            try:
                _, err = pr.communicate(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(pr.pid, signal.SIGKILL)
                pr.wait()
                return None, f"rocprofv3 --pmc {ctr} pass exceeded {timeout_s} s"
            if pr.returncode != 0:
                return None, f"rocprofv3 --pmc {ctr} pass failed ({pr.returncode}): {err.decode(errors='replace')[-300:]}"
            groups = {}                                    # full instantiation name -> values (the x-only last pass of solve() is another
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):      # instantiation of the same kernel: 8 B / element)
                for row in csv.DictReader(open(f)):
                    full = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("dpx::", "")
                    if full.split("<")[0] == kernel and row["Counter_Name"] == ctr:
                        groups.setdefault(full, []).append(float(row["Counter_Value"]))
            if not groups:
                return None, f"no {ctr} rows for {kernel} in the counter CSV"
            inst, got = max(groups.items(), key=lambda kv: len(kv[1]))      # the instantiation the timed iterations launch
            vals[ctr] = (sum(got) / len(got), len(got), inst)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    rd, wr = vals["FETCH_SIZE"][0] * 1024 * 2, vals["WRITE_SIZE"][0] * 1024
    return {"bytes": rd + wr, "read_bytes_corrected": rd, "write_bytes": wr, "launches_sampled": min(vals["FETCH_SIZE"][1], vals["WRITE_SIZE"][1]),
            "instantiation": vals["FETCH_SIZE"][2]}, None


def cache_sweep(dp, synthetic, device):
    """HBM or Infinity Cache?  The two spectra handed between the two kernels of an iteration are 2 x 100 MB at B = 8 -- the 256 MB
    Infinity Cache can hold them -- and FETCH_SIZE / WRITE_SIZE count fabric requests, cache hits included.  The same iteration at
    B = 8 / 16 / 24 (hand-over 200 / 400 / 600 MB: beyond the cache from 16 on) in ps per pixel and iteration: the growth from 8 to 16
    is what the cache contributes at B = 8; from 16 on every byte crosses the HBM interface."""
    out = {}
    for nb in (8, 16, 24):
        rng = np.random.RandomState(77)
        bb = torch.from_numpy(rng.rand(nb, C, H, W).astype(np.float32)).to(device)
        x = dp.Variable()
        s = dp.compile(dp.sum_squares(dp.conv(x, synthetic.point_spread_function(15, 5.0)) - bb) + dp.norm1(dp.grad(x, dim=0)) + dp.norm1(dp.grad(x, dim=1)),
                       method="admm", device=device)
        t = {}
        for n in (20, 120):
            s.solve(x0=bb, rhos=RHO, lams=LAM, max_iter=n)
            best = None
            for _ in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                s.solve(x0=bb, rhos=RHO, lams=LAM, max_iter=n)
                torch.cuda.synchronize()
                dtn = time.perf_counter() - t0
                best = dtn if best is None else min(best, dtn)
            t[n] = best
        per_it = (t[120] - t[20]) / 100
        npx = nb * C * H * W
        out[f"B{nb}"] = {"ms_per_iter": per_it * 1e3, "ps_per_pixel": per_it * 1e12 / npx, "handover_MB": 2 * npx * 4 / 1e6,
                         "frac_of_hbm_peak_on_36B": DESIGN_BYTES_PER_ELEM * npx / per_it / HBM_PEAK}
        del s, bb
        torch.cuda.empty_cache()
    p8, p16 = out["B8"]["ps_per_pixel"], out["B16"]["ps_per_pixel"]
    out["statement"] = (f"{p8:.2f} ps/pixel at B = 8 against {p16:.2f} at B = 16 and {out['B24']['ps_per_pixel']:.2f} at B = 24: "
                        f"{100 * (p16 - p8) / p16:.0f} % of the B = 8 rate comes from spectrum hand-overs served by the Infinity Cache; the "
                        "roofline fractions of this line are fractions of the 8 TB/s HBM peak on bytes that cross the L2 (fabric requests), "
                        "not proven DRAM bytes -- `frac_of_hbm_peak_on_36B` at B = 16 / 24 is the cache-free figure")
    return out


def cpu_baseline(b_host, psf, n_iters=4, sample_b=2):
    """reference-schedule oracle on the host cores: `sample_b` of the 8 images, 1 warm-up + n timed iterations"""
    import oracle as O
    bs = b_host[:sample_b].contiguous()
    terms = [O.sum_squares(O.lin_conv(psf).minus(bs)), O.norm1(O.lin_grad(0)), O.norm1(O.lin_grad(1))]
    stamps = []
    x_ref = O.solve(terms, "admm", x0=bs, rhos=RHO, lams=LAM, max_iter=1 + n_iters,
                    callback=lambda **kw: stamps.append(time.perf_counter()))
    per_iter = (stamps[-1] - stamps[0]) / n_iters            # first iteration = warm-up (OTF build, caches)
    its_batch8 = (sample_b / B) / per_iter                   # iterations/s of a batch-8 problem
    return x_ref, {"value": its_batch8, "unit": "it/s (batch of 8x3x1024x1024)", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{sample_b} of the {B} images, 1 warm-up + {n_iters} timed ADMM iterations of the reference-schedule "
                      f"oracle ({per_iter:.2f} s/iter), scaled x{sample_b}/{B} to the batch-8 rate"}


def _timed(fn, n, rounds=2):
    """seconds per call over n back-to-back calls after one warm-up call; the fastest of `rounds` such rounds (one stall of the caching
    allocator -- the 755 MB history of config 5 -- or a clock ramp in a round would otherwise double or triple a small figure)"""
    fn()
    best, out = None, None
    for _ in range(rounds):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            out = fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        best = dt if best is None else min(best, dt)
    return best, out


def extra_configs(dp, synthetic, device):
    """BASELINE.json configurations 1, 3, 4 (one GPU's shard), 5 through the same drop-in API: wall-clock per iteration /
    step with the roofline figure SURVEY 8(d) names for each.  Short runs (a few seconds in total)."""
    from dprox.linalg import LinearSolveConfig
    from dprox.proxfn.pnp.denoisers import FFDNetColorDenoiser, FFDNetDenoiser
    from dprox.contrib import masked_fft
    from dprox.utils import ifft2
    out = {}
    psnr = lambda a_, b_: float(10 * torch.log10(1.0 / ((a_ - b_) ** 2).mean()))
    # ---- config 1: 1x1x256x256 TV deconvolution, 20 iterations (launch-latency-bound)
    gt, b1, psf = synthetic.deconv_case(1, 1, 256, 256, seed=2023)
    bt, x = torch.from_numpy(b1).to(device), dp.Variable()
    s = dp.compile(dp.sum_squares(dp.conv(x, psf) - bt) + dp.norm1(dp.grad(x, dim=0)) + dp.norm1(dp.grad(x, dim=1)), method="admm", device=device)
    dt, o = _timed(lambda: s.solve(x0=bt, rhos=0.1, lams=0.005, max_iter=20), 20)
    out["config1"] = {"workload": "1x1x256x256 TV-deconv, ADMM 20 it", "ms_per_solve": dt * 1e3, "it_per_s": 20 / dt,
                      "psnr_db": [psnr(bt.cpu(), torch.from_numpy(gt)), psnr(o.cpu(), torch.from_numpy(gt))],
                      "roofline": {"bound": "hbm", "bytes_per_iter": 36.0 * 65536, "frac": 20 / dt * 36.0 * 65536 / HBM_PEAK,
                                   "note": "2.4 MB per iteration is cache resident: launch-latency-bound, reported only"}}
    # ---- the same data term on the other fused solvers / plane sizes (not BASELINE configurations; steady-state ms per iteration from
    #      the difference between a 10- and a 40-iteration solve)
    def per_iter(solver, x0, **kw):
        a, _ = _timed(lambda: solver.solve(x0=x0, max_iter=10, **kw), 2)
        c, _ = _timed(lambda: solver.solve(x0=x0, max_iter=40, **kw), 2)
        return (c - a) / 30
    other = {}
    for tag, shape, method in (("hqs_8x3x1024x1024", (B, C, H, W), "hqs"), ("admm_vxu_8x3x1024x1024", (B, C, H, W), "admm_vxu"),
                               ("pgd_8x3x1024x1024", (B, C, H, W), "pgd"),
                               ("admm_8x3x768x1024", (B, C, 768, 1024), "admm"), ("admm_8x3x768x768", (B, C, 768, 768), "admm"),
                               ("admm_8x3x1000x1000", (B, C, 1000, 1000), "admm"),
                               ("admm_8x3x1024x1024", (B, C, H, W), "admm"), ("admm_1x3x1024x1024", (1, C, H, W), "admm"),
                               ("admm_1x3x768x1024", (1, C, 768, 1024), "admm")):
        gto, bo, psfo = synthetic.deconv_case(*shape, seed=2023)
        bo, x = torch.from_numpy(bo).to(device), dp.Variable()
        reg = dp.norm1(x) if method == "pgd" else dp.norm1(dp.grad(x, dim=0)) + dp.norm1(dp.grad(x, dim=1))
        s = dp.compile(dp.sum_squares(dp.conv(x, psfo) - bo) + reg, method=method, device=device)
        dt = per_iter(s, bo, rhos=0.8 if method == "pgd" else 0.1, lams=0.005)
        npx = float(np.prod(shape))
        other[tag] = {"ms_per_iter": dt * 1e3, "it_per_s": 1 / dt, "ps_per_pixel": dt * 1e12 / npx}
        del s, bo
    # strong scaling of config 2 predicted from one GPU: 8 ranks hold one image each (dprox.distributed deals contiguous batch slices, no
    # collective inside the iteration), so the batch advances at the rate of ONE 1 x 3 x 1024 x 1024 problem per GPU
    other["config2_predicted_speedup_8_gpus"] = other["admm_8x3x1024x1024"]["ms_per_iter"] / other["admm_1x3x1024x1024"]["ms_per_iter"]
    other["config2_shard_note"] = ("admm_1x3x1024x1024 = one rank's shard of the batch of 8 on 8 GPUs (row pass: k_iter_rows_par, the rows of a band side "
                                   "by side in one 16-wave workgroup; bit-identical to the batch run); predicted 8-GPU speed-up = ms per iteration of the "
                                   "whole batch on one GPU / ms per iteration of the shard, both steady-state (40- minus 10-iteration solves); "
                                   "admm_1x3x768x1024 is the reference's own example (examples/applications/deconv.py:1-16)")
    other["note"] = ("hqs: the two-kernel ADMM iteration with DPX_TERM_NO_DUAL (no-dual row kernel, 20 B per pixel); admm_vxu: the same two kernels "
                     "with DPX_TERM_VXU (the planes carry q = u' - v); pgd: dpx_pgd_run (2 launches per iteration, 28 B per pixel); "
                     "768 x 1024 (the reference's example image): column length 3 x 256 on the register-radix path (fft_reg_x3), two-kernel "
                     "iteration; 768 x 768: two-kernel iteration, 384-point rows on one wave; 1000 x 1000: off the register-radix path -- size-generic in-place LDS transforms + merged z / rhs pass, 4 launches per iteration.  hqs, admm_vxu, pgd and the 768 x 1024 ADMM run as two "
                     "sub-batch chains on two streams like the headline (the staged kernels as one)")
    out["other_paths"] = other
    # ---- config 3: config 2's data term + deep_prior(FFDNet-colour, seeded weights), 30 iterations
    rng = np.random.RandomState(2023)
    gt3 = torch.from_numpy(synthetic.synth(rng, B, C, H, W)).to(device)
    b3 = (dp.conv(dp.Variable(), psf).to(device).forward(gt3) + torch.from_numpy((rng.randn(B, C, H, W) * 2 / 255).astype(np.float32)).to(device)).contiguous()
    x = dp.Variable()
    prior = dp.deep_prior(x, denoiser=FFDNetColorDenoiser(synthetic.ffdnet_weights(7)))
    s = dp.compile(dp.sum_squares(dp.conv(x, psf) - b3) + prior, method="admm", device=device)
    rhos, sig = dp.log_descent(35, 5, 30)
    with torch.no_grad():
        dt, _ = _timed(lambda: s.solve(x0=b3, rhos=rhos, lams={prior: sig}, max_iter=30), 1)
    flop = 3.5695e12                                   # denoiser FLOP per iteration at B = 8
    out["config3"] = {"workload": "8x3x1024x1024 PnP ADMM, FFDNet-colour z-update (seeded weights), 30 it", "ms_per_iter": dt / 30 * 1e3,
                      "it_per_s": 30 / dt, "path": s.last_path,
                      "denoiser_arithmetic": getattr(prior.denoiser.model, "compute_mode", "f32") + " (f16x2 = operands split into two binary16 terms, three "
                                             "products on the f16 matrix cores, fp32 accumulation: 2e-7 from the f32-input MFMA path, parity-pinned by "
                                             "G8 / G31 at 1e-5; bf16x3 = three bf16 terms, six products)",
                      "roofline": {"bound": "mfma", "unit": "TFLOP/s", "achieved": flop * 30 / dt / 1e12, "peak": 2500.0 / 3.0,
                                   "frac": flop * 30 / dt / (2500.0e12 / 3.0),
                                   "note": "denoiser FLOP (fp32-equivalent, SURVEY 8(d)) / whole-iteration time against the pipe the default "
                                           "arithmetic runs on: dense f16 MFMA peak 2.5 PFLOP/s / 3 products per fp32-accurate product; "
                                           "the same rate is " + f"{flop * 30 / dt / 157.3e12:.2f}" + " x the fp32-input MFMA peak (157.3 TFLOP/s), "
                                           "which this path does not use"}}
    del s, prior, b3, gt3
    # ---- config 4: one GPU's shard (4 of the 32 images) and the whole batch on one GPU
    for tag, nb in (("config4_shard4", 4), ("config4_batch32", 32)):
        gt4, mask, y = synthetic.csmri_case(nb, 320, 320, seed=2023)
        mask_d, y_d = torch.from_numpy(mask).to(device), torch.from_numpy(y).to(device)
        x = dp.Variable()
        fns = dp.sum_squares(masked_fft(x, mask_d), y_d) + dp.nonneg(x) + dp.deep_prior(x, denoiser=FFDNetDenoiser(synthetic.ffdnet_weights(11, 1, 1, 64, 15)))
        s = dp.compile(fns, method="ladmm", device=device, linear_solve_config=LinearSolveConfig(rtol=1e-6, max_iters=100))
        x0 = ifft2(y_d).real.contiguous()
        with torch.no_grad():
            dt, _ = _timed(lambda: s.solve(x0=x0, rhos=0.5, lams=0.03, max_iter=10), 3, rounds=4)      # (a round of the shard is 17 ms: four of them, the GPU's clocks settle within the first)
        cg = [int(n) for n in s.least_square.cg_iters[-10:]]
        flop4 = 2.480e10 * nb
        out[tag] = {"workload": f"{nb}x1x320x320 CS-MRI, LADMM + CG(rtol 1e-6, <=100) + nonneg + FFDNet-gray, 10 outer it",
                    "ms_per_outer_iter": dt / 10 * 1e3, "cg_iters": cg,
                    "roofline": {"bound": "mfma", "unit": "TFLOP/s", "achieved": flop4 * 10 / dt / 1e12, "peak": 2500.0 / 3.0,
                                 "frac": flop4 * 10 / dt / (2500.0e12 / 3.0),
                                 "cg_bytes_per_iter": 80.0 * nb * 320 * 320,
                                 "note": "denoiser FLOP / whole outer-iteration time against the f16 MFMA peak / 3 (split-f16 arithmetic); the "
                                         "CG part is latency-bound (SURVEY 8(d): 80 B per element and CG iteration)"}}
        del s
    # ---- config 5: unrolled ADMM x10 training step, 4x3x512x512
    gt5, b5, psf = synthetic.deconv_case(4, 3, 512, 512, seed=2023)
    bt, gtt = torch.from_numpy(b5).to(device), torch.from_numpy(gt5).to(device)
    for mode in ("f32", "bf16"):
        x = dp.Variable()
        n0, n1 = dp.norm1(dp.grad(x, dim=0)), dp.norm1(dp.grad(x, dim=1))
        s = dp.compile(dp.sum_squares(dp.conv(x, psf) - bt) + n0 + n1, method="admm", device=device)
        if mode == "bf16" and not getattr(dp, "UNROLL_BF16", False):
            continue
        s = dp.specialize(s, method="unroll", device=device, max_iter=10, **({} if mode == "f32" else {"dtype": "bf16"}))
        prm = [torch.full((10,), v, requires_grad=True, device=device) for v in (0.1, 0.005, 0.005)]

        def step():
            for p_ in prm:
                p_.grad = None
            o = s.solve(x0=bt, rhos=prm[0], lams={n0: prm[1], n1: prm[2]})
            loss = ((o - gtt) ** 2).mean()
            loss.backward()
            return loss
        dt, loss = _timed(step, 30)                        # (5 steps measure the pipeline's start: the first step's host work is not overlapped)
        n5 = 4 * 3 * 512 * 512
        out["config5_" + mode] = {"workload": "4x3x512x512 unrolled ADMM x10 (specialize 'unroll'), MSE loss, fwd + bwd w.r.t. rho_t, lam_t",
                                  "dtype": mode, "ms_per_step": dt * 1e3, "steps_per_s": 1 / dt, "loss": float(loss.detach()),
                                  "roofline": {"bound": "hbm", "note": "forward 10 x 36 B/element + history writes, backward ~3x: launch-bound at "
                                                                       "this size (12.6 MB per tensor)",
                                               "frac_forward_bytes_only": 10 * 36.0 * n5 / dt / HBM_PEAK}}
    return out


def sharded_runs(dp, synthetic, dist, rank, world, device):
    """N > 1 only (or DPX_BENCH_FORCE_DIST on one GPU): the STRONG-scaling companions of the weak-scaling headline -- a fixed batch whose
    independent images are dealt over the ranks (BASELINE.json north_star: RCCL broadcast / all-gather, no collective inside the iteration).
    Per config three timings, each bracketed by barrier + synchronize, max over ranks:
      distribution : the one-time transfers, each on its own -- `scatter_s` (rank 0's batch -> every rank's slice), `allgather_s` (the results
                     back), and under `broadcast_s` the shared constants (PSF / mask, denominator tables, denoiser weights) -- excluded from `loop_s`;
      loop_s       : the iterations alone, every rank's slice already resident in its HBM (SURVEY 8(e) allows per-rank generation from the seed);
      end_to_end_s : scatter + iterations + all-gather in one call (dprox.distributed.solve_sharded);
    and `n1_loop_s` -- the whole batch solved by rank 0 ALONE in the same process (the other ranks wait) -- with
    `speedup_vs_n1 = {loop: n1_loop_s / loop_s, end_to_end: n1_loop_s / end_to_end_s}`: the numbers north_star's ">= 6x at 8 GPUs" is about.
      config2_batch8  : the 8 x 3 x 1024 x 1024 batch of the headline, 50 ADMM iterations, 8 / N images per GPU
      config4_batch32 : BASELINE.json config 4, 32 x 1 x 320 x 320 CS-MRI, LADMM + CG + FFDNet-gray, 10 outer iterations, 32 / N per GPU"""
    from dprox import distributed as dd
    from dprox.contrib import masked_fft
    from dprox.linalg import LinearSolveConfig
    from dprox.proxfn.pnp.denoisers import FFDNetDenoiser
    from dprox.utils import ifft2
    comm = dd.Comm.from_process_group() if os.environ.get("DPX_COMM", "torch") == "abi" else None
    out = {"transport": "dpx_comm_* (RCCL through the C ABI)" if comm is not None else "torch.distributed nccl backend (RCCL)", "world": world,
           "scaling": "strong (fixed batch dealt over the ranks); the top-level `value` is WEAK scaling (every rank its own batch of 8)"}
    # self-check of the communicator(s): one RCCL rank per process of the job
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    check = {"WORLD_SIZE": env_world, "torch_backend": dist.get_backend(), "torch_world_size": dist.get_world_size()}
    assert check["torch_world_size"] == env_world and check["torch_backend"] == "nccl", check
    if comm is not None:
        check["dpx_comm_world"] = int(comm.world)
        assert comm.world == env_world, check
    out["communicator_check"] = check

    def barrier():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    failed = []

    def guarded(fn, out_shape):
        """a rank whose local solve raises still takes part in the collectives (zeros), and reports the failure afterwards"""
        def run(loc):
            try:
                return fn(loc)
            except Exception as e:                         # noqa: BLE001
                failed.append(f"rank {rank}: {type(e).__name__}: {e}")
                n = next(iter(loc.values())).shape[0]
                return torch.zeros((n,) + tuple(out_shape), dtype=torch.float32, device=device)
        return run

    def any_failed():
        t = torch.tensor([float(len(failed))], device=device)
        dist.all_reduce(t)
        return float(t.item()) > 0

    def timed(fn, warm=1):
        """fn() on every rank; seconds = max over ranks of (barrier, fn, synchronize)"""
        res = None
        for _ in range(warm):
            res = fn()                                     # warm-up: tables, workspaces, RCCL channels
        barrier()
        t0 = time.perf_counter()
        res = fn()
        torch.cuda.synchronize()                           # (clock stops at this rank's completion; MAX over ranks below = the job's time)
        dt_local = time.perf_counter() - t0
        barrier()
        t = torch.tensor([dt_local], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()), res

    def legs(tag, full, key, local_solve, full_solve, out_shape, units, unit_name):
        """the three timings of one config; full: rank 0's batch tensor (None elsewhere)"""
        g = guarded(local_solve, out_shape)
        batch = dd._bcast_meta(int(full.shape[0]) if rank == 0 else None, 0, None)
        t_scatter, loc = timed(lambda: dd.scatter_batch(full if rank == 0 else None, 0, None, device, comm))
        t_loop, res = timed(lambda: g({key: loc}) if loc.shape[0] > 0 else None)
        if res is None:
            res = torch.empty((0,) + tuple(out_shape), dtype=torch.float32, device=device)
        t_gather, _ = timed(lambda: dd.all_gather_batch(res, batch, None, comm))
        t_e2e, xs = timed(lambda: dd.solve_sharded(g, {key: full} if rank == 0 else None, src=0, device=device, comm=comm))
        # rank 0 alone on the whole batch (the N = 1 reference of this very process and GPU)
        t_n1 = None
        if rank == 0:
            full_solve(full)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            full_solve(full)
            torch.cuda.synchronize()
            t_n1 = time.perf_counter() - t0
        barrier()
        t = torch.tensor([t_n1 if t_n1 is not None else 0.0], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        t_n1 = float(t.item())
        rec = {"images_per_gpu": batch / world, unit_name: units,
               "distribution": {"scatter_s": t_scatter, "allgather_s": t_gather},
               "loop_s": t_loop, "end_to_end_s": t_e2e, "n1_loop_s": t_n1,
               "speedup_vs_n1": {"loop": t_n1 / t_loop, "end_to_end": t_n1 / t_e2e},
               "seconds": t_e2e}
        return rec, xs

    # ---- config 2, one batch of 8 split over the ranks
    b = None
    t_bc = {}
    if rank == 0:
        rng = np.random.RandomState(2023)
        gt = torch.from_numpy(synthetic.synth(rng, B, C, H, W)).to(device)
        psf0 = synthetic.point_spread_function(15, 5.0)
        b = (dp.conv(dp.Variable(), psf0).to(device).forward(gt) + torch.from_numpy((rng.randn(B, C, H, W) * (2.0 / 255.0)).astype(np.float32)).to(device)).contiguous()
    barrier()
    t0 = time.perf_counter()
    consts = dd.broadcast_constants({"psf": torch.from_numpy(psf0).to(device)} if rank == 0 else None, src=0, device=device, comm=comm)
    barrier()
    t_bc["config2_psf"] = time.perf_counter() - t0
    psf = consts["psf"].cpu().numpy()
    # compiled ONCE per rank, outside the timed calls: the observation is a Placeholder that every call fills with its slice; the
    # OTF / denominator tables are built on rank 0 and broadcast (dprox.distributed.share_tables) instead of rebuilt per rank
    obs2 = dp.Placeholder()
    x2 = dp.Variable()
    s2 = dp.compile(dp.sum_squares(dp.conv(x2, psf) - obs2) + dp.norm1(dp.grad(x2, dim=0)) + dp.norm1(dp.grad(x2, dim=1)), method="admm", device=device)
    barrier()
    t0 = time.perf_counter()
    dd.share_tables(s2, (max(B // world, 1), C, H, W), src=0, device=device, comm=comm)
    barrier()
    t_bc["config2_tables"] = time.perf_counter() - t0

    def solve_c2(loc):
        bb = loc["b"]
        obs2.value = bb
        return s2.solve(x0=bb, rhos=RHO, lams=LAM, max_iter=50)

    rec, xs = legs("config2_batch8", b, "b", solve_c2, lambda full: solve_c2({"b": full}), (C, H, W), 50, "iters")
    if any_failed():
        out["error"] = failed or ["a peer rank failed in config2_batch8"]
        return out
    rec["it_per_s"] = 50 / rec["end_to_end_s"]
    rec["it_per_s_loop"] = 50 / rec["loop_s"]
    rec["note"] = "end_to_end_s: scatter of the 100.7 MB observation + 50 iterations + all-gather of the result; loop_s: the 50 iterations on resident slices"
    if rank == 0:
        rec["psnr_db_mean"] = float(np.mean(psnr_per_image(xs, gt)))
    out["config2_batch8"] = rec
    # ---- config 4, 32 images split over the ranks
    nb = 32
    y_ri = None
    if rank == 0:
        gt4, mask, y = synthetic.csmri_case(nb, 320, 320, seed=2023)
        y_d = torch.from_numpy(y).to(device)
        y_ri = torch.view_as_real(y_d).contiguous()        # complex tensors travel as [.., 2] float32
    barrier()
    t0 = time.perf_counter()
    consts4 = dd.broadcast_constants({"mask": torch.from_numpy(mask).to(device)} if rank == 0 else None, src=0, device=device, comm=comm)
    # the denoiser's weights exist on rank 0 (a real checkpoint would be loaded there) and reach the others by broadcast; solver and
    # denoiser are built ONCE per rank, outside the timed calls; the k-space data is a Placeholder filled per call
    den4 = FFDNetDenoiser(synthetic.ffdnet_weights(11, 1, 1, 64, 15) if rank == 0 else None).to(device)
    sd = dd.broadcast_constants({k: v.detach() for k, v in den4.state_dict().items()} if rank == 0 else None, src=0, device=device, comm=comm)
    barrier()
    t_bc["config4_mask_and_weights"] = time.perf_counter() - t0
    if rank != 0:
        den4.load_state_dict(sd, strict=True)
    y4 = dp.Placeholder()
    x4 = dp.Variable()
    fns4 = dp.sum_squares(masked_fft(x4, consts4["mask"]), y4) + dp.nonneg(x4) + dp.deep_prior(x4, denoiser=den4)
    s4 = dp.compile(fns4, method="ladmm", device=device, linear_solve_config=LinearSolveConfig(rtol=1e-6, max_iters=100))

    def solve_c4(loc):
        yy = torch.view_as_complex(loc["y"].contiguous())
        y4.value = yy
        with torch.no_grad():
            return s4.solve(x0=ifft2(yy).real.contiguous(), rhos=0.5, lams=0.03, max_iter=10)

    rec, _ = legs("config4_batch32", y_ri, "y", solve_c4, lambda full: solve_c4({"y": full}), (1, 320, 320), 10, "outer_iters")
    if any_failed():
        out["error"] = failed or ["a peer rank failed in config4_batch32"]
        return out
    rec["ms_per_outer_iter"] = rec["end_to_end_s"] / 10 * 1e3
    rec["ms_per_outer_iter_loop"] = rec["loop_s"] / 10 * 1e3
    out["config4_batch32"] = rec
    out["broadcast_s"] = t_bc
    if comm is not None:
        comm.close()
    return out


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU path)"
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1 or os.environ.get("DPX_BENCH_FORCE_DIST"):     # (FORCE_DIST: exercise the N > 1 code path on a one-GPU box)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group(backend="nccl", device_id=device, rank=rank, world_size=world)

    import dprox as dp
    import synthetic
    from dprox import _backend as be

    if a.chains > 0:
        os.environ["DPX_CHAINS"] = str(a.chains)
    solver, xvar, b, gt, psf = make_problem(dp, synthetic, rank, device)
    K, Wm = a.steps, a.warmup

    if a.pmc_child:
        # what the rocprofv3 counter passes of pmc_traffic() run: the headline workload's iterations as ONE chain (every launch covers the
        # whole batch and has the GPU to itself), W warm-up + K steps, nothing else
        solver.solve(x0=b, rhos=RHO, lams=LAM, max_iter=max(Wm, 1))
        st = solver.initialize(b)
        _, rs, ls, _ = solver.defaults(b, RHO, LAM, K)
        solver.iters(st, rs.to(device), {k: v.to(device) for k, v in ls.items()}, K)
        torch.cuda.synchronize()
        return

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- schedules of the timed regions, as the caller of iters() holds them: device tensors, one per region length (prepared first:
    #      nothing but kernel launches then sits between the legs below)
    x0, rhos, lams, _ = solver.defaults(b, RHO, LAM, max(K, 200))
    rhos = rhos.to(device)
    lams = {k: v.to(device) for k, v in lams.items()}
    sched = {n: (rhos[..., :n].contiguous(), {k: v[..., :n].contiguous() for k, v in lams.items()}) for n in {K, 200}}

    def timed_region(n_steps):
        """W untimed warm-up steps, then exactly n_steps timed steps between barriers"""
        solver.solve(x0=b, rhos=RHO, lams=LAM, max_iter=max(Wm, 1))
        st = solver.initialize(b)
        rs, ls = sched[n_steps]
        barrier()
        t0 = time.perf_counter()
        st = solver.iters(st, rs, ls, n_steps)
        # the region ends when this rank's GPU has finished its K steps: the clock stops behind the synchronize, the inter-rank barrier
        # follows it, and the job's time is the MAX over the ranks' clocks (all ranks left the opening barrier together) -- the RCCL
        # barrier's own latency (~1 ms, a fifth of a 20-step region) is not part of any step
        torch.cuda.synchronize()
        dt_ = time.perf_counter() - t0
        barrier()
        if dist is not None:
            t = torch.tensor([dt_], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_ = float(t.item())
        return dt_

    # ---- leg 1, cold solve: BASELINE.json config 2 as stated -- compile()d solver, nothing cached (OTF / denominator tables, fp64
    #      data spectrum, workspaces are all built inside), 50 iterations, wall clock around solve().  Its result is the quality figure.
    barrier()
    t0 = time.perf_counter()
    out = solver.solve(x0=b, rhos=RHO, lams=LAM, max_iter=50)
    barrier()
    cold_ms = 1e3 * (time.perf_counter() - t0)
    assert solver.last_path == "fused", "bench must run the fused HIP iteration"

    # (the roofline leg below runs as one chain: its data spectrum -- a layout of its own -- is computed here, not between the timed legs)
    _env = os.environ.get("DPX_CHAINS")
    os.environ["DPX_CHAINS"] = "1"
    solver.solve(x0=b, rhos=RHO, lams=LAM, max_iter=2)
    if _env is None:
        os.environ.pop("DPX_CHAINS", None)
    else:
        os.environ["DPX_CHAINS"] = _env
    # ---- leg 2, steady state: 200 timed steps (a solve's fixed parts -- seed pass, result emission, launch latency of the first
    #      kernel -- spread over 200 iterations)
    dt_steady = timed_region(200)
    K_steady = 200
    # ---- leg 3, the headline: W warm-up steps, then exactly K timed steps.  It follows leg 2 directly (no host work in between), so
    #      the GPU has been under load for ~40 ms when the warm-up starts.  This GPU needs tens of milliseconds of sustained load to
    #      reach its clocks and loses them within a few milliseconds of idling (tools/ramp_probe.py): the last leg shows what the
    #      same region measures when the GPU idled in front of the W warm-up steps.
    dt = timed_region(K) if K != 200 else dt_steady
    # ---- roofline leg: the same K iterations once more with the library's per-kernel timers on (HIP events attached to the dispatch
    #      packets), directly behind the headline region -- same clocks
    #      With sub-batch chains (dprox/algo/fused.py: two halves of the batch iterate independently on two streams) two launches share
    #      the GPU, and the duration of one is no bandwidth measurement: this leg runs as ONE chain (DPX_CHAINS=1) -- every launch
    #      covers the whole batch and has the GPU to itself, which is what `roofline` and `kernels` describe.
    from dprox.algo import fused as _fused
    chains_used = _fused.sub_batch_chains(B, C, H, W)
    if chains_used > 1 and _fused.chain_stream_handles(device, chains_used) is None:      # (no second hardware queue: the library runs one chain)
        chains_used = 1
    rhos_k, lams_k = sched[K]
    chains_env = os.environ.get("DPX_CHAINS")
    os.environ["DPX_CHAINS"] = "1"
    try:
        be.lib().call("dpx_timing_enable", 1)
        state2 = solver.initialize(b)
        timing_report(be)                                   # drop the initialize() launches
        solver.iters(state2, rhos_k, lams_k, K)
        torch.cuda.synchronize()
        rep = timing_report(be)
        be.lib().call("dpx_timing_enable", 0)
    finally:
        if chains_env is None:
            os.environ.pop("DPX_CHAINS", None)
        else:
            os.environ["DPX_CHAINS"] = chains_env
    del state2
    # ---- leg 4: a warm 50-iteration solve (tables and data spectrum cached): cold - warm = what a first solve pays for its setup
    solver.solve(x0=b, rhos=RHO, lams=LAM, max_iter=50)      # (load in front, as for the cold runs of leg 5: the roofline leg's report left the GPU idle)
    barrier()
    t0 = time.perf_counter()
    solver.solve(x0=b, rhos=RHO, lams=LAM, max_iter=50)
    barrier()
    warm_ms = 1e3 * (time.perf_counter() - t0)
    # ---- leg 5: the same cold solve in a warm process: a NEW problem (fresh observation tensor, freshly compiled solver: no table, no data
    #      spectrum cached) while the allocator's pool, the code objects and the FFT twiddle tables of this process are warm -- what a
    #      long-running caller pays per new problem
    cold_runs = []
    for _ in range(2):          # (twice: the first new problem may still make the caching allocator grow its pool -- a hipMalloc of 100 MB
        b2 = b.clone()          #  takes ~10 ms --, the second one is the steady per-problem cost)
        x2 = dp.Variable()
        solver2 = dp.compile(dp.sum_squares(dp.conv(x2, psf) - b2) + dp.norm1(dp.grad(x2, dim=0)) + dp.norm1(dp.grad(x2, dim=1)), method="admm", device=device)
        solver.solve(x0=b, rhos=RHO, lams=LAM, max_iter=50)      # (load in front, as for the headline: clocks)
        barrier()
        t0 = time.perf_counter()
        solver2.solve(x0=b2, rhos=RHO, lams=LAM, max_iter=50)
        barrier()
        cold_runs.append(1e3 * (time.perf_counter() - t0))
        del solver2, b2, x2
    cold2_ms = min(cold_runs)
    # ---- leg 6 (last): the same region after the GPU idled for half a second (the clock ramp falls into the timed steps)
    time.sleep(0.5)
    dt_idle = timed_region(K)

    # ---- quality (of the cold 50-iteration solve) -------------------------------------
    psnr_in, psnr_out = psnr_per_image(b, gt), psnr_per_image(out, gt)
    del out
    setup = setup_profile(dp, be, b, psf, device) if rank == 0 else None
    # ---- second quality leg: the same solver settings on a detail-scaled synthetic (synthetic.synth_detail: the SURVEY generator's
    #      cosines have <= 8 cycles per IMAGE whatever the plane size, so at 1024 x 1024 the blur hardly hurts and the deconvolution
    #      has nothing to win -- 36.8 dB in, 36.7 dB out; with the detail of a 256 x 256 image per 256 x 256 pixels it has)
    quality_detail = None
    if rank == 0 and not a.no_extra_configs:
        rngd = np.random.RandomState(4023)
        gtd = torch.from_numpy(synthetic.synth_detail(rngd, B, C, H, W)).to(device)
        bd = (dp.conv(dp.Variable(), psf).to(device).forward(gtd) + torch.from_numpy((rngd.randn(B, C, H, W) * (2.0 / 255.0)).astype(np.float32)).to(device)).contiguous()
        xd = dp.Variable()
        sd_ = dp.compile(dp.sum_squares(dp.conv(xd, psf) - bd) + dp.norm1(dp.grad(xd, dim=0)) + dp.norm1(dp.grad(xd, dim=1)), method="admm", device=device)
        outd = sd_.solve(x0=bd, rhos=RHO, lams=LAM, max_iter=50)
        pin, pout = psnr_per_image(bd, gtd), psnr_per_image(outd, gtd)
        quality_detail = {"generator": "synthetic.synth_detail (seed 4023): cosines up to 8 cycles per 256 pixels, 128 rectangles of 16..85 pixels per plane",
                          "input_mean": float(np.mean(pin)), "admm50_mean": float(np.mean(pout)), "gain_db": float(np.mean(pout) - np.mean(pin)),
                          "admm50_per_image": pout}
        del gtd, bd, outd, sd_, xd

    # The strong-scaling companions run collectives; a rank that fails or stalls inside them must not take the headline line with
    # it: they run in a worker thread with a deadline, after which every rank goes on (and leaves through os._exit, see below).
    sharded, stalled = None, False
    if dist is not None and not a.no_extra_configs:
        import threading
        box = {}

        def companions():
            try:
                torch.cuda.set_device(local)
                box["out"] = sharded_runs(dp, synthetic, dist, rank, world, device)
            except Exception as e:
                box["out"] = {"error": f"{type(e).__name__}: {e}"}

        th = threading.Thread(target=companions, daemon=True)
        th.start()
        th.join(float(os.environ.get("DPX_BENCH_COMPANION_DEADLINE", "240")))
        stalled = th.is_alive()
        sharded = box.get("out") if not stalled else {"error": "the sharded companion runs did not finish before their deadline"}
    if rank != 0:
        if dist is not None and not stalled:
            dist.destroy_process_group()
        if stalled:
            os._exit(0)
        return

    n_elem = B * C * H * W
    total_ms = sum(t for _, t in rep.values())
    kernels = {k: {"launches": c, "avg_us": 1e3 * t / c, "share": t / total_ms,
                   "GBps": KERNEL_BYTES_PER_ELEM.get(k, 0.0) * n_elem / (1e-3 * t / c) / 1e9}
               for k, (c, t) in rep.items()}
    dom = max(rep, key=lambda k: rep[k][1])
    # HBM traffic of the dominant kernel from the PMC counters: rocprofv3 cannot be run from inside this process, so the
    # figure comes from the committed separate --pmc passes of this same command (tools/pmc_summary.py, profiles/)
    traffic, traffic_src, traffic_note, traffic_detail = None, None, None, None
    import glob
    under_profiler = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", "")
    want_pmc = a.pmc if a.pmc is not None else (world == 1 and not under_profiler)
    if want_pmc:
        traffic_detail, why = pmc_traffic(dom)
        if traffic_detail is not None:
            traffic = traffic_detail["bytes"]
            traffic_src = "measured by this command: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (--kernel-trace only) of the same workload as subprocesses"
            traffic_note = ("per launch, one chain; FETCH_SIZE x 1024 x 2 (gfx950 tallies wide coalesced reads at half their bytes) + WRITE_SIZE x 1024; "
                            "fabric requests incl. Infinity-Cache hits (see hbm_vs_infinity_cache)")
        else:
            traffic_note = f"the in-run counter passes failed ({why}); committed figure used"
    pmc_files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm.json")))
    if traffic is None and pmc_files:
        for name, e in json.load(open(pmc_files[-1])).items():
            if name.split("<")[0] == dom and "hbm_traffic_bytes" in e:
                traffic = e["hbm_traffic_bytes"]
                traffic_src = f"profiles/{os.path.basename(pmc_files[-1])} (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, per launch)"
    dom_bytes = KERNEL_BYTES_PER_ELEM.get(dom, 0.0) * n_elem
    emit_note = None
    if dom.startswith("k_iter_rows") and rep[dom][0] > 0:
        # the LAST row pass of a solve writes the result -- x and the two v_i, 12 B per element -- instead of the next spectrum (4 B):
        # 32 B per element instead of 24 (135 us instead of 106); averaged over the K launches like the duration it is divided by
        dom_bytes += 8.0 * n_elem / rep[dom][0]
        emit_note = (f"24 B/element per launch; the last of the {rep[dom][0]} launches moves 32 (x, v_0, v_1 out, no next spectrum): "
                     f"+ 8 / {rep[dom][0]} B/element on average")
    dom_avg_s = 1e-3 * rep[dom][1] / rep[dom][0]
    achieved = dom_bytes / dom_avg_s
    it_per_s = world * K / dt
    res = {
        "metric": "admm_iters_per_sec", "value": it_per_s, "unit": "it/s (one iteration = one 8x3x1024x1024 batch)",
        "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": 1e3 * dt / K, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "scaling_note": "`value` is WEAK scaling: every rank solves its own batch of 8 (N x the work, ~N x the rate by construction).  The STRONG-scaling "
                        "figures north_star's '>= 6x at 8 GPUs' is about -- one fixed batch dealt over the ranks -- are `sharded.<config>.speedup_vs_n1` "
                        "(loop = iterations on resident slices, end_to_end = incl. scatter and all-gather), present for N > 1",
        "config": {"workload": "config 2: batch-8 3x1024x1024 RGB deconv, sum_squares(conv(x,psf)-b)+norm1(grad_H)+norm1(grad_W), "
                               "ADMM rho=0.1 lam=0.005, Gaussian 15/5 PSF",
                   "batch_per_gpu": B, "global_batch": B * world, "shape": [C, H, W], "parallelism": f"batch-shard x{world}"},
        "headline_protocol": {"warmup": Wm, "timed_steps": K, "runs_directly_after": "the steady-state leg (steady_state: W warm-up + 200 timed steps, ~40 ms of load)",
                              "note": "`value`: W warm-up steps, then exactly K timed steps between barriers; the region follows the steady-state "
                                      "measurement without host work in between, i.e. on a GPU that has just been under load.  "
                                      "`value_after_idle_gpu`: the identical region after the GPU idled for 0.5 s in front of the warm-up -- this GPU "
                                      "needs tens of ms of load to reach its clocks, so a 5-step warm-up (1 ms) leaves the ramp inside the "
                                      "timed steps.  `steady_state`: 200 timed steps.",
                              "clock": "opening: torch.cuda.synchronize + inter-rank barrier + synchronize, then t0; closing: torch.cuda.synchronize, "
                                       "then t1, then the inter-rank barrier; the job's time is the MAX of t1 - t0 over the ranks (N = 1: no barrier).  "
                                       "The closing RCCL barrier itself (~1 ms with the nccl backend) is not inside any rank's clock"},
        "value_after_idle_gpu": world * K / dt_idle, "ms_per_step_after_idle_gpu": 1e3 * dt_idle / K,
        "steady_state": {"steps": K_steady, "it_per_s": world * K_steady / dt_steady, "ms_per_step": 1e3 * dt_steady / K_steady,
                         "roofline_iteration_frac": (K_steady / dt_steady) * DESIGN_BYTES_PER_ELEM * n_elem / HBM_PEAK},
        "cold_solve": {"cold_solve50_ms": cold2_ms, "warm_solve50_ms": warm_ms, "setup_ms": cold2_ms - warm_ms,
                       "first_solve_of_the_process_ms": cold_ms, "cold_solve50_runs_ms": cold_runs,
                       "steady_50_iterations_ms": 50 * 1e3 * dt_steady / K_steady,
                       "cold_over_50_steady_iterations": cold2_ms / (50 * 1e3 * dt_steady / K_steady),
                       "setup_profile": setup,
                       "note": "cold = solve(max_iter=50) of a freshly compiled solver on a new observation (no OTF / denominator table, no data "
                               "spectrum cached), wall clock incl. every table, the fp64 data spectrum, initialize(), seed and result "
                               "emission, in a warm process; warm = the same call on a solver that has solved before; first_solve_of_the_"
                               "process additionally pays code-object loading and ~1.5 GB of first-time device allocations"},
        "psnr_db": {"input_mean": float(np.mean(psnr_in)), "admm50_mean": float(np.mean(psnr_out)), "admm50_per_image": psnr_out,
                    "generator": "SURVEY 8(d) / Appendix B (synthetic.synth, seed 2023): <= 8 cycles per image, 8 rectangles -- at 1024 x 1024 the "
                                 "blur barely degrades it, see psnr_db_detail"},
        "psnr_db_detail": quality_detail,
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "value_after_idle_gpu": world * K / dt_idle,       # (scalar copy of the top-level key: the clock protocol's cold-GPU figure, it/s)
                     "frac": achieved / HBM_PEAK, "frac_of_measured_copy": achieved / HBM_COPY, "traffic": traffic,
                     "traffic_source": traffic_src,
                     "traffic_note": traffic_note or "NOT measured in this run (--no-pmc / under a profiler / N > 1): the per-launch HBM traffic of this "
                                                     "kernel from the committed separate --pmc passes of this same command",
                     "traffic_detail": traffic_detail,
                     "traffic_over_algorithmic": (traffic / dom_bytes) if traffic else None,
                     "algorithmic_bytes_per_launch": dom_bytes, "algorithmic_bytes_note": emit_note, "avg_launch_us": dom_avg_s * 1e6,
                     "measured_with": "one chain (DPX_CHAINS=1): each launch covers the whole batch and runs alone on the GPU; the timed legs "
                                      f"(`value`, `steady_state`) run {chains_used} sub-batch chain(s) whose launches overlap"},
        "roofline_iteration": {"bound": "hbm", "bytes_per_element": DESIGN_BYTES_PER_ELEM,
                               "algorithmic_bytes_per_iter": DESIGN_BYTES_PER_ELEM * n_elem,
                               "achieved_GBps": (it_per_s / world) * DESIGN_BYTES_PER_ELEM * n_elem / 1e9,
                               "frac": (it_per_s / world) * DESIGN_BYTES_PER_ELEM * n_elem / HBM_PEAK,
                               "design_frac": (it_per_s / world) * DESIGN_BYTES_PER_ELEM * n_elem / HBM_PEAK,     # (same figure: the name the round-1 review used)
                               "frac_of_measured_copy": (it_per_s / world) * DESIGN_BYTES_PER_ELEM * n_elem / HBM_COPY,
                               "kernel_time_share_of_step": 1e-3 * total_ms / K / (dt / K) if K else None,
                               "kernel_time_share_note": "sum of the kernels' one-chain (exclusive) durations / wall-clock step; above 1 when "
                                                         "sub-batch chains overlap one chain's column pass with the other's row pass",
                               "sub_batch_chains": chains_used,
                               "note": "whole iteration (wall clock incl. launch gaps) on the 36 B/element the two-kernel schedule "
                                       "moves: k_cols_p2 12 + k_iter_rows 24",
                               "survey_accounting_bytes_per_iter": ITER_BYTES_PER_ELEM * n_elem,
                               "survey_accounting_frac": (it_per_s / world) * ITER_BYTES_PER_ELEM * n_elem / HBM_PEAK,
                               "survey_accounting_note": "comparison only: SURVEY 8(d) budgets 64 B/element for an un-fused 5-kernel "
                                                         "schedule; this schedule does not move those bytes"},
        "kernels": kernels,
    }
    if world == 1 and not a.no_cpu_baseline:
        n_it, sb = 4, 2
        x_ref, res["cpu_baseline"] = cpu_baseline(b.cpu(), psf, n_iters=n_it, sample_b=sb)
        x_gpu = solver.solve(x0=b, rhos=RHO, lams=LAM, max_iter=1 + n_it)[:sb].cpu()
        res["parity_rel_l2"] = float((x_gpu - x_ref).double().norm() / x_ref.double().norm())
        res["parity_note"] = (f"GPU iterate (images 0..{sb - 1} of the batch-8 solve, {1 + n_it} ADMM iterations) vs the CPU oracle run "
                              f"timed above on the same images; bar 1e-5")
    if sharded is not None:
        res["sharded"] = sharded
    if world == 1 and not a.no_extra_configs:
        res["configs"] = extra_configs(dp, synthetic, device)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_train
        tr = {}
        for key, trainable in (("frozen_denoiser", False), ("trainable_denoiser", True)):
            try:
                tr[key] = bench_train.run(dp, synthetic, be, device, bs=2, size=768, iters=10, steps=3, trainable=trainable, kernels=True)
            except Exception as e:                           # noqa: BLE001 (a companion leg must not take the headline line with it)
                tr[key] = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.empty_cache()
        tr["reference_published"] = {"it_per_s": 1.49, "source": "notebooks/quickstart.ipynb:254-257 (train(model=doe_model, step_fn, 'BSD500', epochs=2): bs 2, "
                                     "768 x 768 patches, ADMM unrolled x10, frozen ffdnet_color prior; GPU not stated)",
                                     "ratio_frozen": (tr["frozen_denoiser"].get("steps_per_s", 0.0) / 1.49) if "steps_per_s" in tr["frozen_denoiser"] else None}
        res["configs"]["train_unrolled_pnp"] = tr
        res["roofline"]["hbm_vs_infinity_cache"] = cache_sweep(dp, synthetic, device)
    # the figures a reader needs next to `roofline.frac`, inside the object the round records keep whole
    cfg = res.get("configs", {})
    kc = kernels.get("k_cols_p2")
    res["roofline"]["companions"] = {
        "headline_it_per_s": it_per_s, "headline_iteration_frac": res["roofline_iteration"]["frac"],
        "steady_state_it_per_s": res["steady_state"]["it_per_s"], "steady_state_iteration_frac": res["steady_state"]["roofline_iteration_frac"],
        "value_after_idle_gpu_it_per_s": res["value_after_idle_gpu"],
        "after_idle_iteration_frac": (K / dt_idle) * DESIGN_BYTES_PER_ELEM * n_elem / HBM_PEAK,
        "k_cols_p2": None if kc is None else {"avg_us": kc["avg_us"], "frac": kc["GBps"] * 1e9 / HBM_PEAK},
        "parity_rel_l2": res.get("parity_rel_l2"),
        "config3_frac_of_f16_pipe_over_3": cfg.get("config3", {}).get("roofline", {}).get("frac"),
        "config3_it_per_s": cfg.get("config3", {}).get("it_per_s"),
        "config4_shard4_ms_per_outer_iter": cfg.get("config4_shard4", {}).get("ms_per_outer_iter"),
        "config4_shard4_frac": cfg.get("config4_shard4", {}).get("roofline", {}).get("frac"),
        "config4_batch32_ms_per_outer_iter": cfg.get("config4_batch32", {}).get("ms_per_outer_iter"),
        "config4_predicted_speedup_8_gpus": (cfg["config4_batch32"]["ms_per_outer_iter"] / cfg["config4_shard4"]["ms_per_outer_iter"])
        if "config4_batch32" in cfg and "config4_shard4" in cfg else None,
        "config2_shard_1x3x1024x1024_us_per_iter": 1e3 * cfg["other_paths"]["admm_1x3x1024x1024"]["ms_per_iter"] if "other_paths" in cfg else None,
        "config2_predicted_speedup_8_gpus": cfg.get("other_paths", {}).get("config2_predicted_speedup_8_gpus"),
        "config5_f32_ms_per_step": cfg.get("config5_f32", {}).get("ms_per_step"),
        "config5_bf16_ms_per_step": cfg.get("config5_bf16", {}).get("ms_per_step"),
        "train_unrolled_pnp_steps_per_s": cfg.get("train_unrolled_pnp", {}).get("frozen_denoiser", {}).get("steps_per_s"),
        "note": "iteration fractions: it/s x 36 B/element x 25 165 824 elements / 8 TB/s (wall clock, chains on); `frac` of this object: the dominant "
                "kernel alone, one chain; after_idle: the identical K-step region after the GPU idled 0.5 s in front of its warm-up",
    }
    # RCCL writes a version banner to the C stdout buffer, which would otherwise be flushed at exit BEHIND the JSON line: flush it
    # first, print the one line, then close stdout for everything that follows
    ctypes.CDLL(None).fflush(None)
    sys.stdout.write(json.dumps(res) + "\n")
    sys.stdout.flush()
    os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    if stalled:
        os._exit(0)                                        # (a collective of the companion runs is still pending: no orderly teardown)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
