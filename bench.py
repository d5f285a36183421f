#!/usr/bin/env python
"""Benchmark of the PoET encoder-decoder hot path on MI355X: images/s, forward + loss + backward +
clip + AdamW (the loop body of the reference's engine.py:55-81), YCB-V config (BASELINE.json
configs[1]: 5 enc / 5 dec / 16 heads, 256-d, 4 levels, 640x480, 20 queries, bs=16 per GPU, bf16),
synthetic backbone features of that shape and random-init weights.

    python bench.py --gpus N --steps K --warmup W
N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py`
(one process per GPU, RCCL all-reduce of the flat gradient arena overlapped with backward).
Rank 0 prints ONE JSON line (see README / DESIGN.md section "Measurement").
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# BASELINE.json configs -> geometry; kept here so the product never imports oracle/
CONFIGS = {
    "cfg0": dict(d_model=256, nheads=4, enc_layers=1, dec_layers=1, d_ffn=1024, n_levels=2, n_points=4, num_queries=10,
                 n_classes=21, dropout=0.1, strides=[8, 16], num_channels=[256, 256], image_hw=(128, 128),
                 level_hw=[(16, 16), (8, 8)], batch=2),
    "ycbv": dict(d_model=256, nheads=16, enc_layers=5, dec_layers=5, d_ffn=1024, n_levels=4, n_points=4, num_queries=20,
                 n_classes=21, dropout=0.1, strides=[8, 16, 32], num_channels=[256, 256, 256], image_hw=(480, 640),
                 level_hw=[(60, 80), (30, 40), (15, 20), (8, 10)], batch=16),
    "lmo": dict(d_model=256, nheads=16, enc_layers=5, dec_layers=5, d_ffn=1024, n_levels=4, n_points=4, num_queries=10,
                n_classes=8, dropout=0.1, strides=[16, 32, 64], num_channels=[256, 256, 256], image_hw=(480, 640),
                level_hw=[(30, 40), (15, 20), (8, 10), (4, 5)], batch=32),
    "hires": dict(d_model=256, nheads=16, enc_layers=6, dec_layers=6, d_ffn=1024, n_levels=4, n_points=4, num_queries=50,
                  n_classes=21, dropout=0.1, strides=[8, 16, 32], num_channels=[256, 256, 256], image_hw=(960, 1280),
                  level_hw=[(120, 160), (60, 80), (30, 40), (15, 20)], batch=8),
}

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_BF16_PEAK_TFS = 2500.0  # dense bf16 MFMA peak


def synth_batch(cfg, batch, seed, device):
    """N(0,1) feature maps per level, all-False masks, 3..Q objects per image."""
    rng = np.random.default_rng(seed)
    nb = len(cfg["strides"])
    g = torch.Generator(device="cpu").manual_seed(seed)
    feats = [torch.randn((batch, cfg["num_channels"][l], *cfg["level_hw"][l]), generator=g).to(device) for l in range(nb)]
    q, c = cfg["num_queries"], cfg["n_classes"]
    targets = []
    for _ in range(batch):
        k = int(rng.integers(3, q + 1))
        boxes = np.concatenate([rng.uniform(0.2, 0.8, (k, 2)), rng.uniform(0.05, 0.25, (k, 2))], 1).astype(np.float32)
        labels = rng.integers(1, c + 1, (k,)).astype(np.int64)
        pos = (rng.standard_normal((k, 3)) * np.array([0.2, 0.2, 0.3]) + np.array([0.0, 0.0, 1.0])).astype(np.float32)
        rots = []
        for _ in range(k):
            qm, r = np.linalg.qr(rng.standard_normal((3, 3)))
            qm = qm * np.sign(np.diag(r))
            if np.linalg.det(qm) < 0:
                qm[:, 2] = -qm[:, 2]
            rots.append(qm)
        targets.append({"boxes": torch.from_numpy(boxes), "labels": torch.from_numpy(labels),          # host side (query assembly, matcher)
                        "relative_position": torch.from_numpy(pos).to(device),
                        "relative_rotation": torch.from_numpy(np.stack(rots).astype(np.float32)).to(device)})
    return feats, targets


def gemm_flops_per_image(cfg):
    """Algorithmic GEMM flops, forward; fwd+bwd = 3x."""
    d, f, M, L, P = cfg["d_model"], cfg["d_ffn"], cfg["nheads"], cfg["n_levels"], cfg["n_points"]
    S = sum(h * w for h, w in cfg["level_hw"])
    Q, C1 = cfg["num_queries"], cfg["n_classes"] + 1
    mlp = M * L * P
    enc = cfg["enc_layers"] * 2 * S * d * (3 * mlp + 2 * d + 2 * f)
    dec = cfg["dec_layers"] * (2 * S * d * d + 2 * Q * d * (3 * d + 3 * mlp + 2 * d + 2 * f) + 4 * Q * Q * d)
    heads = cfg["dec_layers"] * 2 * Q * (4 * d * d + 9 * d * C1)
    nb = len(cfg["strides"])
    proj = 2 * d * sum(cfg["level_hw"][l][0] * cfg["level_hw"][l][1] * cfg["num_channels"][l] for l in range(nb))
    if L > nb:
        h, w = cfg["level_hw"][nb]
        proj += 2 * h * w * 9 * cfg["num_channels"][-1] * d
    return enc + dec + heads + proj


# HBM traffic per launch of a kernel family from the committed rocprofv3 PMC summary of THIS command (separate --pmc
# FETCH_SIZE / WRITE_SIZE passes, profiles/collect.sh -> profiles/round<N>_<config>_pmc_hbm.csv, latest round for the
# config): (2 * FETCH_SIZE + WRITE_SIZE) KiB -- the x2 is the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md.  None when
# the family is not in the summary.
# profile tag (poet_amd/ops.py) -> substring of the kernel symbols in the summaries
_PMC_NAMES = {"gemm_pipe": "gemm_pipe_kernel", "msda_bwd_dvalue_scatter_tiled": "msda_bwd_dv_tiled_kernel", "msda_bwd_dq": "msda_bwd_shared_kernel|msda_bwd_kernel<bf16, bf16", "msda_fused_fwd": "msda_fwd_shared_kernel|msda_fwd_kernel<bf16, bf16",
              "gemm_dw_dW": "gemm_dwr_kernel|gemm_dw_kernel", "gemm_dwx_dW_dX": "gemm_dwx_kernel", "gemm_stream_dX": "gemm_ws", "gemm_stream_fwd": "gemm_ws",
              "gemm_tiled_fwd": "gemm_kernel<", "gemm_tiled_dX": "gemm_kernel<", "gemm_tiled_dW": "gemm_kernel<",
              "gemm_small_fwd": "gemm_small_kernel<false", "gemm_small_dX": "gemm_small_kernel<true", "gemm_small_dW": "gemm_small_dw_kernel",
              "ln_fwd": "ln_fwd_kernel<float, float, bf16>|ln_fwd_kernel<bf16", "ln_bwd": "ln_bwd_kernel<bf16"}
# ws kernels: W stored [K][N] (dX) or not; the software-pipelined forward form (gemm_wsp_kernel<TC, ACT, DROP>) is forward only
def _ws_wkm(n):
    import re
    m = re.search(r"gemm_wsk?_kernel<[^,]+(?:::[^,]+)?, \d+, (true|false)", n)      # gemm_ws_kernel<TC, KIND, WKM, ...>, gemm_wsk_kernel<TC, KIND, WKM, ...>
    return None if m is None else m.group(1) == "true"


_PMC_FILTER = {"gemm_stream_dX": lambda n: _ws_wkm(n) is True,
               "gemm_stream_fwd": lambda n: "gemm_wsp_kernel" in n or _ws_wkm(n) is False}


def _pmc_rows(config, kind):
    import csv, glob, re
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", f"round*_{config}_pmc_{kind}.csv")),
                   key=lambda p: int(re.search(r"round(\d+)_", os.path.basename(p)).group(1)))
    if not paths:
        return None, []
    rows = [r for r in csv.reader(l for l in open(paths[-1]) if not l.startswith("#"))]
    return os.path.basename(paths[-1]), rows


def _pmc_key(tag):
    """profile tag -> kernel-symbol substring (GEMM tags carry a _<N>x<K> shape suffix: longest prefix wins)"""
    best = None
    if tag.endswith("_small"):
        return None
    for k in _PMC_NAMES:
        if tag.startswith(k) and (best is None or len(k) > len(best)):
            best = k
    return best


def pmc_source(config="ycbv", kind="hbm"):
    """Where `roofline.traffic` comes from: the committed counter summary (collected by profiles/collect.sh in separate
    --pmc passes on an EARLIER box, not in this run) and a content hash of that file, so a reader can tell which one."""
    import glob, hashlib, re
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", f"round*_{config}_pmc_{kind}.csv")),
                   key=lambda p: int(re.search(r"round(\d+)_", os.path.basename(p)).group(1)))
    if not paths:
        return None
    return {"file": "profiles/" + os.path.basename(paths[-1]), "sha256_12": hashlib.sha256(open(paths[-1], "rb").read()).hexdigest()[:12],
            "measured_in_this_run": False, "fetch_correction": FETCH_CORRECTION_NOTE}


FETCH_CORRECTION_NOTE = ("traffic = (2 x FETCH_SIZE + WRITE_SIZE) KiB: the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE reports half of a wide "
                         "coalesced read stream).  Calibration on known byte counts, profiles/round5_fetch_calib.csv (profiles/probes/fetch_calib.hip): "
                         "counter / known bytes = 0.50 for 16-byte-per-lane streams, 1.00 for 64-byte segments of strided rows, 2.00 for 32-byte "
                         "segments (fetched as 64-byte sectors) -- so the x2 is exact for streams and for the scatter since its 16 heads share an XCD "
                         "(whole rows arrive as wide requests), and an upper bound for the L2-served gathers; traffic_raw = FETCH_SIZE + WRITE_SIZE")


def pmc_traffic_bytes(tag, config="ycbv", fetch_factor=None):
    if fetch_factor is None:
        fetch_factor = 2.0
    tag = _pmc_key(tag)
    key = _PMC_NAMES.get(tag)
    _, rows = _pmc_rows(config, "hbm")
    if key is None or len(rows) < 2:
        return None
    flt = _PMC_FILTER.get(tag, lambda n: True)
    tot = n = 0.0
    for alt in key.split("|"):                          # first alternative present in the summary; launch-weighted mean over
        for r in rows[1:]:                              # its symbols
            if alt in r[0] and flt(r[0]):
                tot += float(r[1]) * (fetch_factor * float(r[2]) + float(r[4])) * 1024
                n += float(r[1])
        if n:
            break
    return int(tot / n) if n else None


def pmc_mfma_summary(config="ycbv"):
    """Matrix-pipe view of the GEMM kernels from the committed SQ counter summary (profiles/round<N>_<config>_pmc_sq.csv):
    time-weighted MFMA rate (SQ_INSTS_MFMA * flop per instruction / kernel-trace duration) and the share of the busy CU
    cycles the matrix pipe was busy.  None when no summary for this config is committed."""
    src, rows = _pmc_rows(config, "sq")
    if len(rows) < 2:
        return None
    h = {c: i for i, c in enumerate(rows[0])}
    us = tf = busy = 0.0
    top = None
    for r in rows[1:]:
        if "gemm" not in r[0].split("<")[0] or float(r[h["SQ_INSTS_MFMA"]]) == 0:
            continue
        w = float(r[h["avg_us"]])                       # per-launch time; launches per step are in the kernel_stats summary
        us += w
        tf += w * float(r[h["mfma_TFLOPs"]])
        busy += w * float(r[h["mfma_busy_share"]])
        if top is None or float(r[h["mfma_TFLOPs"]]) > top[1]:
            top = (r[0], float(r[h["mfma_TFLOPs"]]), float(r[h["mfma_busy_share"]]))
    if not us:
        return None
    return {"source": "profiles/" + src, "gemm_kernels_mean_TFLOPs": round(tf / us, 1), "gemm_kernels_mean_frac_of_peak": round(tf / us / MFMA_BF16_PEAK_TFS, 4),
            "gemm_kernels_mean_matrix_pipe_busy_share": round(busy / us, 3),
            "best_kernel": {"kernel": top[0], "TFLOPs": top[1], "frac_of_peak": round(top[1] / MFMA_BF16_PEAK_TFS, 4), "matrix_pipe_busy_share": top[2]}}


def build_model(cfg, feats, precision, device):
    import poet_amd
    from poet_amd.synthetic import SyntheticBackbone
    bb = SyntheticBackbone(feats, cfg["strides"], cfg["num_channels"], cfg["d_model"] // 2)
    tr = poet_amd.DeformableTransformer(d_model=cfg["d_model"], nhead=cfg["nheads"], num_encoder_layers=cfg["enc_layers"],
                                        num_decoder_layers=cfg["dec_layers"], dim_feedforward=cfg["d_ffn"],
                                        dropout=cfg["dropout"], activation="relu", return_intermediate_dec=True,
                                        num_feature_levels=cfg["n_levels"], dec_n_points=cfg["n_points"],
                                        enc_n_points=cfg["n_points"])
    tr.set_precision(precision)
    model = poet_amd.PoET(bb, tr, num_queries=cfg["num_queries"], num_feature_levels=cfg["n_levels"],
                          n_classes=cfg["n_classes"], bbox_mode="gt", class_mode="specific", aux_loss=True).to(device)
    # assignment + target gather on the GPU (poet_lsa_boxes; POET_HOST_MATCHER=1: the SciPy path on the host copy of the boxes)
    host = os.environ.get("POET_HOST_MATCHER", "0") not in ("", "0")
    crit = poet_amd.SetCriterion(poet_amd.PoseMatcher(device_assign=not host), poet_amd.build_weight_dict(cfg["dec_layers"]))
    return model, crit


def synth_detections(cfg, batch, seed):
    """Detector rows (x0, y0, x1, y1, score, class) in pixels for the inference mode: 3..Q per image, as synth_batch's boxes."""
    rng = np.random.default_rng(seed)
    ih, iw = cfg["image_hw"]
    preds = []
    for _ in range(batch):
        k = int(rng.integers(3, cfg["num_queries"] + 1))
        cx, cy = rng.uniform(0.2, 0.8, k) * iw, rng.uniform(0.2, 0.8, k) * ih
        w, h = rng.uniform(0.05, 0.25, k) * iw, rng.uniform(0.05, 0.25, k) * ih
        rows = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2, rng.uniform(0.1, 1.0, k), rng.integers(1, cfg["n_classes"] + 1, k)], 1)
        preds.append(torch.from_numpy(rows.astype(np.float32)))
    return preds


def run_inference(args, cfg, device):
    """`--infer`: forward-only rate of the inference path (bbox_mode='backbone', eval, HIP-graph replay).  A secondary
    number; the default run and the driver's contract are the training metric."""
    import poet_amd
    batch = args.batch or 1
    feats, _ = synth_batch(cfg, batch, 1234, device)
    model, _ = build_model(cfg, feats, args.precision, device)
    model.bbox_mode = "backbone"
    model.backbone.predictions = synth_detections(cfg, batch, 99)
    runner = model.eval() if args.no_graphs else poet_amd.GraphedInference(model, warm=1)
    ih, iw = cfg["image_hw"]
    samples = poet_amd.NestedTensor(None, torch.zeros((batch, ih, iw), dtype=torch.bool, device=device))
    with torch.no_grad():
        for _ in range(max(args.warmup, 3)):
            runner(samples)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out, _ = runner(samples)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print(json.dumps({"metric": "images/sec inference (forward only, queries from detector rows)", "value": round(batch / dt, 1),
                      "unit": "images/s", "n_gpus": 1, "steps": args.steps, "warmup": max(args.warmup, 3), "ms_per_step": round(dt * 1e3, 3),
                      "higher_is_better": True, "dtype": args.precision, "data": "synthetic",
                      "config": {"workload": f"{args.config}: inference, bs={batch}, {'eager' if args.no_graphs else 'HIP graph'}"},
                      "finite": bool(torch.isfinite(out["pred_translation"]).all())}), flush=True)


def usable_cores():
    """Cores this process may actually use: the cgroup CPU quota when there is one (threads beyond it only get the whole
    process descheduled for the rest of each 100 ms period), else the visible CPU count."""
    n = os.cpu_count() or 8
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    return n


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def _cpu_steps(cfg, batch, warm, steps, max_seconds, cores):
    """seconds per optimisation step of the CPU oracle (fwd + loss + bwd + clip + AdamW) at `batch` images of `cfg` geometry"""
    from oracle import poet_ref
    feats, targets = synth_batch(cfg, batch, 99, "cpu")
    torch.manual_seed(0)
    model, crit = poet_ref.build_poet(cfg, feats)
    model.train()
    opt = torch.optim.AdamW(poet_ref.param_groups(model), lr=2e-4, weight_decay=1e-4)
    ih, iw = cfg["image_hw"]
    samples = poet_ref.nested_from_list([torch.zeros(3, ih, iw) for _ in range(batch)])
    times, t_begin = [], time.perf_counter()
    for it in range(warm + steps):
        t0 = time.perf_counter()
        out, nb = model(samples, targets)
        ls = crit(out, targets, nb)
        tot = sum(ls[k] * crit.weight_dict[k] for k in ls)
        opt.zero_grad()
        tot.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 0.1)
        opt.step()
        dt = time.perf_counter() - t0
        if it >= warm:
            times.append(dt)
        if time.perf_counter() - t_begin > max_seconds and len(times) >= min(3, steps):
            break
    return float(np.mean(times)), len(times)


def cpu_baseline(cfg, max_seconds=40.0):
    """The oracle (CPU restatement of the reference's PyTorch path, grid_sample MSDA) timed on the host cores of this box on a
    BOUNDED sample: the metric's geometry at bs = 2, 1 warm-up + 3 timed optimisation steps; and BASELINE.json
    configs[0] (the reference's own CPU-runnable case: 1 enc / 1 dec / 4 heads, 2 levels, 128x128, bs = 2), 3 + 20 steps."""
    cores = min(usable_cores(), 32)
    torch.set_num_threads(cores)
    sec, n = _cpu_steps(cfg, 2, 1, 3, max_seconds, cores)
    sec0, n0 = _cpu_steps(CONFIGS["cfg0"], 2, 3, 20, 15.0, cores)
    cpu = _cpu_model()
    return {"value": round(2.0 / sec, 4), "unit": "images/s", "cores": cores, "kind": "port", "cpu": cpu,
            "sample": f"metric geometry, bs=2, {n} timed fwd+loss+bwd+clip+AdamW steps of the CPU oracle after 1 warm-up "
                      f"({sec:.2f} s/step, torch {torch.__version__} CPU, {cores} threads, {cpu})",
            "cfg0": {"value": round(2.0 / sec0, 2), "unit": "images/s", "sample": f"BASELINE.json configs[0] (128x128, 1 enc / 1 dec / 4 heads, 2 levels), bs=2, "
                     f"{n0} timed steps after 3 warm-up ({1e3 * sec0:.1f} ms/step)"}}


def self_launch(n):
    """`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment): re-execute this command line as N ranks
    of ONE node through torch.distributed.run on a free loopback port -- exactly the command the driver uses (one process per GPU,
    launch_distributed.py:74-92 / main.py:219-231 in the reference).  The ranks inherit stdout: rank 0 still prints the ONE JSON
    line.  The exit code is the launcher's."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n:
        raise SystemExit(f"--gpus {n}: this node shows {have} GPU(s)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // n)))
    raise SystemExit(subprocess.call(cmd, env=env))


def time_steps(trainer, samples, targets, n, sync):
    sync()
    t0 = time.perf_counter()
    for _ in range(n):
        trainer.step(samples, targets)
    sync()
    return 1000.0 * (time.perf_counter() - t0) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="ycbv", choices=sorted(CONFIGS))
    ap.add_argument("--precision", default="bf16", choices=["bf16", "bf16_nosplit", "bf16_pure", "fp32"])
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-graphs", action="store_true", help="enqueue every kernel from Python instead of replaying HIP graphs")
    ap.add_argument("--infer", action="store_true", help="secondary number: forward-only rate of the inference path (1 GPU; --batch, default 1)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or os.environ.get("POET_BENCH_SELF_LAUNCH", "0") not in ("", "0")):
        # (POET_BENCH_SELF_LAUNCH=1 takes the launcher path at --gpus 1 too: with POET_FORCE_COLLECTIVES=1 a one-GPU box then runs
        # the whole N > 1 code path -- launcher, RCCL group, DP mode trial, exposed-collective timing -- with one rank)
        return self_launch(args.gpus)                          # plain `python bench.py --gpus N`: become the launcher of N ranks
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch {args.gpus} ranks (torch.distributed.run --nproc-per-node {args.gpus})")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    forced = os.environ.get("POET_FORCE_COLLECTIVES", "0") not in ("", "0")    # 1-rank RCCL group: exercises the N > 1 code path on one GPU
    if world > 1 or (forced and "RANK" in os.environ):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)      # 'nccl' == RCCL on ROCm

    import poet_amd
    from poet_amd import ops
    cfg = CONFIGS[args.config]
    if args.infer:
        return run_inference(args, cfg, device)
    batch = args.batch or cfg["batch"]
    torch.manual_seed(1234)                                      # identical init on every rank (then broadcast anyway)
    poet_amd.manual_seed(1234 + rank)
    feats, targets = synth_batch(cfg, batch, 1234 + rank, device)
    ih, iw = cfg["image_hw"]
    samples = poet_amd.NestedTensor(None, torch.zeros((batch, ih, iw), dtype=torch.bool, device=device))

    def sync():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    def make_trainer():
        torch.manual_seed(1234)
        model, crit = build_model(cfg, feats, args.precision, device)
        model.train()
        if args.no_graphs:
            return poet_amd.Trainer(model, crit, lr=2e-4, weight_decay=1e-4, max_norm=0.1)
        return poet_amd.GraphedTrainer(model, crit, lr=2e-4, weight_decay=1e-4, max_norm=0.1, warm=2)

    dp_trial = None
    if dist.is_initialized() and not args.no_graphs and "POET_DP_SINGLE_COLLECTIVE" not in os.environ:
        # Data-parallel launch mode chosen by MEASUREMENT on this node, not by a 1-rank trace: both modes are captured (each on
        # its own model instance, same seed), 3 replayed steps of each are timed over all ranks (max), the faster one runs the
        # timed region.  "single" = one backward graph + ONE all-reduce of the flat gradient arena behind it; "buckets" = one
        # backward segment per gradient bucket with its all-reduce on the comm stream under the later segments (main.py:282's
        # DDP overlap).  POET_DP_SINGLE_COLLECTIVE=0/1 in the environment skips the trial.
        # (round 6) a third candidate unless POET_DP_GRAD_DTYPE is pinned in the environment: the single collective with the gradients
        # travelling as bf16 (24.5 MB instead of 49 MB per step on the ring; summed in fp32 on arrival -- engine.BucketReducer).
        dp_trial, cands = {}, {}
        modes = [("single", "1", None), ("buckets", "0", None)]
        if "POET_DP_GRAD_DTYPE" not in os.environ and os.environ.get("POET_DP_TRIAL_BF16", "1") not in ("", "0"):
            modes.append(("single_bf16", "1", "bf16"))
        for mode, flag, gdt in modes:
            os.environ["POET_DP_SINGLE_COLLECTIVE"] = flag
            if gdt:
                os.environ["POET_DP_GRAD_DTYPE"] = gdt
            t = make_trainer()
            for _ in range(4):                                  # 2 eager steps + capture + 1 replay
                t.step(samples, targets)
            ms_mode = torch.tensor([time_steps(t, samples, targets, 3, sync)], dtype=torch.float64, device=device)
            dist.all_reduce(ms_mode, op=dist.ReduceOp.MAX)
            dp_trial[mode + "_ms_per_step"] = round(float(ms_mode.item()), 3)
            cands[mode] = t
            if gdt:
                del os.environ["POET_DP_GRAD_DTYPE"]
        del os.environ["POET_DP_SINGLE_COLLECTIVE"]
        best = min([m for m, _, _ in modes], key=lambda m: dp_trial[m + "_ms_per_step"])      # identical on every rank (all-reduced)
        dp_trial["chosen"] = best
        trainer = cands.pop(best)
        grad_transport = "bf16" if best == "single_bf16" else os.environ.get("POET_DP_GRAD_DTYPE", "fp32")
        cands.clear()
        torch.cuda.empty_cache()
    else:
        trainer = make_trainer()
        grad_transport = os.environ.get("POET_DP_GRAD_DTYPE", "fp32")
    model = trainer.model
    if not args.no_graphs:
        args.warmup = max(args.warmup, 4)                       # 2 eager steps + capture + 1 replay before timing
    for _ in range(args.warmup):
        trainer.step(samples, targets)
    # Input hand-over inside the timed region: the batch is staged AHEAD of its step in the graphs' static-input layout
    # (GraphedTrainer.pack: what a prefetcher does under the previous step -- data_prefetcher.py:22-78 in the reference), and every
    # timed step moves it into the graphs' input area with ONE streaming copy (~103 MB at 640x480, bs 16).  POET_BENCH_NO_PACK=1: the
    # per-field staging of rounds 1-5 (3 feature copies + ~15 small launches per step) inside the timed region instead (A/B aid).
    packed = None
    if not args.no_graphs and os.environ.get("POET_BENCH_NO_PACK", "0") in ("", "0") and getattr(trainer, "graph_loss", False):
        packed = trainer.pack(samples, targets)
        trainer.step(packed)
    batch_in = (packed,) if packed is not None else (samples, targets)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        total, loss_dict = trainer.step(*batch_in)
        poet_amd.reduce_dict(loss_dict)                          # engine.py:61 (logging all-reduce), no .item()
    t_enq = time.perf_counter() - t0                              # host time to ENQUEUE the K steps (no device sync inside)
    sync()
    elapsed = time.perf_counter() - t0
    # host cost of ONE step with an empty GPU queue (the in-loop figure above includes back-pressure: the runtime lets the host
    # run only about one graph launch ahead, so at steady state the enqueue loop is paced by the GPU)
    t1 = time.perf_counter()
    trainer.step(*batch_in)
    t_host = time.perf_counter() - t1
    sync()
    exposed = None
    if dist.is_initialized() and not args.no_graphs and getattr(trainer, "segs", None) is not None:
        # what the gradient all-reduces cost on the critical path: the same replayed step with the collectives left out (the ranks'
        # parameters drift apart for these 3 steps -- AFTER the timed region, nothing measured depends on them)
        with_ms = time_steps(trainer, samples, targets, 3, sync)
        trainer.skip_collectives = True
        without_ms = time_steps(trainer, samples, targets, 3, sync)
        trainer.skip_collectives = False
        tt = torch.tensor([with_ms, without_ms], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        exposed = {"with_collectives_ms": round(float(tt[0]), 3), "without_collectives_ms": round(float(tt[1]), 3),
                   "allreduce_exposed_ms": round(float(tt[0] - tt[1]), 3)}
    # SURVEY 8(d) secondary line, STRONG scaling: the same GLOBAL batch as the 1-GPU run (cfg batch) cut over the ranks -- 16 / N images
    # per GPU at YCB-V (2 at 8 GPUs), where the gradient all-reduce is hardest to hide.  A second trainer (same model seed, same
    # data-parallel mode), captured and timed after the weak-scaling region; its own field of the ONE line, `value` stays the weak
    # figure.  world == 1: identical to `value` by definition (POET_BENCH_STRONG=<per-GPU batch> forces the leg for testing).
    strong = None
    forced_b = int(os.environ.get("POET_BENCH_STRONG", "0") or 0)
    if not args.no_graphs and not args.batch and ((world > 1 and cfg["batch"] % world == 0 and cfg["batch"] // world >= 1) or forced_b):
        try:
            b2 = forced_b or cfg["batch"] // world
            feats2, targets2 = synth_batch(cfg, b2, 4321 + rank, device)
            samples2 = poet_amd.NestedTensor(None, torch.zeros((b2, ih, iw), dtype=torch.bool, device=device))
            if dp_trial:
                os.environ["POET_DP_SINGLE_COLLECTIVE"] = "0" if dp_trial["chosen"] == "buckets" else "1"
                if dp_trial["chosen"] == "single_bf16":
                    os.environ["POET_DP_GRAD_DTYPE"] = "bf16"
            torch.manual_seed(1234)
            model2, crit2 = build_model(cfg, feats2, args.precision, device)
            model2.train()
            t2 = poet_amd.GraphedTrainer(model2, crit2, lr=2e-4, weight_decay=1e-4, max_norm=0.1, warm=2)
            for _ in range(4):
                t2.step(samples2, targets2)
            k2 = max(5, min(args.steps, 20))
            ms2 = torch.tensor([time_steps(t2, samples2, targets2, k2, sync)], dtype=torch.float64, device=device)
            if dist.is_initialized():
                dist.all_reduce(ms2, op=dist.ReduceOp.MAX)
            ms2 = float(ms2.item())
            strong = {"scaling": "strong", "global_batch": b2 * world, "per_gpu_batch": b2, "steps": k2, "ms_per_step": round(ms2, 3),
                      "images_per_s": round(world * b2 / (ms2 / 1e3), 2)}
            del t2, model2, crit2
            torch.cuda.empty_cache()
        except Exception as e:                                   # the secondary leg must never cost the primary line
            strong = {"scaling": "strong", "error": f"{type(e).__name__}: {e}"[:300]}
        finally:
            if dp_trial:
                os.environ.pop("POET_DP_SINGLE_COLLECTIVE", None)
                if dp_trial["chosen"] == "single_bf16":
                    os.environ.pop("POET_DP_GRAD_DTYPE", None)
    prof, prof_steps = None, 3
    if not args.no_roofline:
        # per-kernel HIP-event timing on the launch stream, over 3 EXTRA steps of the same workload right after the
        # timed region (kept out of it so the events do not perturb `value`).  At N > 1 every rank runs the extra steps
        # (they contain the gradient all-reduces); rank 0 alone records events.
        eager = trainer if args.no_graphs else super(poet_amd.GraphedTrainer, trainer)     # per-launch events need eager launches
        ops.SEED_DEV[0] = None
        if rank == 0:
            ops.PROFILE.start()
        for _ in range(prof_steps):
            # keep the GPU busy while the host enqueues the eager step: with an idle queue the begin-event of a launch
            # is stamped when it is recorded and the pair would time host latency, not the kernel
            torch.cuda._sleep(int(0.25 * 2.4e9))
            eager.step(samples, targets)
            sync()
        if rank == 0:
            prof = ops.PROFILE.stop()
    if dist.is_initialized():
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_val = float(total)
    if not np.isfinite(loss_val):
        raise SystemExit(f"non-finite loss {loss_val}")

    if rank == 0:
        ms = 1000.0 * elapsed / args.steps
        value = world * batch * args.steps / elapsed
        fl = 3 * gemm_flops_per_image(cfg) * batch
        out = {
            "metric": "images/sec fwd+bwd (loss + backward + clip + AdamW), PoET encoder-decoder",
            "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"bf16": "bf16", "bf16_nosplit": "bf16", "bf16_pure": "bf16", "fp32": "f32"}[args.precision], "data": "synthetic",
            "config": {"workload": f"{args.config}: {cfg['enc_layers']} enc / {cfg['dec_layers']} dec / {cfg['nheads']} heads, "
                                   f"{cfg['d_model']}-d, {cfg['n_levels']} levels, {iw}x{ih}, {cfg['num_queries']} queries, "
                                   f"bs={batch} per GPU, dropout {cfg['dropout']}, AdamW + clip 0.1, random init",
                       "global_batch": world * batch, "tokens_per_image": sum(h * w for h, w in cfg["level_hw"]),
                       "parallelism": f"dp{world}", "precision_policy": args.precision,
                       **({"dp_mode": ("one backward graph + ONE all-reduce of the flat gradient arena" if getattr(trainer, "single_collective", False)
                                       else "one backward segment + one all-reduce per gradient bucket (comm stream)") +
                                      f", gradients travel as {grad_transport}"} if dist.is_initialized() else {}),
                       **({"dp_mode_trial": dp_trial} if dp_trial else {}), **(exposed or {}),
                       "launch": "eager" if args.no_graphs else ("hipGraph replay (fwd + matcher + loss graph, bwd + clip + AdamW graph)" if getattr(trainer, "graph_loss", False) else "hipGraph replay (fwd graph, eager loss, bwd+opt graph)"),
                       "input_handover": ("one packed streaming copy per step (batch staged ahead by GraphedTrainer.pack)" if packed is not None
                                          else "per-field staging inside the step"),
                       "gemm_tflops_per_step_algorithmic": round(fl / 1e12, 3),
                       "gemm_tflops_achieved_whole_step": round(fl / 1e12 / (ms / 1e3), 1), "final_loss": round(loss_val, 4),
                       "host_enqueue_ms_per_step": round(1000.0 * t_enq / args.steps, 3),
                       "host_enqueue_ms_one_step_idle_queue": round(1000.0 * t_host, 3)},
        }
        if strong is not None:
            out["strong_scaling"] = strong
        if prof is not None:
            out["roofline"] = ops.PROFILE.roofline(prof, prof_steps, HBM_PEAK_GBS, MFMA_BF16_PEAK_TFS)
            out["roofline"]["traffic"] = pmc_traffic_bytes(out["roofline"]["kernel"], args.config)
            out["roofline"]["traffic_raw"] = pmc_traffic_bytes(out["roofline"]["kernel"], args.config, fetch_factor=1.0)
            out["roofline"]["traffic_source"] = pmc_source(args.config)
            # the three kernels (symbol x shape) that take the most time, each against its own roof, with the counter traffic
            # of the committed PMC summary next to the algorithmic bytes
            top = []
            for tg in sorted(prof, key=lambda k: -prof[k]["total_ms"])[:3]:
                r3 = ops.PROFILE.roofline(prof, prof_steps, HBM_PEAK_GBS, MFMA_BF16_PEAK_TFS, tag=tg)
                tr3 = pmc_traffic_bytes(tg, args.config)
                top.append({"kernel": tg, "ms_per_step": r3["ms_per_step"], "avg_launch_us": r3["avg_launch_us"], "bound": r3["bound"], "frac": r3["frac"],
                            "algorithmic_bytes_per_launch": r3["algorithmic_bytes_per_launch"], "traffic": tr3,
                            "traffic_over_algorithmic": round(tr3 / r3["algorithmic_bytes_per_launch"], 2) if tr3 and r3["algorithmic_bytes_per_launch"] else None})
            out["roofline"]["top3"] = top
            # the step's matrix work next to the dominant (HBM / LDS-atomic bound) kernel: algorithmic GEMM flops of the
            # step over the MFMA kernels' own time, and what the SQ counters say about them
            gemm_ms = sum(v["total_ms"] for k, v in prof.items() if k.startswith("gemm")) / prof_steps
            out["roofline"]["mfma"] = {"peak": MFMA_BF16_PEAK_TFS, "unit": "TFLOP/s",
                                       "gemm_kernels_ms_per_step": round(gemm_ms, 3),
                                       "achieved_over_gemm_kernel_time": round(fl / 1e12 / (gemm_ms / 1e3), 1) if gemm_ms else None,
                                       "frac": round(fl / 1e12 / (gemm_ms / 1e3) / MFMA_BF16_PEAK_TFS, 4) if gemm_ms else None,
                                       "pmc": pmc_mfma_summary(args.config)}
            if out["roofline"]["kernel"].startswith("msda_bwd_dvalue_scatter"):
                # the contract's two bounds do not name this kernel's real limiter; say so next to the HBM fraction
                out["roofline"]["limiter"] = ("two on-chip pipes, not HBM: VALU issue (~310 instructions per 4 queries x 64 corners x 16 channels = ~258 us) AND the LDS "
                                              "atomic pipe (1.67 G lane-atomics per launch at its measured 6.8 T/s = 246 us) at ~70 % each, + ~65 us of max pass and flush; "
                                              "round 6 moved the 64 x 16 weight x gradient products per query-head to the matrix pipe (v_mfma_f32_4x4x1 outer products: "
                                              "VALU per iteration 277 -> 126 ns, parity-exact) and the kernel got 5 % SLOWER: DESIGN.md sections 0 and 8b")
            out["kernel_breakdown_ms_per_step"] = {k: round(v["total_ms"] / prof_steps, 3)
                                                   for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])}
            bsum = sum(v["total_ms"] for v in prof.values()) / prof_steps
            out["kernel_breakdown_note"] = (f"EAGER launches timed by per-launch HIP events over {prof_steps} extra steps after the timed region; "
                                            f"NOT additive: the sum ({bsum:.3f} ms) exceeds ms_per_step ({ms:.3f} ms, hipGraph replay) because "
                                            "each of the ~400 small launches carries event overhead and replayed kernels overlap head-to-tail; "
                                            "the large kernels agree with the rocprofv3 kernel trace under profiles/")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
