#!/usr/bin/env python3
"""Headline benchmark: train images/sec of UNet3p_SNws (14 bands -> 10 classes, n_filters 64) at 512x512 in bf16,
batch 16 per GPU, synthetic FY-4A-shaped tiles, data-parallel over N GPUs of one node (one process per GPU).

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts its own N ranks as child processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = zero_grad -> forward -> CE/N loss -> backward -> (bucketed gradient all-reduce) -> AdamW step with the poly
LR, i.e. Trainer.training's inner loop.  Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON
line.  `roofline` is for the dominant kernel (the 3x3 MFMA implicit-GEMM kernel conv_fwd_kernel<32,9,...> that runs
forward and dgrad, in both of its epilogue variants): algorithmic FLOPs of its launches / their HIP-event durations
inside the timed region, against the dense bf16 MFMA peak.  (`kernels` lists the variants separately:
conv_mfma_dgradsn_* are the dgrad launches whose epilogue also runs the SwitchNorm backward reduction; sn_apply /
sn_bwd_apply are the HBM-bound SwitchNorm streaming passes, in GB/s of their algorithmic 2 / 3 tensor passes.)
`cpu_baseline` times the CPU oracle (a port of the reference's arithmetic) on rank 0's host cores on a bounded
sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "semantic-segmentation_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 matrix peak of MI355X
MFMA_F32_PEAK_TFLOPS = 157.3     # f32-input matrix instruction (= the f32 vector rate)
HBM_PEAK_GBS = 8000.0


def kernel_source_hash():
    import hashlib
    with open(os.path.join(ROOT, "semantic-segmentation_amd", "csrc", "conv_mfma.hip"), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def pmc_traffic():
    """(HBM bytes per launch of the roofline kernel, where the figure comes from) from the newest
    profiles/r*_pmc_bench_conv.json - or (None, why) when that summary was measured on another version of the kernel
    source than the one in the tree (a stale figure is worse than none)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_bench_conv.json")))
    if not files:
        return None, "no profiles/r*_pmc_bench_conv.json"
    with open(files[-1]) as f:
        d = json.load(f)
    name = "profiles/" + os.path.basename(files[-1])
    if d.get("source_sha256") != kernel_source_hash():
        print("bench.py: %s was measured on a different conv_mfma.hip than the one in the tree; roofline.traffic left "
              "null (re-run tools/run_pmc_bench.sh)" % name, file=sys.stderr)
        return None, name + " is stale: taken from another version of csrc/conv_mfma.hip"
    return d.get("hbm_bytes_per_launch"), name + " (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)"


def cpu_baseline(n_filters, size, seconds_budget=30.0):
    """CPU oracle (oracle/snws_oracle.py) forward + loss + backward of the same model on ONE image of the bench
    resolution (if that is estimated to exceed the budget, a square crop whose side is halved until it fits)."""
    from oracle import snws_oracle as O
    torch.manual_seed(1)
    state = O.make_state("unet3p", 14, n_filters, 10)
    for k, v in state.items():
        if k.endswith(".0.weight") or k == "output_seg_map.weight":
            v.copy_(torch.randn_like(v) * 0.1)
    side = size
    # probe at 64x64 to scale the sample so the timed run stays within the budget (cost ~ pixels)
    img = torch.rand(1, 14, 64, 64) * 255.0
    lab = torch.randint(0, 10, (1, 64, 64)).float()
    t0 = time.perf_counter()
    O.model_step("unet3p", state, img, lab)
    probe = time.perf_counter() - t0
    while side > 64 and probe * (side / 64.0) ** 2 > seconds_budget:
        side //= 2
    img = torch.rand(1, 14, side, side) * 255.0
    lab = torch.randint(0, 10, (1, side, side)).float()
    # repeat the step until about 10 s of CPU work have been timed (never beyond the budget)
    reps, dt = 0, 0.0
    t0 = time.perf_counter()
    while True:
        O.model_step("unet3p", state, img, lab)
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= 10.0 or dt * (reps + 1) / reps > seconds_budget:
            break
    frac = (side * side) / float(size * size)
    return {"value": reps * frac / dt, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "CPU oracle fwd+CE+bwd of UNet3p_SNws(n_filters=%d) f32, %d step(s) on one synthetic %dx%d tile "
                      "(=%.4g of a %dx%d image each), %.1f s" % (n_filters, reps, side, side, frac, size, size, dt)}


def evaluator_throughput(device, reps=24, pairs=6):
    """Evaluator.add_batch on full FY-4A disks (1 x 2748 x 2748, f32 labels + i64 predictions, 10 classes, resident
    in HBM): 12 algorithmic bytes per pixel.  Two figures: `hbm` rotates `pairs` distinct (label, prediction) disks -
    6 x 90.6 MB = 544 MB, more than the 256 MB Infinity Cache holds, so every call streams from HBM; `cache_resident`
    repeats ONE disk pair (90.6 MB stays in the Infinity Cache: an on-die figure, not HBM evidence)."""
    from utils.metrics import Evaluator
    g = torch.Generator(device="cpu").manual_seed(0)
    gts = [torch.randint(0, 10, (1, 2748, 2748), generator=g).float().to(device) for _ in range(pairs)]
    prs = [torch.randint(0, 10, (1, 2748, 2748), generator=g).to(device) for _ in range(pairs)]
    npix = gts[0].numel()
    ev = Evaluator(10)

    def timed(order, gts=gts, prs=prs):
        ev.reset()
        for i in order[:len(gts)]:
            ev.add_batch(gts[i], prs[i])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in order:
            ev.add_batch(gts[i], prs[i])
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / len(order)
    sec_hbm = timed([i % pairs for i in range(reps)])
    miou = float(ev.Mean_Intersection_over_Union())
    sec_cache = timed([0] * reps)

    def fig(sec, px=None, key="us_per_disk"):
        px = npix if px is None else px
        return {key: sec * 1e6, "Mpix/s": px / sec / 1e6, "GB/s": 12.0 * px / sec / 1e9,
                "frac_of_hbm_peak": 12.0 * px / sec / 1e9 / HBM_PEAK_GBS}
    # the other shape the reference's validation loop feeds it (train.py:294-298): one 16 x 512 x 512 batch per call;
    # 11 distinct batch pairs (553 MB) rotated so that they stream from HBM as well
    vb = 11
    vgts = [torch.randint(0, 10, (16, 512, 512), generator=g).float().to(device) for _ in range(vb)]
    vprs = [torch.randint(0, 10, (16, 512, 512), generator=g).to(device) for _ in range(vb)]
    sec_val = timed([i % vb for i in range(4 * vb)], vgts, vprs)
    val_px = vgts[0].numel()
    return {"workload": "Evaluator.add_batch, 1x2748x2748 f32 gt + i64 pred, 10 classes",
            "validation_batch": dict(fig(sec_val, val_px, "us_per_batch"),
                                     note="16x512x512 f32 gt + i64 pred per call, %d distinct batch pairs rotated (%.0f MB)"
                                          % (vb, vb * 12.0 * val_px / 1e6)),
            "hbm": dict(fig(sec_hbm), note="%d distinct disk pairs rotated (%.0f MB > 256 MB Infinity Cache)"
                                            % (pairs, pairs * 12.0 * npix / 1e6)),
            "cache_resident": dict(fig(sec_cache), note="one disk pair repeated (90.6 MB, served by the Infinity Cache)"),
            "mIoU": miou}


def stream_ceiling(device, reps=10):
    """What a plain streaming pass reaches on this box: the library's own copy kernel (snws_resample_fwd, mode copy)
    over a 537 MB bf16 NHWC tensor - 1 read + 1 write, the same shape of traffic as sn_apply.  The SwitchNorm passes
    are reported against the 8 TB/s data-sheet peak AND against this measured figure."""
    import ctypes
    from snws import _lib as L
    x = torch.randn(16, 512, 512, 64, device=device).to(torch.bfloat16)
    y = torch.empty_like(x)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run():
        L.call("snws_resample_fwd", x.data_ptr(), 64, y.data_ptr(), 64, L.SNWS_BF16, 16, 512, 512, 64, 0, 1, st)
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * x.numel() * 2 * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def disk_inference(model, device, rank, world, reps=3, tiles_per_forward=12):
    """Full-disk inference: one synthetic (14, 2748, 2748) FY-4A disk resident in HBM -> 36 tiles of 512^2 in eval
    mode, arg-max and stitch on the device, Evaluator against a synthetic label map; tiles are dealt round-robin
    to the ranks."""
    from snws import tiles
    from utils.metrics import Evaluator
    g = torch.Generator(device="cpu").manual_seed(0)
    disk = (torch.rand((14, 2748, 2748), generator=g) * 255.0).to(device=device, dtype=model.compute_dtype or torch.float32)
    gt = torch.randint(0, 10, (1, 2748, 2748), generator=g).float().to(device)
    pred = tiles.predict_disk(model, disk, block=512, overlap=0, tiles_per_forward=tiles_per_forward)     # warm-up
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        pred = tiles.predict_disk(model, disk, block=512, overlap=0, tiles_per_forward=tiles_per_forward)
        ev = Evaluator(10)
        ev.add_batch(gt, pred[None])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    sec = (time.perf_counter() - t0) / reps
    return {"workload": "one 14x2748x2748 disk -> 36 tiles of 512x512 (eval, %d tiles per forward), arg-max + stitch + "
                        "Evaluator on device" % tiles_per_forward,
            "seconds_per_disk": sec, "disks_per_s": 1.0 / sec, "tiles_per_s": 36.0 / sec, "Mpix/s": 2748 * 2748 / sec / 1e6,
            "mIoU_vs_random_labels": float(ev.Mean_Intersection_over_Union())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--filters", type=int, default=64)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--model", default="unet3p", choices=["unet3p", "unet3p_res_ocr"],
                    help="unet3p = the headline UNet3p_SNws; unet3p_res_ocr = UNet3p_res_ocr_SNws (residual encoder + OCR head)")
    ap.add_argument("--disk", action="store_true", help="also time full-disk tiled inference (36 tiles of 512^2)")
    ap.add_argument("--disk-tiles", type=int, default=12, help="tiles per forward pass of the full-disk inference")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--all-kernel-events", action="store_true",
                    help="bracket EVERY kernel launch with HIP events (fills the `kernels` table; costs ~5 ms of host "
                         "time per step); default: the roofline kernel's launches and the SwitchNorm apply passes")
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, as children of this process, BEFORE
    # anything here touches the GPU, and relay rank 0's result line (train.py:210-214 needs no launcher either)
    from snws import launch
    if args.gpus > 1 and not launch.launched():
        sys.exit(launch.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))

    # stdout carries exactly one line, the JSON result: whatever libraries print there (RCCL announces its version
    # on stdout when a communicator is created) is sent to stderr instead, on every rank
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    from snws import harness, ops
    from modeling.UNet3p_SNws import UNet3p_SNws
    from modeling.UNet3p_res_ocr_SNws import UNet3p_res_ocr_SNws

    rank, world, device = harness.init_distributed()
    grouped = dist.is_initialized()      # world > 1, or a one-rank self-test group
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))

    torch.manual_seed(1)
    cls = UNet3p_SNws if args.model == "unet3p" else UNet3p_res_ocr_SNws
    model = cls(14, args.filters, 10, 1, 1).to(device)
    model.compute_dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    if grouped:   # identical replicas
        for t in list(model.parameters()) + list(model.buffers()):
            ops.broadcast(t.data, 0)
    trainer = harness.Trainer(model, lr=0.002, weight_decay=5e-4, total_iters=max(1000, args.steps + args.warmup))
    img, lab = harness.synthetic_batch(args.batch, 14, args.size, 10, device, seed=1 + rank,
                                       dtype=model.compute_dtype)

    def barrier():
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = trainer.step(img, lab)
    barrier()
    ops.PROFILE = None if args.no_kernel_events else []
    # default: the roofline kernel's launches plus the two SwitchNorm streaming passes (HBM-bound: GB/s in `kernels`)
    roof_tags = ("conv_mfma_fwd_k3_kc32", "conv_mfma_dgrad_k3_kc32", "conv_mfma_dgradsn_k3_kc32") if args.dtype == "bf16" \
        else ("conv_f32_fwd_k3", "conv_f32_dgrad_k3")
    ops.PROFILE_ONLY = None if args.all_kernel_events else roof_tags + ("sn_apply", "sn_bwd_apply")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.step(img, lab)
    barrier()
    elapsed = time.perf_counter() - t0
    prof, ops.PROFILE = ops.PROFILE, None
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if grouped:
        ops.all_reduce(tmax, "max")
    elapsed = float(tmax.item())
    loss_val = float(loss)
    if grouped:
        lt = loss.detach().clone()
        ops.all_reduce(lt, "sum")
        loss_val = float(lt)

    disk = disk_inference(model, device, rank, world, tiles_per_forward=args.disk_tiles) if args.disk else None
    if rank == 0:
        kernels = {}
        for tag, work, e0, e1 in (prof or []):
            k = kernels.setdefault(tag, [0, 0.0, 0.0])
            k[0] += 1
            k[1] += work
            k[2] += e0.elapsed_time(e1) * 1e-3
        fam = [v for t, v in kernels.items() if t in roof_tags]
        peak = MFMA_BF16_PEAK_TFLOPS if args.dtype == "bf16" else MFMA_F32_PEAK_TFLOPS
        roofline = None
        if fam:
            fl, tt, cnt = sum(v[1] for v in fam), sum(v[2] for v in fam), sum(v[0] for v in fam)
            ach = fl / tt / 1e12
            # HBM bytes per launch: PMC counters cannot be read from inside this process; the figure comes from the
            # committed summary of the rocprofv3 --pmc passes over this same command (tools/run_pmc_bench.sh), and
            # only while that summary was taken from the kernel source that is built now (it records the source's hash)
            traffic, tsrc = None, None
            if args.model == "unet3p" and args.batch == 16 and args.size == 512 and args.filters == 64 and args.dtype == "bf16":
                traffic, tsrc = pmc_traffic()
            kname = "conv_fwd_kernel<KC=32,TAPS=9> (forward + dgrad launches, both epilogue variants)" if args.dtype == "bf16" \
                else "conv_f32_fwd_kernel<TAPS=9> (forward + dgrad launches; v_mfma_f32_32x32x2_f32)"
            roofline = {"bound": "mfma", "kernel": kname,
                        "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                        "frac": ach / peak, "traffic": traffic, "traffic_source": tsrc, "launches": cnt,
                        "avg_launch_ms": tt / cnt * 1e3, "flops_per_launch": fl / cnt}
        summary = {}
        ceiling = stream_ceiling(device)
        for tag, (cnt, work, tt) in sorted(kernels.items()):
            if tag.startswith(("conv", "gemm")):
                summary[tag] = {"launches": cnt, "ms_per_step": tt / args.steps * 1e3, "TFLOP/s": work / tt / 1e12}
            else:
                summary[tag] = {"launches": cnt, "ms_per_step": tt / args.steps * 1e3, "GB/s": work / tt / 1e9,
                                "frac_of_hbm_peak": work / tt / 1e9 / HBM_PEAK_GBS,
                                "frac_of_measured_stream_ceiling": work / tt / 1e9 / ceiling}
        out = {
            "metric": "train images/sec (node) %s 512x512" % cls.__name__, "value": args.batch * world * args.steps / elapsed,
            "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "%s(n_channels=14, n_filters=%d, n_class=10) training step, "
                                   "%dx%d synthetic 14-band tiles, batch %d per GPU, AdamW + poly LR, CE/N loss"
                                   % (cls.__name__, args.filters, args.size, args.size, args.batch),
                       "global_batch": args.batch * world, "parallelism": "dp%d" % world},
            "loss": loss_val, "roofline": roofline, "kernels": summary,
            "stream_ceiling": {"GB/s": ceiling, "what": "snws_resample_fwd copy, 537 MB bf16 read + write, measured in this run"},
        }
        out["evaluator"] = evaluator_throughput(device)
        if disk is not None:
            out["disk_inference"] = disk
        if world == 1 and not args.no_cpu_baseline and args.model == "unet3p":
            out["cpu_baseline"] = cpu_baseline(args.filters, args.size)
        print(json.dumps(out), file=result_out, flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
