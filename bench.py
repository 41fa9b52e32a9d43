#!/usr/bin/env python3
"""bench.py — headline benchmark: 112x112 face embeddings/s (IR-ResNet-100 "ArcFace") on MI355X.

A "step" embeds one batch of synthetic 112x112x3 faces (uniform integer pixels 0..255, seed 0;
synthetic weights seed 1 — SURVEY.md §8d) already resident in HBM: stem -> 49 residual units -> FC ->
L2 normalise, exactly the launch chain ArcFace.process runs.  N GPUs = N independent shards of the
pool (weak scaling, no data-path collective: images are independent).

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (implicit-GEMM conv
kernel: algorithmic FLOPs / HIP-event kernel time, vs the 2.5 PFLOP/s dense bf16 MFMA peak) and
`cpu_baseline` (the CPU oracle timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class _stdout_to_stderr(object):
    """fd-level redirect: RCCL prints a start-up banner on STDOUT when the first communicator is made;
    the contract is ONE JSON line on stdout."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *a):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)

MFMA_PEAK_TFLOPS = 2516.6    # 256 CU x 2.4 GHz x 4096 FLOP/clk/CU dense bf16/f16 (MI355X_MICROARCH.md)
F32_MFMA_PEAK_TFLOPS = 157.3  # 256 CU x 2.4 GHz x 256 FLOP/clk/CU, v_mfma_f32_32x32x2_f32 (the float32 mode's pipe)


def _launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, one process per GPU, through
    torch.distributed.run — as a CHILD process, before this one has imported torch or touched the GPU (a process that
    has initialised the GPU must never exec or be replaced) — relay rank 0's single JSON line, return the child's code."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC only on this driver (RCCL across processes)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in proc.stdout:
        t = ln.strip()
        if t.startswith("{") and '"metric"' in t:
            line = t
        elif t:
            print(t, file=sys.stderr)
    rc = proc.wait()
    if line is not None and rc == 0:
        print(line)
    elif rc == 0:
        print("bench.py: the ranks finished without printing a result line", file=sys.stderr)
        rc = 1
    return rc



def _identity_pool(n, seed, per_person=32, n_gallery=16, person_seed=7):
    """Synthetic pool for the config-3 / config-4 legs, made on the device: n uint8 112x112 images of n / per_person identities (a
    blocky random base face of 8x8-pixel blocks + per-image pixel noise and a brightness shift, like
    tests/golden/make_golden_config3.py) and one further image of each of the first n_gallery identities as the gallery — (pool,
    gallery) pairs then range from "same person" to "unrelated" like an unlabeled pool against enrolled faces.  The PERSONS come
    from person_seed (the same on every rank and for the calibration sample: the first n_gallery of them are the enrolled ones),
    the images of them from `seed`."""
    import torch
    gp = torch.Generator(device="cuda").manual_seed(person_seed)
    g = torch.Generator(device="cuda").manual_seed(seed)
    persons = max((n + per_person - 1) // per_person, n_gallery)
    coarse = torch.randint(40, 216, (persons, 14, 14, 3), generator=gp, device="cuda", dtype=torch.int16)
    bases = coarse.repeat_interleave(8, dim=1).repeat_interleave(8, dim=2)

    def draw(base_idx, gen):
        m = base_idx.numel()
        out = torch.empty((m, 112, 112, 3), dtype=torch.uint8, device="cuda")
        for i in range(0, m, 2048):
            b = bases[base_idx[i:i + 2048]]
            nz = torch.randint(-40, 41, b.shape, generator=gen, device="cuda", dtype=torch.int16)
            sh = torch.randint(-20, 21, (b.shape[0], 1, 1, 1), generator=gen, device="cuda", dtype=torch.int16)
            out[i:i + 2048] = (b + nz + sh).clamp_(0, 255).to(torch.uint8)
        return out
    pool = draw(torch.arange(n, device="cuda") // per_person, g)
    gallery = draw(torch.arange(n_gallery, device="cuda"), torch.Generator(device="cuda").manual_seed(person_seed + 1))
    return pool, gallery


def _spread_head(head, L, R, li, ri):
    """Rescale the head's last layer so that its logit difference runs from about -2 to +2 (p from 0.12 to 0.88) between the
    1st and the 99th percentile of these pairs, centred between them — a fresh glorot head puts every pair at 0.5 +- 0.02,
    where no selection rule has anything to decide (same construction as the golden fixture's heads)."""
    import numpy as np
    p = head.predict_device(L, R, li, ri).double().cpu().numpy()
    t = np.log(p[:, 1]) - np.log(p[:, 0])
    lo, hi = np.percentile(t, [1, 99])
    gain = 4.0 / max(hi - lo, 1e-12)
    ws = head.get_weights()
    ws[4] = (ws[4] * np.float32(gain)).astype(np.float32)
    ws[5] = np.array([0, -gain * 0.5 * (lo + hi)], np.float32) + ws[5] * np.float32(gain)
    head.set_weights(ws)


def _timed(fn, reps, barrier, dist):
    import torch
    fn()                                   # untimed: workspaces of every stream slot exist afterwards
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    barrier()
    t = torch.tensor([(time.perf_counter() - t1) / reps], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item()), r


def _timed_once(fn, barrier, dist):
    """one timed call (already warm), max over ranks"""
    import torch
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    barrier()
    t = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item()), r


def _preflight(dist, rank, world, local_rank, quiet=False):
    """Every collective shape the timed legs use, on tiny known data, each under a watchdog: a rank that cannot reach its peers
    fails HERE, within seconds and with the step's name, instead of hanging the timed run.  Returns 0 / raises."""
    import threading
    import numpy as np
    import torch
    from a_link_amd import distributed as D
    from a_link_amd.backbone import IRBackbone
    from a_link_amd import weights as W
    step = {"name": "start"}
    done = threading.Event()

    def watchdog():
        if not done.wait(60.0):
            sys.stderr.write("[preflight] rank %d of %d (cuda:%d) STUCK in step %r for 60 s: a peer is unreachable or has died — "
                             "check HSA_ENABLE_IPC_MODE_LEGACY=0, MASTER_ADDR/PORT, one process per GPU\n" % (rank, world, local_rank, step["name"]))
            sys.stderr.flush()
            os._exit(3)
    threading.Thread(target=watchdog, daemon=True).start()
    ok = []
    try:
        if dist is None:
            step["name"] = "no process group (one process): nothing to exchange"
        else:
            g = dist.group.WORLD
            step["name"] = "rank placement (one process per GPU)"
            D.check_rank_placement(g)                        # two ranks on one device: every rank raises, before any collective
            ok.append("placement")
            step["name"] = "all_reduce of a known vector"
            t = torch.arange(4, dtype=torch.float64, device="cuda") + rank
            dist.all_reduce(t)
            want = world * np.arange(4) + world * (world - 1) / 2.0
            assert np.array_equal(t.cpu().numpy(), want), (t, want)
            ok.append("all_reduce")
            step["name"] = "merge_topk on known candidates"
            vals = torch.tensor([10.0 - rank, 1.0], device="cuda")
            v, i = D.merge_topk(vals, torch.tensor([2 * rank, 2 * rank + 1], device="cuda"), 3, largest=True, group=g)
            want_i = [j for _, j in sorted([(-(10.0 - r), 2 * r) for r in range(world)] + [(-1.0, 2 * r + 1) for r in range(world)])[:3]]
            assert i.cpu().tolist() == want_i, (i, want_i)
            ok.append("merge_topk")
            step["name"] = "RowShards gathers"
            sh = D.RowShards(5 * world + 3, g)
            table = np.arange((5 * world + 3) * 2, dtype=np.float32).reshape(-1, 2)
            assert np.array_equal(sh.all_rows(table[sh.lo:sh.hi]), table)
            req = [np.array([0, 4, 5 * world + 2]), np.arange(sh.P)[::3]]
            got = sh.subsets(req, [table[sh.lo:sh.hi][sh.owned(r)] for r in req], (2,))
            assert all(np.array_equal(a, table[r]) for a, r in zip(got, req))
            assert sh.bcast(rank) == 0 and sh.all_true(True) and (world == 1 or not sh.same_everywhere(rank))
            ok.append("row_shards")
            step["name"] = "calibration broadcast"
            pr = W.synthetic_ir_params((1, 1, 1, 1), size=(32, 32), seed=3)
            bb = IRBackbone(pr, image_size=(32, 32), max_batch=8, dtype="f16x2", device=local_rank)
            rng = np.random.default_rng(rank)
            bb.calibrate(rng.integers(0, 256, (8, 32, 32, 3)).astype(np.float32) * (1.0 + 3.0 * rank))
            D.broadcast_calibration([bb], group=g)
            assert sh.same_everywhere(bb.state()), "calibration differs between ranks after the broadcast"
            ok.append("broadcast_calibration")
            step["name"] = "barrier"
            dist.barrier()
    finally:
        done.set()
    if not quiet:
        print("[preflight] rank %d of %d on cuda:%d: %s ok" % (rank, world, local_rank, ", ".join(ok) or "single process"), file=sys.stderr if rank else sys.stdout)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed steps (default 100: ~2.6 s of the headline workload, long enough for "
                    "a power / utilisation sampler beside the run to see it)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--model", default="r100", choices=["r100", "r50", "r34", "r18"])
    ap.add_argument("--batch", type=int, default=1168, help="images per step per GPU")
    ap.add_argument("--chunk", type=int, default=292, help="images per alink_embed call: 292 x 196 pixels = 511 workgroups of the "
                    "14-wide linear-tile kernel for the chip's 512 slots, 1022 for 1024 at 28 wide (chunks of a step are "
                    "issued round-robin on --streams streams)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32", "f16x2"], help="activation / weight storage of the backbone; f32 = the "
                    "reference's own precision on the exact-f32 MFMA GEMM (use --batch 256 --chunk 128 --no-extras)")
    ap.add_argument("--weights", default="survey", choices=["survey", "normalized"], help="survey: the SURVEY §8d draw (BatchNorm "
                    "statistics random: activations grow to ~1e8, bf16 only); normalized: the same draw with BatchNorm statistics "
                    "matching the activations like a trained checkpoint's (a-link_amd/weights.py) — float16 storage works there")
    ap.add_argument("--input", default="f32", choices=["f32", "u8"], help="pixel type resident in HBM")
    ap.add_argument("--streams", type=int, default=2, help="streams the chunks of one step are spread over (IRBackbone's default)")
    ap.add_argument("--linear", type=int, default=-1, help="A/B: widths on linear pixel tiles (bit0 56, bit1 28, bit2 14, bit3 7)")
    ap.add_argument("--shards", type=int, default=0, help="A/B: image shards one alink_embed call is split into on the "
                    "library's internal streams (alink_backbone_set_streams)")
    ap.add_argument("--fine-max", type=int, default=-1, help="A/B: largest 128-channel grid that still takes the 64-channel form")
    ap.add_argument("--stagger", type=int, default=-1, help="A/B: start delay (x 1024 cycles) of the second workgroup on a CU in the linear-tile kernel")
    ap.add_argument("--shard-stagger", type=int, default=-1, help="A/B: in-call shards start one after the other's front (stem + stage 1)")
    ap.add_argument("--no-fuse-stem", action="store_true", help="A/B: stem and stage1_unit1 conv1 as two launches instead of the fused front kernel")
    ap.add_argument("--no-sibling-aware", action="store_true", help="A/B: the 64- / 128-channel form of the linear-tile kernel chosen per shard, blind to the call's other shards")
    ap.add_argument("--no-s2direct", action="store_true", help="A/B: stage1_unit1's stride-2 conv2 + shortcut on the implicit-GEMM kernel instead of the direct stride-2 kernel")
    ap.add_argument("--no-c64", action="store_true", help="A/B: without the rolling-row kernel for the 64 -> 64 channel front layers")
    ap.add_argument("--no-fuse-sc", action="store_true", help="A/B: projection shortcuts as launches of their own instead of extra K-steps of conv2")
    ap.add_argument("--generic-epilogue", action="store_true", help="A/B: linear-tile kernel with the run-time-flag epilogue everywhere")
    ap.add_argument("--config3", action="store_true", help="(accepted for older scripts: the config-3 leg now always runs unless --no-config3)")
    ap.add_argument("--no-config3", action="store_true", help="skip the config-3 leg (3 x IR-50 committee over a pool shard: screening, "
                    "all-exact and screen-then-settle)")
    ap.add_argument("--config3-shard", type=int, default=12500, help="pool images per GPU in the config-3 leg (BASELINE configs[2]: 100k / 8)")
    ap.add_argument("--no-config4", action="store_true", help="skip the config-4 leg (one A-LINK iteration, IR-100 teacher: all-exact and screen-then-settle)")
    ap.add_argument("--no-config5", action="store_true", help="skip the config-5 leg (the A2-LINK few-pixel attack, K pairs' searches in lock-step)")
    ap.add_argument("--config5-pairs", type=int, default=16, help="pairs per GPU the config-5 leg attacks in its 16-bit search modes (half as many in the exact mode)")
    ap.add_argument("--no-configs1", action="store_true", help="skip the configs[1] leg (IR-50, one 256-image batch per step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-small", action="store_true", help="skip the N = 256 CPU forward (~30 s of host time): batches <= 128 only")
    ap.add_argument("--dry-ranks", action="store_true", help="pre-flight of a multi-GPU run: every rank creates its communicator, all-reduces a known "
                    "vector, runs merge_topk on a known candidate set, RowShards gathers and a calibration broadcast, prints one line per rank "
                    "and exits — seconds, before any timing (a first 8-GPU run should fail here with a reason, not hang a lease)")
    ap.add_argument("--strict", action="store_true", help="make the identity checks of the screen-then-settle legs fatal instead of reported")
    ap.add_argument("--no-extras", action="store_true", help="skip roofline profile / fine-tune timing")
    ap.add_argument("--select-dtype", default="f16x2", choices=["f16x2", "f32", "none"], help="the exact-selection leg: the same "
                    "workload in the mode whose active-learning selection sets equal the f32 arithmetic's (DESIGN.md §5), reported "
                    "as `exact_selection` beside the headline; also the committee's dtype in the config-3 leg")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_launch_ranks(args.gpus))

    import numpy as np
    import torch
    import a_link_amd  # noqa: F401
    from a_link_amd import weights as W
    from a_link_amd.backbone import IRBackbone

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "WORLD_SIZE=%d but --gpus %d" % (world, args.gpus)
    assert torch.cuda.is_available(), "bench.py needs a ROCm device"
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:          # launched by torch.distributed.run: rendezvous even when alone
        import torch.distributed as dist
        with _stdout_to_stderr():
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            dist.barrier()                                   # creates the communicator (and its banner) now
            torch.cuda.synchronize()

    if args.dry_ranks:
        sys.exit(_preflight(dist, rank, world, local_rank))
    if dist is not None and world > 1:
        _preflight(dist, rank, world, local_rank, quiet=True)       # the same checks, silently, before every multi-rank run (~1 s)

    if args.linear >= 0:
        from a_link_amd import _abi
        _abi.load().alink_debug_set_linear(args.linear)
    if args.stagger >= 0:
        from a_link_amd import _abi
        _abi.load().alink_debug_set_stagger(args.stagger)
    if args.generic_epilogue:
        from a_link_amd import _abi
        _abi.load().alink_debug_set_generic_epilogue(1)
    if args.no_fuse_sc:
        from a_link_amd import _abi
        _abi.load().alink_debug_set_fuse_shortcut(0)
    if args.no_fuse_stem:
        from a_link_amd import _abi
        _abi.load().alink_debug_set_fuse_stem(0)
    if args.no_sibling_aware:
        from a_link_amd import _abi
        _abi.load().alink_debug_set_sibling_aware(0)
    if args.no_s2direct:
        from a_link_amd import _abi
        _abi.load().alink_debug_set_s2direct(0)
    if args.no_c64:
        from a_link_amd import _abi
        _abi.load().alink_debug_set_c64(0)
    if args.shard_stagger >= 0:
        from a_link_amd import _abi
        _abi.load().alink_debug_set_shard_stagger(args.shard_stagger)
    if args.fine_max >= 0:
        from a_link_amd import _abi
        _abi.load().alink_debug_set_fine_max(args.fine_max)
    units = W.ARCH_UNITS[args.model]
    params = W.synthetic_ir_params(units, seed=1, normalized=args.weights == "normalized")
    # lazy_range_check: 16-bit float storage can leave its range; the flag the last kernel raises is read once per
    # timed region (bb.check_range()) instead of after every call
    bb = IRBackbone(params, image_size=(112, 112), dtype=args.dtype, device=local_rank, max_batch=args.chunk,
                    streams=args.streams, shards_per_call=args.shards or None, lazy_range_check=True)
    B = args.batch
    g = torch.Generator(device="cpu").manual_seed(rank)           # rank 0 == seed 0
    px = torch.randint(0, 256, (B, 112, 112, 3), generator=g, dtype=torch.uint8)
    x = (px if args.input == "u8" else px.to(torch.float32)).cuda()
    out = torch.empty((B, 512), dtype=torch.float32, device="cuda")

    def barrier():
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        bb.embed_device(x, out)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs_ = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for i_ in range(args.steps):
        if i_ == 0:
            evs_[0].record()
        if i_ == args.steps // 2:
            evs_[1].record()
        bb.embed_device(x, out)
    evs_[2].record()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sustained = None
    if args.steps >= 4:
        a_ms, b_ms = evs_[0].elapsed_time(evs_[1]), evs_[1].elapsed_time(evs_[2])
        h_ = args.steps // 2
        sustained = {"first_half_ms_per_step": a_ms / h_, "second_half_ms_per_step": b_ms / (args.steps - h_),
                     "second_over_first": (b_ms / (args.steps - h_)) / (a_ms / h_),
                     "timing": "HIP events on the launch stream at step 0, steps/2 and the end of the timed region"}
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    bb.check_range()
    assert torch.isfinite(out).all()
    # parity of what was just timed (cheap part, every rank): images embedded ALONE (batch 1: the 64-channel form of
    # the linear-tile kernel) must equal their rows of the timed batch (128-channel form, launches on several streams) bit for bit
    probe = sorted(set([0, 1, min(B - 1, args.chunk - 1), min(B - 1, args.chunk), B - 1]))
    alone = torch.cat([bb.embed_device(x[i:i + 1]) for i in probe])
    parity = {"batch1_rows_bit_equal_to_timed_batch": bool(torch.equal(alone, out[probe])), "rows_checked": probe}
    assert parity["batch1_rows_bit_equal_to_timed_batch"], "embedding depends on the batch it was computed in"

    from oracle import ir_resnet     # FLOP count + cpu_baseline leg only
    gflop_per_emb = ir_resnet.flops_per_image(units) / 1e9
    emb_per_s = world * args.steps * B / dt
    line = {
        "metric": "112x112 face embeddings/sec (IR-ResNet-%s ArcFace, 512-d, L2-normalised)" % args.model[1:],
        "value": emb_per_s, "unit": "embeddings/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        # what a reader of `value` alone must know: the two at-reference-precision rates are value_exact and
        # value_identical_selection below, filled in by their legs
        "dtype_note": ("bf16 < reference f32; within 1e-3 cosine, not selection-identical" if args.dtype == "bf16" else
                       "f16 < reference f32; within 1e-3 cosine, not selection-identical" if args.dtype == "f16" else
                       "the reference's float32 accuracy (selection-identical)"),
        "value_exact": None, "value_identical_selection": None,
        "config": {"workload": "LResNet%sE-IR embed, %d x 3x112x112 images per step per GPU (as %d-image launches "
                               "round-robin on %d streams), %s pixels resident in HBM"
                               % (args.model[1:], B, args.chunk, args.streams, args.input),
                   "arch": args.model, "weights": args.weights, "units": list(units), "batch_per_gpu": B, "global_batch": B * world,
                   "gflop_per_embedding": gflop_per_emb, "sharding": "images (dp%d), no collective" % world,
                   "images_per_launch": args.chunk, "intra_gpu_streams": args.streams},
        "tflops_end_to_end": emb_per_s * gflop_per_emb / 1e3,
        "frac_mfma_peak_end_to_end": emb_per_s * gflop_per_emb / 1e3 / (MFMA_PEAK_TFLOPS * world),
        "rccl_world_size": world if dist is not None else 0,
        "sustained_check": sustained,
        # how long the number above was measured for: with the driver's flags (--steps 20) the timed region is half a second — the
        # 60-second runs under profiles/ (r05d_sustained_*.json: 47.6 k bf16 / 16.3 k f16x2, halves within 0.3 %) are what say it holds
        "timed_region_s": dt,
        # what a model built through the reference's API computes in when the caller names no dtype (ArcFace / FaceModel)
        "default_api_dtype": __import__("a_link_amd.face_model", fromlist=["x"]).default_dtype(),
        # CITATIONS of the committed test results (tests/test_gpu_pool.py, golden config3_r50.npz), not measurements of this
        # run — this run's own count for its config-3 pool is config3.screening.selected_pairs_that_differ_from_exact_all
        "selection_identity_cited": {"bf16": "screening only: 394 of 1,024 golden config-3 selections differ from the f32 oracle's",
                                     "f16": "41 of 1,024 differ", "f32": "identical (exact-f32 MFMA, ~3.8 k IR-100 embeddings/s)",
                                     "f16x2": "identical (split precision on the f16 matrix cores)",
                                     "screen-then-settle": "identical on every workload measured (16-bit screening + split precision near the cuts, under "
                                                           "a MEASURED error bound with a sampled audit: config3 / config4 .audit; not a theorem)"},
        "headline_mode_note": "value is the %s SCREENING rate; the rate with selection sets identical to the float32 arithmetic is "
                              "exact_selection (every image exact) and config3.screen_settle (exact only near a cut)" % args.dtype
                              if args.dtype in ("bf16", "f16") else None,
    }

    def roofline_of(bbx, dt_name):
        """roofline of the dominant kernel of one forward of `bbx`: HIP events around every launch (alink_embed_profile)"""
        conv_ms, conv_fl, other_ms, profs = [], [], [], []
        for _ in range(3):
            prof = bbx.profile(x[:args.chunk])
            profs.append(prof)
            conv_ms.append(sum(ms for k, ms, f in prof if k == 1))
            conv_fl.append(sum(f for k, ms, f in prof if k == 1))
            other_ms.append(sum(ms for k, ms, f in prof if k != 1))
        n_conv = sum(1 for k, _, _ in prof if k == 1)
        cms = float(np.median(conv_ms))
        achieved_all = conv_fl[0] / (cms * 1e-3) / 1e12
        # the DOMINANT kernel: conv launches of one forward grouped by their algorithmic FLOPs (= same
        # shape = same kernel instantiation); the group with the most time.  For r100/r50 that is the
        # 14x14x256->256 stage-3 convolution (conv3x3_linear_kernel; conv3x3_direct_kernel with --linear 0).
        # launch order of the chain: per stage s, unit u: conv1, [shortcut], conv2 (csrc/backbone.hip)
        n_conv_launches = sum(1 for k, _, _ in profs[0] if k == 1)
        fused_sc = n_conv_launches == 2 * sum(units)        # projection shortcuts inside the conv2 launch (16-bit modes)
        shape_of = []
        for s_ in range(4):
            for u_ in range(units[s_]):
                shape_of.append("stage%d %s" % (s_ + 1, "unit1 conv1" if u_ == 0 else "3x3 s1 C->C"))
                if u_ == 0 and not fused_sc:
                    shape_of.append("stage%d unit1 shortcut" % (s_ + 1))
                shape_of.append("stage%d %s" % (s_ + 1, ("unit1 conv2 (stride 2%s)" % (" + 1x1 shortcut" if fused_sc else "")) if u_ == 0 else "3x3 s1 C->C"))
        groups = {}
        for run in profs:
            convs = [(ms, f) for k, ms, f in run if k == 1]
            assert len(convs) == len(shape_of)
            for name, (ms, f) in zip(shape_of, convs):
                g = groups.setdefault(name, [f, []])
                g[1].append(ms)
        dom_name, (dom_f, dom_ms) = max(groups.items(), key=lambda kv: sum(kv[1][1]))
        dom_avg = float(np.mean(dom_ms))
        dom_achieved = dom_f / (dom_avg * 1e-3) / 1e12
        # HBM traffic per launch: PMC counters are collected in separate rocprofv3 --pmc passes
        # (tools/pmc_summary.py -> profiles/*pmc_hbm_traffic*.csv); read the newest committed summary
        traffic = None
        try:
            import glob
            pm = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*pmc_hbm_traffic_%s_%s_b*.csv" % (args.model, dt_name))))
            if not pm and dt_name == "bf16":        # rounds 1-2 named the bf16 summaries without the dtype
                pm = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*pmc_hbm_traffic_%s_b*.csv" % args.model)))
            if pm:
                rows = [ln.split(",") for ln in open(pm[-1]).read().splitlines()[1:]]
                # the dominant instantiation is the 3x3 conv row with the most launches per forward
                rows = [r for r in rows if r[0] in ("conv3x3_linear_kernel", "conv3x3_direct_kernel") and r[1]]
                r = max(rows, key=lambda r: float(r[2]))
                import re
                nimg = int(re.search(r"_b(\d+)\.csv$", pm[-1]).group(1))       # images per launch of that profile
                traffic = {"bytes_per_launch": (float(r[3]) + float(r[4])) * 1e6, "fetch_MB": float(r[3]),
                           "write_MB": float(r[4]), "launches_per_forward": float(r[2]), "per_images": nimg,
                           "algorithmic_bytes_per_launch": (nimg * 196 * 256 * 2 * 2.5 + 256 * 2304 * 2) * (2 if dt_name == "f16x2" else 1),
                           "note": "rocprofv3 --pmc FETCH_SIZE (x2 gfx950 correction) and WRITE_SIZE, separate passes; "
                                   "algorithmic = input + output (+ residual on every second launch) + weights",
                           "source": os.path.basename(pm[-1]),
                           # PMC counters need rocprofv3 passes of their own (tools/profile_round.sh): this is the newest
                           # committed summary for this network and dtype, NOT a measurement of this run
                           "measured_in_this_run": False}
        except Exception:
            traffic = None
        # the same kernel by rocprofv3: the newest committed --kernel-trace summary for this network and dtype (a profiled pass
        # runs ~2-4 % slower than an un-profiled one: MI355X_MICROARCH.md, DVFS) — printed BESIDE the HIP-event figure, not merged
        rocprof = None
        try:
            import glob
            ks = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_%s_%s_streams1_kernel_stats.csv" % (args.model, dt_name))))
            if ks and "stage3" in dom_name:
                import csv
                rows = [r_ for r_ in csv.DictReader(open(ks[-1])) if "conv3x3_linear_kernel" in r_["Name"] and "Li14ELi4E" in r_["Name"]]
                if rows:
                    by_grid = {}
                    for r_ in rows:
                        by_grid[int(r_["GridX"])] = by_grid.get(int(r_["GridX"]), 0) + int(r_["Calls"])
                    gmax = max(by_grid, key=by_grid.get)                    # the launch shape with the most calls: the dominant layer's
                    rows = [r_ for r_ in rows if int(r_["GridX"]) == gmax]
                    calls = sum(int(r_["Calls"]) for r_ in rows)
                    avg_ns = sum(float(r_["TotalDurationNs"]) for r_ in rows) / calls
                    rocprof = {"avg_launch_us": avg_ns / 1e3, "calls": calls, "source": os.path.basename(ks[-1]),
                               "instantiations": sorted(set(r_["Name"][:96] for r_ in rows)),
                               "note": "mean over the kernel's epilogue forms, weighted by calls; the profile's launch batch must equal --chunk (%d) for "
                                       "frac_rocprof to price the same launch" % args.chunk}
                    rocprof["achieved"] = dom_f / (avg_ns * 1e-9) / 1e12
                    rocprof["frac"] = rocprof["achieved"] / MFMA_PEAK_TFLOPS
        except Exception as e_:
            rocprof = {"error": repr(e_)}
        return {"bound": "mfma",
                            "kernel": "%s, %s: %d launches per %d-image forward, %.1f GFLOP each"
                                      % ("conv3x3_linear_kernel (linear 16-pixel tiles, 4 waves, 2 workgroups/CU)"
                                         if (args.linear < 0 or args.linear & 4) else "conv3x3_direct_kernel", dom_name, len(dom_ms) // len(profs), args.chunk, dom_f / 1e9),
                            "achieved": dom_achieved, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": dom_achieved / MFMA_PEAK_TFLOPS, "traffic": traffic,
                            "frac_hip_events": dom_achieved / MFMA_PEAK_TFLOPS,
                            "frac_rocprof": rocprof.get("frac") if rocprof else None, "rocprof": rocprof,
                            "flops_per_launch": dom_f, "avg_launch_ms": dom_avg,
                            # split precision issues three MFMA FLOPs per algorithmic FLOP (hi*hi, hi*lo, lo*hi): what the matrix
                            # pipes actually do, beside the algorithmic fraction above
                            "mfma_flops_issued_per_algorithmic_flop": 3 if dt_name == "f16x2" else 1,
                            "frac_issued": dom_achieved * (3 if dt_name == "f16x2" else 1) / MFMA_PEAK_TFLOPS,
                            "timing": "HIP events on the launch stream around 4 back-to-back launches of every kernel of "
                                      "the chain, kernels run one at a time (alink_embed_profile); rocprofv3 --kernel-trace of "
                                      "`bench.py --streams 1 --batch %d` gives the same averages (profiles/)" % args.chunk,
                            "all_conv_launches": {"launches": n_conv, "achieved": achieved_all,
                                                  "frac": achieved_all / MFMA_PEAK_TFLOPS, "flops_per_forward": conv_fl[0],
                                                  "kernel_ms_per_forward": cms,
                                                  "non_conv_ms_per_forward": float(np.median(other_ms))}}

    def single_image_ms(bbx):
        """the reference's own call shape (FaceModel.get_feature, code/face_model.py:86-93): ONE image, host array in, host array out"""
        one = px[:1].numpy() if args.input == "u8" else px[:1].to(torch.float32).numpy()
        for _ in range(5):
            bbx.embed(one)
        ts = []
        for _ in range(30):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            bbx.embed(one)
            ts.append(time.perf_counter() - t1)
        return 1e3 * float(np.median(ts))

    if rank == 0 and not args.no_extras:
        line["single_image_ms"] = single_image_ms(bb)
        line["single_image_note"] = ("one image, host in / host out, %s; launches of a handful of images take the latency form of the 3x3 convolution "
                                     "(csrc/conv3x3_lat.hip), bit-identical to the batched kernels (parity.batch1_rows_bit_equal_to_timed_batch)" % args.dtype)

    sel_out = None
    smallres_weights = None
    if args.select_dtype != "none" and args.select_dtype != args.dtype:
        # ---- the exact-selection leg (north_star: ">= 10 k embeddings/s ... with selection sets identical to the
        # reference"): the SAME images through the mode whose selection sets equal the f32 arithmetic's — split precision
        # (f16 pairs, three products on the f16 matrix cores, power-of-two scales calibrated per tensor) unless --select-dtype f32
        sel_chunk = args.chunk if args.select_dtype == "f16x2" else 128
        bs = IRBackbone(params, image_size=(112, 112), dtype=args.select_dtype, device=local_rank, max_batch=sel_chunk,
                        streams=args.streams, lazy_range_check=True)
        if args.select_dtype == "f16x2":
            bs.calibrate(x[:64])
        sel_out = torch.empty((B, 512), dtype=torch.float32, device="cuda")
        sel_steps = max(2, args.steps // (4 if args.select_dtype == "f16x2" else 25))
        bs.embed_device(x, sel_out)
        torch.cuda.synchronize()
        barrier()
        t1 = time.perf_counter()
        for _ in range(sel_steps):
            bs.embed_device(x, sel_out)
        torch.cuda.synchronize()
        barrier()
        tsel = time.perf_counter() - t1
        if dist is not None:
            t = torch.tensor([tsel], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            tsel = float(t.item())
        bs.check_range()
        sel_rate = world * sel_steps * B / tsel
        peak = MFMA_PEAK_TFLOPS if args.select_dtype == "f16x2" else F32_MFMA_PEAK_TFLOPS
        line["exact_selection"] = {
            "dtype": args.select_dtype, "embeddings_per_s": sel_rate, "steps": sel_steps, "ms_per_step": 1e3 * tsel / sel_steps,
            "tflops_algorithmic": sel_rate * gflop_per_emb / 1e3,
            "frac_of_peak": sel_rate * gflop_per_emb / 1e3 / (peak * world), "peak": peak,
            "mfma_flops_issued_per_algorithmic_flop": 3 if args.select_dtype == "f16x2" else 1,
            "max_abs_diff_vs_headline_dtype": float((sel_out - out).abs().max()),
            "note": "selection sets identical to the f32 oracle's: tests/test_gpu_pool.py (config 3: 0 of 1,024 differ; config 4: "
                    "both columns equal at IR-50 and IR-100 depth)"}
        line["value_exact"] = {"embeddings_per_s": sel_rate, "dtype": args.select_dtype, "same_workload_as_value": True,
                               "note": "every image in the mode whose embeddings equal the f32 oracle's to float32 accuracy (exact_selection)"}
        if rank == 0 and not args.no_extras and args.select_dtype == "f16x2":
            # the exact mode's own dominant kernel: algorithmic fraction, issued fraction (3 MFMA FLOPs per algorithmic FLOP), traffic
            line["exact_selection"]["roofline"] = roofline_of(bs, "f16x2")
        if rank == 0 and not args.no_extras:
            line["exact_selection"]["single_image_ms"] = single_image_ms(bs)
        del bs
    del params

    if dist is not None and not args.no_extras:
        # ---- fine-tune step in a one-process-per-GPU job, both forms (a-link_amd/distributed.py):
        # sharded = the 16-pair batch split over the ranks, ONE RCCL all-reduce of the flat gradient buffer (+ metrics),
        # identical Adadelta update on every rank; replicated = every rank runs the whole batch, no collective
        # (what dp_train_on_batch picks by itself below DP_SHARD_MIN_ROWS rows)
        from a_link_amd import distributed as D
        from a_link_amd.head import DenseHead
        rng = np.random.RandomState(0)
        Ld = torch.from_numpy(rng.randn(16, 512).astype(np.float32)).cuda()
        Rd = torch.from_numpy(rng.randn(16, 512).astype(np.float32)).cuda()
        yh = np.zeros((16, 2), np.float32)
        yh[np.arange(16), rng.randint(0, 2, 16)] = 1
        ydd = torch.from_numpy(yh).cuda()
        for mode, key in (("sharded", "finetune_step_dp_ms"), ("replicated", "finetune_step_replicated_ms")):
            hdp = DenseHead(512, lr=0.1, seed=0, device=local_rank)
            # RCCL's first few hundred all-reduces on a fresh communicator run 3-5x slower than its steady state
            # (tools/experiments/dp_step_time.py: 0.27-0.42 ms for the first 220 steps, 0.085 ms after): warm up past that
            for _ in range(400):
                D.dp_train_on_batch(hdp, [Ld, Rd], ydd, mode=mode)
            torch.cuda.synchronize()
            barrier()
            t1 = time.perf_counter()
            for _ in range(100):
                D.dp_train_on_batch(hdp, [Ld, Rd], ydd, mode=mode)
            torch.cuda.synchronize()
            barrier()
            tdp = torch.tensor([(time.perf_counter() - t1) / 100], dtype=torch.float64, device="cuda")
            dist.all_reduce(tdp, op=dist.ReduceOp.MAX)
            line[key] = 1e3 * float(tdp.item())
            del hdp
        line["finetune_step_dp_policy"] = "replicated below %d rows (a-link_amd/distributed.py)" % D.DP_SHARD_MIN_ROWS

    if not args.no_config3 and not args.no_extras:
        # ---- BASELINE configs[2] shape, this rank's share: a committee of THREE IR-50 backbones + heads over a pool shard
        # against a 16-image gallery, entropy, top-1024, one candidate exchange — three ways:
        #   screening      everything in the 16-bit mode (f16 where the activations fit, else bf16): the fast number, whose
        #                  selection is NOT the float32 arithmetic's (the count that differs is measured below);
        #   exact_all      everything in split precision: the float32 arithmetic's selection;
        #   screen_settle  screening + exact re-embedding of only the images that own a pair near the cut (settle.py):
        #                  the SAME scores, order and indices as exact_all (asserted bit for bit here).
        # Weights: the SURVEY draw with BatchNorm statistics matching the activations, like a trained checkpoint's (the raw
        # draw's activations leave float16); pool: synthetic identities; heads: glorot, last layer rescaled so that
        # probabilities spread over (0, 1).  The same calibration sample (seed 999) on every rank: scales and head
        # rescaling are rank-consistent.
        from a_link_amd import distributed as D
        from a_link_amd import _abi as _A
        from a_link_amd.head import DenseHead
        n_shard = args.config3_shard
        exact_dt = args.select_dtype if args.select_dtype != "none" else "f16x2"
        cal, cal_gal = _identity_pool(512, 999)
        shard, _g = _identity_pool(n_shard, 1000 + rank)
        gal = cal_gal                                                           # the gallery is replicated: one draw for all ranks
        exa, scr = [], []
        for s_ in (1, 2, 3):
            pr = W.synthetic_ir_params(W.R50_UNITS, seed=s_, normalized=True)
            e_ = IRBackbone(pr, dtype=exact_dt, device=local_rank, max_batch=args.chunk if exact_dt != "f32" else 128, streams=args.streams)
            if exact_dt == "f16x2":
                e_.calibrate(cal[:args.chunk])
            exa.append(e_)
            s16 = IRBackbone(pr, dtype="auto", device=local_rank, max_batch=args.chunk, streams=args.streams)
            if s16.dtype == "f16":
                try:                                                          # the probe images passed; do these?
                    s16.embed_device(cal)
                except _A.AlinkError:
                    s16 = None
            if (s16 is None or s16.dtype != "f16") and exact_dt == "f16x2":
                s16 = e_.screening_view()        # the one-product form of the exact handle: nothing to overflow, 8x finer than bf16
            elif s16 is None:
                s16 = IRBackbone(pr, dtype="bf16", device=local_rank, max_batch=args.chunk, streams=args.streams)
            scr.append(s16)
            del pr
        ncal = cal.shape[0]
        lic = torch.arange(ncal, dtype=torch.int32, device="cuda").repeat_interleave(16)
        ric = torch.arange(16, dtype=torch.int32, device="cuda").repeat(ncal)
        Ecal = [e_.embed_device(cal) for e_ in exa]
        Egal = [e_.embed_device(cal_gal) for e_ in exa]
        lo_ = rank * n_shard
        k3 = 1024
        pool_n = world * n_shard

        def three_rates(hds, full):
            D.committee_pool_topk(exa, hds, shard[:args.chunk], gal, 16, lo_)          # warm-up of every handle
            D.committee_pool_topk(scr, hds, shard[:args.chunk], gal, 16, lo_)
            t_x, (xv, xi) = _timed(lambda: D.committee_pool_topk(exa, hds, shard, gal, k3, lo_), 2 if full else 1, barrier, dist)
            t_s, (sv, si) = _timed(lambda: D.committee_pool_topk(scr, hds, shard, gal, k3, lo_), 2 if full else 1, barrier, dist)
            inf3 = {}
            t_ss, (ssv, ssi) = _timed(lambda: D.committee_pool_topk_settled(scr, exa, hds, shard, gal, k3, lo_, info=inf3),
                                      2 if full else 1, barrier, dist)
            # checked in the run and REPORTED (a line that says "false" is worth more than no line); --strict makes them fatal
            identical = bool(torch.equal(ssi, xi) and torch.equal(ssv, xv))
            assert identical or not args.strict, "screen-then-settle returned a different selection than the all-exact pass"
            xs = xv.cpu().numpy()
            out_ = {
                "screening": {"pool_images_per_s": pool_n / t_s, "ms_per_pass": 1e3 * t_s,
                              "selected_pairs_that_differ_from_exact_all": len(set(si.cpu().numpy().tolist()) - set(xi.cpu().numpy().tolist())),
                              "measured_in_this_run": True},
                "exact_all": {"pool_images_per_s": pool_n / t_x, "ms_per_pass": 1e3 * t_x},
                "screen_settle": {"pool_images_per_s": pool_n / t_ss, "ms_per_pass": 1e3 * t_ss,
                                  "fraction_re_embedded": inf3["fraction_re_embedded"], "rounds": inf3["rounds"],
                                  "delta": inf3["delta"], "largest_dp_seen": inf3["d_max"], "widened": inf3["widened"],
                                  "audit": inf3.get("audit"), "recalibrated": inf3.get("recalibrated"),
                                  "identical_to_exact_all": identical,
                                  "identical_means": "scores, order and indices of the top-%d equal the all-exact pass's bit for bit (compared in this run)" % k3,
                                  "speedup_over_exact_all": t_x / t_ss},
                "entropy_of_the_selected": {"largest": float(xs[0]), "smallest": float(xs[-1])},
                "backbone_forwards_per_s_exact_all": 3 * world * (n_shard + 16) / t_x}
            if full:
                inf3b = {}
                t_set, (_v, sei) = _timed(lambda: D.committee_pool_topk_settled(scr, exa, hds, shard, gal, k3, lo_, settle_selected=False,
                                                                                 info=inf3b), 1, barrier, dist)
                same_set = set(sei.cpu().numpy().tolist()) == set(xi.cpu().numpy().tolist())
                assert same_set or not args.strict
                out_["screen_settle_set_only"] = {"pool_images_per_s": pool_n / t_set, "ms_per_pass": 1e3 * t_set,
                                                  "fraction_re_embedded": inf3b["fraction_re_embedded"], "same_set_as_exact_all": same_set,
                                                  "note": "members certain by interval keep their screened score: same SET (compared in this run), no exact scores for them"}
            return out_

        # (a) TRAINED heads, like the reference's committee (its ensemble models are pre-trained pair scorers: code/ALINK_arc.py:96-137):
        # each member's head fine-tuned by the product's own fit() on its exact embeddings of the calibration sample — 16 enrolled
        # persons x 32 images against the 16 gallery images, every same-person pair + three times as many others, 6 epochs
        yc = (lic.cpu().numpy() // 32 == ric.cpu().numpy())
        rs = np.random.RandomState(0)
        pick = np.concatenate([np.flatnonzero(yc), rs.choice(np.flatnonzero(~yc), 3 * int(yc.sum()), replace=False)])
        rs.shuffle(pick)
        yoh = np.stack([~yc[pick], yc[pick]], 1).astype(np.float32)
        trained = []
        for m_ in range(3):
            h_ = DenseHead(512, lr=1.0, seed=10 + m_, device=local_rank)
            np.random.seed(100 + m_)
            h_.fit([Ecal[m_].cpu().numpy()[lic.cpu().numpy()[pick]], Egal[m_].cpu().numpy()[ric.cpu().numpy()[pick]]], yoh,
                   batch_size=64, epochs=6, verbose=0)
            trained.append(h_)
        c3 = three_rates(trained, True)
        # (b) the worst case for screen-then-settle: untrained (glorot) heads whose last layer is rescaled so that probabilities spread
        # over (0, 1) — the DENSEST part of their distribution sits at 1/2, exactly where the cut of a most-uncertain top-k is
        spread = [DenseHead(512, lr=0.1, seed=10 + i, device=local_rank) for i in range(3)]
        for h_, el, eg in zip(spread, Ecal, Egal):
            _spread_head(h_, el, eg, lic, ric)
        c3b = three_rates(spread, False)
        line["config3"] = dict(c3, **{
            "workload": "committee of 3 IR-50 (BatchNorm statistics matching the activations) + 3 pair heads TRAINED on 16 enrolled persons (the "
                        "product's own fit(), 6 epochs), %d synthetic-identity pool images per GPU (32 per person, the first 16 persons enrolled) x 16 "
                        "gallery images = %d pairs per GPU, entropy, top-%d, candidate all-gather + device merge" % (n_shard, n_shard * 16, k3),
            "exact_dtype": exact_dt, "screening_dtype": scr[0].dtype, "selected": k3,
            "untrained_heads_worst_case": dict(c3b, note="glorot heads, last layer rescaled so that probabilities spread over (0,1): the densest part of "
                                                          "their distribution is AT the cut of a most-uncertain top-k; 16 pairs per image then put a third of "
                                                          "the images inside the band")})
        hds = trained
        del exa, scr, hds, shard, cal

    if not args.no_configs1 and not args.no_extras:
        # ---- BASELINE configs[1] AS WORDED: "ResNet-50 ArcFace 512-d feature extraction, 112x112 batch=256, 1 MI355X" — IR-50,
        # ONE 256-image batch per step (one alink_embed call, split into two in-call shards by the library), in the screening
        # dtype and in the exact mode
        c1 = {}
        p50 = W.synthetic_ir_params(W.R50_UNITS, seed=1, normalized=args.weights == "normalized")
        g50 = ir_resnet.flops_per_image(W.R50_UNITS) / 1e9
        x256 = x[:256] if B >= 256 else x
        for dt_ in ([args.dtype] if args.dtype != "f32" else []) + ([args.select_dtype] if args.select_dtype == "f16x2" and args.dtype != "f16x2" else []):
            b50 = IRBackbone(p50, image_size=(112, 112), dtype=dt_, device=local_rank, max_batch=256, streams=1, lazy_range_check=True)
            if dt_ == "f16x2":
                b50.calibrate(x256[:64])
            o50 = torch.empty((x256.shape[0], 512), dtype=torch.float32, device="cuda")
            n50 = 60 if dt_ != "f16x2" else 20
            t50, _ = _timed(lambda: [b50.embed_device(x256, o50) for _ in range(n50)], 1, barrier, dist)
            b50.check_range()
            r50 = world * n50 * x256.shape[0] / t50
            c1[dt_] = {"embeddings_per_s": r50, "ms_per_step": 1e3 * t50 / n50, "steps": n50, "tflops_algorithmic": r50 * g50 / 1e3,
                       "frac_mfma_peak": r50 * g50 / 1e3 / (MFMA_PEAK_TFLOPS * world)}
            del b50
        c1["workload"] = "LResNet50E-IR, ONE %d-image batch per step per GPU, f32 pixels resident in HBM, %.3f GFLOP per embedding" % (x256.shape[0], g50)
        line["configs1"] = c1
        del p50

    if rank == 0 and world == 1 and not args.no_extras:
        # ---- the boundary as the reference's callers use it: HOST arrays in, host arrays out (ArcFace.process on a NumPy pool,
        # reference code/siamese.py:232-234) — PCIe-inclusive, never `value`: uint8 pixels, uploads overlapped with compute
        hp = px[:B].numpy()
        bb.embed(hp)
        t1 = time.perf_counter()
        for _ in range(3):
            eh = bb.embed(hp)
        line["pool_from_host_u8_embeddings_per_s"] = 3 * B / (time.perf_counter() - t1)
        line["pool_from_host_note"] = "IRBackbone.embed on a (%d,112,112,3) uint8 host array -> (N,512) host array, %s; PCIe-inclusive, not `value`" % (B, args.dtype)
        assert np.isfinite(eh).all()

    if not args.no_config4 and not args.no_extras:
        # ---- BASELINE configs[3] shape: ONE A-LINK iteration (reference code/ALINK_arc.py:142-254) with an IR-100
        # teacher — 16 persons x (2 plain + 3 disguised) = 80 unique images, P = 3,840 pairs, four noises drawn per pair
        # occurrence = 30,720 noisy embeddings (the bulk of an iteration), selection, fine-tune —
        # twice from identical seeds: every embedding exact, and screen-then-settle (noisy copies in the 16-bit mode, only
        # the pairs near a cut of the rule and the selected ones again in the exact mode).  Must agree on the oracle-query
        # count, the number of fine-tunes and the student's weights afterwards, bit for bit (compared here).
        # With N ranks (a-link_amd/alink_loop.py, `group`): ONE iteration of the same size shared by all ranks — the pair
        # rows split contiguously, every rank perturbs / embeds / scores its rows, one all-gather of the predictions,
        # selection replicated, settle requests and fine-tune rows served by their owners (STRONG scaling of this leg: the
        # job is fixed); every rank must end with the same student weights (hash compared across ranks here).
        from a_link_amd import alink_loop as AL, committee, noise as NZ, pairs as PR, siamese
        import hashlib
        import tempfile
        names = ("gaussian", "saltpepper", "poisson", "speckle")
        ppl, _ = _identity_pool(16 * 5, 4242, per_person=5)
        ppl = ppl.float().cpu().numpy().reshape(16, 5, 112, 112, 3)
        X_plain, X_dig = [q[:2] for q in ppl], [q[2:] for q in ppl]
        res4 = {}
        tmpd = tempfile.mkdtemp()
        grp4 = dist.group.WORLD if dist is not None else None        # under a launcher the multi-rank form runs even with one rank
        for mode in ("exact_all", "screen_settle"):
            conv = siamese.ArcFace((112, 112), "synthetic:r100:1:normalized", dtype="f16x2",
                                   screen_dtype="auto" if mode == "screen_settle" else None)
            conv.model.model.calibrate(torch.from_numpy(ppl.reshape(-1, 112, 112, 3)).cuda())     # the same images on every rank
            student = siamese.SiameseNetwork((512,), os.path.join(tmpd, "student"), 0.1, seed=1)
            ens = [siamese.SiameseNetwork((512,), "e%d" % i, 0.1, seed=2 + i) for i in range(2)]
            nzs = [NZ.get_relevant_noise(n_)(model=student, sess=None, feature_model=conv) for n_ in names]
            feats = [conv.process(q) for q in X_plain]
            allf = torch.from_numpy(np.concatenate(feats)).cuda()
            li4 = torch.arange(len(allf), dtype=torch.int32, device="cuda").repeat_interleave(len(allf))
            ri4 = torch.arange(len(allf), dtype=torch.int32, device="cuda").repeat(len(allf))
            for h_ in [student] + ens:
                _spread_head(h_.siamese_net, allf, allf, li4, ri4)
            bag = committee.Bagging(ens, nzs)
            flags = AL.Flags(out_model="", screen_settle=mode == "screen_settle")
            ts4 = []
            for rep in range(2):                       # rep 0 warms every handle and workspace; both reps start from the same state
                student.siamese_net.set_weights(w0) if rep else None
                w0 = student.siamese_net.get_weights()
                for i_, z in enumerate(nzs):
                    z._seed, z._calls = 1000 + i_, 0
                np.random.seed(0)
                gen = PR.getGenerator(PR.getNormalGenerator(feats, 16), PR.getNormalGenerator(feats, 16),
                                      PR.getImposterGenerator(feats, feats, 16), 16)
                torch.cuda.synchronize()
                barrier()
                t1 = time.perf_counter()
                st4 = AL.run_alink_dfw(flags, conv, bag, nzs, student, X_plain, X_dig, gen, (112, 112), col=0, verbose=0, group=grp4)
                torch.cuda.synchronize()
                barrier()
                t4 = time.perf_counter() - t1
                if dist is not None:
                    tt = torch.tensor([t4], dtype=torch.float64, device="cuda")
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    t4 = float(tt.item())
                ts4.append(t4)
            res4[mode] = (ts4[-1], st4, student.siamese_net.get_weights(), conv.screen.model.dtype if conv.screen else None)
            del conv, student, ens, bag, nzs
        (t_e, st_e, w_e, _), (t_q, st_q, w_q, sdt) = res4["exact_all"], res4["screen_settle"]
        same4 = (st_e.active_count == st_q.active_count and st_e.finetunes == st_q.finetunes and st_e.un_size == st_q.un_size
                 and all(np.array_equal(a_, b_) for a_, b_ in zip(w_e, w_q)))
        assert same4 or not args.strict, "screen-then-settle A-LINK iteration differs from the all-exact one"
        # every rank's student after the iteration: one hash per rank and mode, gathered
        digest = [hashlib.sha256(b"".join(np.ascontiguousarray(w_).tobytes() for w_ in ws_)).hexdigest()[:16] for ws_ in (w_e, w_q)]
        digests = [digest]
        if dist is not None:
            digests = [None] * world
            dist.all_gather_object(digests, digest)
        ranks_agree = all(d_ == digests[0] for d_ in digests)
        assert ranks_agree or not args.strict, "ranks ended the A-LINK iteration with different student weights: %s" % digests
        P4 = st_e.un_size
        n_emb = 80 + 2 * P4 * len(names)
        g100 = ir_resnet.flops_per_image(W.ARCH_UNITS["r100"]) / 1e9
        inf4 = st_q.settle_info[-1]
        rows4 = [inf4.get("rows_of_this_rank", P4)]
        if dist is not None:
            rows4 = [None] * world
            dist.all_gather_object(rows4, inf4.get("rows_of_this_rank", P4))
        if rank == 0:
            line["config4"] = {
                "workload": "one A-LINK iteration, IR-100 teacher (BatchNorm statistics matching the activations), 16 persons, 80 unique "
                            "images, %d pairs, noises %s per pair occurrence = %d embeddings, heads' last layers rescaled (probabilities "
                            "spread), selection + fine-tune" % (P4, "/".join(names), n_emb),
                "n_gpus": world, "scaling_of_this_leg": "strong (ONE iteration of fixed size shared by all ranks)" if world > 1 else "n/a (one rank)",
                "pair_rows_per_rank": rows4,
                "sharding": ("pair rows split contiguously over %d ranks; clean pass (80 images), committee predictions, selection and "
                             "the batch-16 fine-tune replicated; one all-gather of the (P, n_noise, 2) predictions, settle replies and "
                             "fine-tune rows from their owners" % world) if world > 1 else "one process",
                "ranks_end_with_identical_student_weights": bool(ranks_agree),
                "exact_all": {"s_per_iteration": t_e, "embeddings_per_s": n_emb / t_e},
                "screen_settle": {"s_per_iteration": t_q, "embeddings_per_s": n_emb / t_q, "screening_dtype": sdt,
                                  "fraction_pair_noise_rows_settled": inf4["fraction_settled"], "rounds": inf4["rounds"], "delta": inf4["delta"],
                                  "audit": inf4.get("audit"),
                                  "tflops_algorithmic": n_emb / t_q * g100 / 1e3, "frac_mfma_peak": n_emb / t_q * g100 / 1e3 / (MFMA_PEAK_TFLOPS * world),
                                  "speedup_over_exact_all": t_e / t_q,
                                  "identical_to_exact_all": bool(same4),
                                  "identical_means": "oracle-query count, number of fine-tunes and the student's weights afterwards equal the all-exact iteration's bit for bit (compared in this run)"},
                "oracle_queries": st_e.active_count, "finetunes": st_e.finetunes,
                # which form the fine-tune's steps took: distributed.dp_train_on_batch in mode "auto"
                "finetune_steps": {"rows_per_step": 16, "mode": ("replicated on every rank, no collective (16 rows < DP_SHARD_MIN_ROWS = %d: a 0.04 ms "
                                                                 "step against a 1.18 MB exchange)" % __import__("a_link_amd.distributed", fromlist=["x"]).DP_SHARD_MIN_ROWS)
                                   if world > 1 else "one process", "gradient_all_reduce_exercised_by": "finetune_step_dp_ms (this line, N > 1) and tests/test_gpu_distributed.py"},
                "note": "whole iteration on the wall clock (max over ranks): noise kernels, embeddings, heads, host-side selection, collectives, fine-tune"}
            line["value_identical_selection"] = {"embeddings_per_s": n_emb / t_q, "workload": "config4.screen_settle: one A-LINK iteration, results bit-equal "
                                                 "to the all-exact iteration's in this run" if same4 else "config4.screen_settle (NOT identical in this run)",
                                                 "all_exact_embeddings_per_s": n_emb / t_e}

    if not args.no_config5 and not args.no_extras:
        # ---- BASELINE configs[4] ("A2-LINK: adversarial-noise batch"): the reference's adversarial noise is the black-box few-pixel
        # attack (code/noise.py:171-188 -> code/attack.py:91-103: 40 pixels, population 200, 50 generations = 20,400 backbone forwards per
        # pair), `adversarial` is in its default --noise list (code/ALINK_arc.py:41).  attack_all advances the pairs' searches in
        # LOCK-STEP (a-link_amd/attack.py: 400 x K images per launch, the success test read off the generation's own scores, host
        # bookkeeping hidden behind the other lane).  Timed: every search run to maxiter (early_stop=False — the cost of an attack that
        # does not succeed, the worst case) in the three search arithmetics; beside each the SAME handle's plain in-batch embedding rate
        # in this run, whose fraction is what the lock-step engine leaves on the table.  Each rank attacks its own pairs (no collective).
        from a_link_amd import attack as ATK, noise as NZ5, siamese as S5
        conv5 = S5.ArcFace((112, 112), "synthetic:r100:1:normalized", dtype="f16x2", screen_dtype="auto")
        conv5.model.model.calibrate(x[:64])
        stu5 = S5.SiameseNetwork((512,), "/tmp/alink_student5", 0.1, seed=1)
        wrapped5 = NZ5.PredictionWrappedModel(stu5, conv5)
        rs5 = np.random.RandomState(77 + rank)
        n5 = max(2, args.config5_pairs)
        imgs5 = [rs5.randint(0, 256, (224, 112, 3)).astype(np.float32) for _ in range(n5)]
        tg5 = [[0, 1]] * n5
        sd5 = [1000 * rank + i for i in range(n5)]
        c5 = {}
        for mode5, npairs in (("screen", n5), ("bf16", n5), ("exact", max(2, n5 // 2))):
            att5 = ATK.PixelAttacker(wrapped5, search=mode5)
            bb5 = ATK._device_parts(wrapped5, mode5)[0]
            att5.attack_all(imgs5[:2], tg5[:2], (224, 112), seeds=sd5[:2], maxiter=2, early_stop=False)          # warm-up
            t5, got5 = _timed_once(lambda: att5.attack_all(imgs5[:npairs], tg5[:npairs], (224, 112), seeds=sd5[:npairs], early_stop=False), barrier, dist)
            if mode5 == "screen":
                first5 = got5[0]
            ev5 = sum(int(r_.nfev) for r_ in att5.last_results)
            o5 = torch.empty((B, 512), dtype=torch.float32, device="cuda")
            bb5.embed_device(x, out=o5)
            tb5, _ = _timed_once(lambda: [bb5.embed_device(x, out=o5) for _ in range(3)], barrier, dist)
            inb = world * 3 * B / tb5
            fps = world * 2 * ev5 / t5
            c5[mode5] = {"search_dtype": bb5.dtype, "pairs_per_gpu": npairs, "lockstep": att5.lockstep, "s_per_pair": t5 / npairs,
                         "generations_per_pair": float(np.mean([r_.nit for r_ in att5.last_results])),
                         "backbone_forwards_per_pair": 2 * ev5 / npairs, "backbone_forwards_per_s": fps,
                         "same_handle_in_batch_embeddings_per_s": inb, "frac_of_in_batch_rate": fps / inb,
                         "frac_mfma_peak_algorithmic": fps * gflop_per_emb / 1e3 / (MFMA_PEAK_TFLOPS * world)}
            del o5
        # one pair after another (the reference's shape, rounds 1-5 of this package): one pair with a success test that never stops
        # the search (the same 50 generations as above: its image must equal the lock-step run's for that pair), then two pairs
        # with the real test on (searches stop early) against the lock-step form
        seq5 = ATK.PixelAttacker(wrapped5, search="screen", lockstep=0)
        seq5.attack_success = lambda *a_, **k_: None
        seq5.attack_all(imgs5[:1], tg5[:1], (224, 112), seeds=sd5[:1], maxiter=2)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        w5 = seq5.attack_all(imgs5[:1], tg5[:1], (224, 112), seeds=sd5[:1])
        tseq = time.perf_counter() - t1
        same5 = bool(np.array_equal(w5[0], first5))
        w5 = ATK.PixelAttacker(wrapped5, search="screen", lockstep=0).attack_all(imgs5[:2], tg5[:2], (224, 112), seeds=sd5[:2])
        g5 = ATK.PixelAttacker(wrapped5, search="screen").attack_all(imgs5[:2], tg5[:2], (224, 112), seeds=sd5[:2])
        same5 = same5 and bool(np.array_equal(np.stack(w5), np.stack(g5)))
        assert same5 or not args.strict, "lock-step few-pixel attack differs from the sequential one"
        if rank == 0:
            line["config5"] = dict(c5, **{
                "workload": "A2-LINK adversarial noise = the few-pixel differential-evolution attack at the reference's defaults (40 pixels, "
                            "population 200, 50 generations + the initial population = 20,400 IR-100 forwards per 112x112 pair), every search run "
                            "to maxiter; pairs' searches advanced in lock-step, 2 lanes x 16 searches = 6,400 images per launch chain",
                "n_gpus": world,
                "one_pair_after_another_screen": {"pairs": 1, "s_per_pair": tseq, "backbone_forwards_per_s": (20400 + 100) / tseq,
                                                  "note": "lockstep=0: a 400-image launch, a synchronisation and a 2-image success-test forward per generation"},
                "lockstep_images_identical_to_one_after_another": same5,
                "identical_means": "attacked image of pair 0 after 50 generations, and of two pairs whose searches stop on success, equal the one-after-another form's bit for bit (compared in this run)",
                "per_iteration_note": "a config-4-sized iteration has 3,840 pairs: x s_per_pair / n_gpus"})
        del conv5, stu5, wrapped5

    if rank == 0 and not args.no_extras and args.dtype == "f32":
        # ---- float32 mode: every convolution and the FC on gemm32_kernel (v_mfma_f32_32x32x2_f32); no per-launch
        # profile entry for this mode, so the roofline is the whole forward of one launch batch between two HIP events
        # on the stream the kernels are launched on (torch's current stream), against the f32 matrix-core peak
        evs = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            bb.embed_device(x[:args.chunk])
            e1.record()
            torch.cuda.synchronize()
            evs.append(e0.elapsed_time(e1))
        fms = float(np.median(evs))
        ach = args.chunk * gflop_per_emb / fms                      # GFLOP / ms = TFLOP/s
        line["roofline"] = {"bound": "mfma", "kernel": "gemm32_kernel: all launches of one %d-image forward (implicit-im2col exact-f32 GEMM "
                                                        "+ its elementwise passes)" % args.chunk,
                            "achieved": ach, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / F32_MFMA_PEAK_TFLOPS,
                            "traffic": None, "flops_per_launch": args.chunk * gflop_per_emb * 1e9, "avg_launch_ms": fms,
                            "timing": "HIP events on the launch stream around one forward, median of 3"}
    if rank == 0 and not args.no_extras and args.dtype != "f32":
        line["roofline"] = roofline_of(bb, args.dtype)
        # ---- fine-tune step (second half of BASELINE.json's metric): head-512, batch 16
        from a_link_amd.head import DenseHead
        hd = DenseHead(512, lr=0.1, seed=0, device=local_rank)
        rng = np.random.RandomState(0)
        L = torch.from_numpy(rng.randn(16, 512).astype(np.float32)).cuda()
        R = torch.from_numpy(rng.randn(16, 512).astype(np.float32)).cuda()
        y = np.zeros((16, 2), np.float32)
        y[np.arange(16), rng.randint(0, 2, 16)] = 1
        yd = torch.from_numpy(y).cuda()
        for _ in range(20):
            hd.train_on_batch([L, R], yd)
        ts = []
        for _ in range(200):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            hd.train_on_batch([L, R], yd)      # fwd + BCE + bwd + Adadelta + metrics read-back
            ts.append(time.perf_counter() - t1)
        line["finetune_step_ms"] = 1e3 * float(np.median(ts))
        # the same step in the head's bf16 compute mode (configs[4]: f32 masters + Adadelta, bf16 GEMM operands)
        hq = DenseHead(512, lr=0.1, seed=0, device=local_rank, compute_dtype="bf16")
        for _ in range(20):
            hq.train_on_batch([L, R], yd)
        ts = []
        for _ in range(200):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            hq.train_on_batch([L, R], yd)
            ts.append(time.perf_counter() - t1)
        line["finetune_step_bf16_ms"] = 1e3 * float(np.median(ts))
        # the drivers' PRE-TRAINING loop (SiameseNetwork.customTrainModel, code/siamese.py:81-112; 20,000 steps an epoch x 100 epochs at
        # the reference's settings): train_on_batch + test_on_batch per step over the balanced generator, 200 persons, batch 16 —
        # the generator's feature table on the device, steps shipped as row indices in blocks (round 6)
        from a_link_amd import pairs as PRc, siamese as SIc
        rsc = np.random.RandomState(0)
        fc_ = [rsc.randn(rsc.randint(3, 6), 512).astype(np.float32) for _ in range(200)]
        genc = PRc.getGenerator(PRc.getNormalGenerator(fc_, 16), PRc.getNormalGenerator(fc_, 16), PRc.getImposterGenerator(fc_, fc_, 16), 16)
        netc = SIc.SiameseNetwork((512,), "/tmp/alink_ctm", 0.1, seed=1)
        np.random.seed(0)
        netc.customTrainModel(genc, 1, 16, 0.2, n_steps=16 * 300, verbose=0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        netc.customTrainModel(genc, 1, 16, 0.2, n_steps=16 * 3000, verbose=0)
        torch.cuda.synchronize()
        line["custom_train_step_ms"] = 1e3 * (time.perf_counter() - t1) / 3000
        del netc, genc, fc_
        # pair scoring throughput (K6): 1M pairs gathered from a 100k x 512 embedding matrix
        E = torch.randn(100000, 512, device="cuda")
        E = E / E.norm(dim=1, keepdim=True)
        li = torch.randint(0, 100000, (1 << 20,), device="cuda", dtype=torch.int32)
        ri = torch.randint(0, 100000, (1 << 20,), device="cuda", dtype=torch.int32)
        po = torch.empty((1 << 20, 2), device="cuda")
        hd.predict_device(E, E, li, ri, out=po)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            hd.predict_device(E, E, li, ri, out=po)
        torch.cuda.synchronize()
        line["pair_scores_per_s"] = 3 * (1 << 20) / (time.perf_counter() - t1)
        hq.predict_device(E, E, li, ri, out=po)          # the same 1M pairs in the bf16 compute mode (bf16 matrix cores)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            hq.predict_device(E, E, li, ri, out=po)
        torch.cuda.synchronize()
        line["pair_scores_per_s_bf16"] = 3 * (1 << 20) / (time.perf_counter() - t1)
        # BASELINE configs[0] / BASELINE.md §4 leg 3: the SmallRes 32 x 32 siamese (reference code/siamese.py:134-184) trained END TO END,
        # one train_on_batch of 16 pairs (tower forward + backward on both sides, head, Adadelta; dropout masks drawn on the host)
        from a_link_amd.smallres import SmallResNet
        srn = SmallResNet((32, 32, 3), 2048, lr=0.1, seed=1, device=local_rank)
        rs_ = np.random.RandomState(0)
        sL = ((rs_.randint(0, 256, (16, 32, 32, 3)) - 128.0) / 128.0).astype(np.float32)
        sR = ((rs_.randint(0, 256, (16, 32, 32, 3)) - 128.0) / 128.0).astype(np.float32)
        sy = np.eye(2, dtype=np.float32)[rs_.randint(0, 2, 16)]
        np.random.seed(0)
        for _ in range(5):
            srn.train_on_batch([sL, sR], sy)
        ts = []
        for _ in range(50):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            srn.train_on_batch([sL, sR], sy)
            ts.append(time.perf_counter() - t1)
        # host (NumPy) operands, as the reference's Keras call takes them: one staging copy + one upload per step inside the time
        line["smallres32_train_step_ms"] = 1e3 * float(np.median(ts))
        # the same step with its operands already in HBM (the bench contract's convention for the timed region; the loop's own
        # generator hands SmallRes device tensors): what the device and its launches cost
        dL, dR, dy = (torch.from_numpy(a).to("cuda:%d" % local_rank) for a in (sL, sR, sy))
        for _ in range(5):
            srn.train_on_batch([dL, dR], dy)
        ts = []
        for _ in range(50):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            srn.train_on_batch([dL, dR], dy)
            ts.append(time.perf_counter() - t1)
        line["smallres32_train_step_resident_ms"] = 1e3 * float(np.median(ts))
        smallres_weights = srn.get_weights()
        del srn

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # ---- CPU baseline: the oracle (kind "port": our CPU restatement; the reference's MXNet /
        # Keras path cannot be installed) on a bounded sample of the same workload, four legs
        from oracle import siamese_head as OH
        params = W.synthetic_ir_params(units, seed=1, normalized=args.weights == "normalized")
        cores = torch.get_num_threads()
        xs = x[:8].float().cpu().numpy()  # bounded sample of the same pixels
        t1 = time.perf_counter()
        ref8 = ir_resnet.embed(params, xs, batch=8)
        one = time.perf_counter() - t1
        # (i) batched forward: BASELINE.md §4 asks for ONE N = 256 forward — timed as such (`value`; ~25-35 s on a 128-thread
        # host: the driver's run has the headroom).  Before it the smaller batches 32 / 64 / 128 while they fit a ~12 s
        # budget: the batch the host is FASTEST at is reported beside it (`fastest_batch`), and the oracle rows of the
        # parity check below come out of these forwards at no extra cost.
        best_b, best_rate, spent, trials = 8, 8 / one, one, [(8, 8 / one)]
        oracle_rows = ref8
        for b in (32, 64, 128):
            predicted = b / best_rate * 0.8
            if spent + predicted > 12.0 or b > B:
                break
            xb = x[:b].float().cpu().numpy()
            t1 = time.perf_counter()
            rb = ir_resnet.embed(params, xb, batch=b)
            tb = time.perf_counter() - t1
            spent += tb
            trials.append((b, b / tb))
            if b >= 32 and oracle_rows.shape[0] < 32:
                oracle_rows = rb[:32]
            if b / tb > best_rate:
                best_b, best_rate = b, b / tb
        n256 = None
        if B >= 256 and not args.cpu_baseline_small:
            x256c = x[:256].float().cpu().numpy()
            t1 = time.perf_counter()
            r256 = ir_resnet.embed(params, x256c, batch=256)
            t256 = time.perf_counter() - t1
            n256 = {"batch": 256, "seconds": t256, "value": 256 / t256}
            trials.append((256, 256 / t256))
            if oracle_rows.shape[0] < 32:
                oracle_rows = r256[:32]
        reps, cpu_dt = 1, best_b / best_rate
        # (ii) the reference's own shape: one image per call, normalised per image (FaceModel.get_feature,
        # reference code/face_model.py:86-93, looped by ArcFace.process, code/siamese.py:232-234)
        n_one = 8
        t1 = time.perf_counter()
        for i in range(n_one):
            ir_resnet.embed(params, xs[i:i + 1], batch=1)
        one_dt = time.perf_counter() - t1
        # (iii) head fine-tune step, batch 16 (Keras train_on_batch semantics, NumPy f32), median of 200
        oh = OH.HeadModel(512, lr=0.1, seed=0)
        rng = np.random.RandomState(0)
        Lh, Rh = rng.randn(16, 512).astype(np.float32), rng.randn(16, 512).astype(np.float32)
        yh = np.zeros((16, 2), np.float32)
        yh[np.arange(16), rng.randint(0, 2, 16)] = 1
        ts = []
        for _ in range(200):
            t1 = time.perf_counter()
            oh.train_on_batch([Lh, Rh], yh)
            ts.append(time.perf_counter() - t1)
        # (iv) pair scoring: 2^18 pairs gathered from a 100k x 512 matrix (the GPU leg scores 2^20)
        Ec = rng.randn(100000, 512).astype(np.float32)
        lc, rc_ = rng.randint(0, 100000, 1 << 18), rng.randint(0, 100000, 1 << 18)
        t1 = time.perf_counter()
        for s0 in range(0, 1 << 18, 1 << 14):
            oh.predict([Ec[lc[s0:s0 + (1 << 14)]], Ec[rc_[s0:s0 + (1 << 14)]]])
        ps_dt = time.perf_counter() - t1
        # (v) the SmallRes 32 x 32 end-to-end step on the host (torch-CPU autograd oracle), median of 20
        sm_cpu = None
        try:
            from oracle import smallres as OSM
            if smallres_weights is None:
                from a_link_amd.smallres import SmallResNet as _SRN
                smallres_weights = _SRN((32, 32, 3), 2048, lr=0.1, seed=1, device=local_rank).get_weights()
            osm = OSM.SmallResModel(smallres_weights, lr=0.1)
            rs_ = np.random.RandomState(0)
            sL = ((rs_.randint(0, 256, (16, 32, 32, 3)) - 128.0) / 128.0).astype(np.float32)
            sR = ((rs_.randint(0, 256, (16, 32, 32, 3)) - 128.0) / 128.0).astype(np.float32)
            sy = np.eye(2, dtype=np.float32)[rs_.randint(0, 2, 16)]
            tsm = []
            for _ in range(22):
                t1 = time.perf_counter()
                osm.train_on_batch([sL, sR], sy)
                tsm.append(time.perf_counter() - t1)
            sm_cpu = 1e3 * float(np.median(tsm[2:]))
        except Exception as e_:
            sm_cpu = repr(e_)
        line["cpu_baseline"] = {"value": n256["value"] if n256 else best_rate, "unit": "embeddings/s", "cores": cores, "kind": "port",
                                "sample": ("ONE N = 256 forward of the same %s network through the torch-CPU f32 oracle (BASELINE.md §4, leg 1): %.1f s; "
                                           "smaller batches tried before it: %s" % (args.model, n256["seconds"], ", ".join("batch %d: %.2f/s" % t for t in trials)))
                                          if n256 else
                                          ("one batch-%d forward of the same %s network (torch-CPU f32 oracle), the fastest of %s; the N = 256 forward of "
                                           "BASELINE.md §4 was skipped (--cpu-baseline-small or --batch < 256)" % (best_b, args.model, ", ".join("batch %d: %.2f/s" % t for t in trials))),
                                "batch": 256 if n256 else best_b,
                                "fastest_batch": {"batch": best_b, "value": best_rate},
                                "reference_shaped": {"value": n_one / one_dt, "unit": "embeddings/s",
                                                     "sample": "%d batch-1 forwards, each L2-normalised on its own: the "
                                                               "reference's get_feature loop" % n_one},
                                "finetune_step_ms_cpu": 1e3 * float(np.median(ts)),
                                "smallres32_train_step_ms_cpu": sm_cpu,
                                "pair_scores_per_s_cpu": (1 << 18) / ps_dt}
        # parity of the timed output against the oracle on the same images — 32 rows where the CPU legs above produced them
        # (8 rows at a 1.6x margin was thin), max AND mean (north_star: 1e-3 cosine)
        nrow = oracle_rows.shape[0]
        gotn = out[:nrow].cpu().numpy().astype(np.float64)
        omc = 1.0 - (gotn * oracle_rows).sum(1)
        parity["one_minus_cos_vs_cpu_oracle_max"] = float(omc.max())
        parity["one_minus_cos_vs_cpu_oracle_mean"] = float(omc.mean())
        parity["rows_above_8e-4"] = int((omc > 8e-4).sum())
        parity["oracle_rows"] = int(nrow)
        parity["weights"] = args.weights
        assert parity["one_minus_cos_vs_cpu_oracle_max"] < 1e-3, parity
        if args.weights == "survey" and args.dtype in ("bf16", "f16") and not args.cpu_baseline_small:
            # the same check on the OTHER weight draw (BatchNorm statistics matching the activations, like a trained checkpoint's:
            # the draw the config-3 / 4 / 5 legs use) — 8 images, one ~1.3 s CPU forward: the headline dtype's error is smaller there
            pn = W.synthetic_ir_params(units, seed=1, normalized=True)
            bn_ = IRBackbone(pn, image_size=(112, 112), dtype=args.dtype, device=local_rank, max_batch=8)
            gn = bn_.embed_device(x[:8]).cpu().numpy().astype(np.float64)
            on = ir_resnet.embed(pn, x[:8].float().cpu().numpy(), batch=8)
            omn = 1.0 - (gn * on).sum(1)
            parity["normalized_weights"] = {"one_minus_cos_vs_cpu_oracle_max": float(omn.max()), "one_minus_cos_vs_cpu_oracle_mean": float(omn.mean()),
                                            "oracle_rows": 8}
            del bn_, pn
        if sel_out is not None:
            sn = sel_out[:nrow].cpu().numpy().astype(np.float64)
            line["exact_selection"]["max_abs_diff_vs_cpu_oracle"] = float(np.abs(sn - oracle_rows).max())
            line["exact_selection"]["one_minus_cos_vs_cpu_oracle_max"] = float((1.0 - (sn * oracle_rows).sum(1)).max())
            line["exact_selection"]["oracle_rows"] = int(nrow)
            assert line["exact_selection"]["max_abs_diff_vs_cpu_oracle"] < 2e-5, line["exact_selection"]

    # which device every rank's handles live on (the C ABI's device rule: a handle belongs to the device current at its create
    # call; rank k of a one-process-per-GPU job must see its models on GPU LOCAL_RANK), gathered from all ranks
    from a_link_amd import _abi as _AB
    mine = torch.tensor([rank, local_rank, torch.cuda.current_device(), int(bb.device),
                         int(_AB.load().alink_backbone_device(bb.h))],
                        dtype=torch.int32, device="cuda")
    seen = [mine.clone() for _ in range(world)]
    if dist is not None:
        dist.all_gather(seen, mine)
    seen = [t.cpu().tolist() for t in seen]
    assert all(r[1] == r[2] == r[3] == r[4] for r in seen), "a rank's handle is not on its own device: %s" % seen
    if rank == 0:
        line["devices_seen"] = [{"rank": r[0], "local_rank": r[1], "current_device": r[2], "backbone_device": r[3], "handle_device": r[4]} for r in seen]
        line["parity"] = parity
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
