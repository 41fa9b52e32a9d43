"""GPU, one-rank RCCL group: the multi-GPU control path of distributed.py on the real device kernels — the
candidate exchange with the device-side merge, and the whole config-3 leg (shard -> committee of backbones ->
uncertainty -> local top-k -> merge) that bench.py times under torchrun.  (World size 2 runs on CPU / gloo in
tests/test_distributed.py; an 8-GPU node is the driver's.)"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture()
def rccl_group(gpu):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        yield dist
    finally:
        dist.destroy_process_group()


def test_merge_topk_on_device(rccl_group):
    from a_link_amd import distributed as D
    from a_link_amd import uncertainty as U
    rng = np.random.RandomState(0)
    s = rng.rand(50000).astype(np.float32)
    s[rng.randint(0, 50000, 5000)] = 0.75                       # many exact ties
    for largest in (True, False):
        for k, have in ((1024, 1024), (1024, 300), (16, 16)):
            sd = torch.from_numpy(s).cuda()
            idx, vals = U.topk_device(sd, have, largest=largest)
            v, i = D.merge_topk(vals, idx.to(torch.int64) + 7000, k, largest=largest)
            want = np.lexsort((np.arange(len(s)), -s if largest else s))[:have]
            assert v.is_cuda and i.dtype == torch.int64
            assert np.array_equal(i.cpu().numpy(), want + 7000)
            assert np.array_equal(v.cpu().numpy(), s[want])


def test_config3_leg_on_one_rank(rccl_group):
    """committee_pool_topk == the same computation spelled out call by call."""
    from a_link_amd import committee, distributed as D, siamese, uncertainty as U, weights as W
    from a_link_amd.backbone import IRBackbone
    size = (32, 32)
    bbs = [IRBackbone(W.synthetic_ir_params((1, 1, 1, 1), size=size, seed=s), image_size=size, max_batch=64) for s in (1, 2, 3)]
    nets = [siamese.SiameseNetwork((512,), "m%d" % i, 0.1, seed=20 + i) for i in range(3)]
    rng = np.random.default_rng(0)
    pool = rng.integers(0, 256, (100, 32, 32, 3), dtype=np.uint8)
    gallery = rng.integers(0, 256, (16, 32, 32, 3), dtype=np.uint8)
    vals, gidx = D.committee_pool_topk(bbs, [n.siamese_net for n in nets], torch.from_numpy(pool).cuda(),
                                       torch.from_numpy(gallery).cuda(), 64, shard_offset=1000)
    Ep = [torch.from_numpy(bb.embed(pool)).cuda() for bb in bbs]
    Eg = [torch.from_numpy(bb.embed(gallery)).cuda() for bb in bbs]
    li = np.repeat(np.arange(100, dtype=np.int32), 16)
    ri = np.tile(np.arange(16, dtype=np.int32), 100)
    probs = committee.Bagging(nets, []).predict_indexed(Ep, Eg, li, ri)
    ent = U.score_device(probs, "entropy").cpu().numpy()
    want = np.lexsort((np.arange(len(ent)), -ent))[:64]
    assert np.array_equal(gidx.cpu().numpy(), want + 1000 * 16)
    assert np.array_equal(vals.cpu().numpy(), ent[want])
    # per-member matrices really are per member: a shared-embedding committee gives a different answer
    shared = committee.Bagging(nets, []).predict_indexed(Ep[0], Eg[0], li, ri)
    assert not torch.equal(shared, probs)


def test_handles_live_on_their_own_device(gpu):
    """Device rule of the C ABI (include/alink_hip.h): a handle lives on the device current at its create call, every
    entry point switches to it for the call and back.  Needs two visible GPUs (a one-GPU box skips): models built
    while device 1 is current run there and leave the caller's current device alone; ArcFace's gpu=None means "this
    process's device", which is what rank 1 of a one-process-per-GPU job has set."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    from a_link_amd import siamese, weights as W
    from a_link_amd.backbone import IRBackbone
    from a_link_amd.head import DenseHead
    size = (32, 32)
    params = W.synthetic_ir_params((1, 1, 1, 1), size=size, seed=3)
    x = np.random.default_rng(0).integers(0, 256, (5, 32, 32, 3)).astype(np.float32)
    ref = IRBackbone(params, image_size=size, device=0, max_batch=8).embed(x)
    with torch.cuda.device(1):
        fm = siamese.ArcFace(size, "synthetic:r18:3", max_batch=8)           # gpu=None -> current device = 1
        assert fm.model.model.device == 1
        bb1 = IRBackbone(params, image_size=size, max_batch=8)
        h1 = DenseHead(512, seed=1)
    assert torch.cuda.current_device() == 0
    bb1_from0 = IRBackbone(params, image_size=size, device=1, max_batch=8)      # explicit device while 0 is current
    assert torch.cuda.current_device() == 0
    for bb in (bb1, bb1_from0):
        got = bb.embed_device(torch.from_numpy(x).to("cuda:1"))
        assert got.device.index == 1 and np.array_equal(got.cpu().numpy(), ref)
    assert torch.cuda.current_device() == 0
    h0 = DenseHead(512, seed=1, device=0)
    L = np.random.RandomState(0).randn(8, 512).astype(np.float32)
    p1 = h1.predict([L, L[::-1].copy()])
    assert np.array_equal(p1, h0.predict([L, L[::-1].copy()])) and torch.cuda.current_device() == 0


def _sharded_worker(rank, world, port, path):
    """One rank of a two-process job that SHARES cuda:0 (a one-GPU box): gloo carries the exchange."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        import a_link_amd  # noqa: F401
        from a_link_amd import distributed as D
        from a_link_amd.head import DenseHead
        L, R, y, cw, sw = _sharded_case()
        out = {}
        for exch in ("allreduce", "host"):
            try:
                hd = DenseHead(512, lr=0.1, seed=0, device=0)
                ms = []
                for step in range(3):
                    ms.append(D.dp_train_on_batch(hd, [L, R], y, class_weight=cw, sample_weight=sw, mode="sharded", exchange=exch))
                # a 1-row batch: rank 1's shard is EMPTY (its gradient buffer must be zeros, not stale)
                ms.append(D.dp_train_on_batch(hd, [L[:1], R[:1]], y[:1], mode="sharded", exchange=exch))
                out[exch] = (np.concatenate([w.ravel() for w in hd.get_weights()]), np.asarray(ms, np.float64))
            except RuntimeError as e:        # gloo built without device tensors: the host-staged exchange still has to work
                if exch == "host":
                    raise
                out[exch] = str(e)
        np.savez(path % rank, **{"%s_%s" % (k, n): v for k, val in out.items() if not isinstance(val, str)
                                 for n, v in (("w", val[0]), ("m", val[1]))},
                 skipped=np.array([k for k, val in out.items() if isinstance(val, str)]))
    finally:
        dist.destroy_process_group()


def _sharded_case():
    rs = np.random.RandomState(1)
    L, R = rs.randn(13, 512).astype(np.float32), rs.randn(13, 512).astype(np.float32)
    y = np.zeros((13, 2), np.float32)
    y[np.arange(13), rs.randint(0, 2, 13)] = 1
    cw = {0: 1.0 / 3, 1: 2.0 / 3}
    sw = np.ones(13, np.float32)
    sw[4] = 0.0                                   # one zero weight: Keras divides by the count of non-zero weights
    return L, R, y, cw, sw


def test_sharded_finetune_step_world_size_2(gpu, tmp_path):
    """distributed.dp_train_on_batch(mode="sharded") with TWO ranks (reference step: code/siamese.py:52-58): two
    spawned processes share cuda:0, the exchange goes through gloo — exchange="allreduce" on the device buffer where
    this gloo build reduces device tensors, and exchange="host" (staged through the host) always.  13 rows with class
    weights and one zero sample weight (rank 0: 7 rows, rank 1: 6), then a 1-row batch (rank 1's shard is empty): the
    weights after the four steps equal the single-process steps' to 2e-6 and the metrics agree, on both ranks."""
    import socket
    import torch.multiprocessing as mp
    from a_link_amd.head import DenseHead
    L, R, y, cw, sw = _sharded_case()
    hd = DenseHead(512, lr=0.1, seed=0, device=0)
    want_m = [hd.train_on_batch([L, R], y, class_weight=cw, sample_weight=sw) for _ in range(3)]
    want_m.append(hd.train_on_batch([L[:1], R[:1]], y[:1]))
    want_w = np.concatenate([w.ravel() for w in hd.get_weights()])
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    path = str(tmp_path / "rank%d.npz")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, path)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    ran = set()
    for r in range(2):
        z = np.load(path % r)
        for exch in ("allreduce", "host"):
            if exch in z["skipped"].tolist():
                continue
            ran.add(exch)
            assert np.abs(z[exch + "_w"] - want_w).max() < 2e-6, (r, exch, np.abs(z[exch + "_w"] - want_w).max())
            assert np.abs(z[exch + "_m"] - np.asarray(want_m, np.float64)).max() < 2e-6, (r, exch, z[exch + "_m"], want_m)
    assert "host" in ran
    print("sharded fine-tune step at world size 2: exchanges exercised = %s" % sorted(ran))


def _calib_worker(rank, world, port, path):
    """One rank of a two-process job sharing cuda:0: calibrates its split-precision backbone on ITS shard, embeds the shared
    gallery, then takes rank 0's scales (distributed.broadcast_calibration) and embeds it again."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        import a_link_amd  # noqa: F401
        from a_link_amd import distributed as D, weights as W
        from a_link_amd.backbone import IRBackbone
        size = (32, 32)
        params = W.synthetic_ir_params((1, 2, 1, 1), size=size, seed=3)
        rng = np.random.default_rng(0)
        gallery = rng.integers(0, 256, (16, 32, 32, 3)).astype(np.float32)
        shard = rng.integers(0, 256, (2, 24, 32, 32, 3)).astype(np.float32)[rank] * (1.0 if rank == 0 else 40.0)   # rank 1: far brighter images
        bb = IRBackbone(params, image_size=size, max_batch=16, dtype="f16x2")
        bb.calibrate(shard)                                      # 24 images: two chunks of <= 16
        own = bb.embed(gallery)
        st_own = bb.state()
        D.broadcast_calibration([bb])
        shared = bb.embed(gallery)
        np.savez(path % rank, own=own, shared=shared, e_own=np.asarray(st_own["scale_exponents"]),
                 e_shared=np.asarray(bb.state()["scale_exponents"]))
    finally:
        dist.destroy_process_group()


def test_calibration_state_is_portable_across_ranks_and_processes(gpu, tmp_path):
    """Portable, rank-consistent calibration of the split-precision mode (include/alink_hip.h: alink_backbone_get_scales /
    set_scales; the persistence contract it serves is reference code/siamese.py:114-125).  Two spawned processes calibrate on
    different shards: their scales differ and the shared gallery embeds to different bits (drift far below the mode's
    error); after the broadcast both embed it to the SAME bits.  Then a save / load round trip: a fresh handle that loads
    the saved state reproduces the embeddings bit for bit, without calibrating."""
    import json
    import socket
    import torch.multiprocessing as mp
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    path = str(tmp_path / "rank%d.npz")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_calib_worker, args=(r, 2, port, path)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    z0, z1 = np.load(path % 0), np.load(path % 1)
    assert not np.array_equal(z0["e_own"], z1["e_own"])                       # different shards -> different scales
    drift = float(np.abs(z0["own"] - z1["own"]).max())
    assert 0 < drift < 2e-6, drift                                            # different bits, far below the mode's error
    assert np.array_equal(z0["e_shared"], z0["e_own"]) and np.array_equal(z1["e_shared"], z0["e_own"])
    assert np.array_equal(z0["shared"], z1["shared"]) and np.array_equal(z0["shared"], z0["own"])
    print("calibration on different shards: max |d embedding| %.2e; after the broadcast: bit-equal" % drift)
    # save / load round trip in THIS process (a third one): no calibration run, same bits
    from a_link_amd import siamese, weights as W
    from a_link_amd.backbone import IRBackbone
    size = (32, 32)
    params = W.synthetic_ir_params((1, 2, 1, 1), size=size, seed=3)
    gallery = np.random.default_rng(0).integers(0, 256, (16, 32, 32, 3)).astype(np.float32)
    bb = IRBackbone(params, image_size=size, max_batch=16, dtype="f16x2")
    st = json.loads(json.dumps({"dtype": "f16x2", "units": [1, 2, 1, 1], "image_size": [32, 32], "scale_exponents": z0["e_own"].tolist()}))
    assert not np.array_equal(bb.embed(gallery), z0["own"]) or bb.state()["scale_exponents"] == st["scale_exponents"]
    bb.load_state(st)
    assert np.array_equal(bb.embed(gallery), z0["own"])
    with pytest.raises(Exception):
        bb.load_state(dict(st, units=[1, 1, 1, 1]))
    # ... and through the feature model's own save / load next to a checkpoint path
    fm = siamese.ArcFace(size, "synthetic:r18:3", max_batch=16)
    fm.calibrate(gallery * 3.0)
    e1 = fm.process(gallery)
    fm.save_calibration(str(tmp_path / "scales.json"))
    fm2 = siamese.ArcFace(size, "synthetic:r18:3", max_batch=16)
    assert fm2.maybeLoadCalibration(str(tmp_path / "scales.json")) and not fm2.maybeLoadCalibration(str(tmp_path / "absent.json"))
    assert np.array_equal(fm2.process(gallery), e1)


def _settle_case():
    """a small committee workload: 3 members ((1,1,1,1) nets at 32x32), 120 pool images of 12 identities x 16 gallery images"""
    from a_link_amd import weights as W
    rng = np.random.default_rng(4)
    bases = rng.integers(40, 216, (12, 4, 4, 3)).repeat(8, axis=1).repeat(8, axis=2)
    pool = np.clip(bases[np.arange(120) // 10] + rng.integers(-40, 41, (120, 32, 32, 3)), 0, 255).astype(np.uint8)
    gallery = np.clip(bases[:12] + rng.integers(-40, 41, (12, 32, 32, 3)), 0, 255).astype(np.uint8)
    params = [W.synthetic_ir_params((1, 1, 1, 1), size=(32, 32), seed=s, normalized=True) for s in (1, 2, 3)]
    return pool, gallery, params


def _settle_members(params, pool, gallery):
    from a_link_amd.backbone import IRBackbone
    from a_link_amd.head import DenseHead
    exa, scr, heads = [], [], []
    for m, p in enumerate(params):
        e = IRBackbone(p, image_size=(32, 32), max_batch=64, dtype="f16x2")
        e.calibrate(pool[:32])                                      # the SAME images on every rank: rank-consistent scales
        exa.append(e)
        scr.append(IRBackbone(p, image_size=(32, 32), max_batch=64, dtype="f16"))
        h = DenseHead(512, lr=0.1, seed=10 + m)
        ws = h.get_weights()
        ws[4] = ws[4] * np.float32(40.0)                            # spread the probabilities (a fresh head: 0.5 +- 0.02)
        h.set_weights(ws)
        heads.append(h)
    return exa, scr, heads


def _settle_worker(rank, world, port, path):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        import a_link_amd  # noqa: F401
        from a_link_amd import distributed as D
        pool, gallery, params = _settle_case()
        exa, scr, heads = _settle_members(params, pool, gallery)
        lo, hi = D.shard_range(len(pool), rank, world)
        info = {}
        v, i = D.committee_pool_topk_settled(scr, exa, heads, torch.from_numpy(pool[lo:hi]).cuda(), torch.from_numpy(gallery).cuda(),
                                             64, shard_offset=lo, info=info, min_sample=8)
        np.savez(path % rank, v=v.cpu().numpy(), i=i.cpu().numpy(), settled=info["images_settled"], n=hi - lo, rounds=info["rounds"])
    finally:
        dist.destroy_process_group()


def test_screen_then_settle_over_two_ranks_equals_single_process_exact(gpu, tmp_path):
    """distributed.committee_pool_topk_settled with TWO ranks (two processes on one card, gloo carrying the candidate exchange
    and the two scalars per round): both ranks return the scores, order and indices of the single-process all-exact pass."""
    import socket
    import torch.multiprocessing as mp
    from a_link_amd import distributed as D
    pool, gallery, params = _settle_case()
    exa, scr, heads = _settle_members(params, pool, gallery)
    want_v, want_i = D.committee_pool_topk(exa, heads, torch.from_numpy(pool).cuda(), torch.from_numpy(gallery).cuda(), 64, shard_offset=0)
    only_screen = D.committee_pool_topk(scr, heads, torch.from_numpy(pool).cuda(), torch.from_numpy(gallery).cuda(), 64, shard_offset=0)[1]
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    path = str(tmp_path / "rank%d.npz")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_settle_worker, args=(r, 2, port, path)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    z = [np.load(path % r) for r in range(2)]
    for r in range(2):
        assert np.array_equal(z[r]["i"], want_i.cpu().numpy()) and np.array_equal(z[r]["v"], want_v.cpu().numpy()), r
    assert z[0]["settled"] + z[1]["settled"] < z[0]["n"] + z[1]["n"]                  # not the trivial "settle everything"
    print("two ranks: settled %d + %d of %d + %d images in %d rounds; screening alone differs in %d of 64"
          % (z[0]["settled"], z[1]["settled"], z[0]["n"], z[1]["n"], z[0]["rounds"],
             len(set(only_screen.cpu().numpy().tolist()) ^ set(want_i.cpu().numpy().tolist())) // 2))


# ---- the multi-rank A-LINK loop on the device models (BASELINE configs[3] / configs[4]) -------------------------------
_LOOP_SIZE = (32, 32)
_LOOP_NOISES = ("gaussian", "saltpepper", "poisson", "speckle")


def _loop_people(n, seed, lo=2, hi=3):
    rng = np.random.RandomState(seed)
    return [rng.randint(0, 256, (rng.randint(lo, hi + 1),) + _LOOP_SIZE + (3,)).astype(np.float32) for _ in range(n)]


def _loop_run(group, rank, screen, tmp):
    """run_alink_dfw on the HIP models (ArcFace in split precision with a 16-bit screening form, DenseHead student and
    committee, the device noise kernels); returns what every rank — and the single-process loop — must agree on"""
    from a_link_amd import alink_loop as AL, committee, noise, pairs, settle, siamese
    flags = AL.Flags(alink_bs=3, batch_send=6, disparity_ratio=0.6, eps=0.0005, ft_epochs=2, mixture_ratio=2,
                     out_model=os.path.join(tmp, "post%d" % rank), screen_settle=screen is not None)
    X_plain, X_dig = _loop_people(6, 1), _loop_people(6, 2)
    conv = siamese.ArcFace(_LOOP_SIZE, "synthetic:r18:3", screen_dtype=screen)
    conv.calibrate(np.concatenate(X_plain + X_dig))                 # the SAME images on every rank: rank-consistent scales
    student = siamese.SiameseNetwork((512,), "student", 0.1, seed=7)
    ens = [siamese.SiameseNetwork((512,), "ens%d" % i, 0.1, seed=100 + i) for i in range(2)]
    # rank 0 carries the single-process run's seeds; the other rank starts from different noise streams and host randomness
    nz = [noise.get_relevant_noise(n)(model=student, sess=None, feature_model=conv, seed=1000 + i + 50 * rank) for i, n in enumerate(_LOOP_NOISES)]
    bag = committee.Bagging(ens, nz)
    feats_plain = [conv.process(p) for p in X_plain]
    gen = pairs.getGenerator(pairs.getNormalGenerator(feats_plain, 8), pairs.getNormalGenerator(feats_plain, 8),
                             pairs.getImposterGenerator(feats_plain, feats_plain, 8), 8)
    np.random.seed(5 + 31 * rank)
    sets = []
    o1, o2 = AL.selection.select_queries, settle.select_queries_settled
    AL.selection.select_queries = lambda *a, **k: (lambda r: (sets.append(list(r[0])), r)[1])(o1(*a, **k))
    settle.select_queries_settled = lambda *a, **k: (lambda r: (sets.append(list(r[0])), r)[1])(o2(*a, **k))
    try:
        st = AL.run_alink_dfw(flags, conv, bag, nz, student, X_plain, X_dig, gen, _LOOP_SIZE, col=0, verbose=0, group=group)
    finally:
        AL.selection.select_queries, settle.select_queries_settled = o1, o2
    return {"counts": np.array([st.active_count, st.un_size, st.finetunes]), "sets": np.array([len(s) for s in sets] + sum(sets, [])),
            "w": np.concatenate([w.ravel() for w in student.siamese_net.get_weights()]),
            "rows": np.array([i.get("rows_of_this_rank", -1) for i in st.settle_info]),
            "saved": np.array([os.path.exists(flags.out_model + ".h5")])}


def _loop_worker(rank, world, port, path, tmp):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        import a_link_amd  # noqa: F401
        for tag, screen in (("exact", None), ("settle", "f16")):
            np.savez(path % (tag, rank), **_loop_run(dist.group.WORLD, rank, screen, tmp))
    finally:
        dist.destroy_process_group()


def test_multirank_alink_loop_equals_single_process_loop(gpu, tmp_path):
    """alink_loop.run_alink_dfw(group=...) with TWO ranks (two processes on one card, gloo carrying the prediction all-gather,
    the settle replies and the fine-tune rows): each rank perturbs, embeds and scores its own rows of every iteration's pair
    batch with the device kernels.  Every iteration's query list, the oracle count, the number of fine-tunes and the
    student's weights afterwards must equal the single-process loop's BIT FOR BIT on both ranks — all-exact and
    screen-then-settle — and only rank 0 writes the model file.  (Reference loop: code/ALINK_arc.py:142-254.)"""
    import socket
    import torch.multiprocessing as mp
    want = {tag: _loop_run(None, 0, screen, str(tmp_path)) for tag, screen in (("exact", None), ("settle", "f16"))}
    assert want["exact"]["counts"][2] >= 1 and want["exact"]["sets"].size > 8, "test data must select queries and fine-tune"
    assert np.array_equal(want["exact"]["sets"], want["settle"]["sets"]) and np.array_equal(want["exact"]["w"], want["settle"]["w"])
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    path = str(tmp_path / "%s_rank%d.npz")
    for f in os.listdir(str(tmp_path)):
        if f.endswith(".h5"):
            os.remove(os.path.join(str(tmp_path), f))
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_loop_worker, args=(r, 2, port, path, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    for tag in ("exact", "settle"):
        for r in range(2):
            z = np.load(path % (tag, r))
            assert np.array_equal(z["counts"], want[tag]["counts"]), (tag, r, z["counts"], want[tag]["counts"])
            assert np.array_equal(z["sets"], want[tag]["sets"]), (tag, r)
            assert np.array_equal(z["w"], want[tag]["w"]), (tag, r, np.abs(z["w"] - want[tag]["w"]).max())
            assert bool(z["saved"][0]) == (r == 0), (tag, r)
        z0, z1 = np.load(path % (tag, 0)), np.load(path % (tag, 1))
        if tag == "settle":
            assert (z0["rows"] > 0).all() and (np.abs(z0["rows"] - z1["rows"]) <= 1).all()
    print("two ranks: per-iteration rows %s / %s; queries per iteration %s"
          % (z0["rows"].tolist(), z1["rows"].tolist(), want["exact"]["sets"][:int(want["exact"]["counts"][1] > 0) * 3].tolist()))


def test_multirank_alink_loop_on_a_one_rank_rccl_group(rccl_group, tmp_path):
    """The same loop under "nccl" (= RCCL) with one rank: the collectives of distributed.RowShards are staged on the DEVICE
    there (all_gather_into_tensor on device buffers, object collectives through the GPU) — the path an 8-GPU job takes —
    and must leave the single-process results untouched."""
    want = _loop_run(None, 0, "f16", str(tmp_path))
    got = _loop_run(rccl_group.group.WORLD, 0, "f16", str(tmp_path))
    assert np.array_equal(got["counts"], want["counts"]) and np.array_equal(got["sets"], want["sets"]) and np.array_equal(got["w"], want["w"])
    assert (got["rows"] > 0).all()


def _mtp_and_attack_run(group, rank, tmp):
    """(1) run_alink_mtp: SmallRes student trained on low-res pixels, replicated train steps, dropout masks and shuffles from
    rank 0's host randomness; (2) run_alink_dfw with the few-pixel attack (a short search) beside Gaussian noise: the
    differential-evolution searches shard by pair, each with the stream of its global row."""
    from a_link_amd import alink_loop as AL, committee, noise, pairs, siamese
    out = {}
    conv = siamese.ArcFace(_LOOP_SIZE, "synthetic:r18:3", screen_dtype=None)
    conv.calibrate(np.concatenate(_loop_people(6, 1) + _loop_people(6, 2)))
    # ---- (1) Multi-PIE shape
    low = (16, 16)
    student = siamese.SmallRes(low + (3,), (64,), os.path.join(tmp, "lowres%d" % rank), 0.1, seed=2)
    ens = [siamese.SiameseNetwork((512,), "e%d" % i, 0.1, seed=50 + i) for i in range(2)]
    nz = [noise.Gaussian(seed=1 + 9 * rank), noise.Noise()]
    bag = committee.Bagging(ens, nz)
    rng = np.random.RandomState(3)
    people = [rng.randint(0, 256, (2, 40, 40, 3)).astype(np.float32) for _ in range(6)]
    gen = pairs.getGeneratorMTP(pairs.getNormalGenerator(people, 16), 8, resize_res=low)
    flags = AL.Flags(alink_bs=3, batch_send=4, disparity_ratio=1.0, eps=0.0, ft_epochs=1, active_ratio=2.0, out_model="")
    np.random.seed(rank * 17)
    st = AL.run_alink_mtp(flags, conv, bag, nz, student, people, gen, _LOOP_SIZE, low, verbose=0, group=group)
    out["mtp_counts"] = np.array([st.iterations, st.un_size, st.active_count, st.finetunes])
    out["mtp_w"] = np.concatenate([np.asarray(w).ravel() for w in student.siamese_net.get_weights()])
    # ---- (2) A2-LINK: few-pixel attack among the noises
    student2 = siamese.SiameseNetwork((512,), "student", 0.1, seed=7)
    ens2 = [siamese.SiameseNetwork((512,), "ens%d" % i, 0.1, seed=100 + i) for i in range(2)]
    nz2 = [noise.Gaussian(seed=40 + rank), noise.AdversarialNoise(student2, None, conv, seed=41 + 5 * rank, pixel_count=3, maxiter=2, popsize=15)]
    bag2 = committee.Bagging(ens2, nz2)
    X_plain, X_dig = _loop_people(3, 1), _loop_people(3, 2)
    feats_plain = [conv.process(p) for p in X_plain]
    gen2 = pairs.getGenerator(pairs.getNormalGenerator(feats_plain, 8), pairs.getNormalGenerator(feats_plain, 8),
                              pairs.getImposterGenerator(feats_plain, feats_plain, 8), 8)
    flags2 = AL.Flags(alink_bs=3, batch_send=6, disparity_ratio=0.6, eps=0.0005, ft_epochs=1, mixture_ratio=2, out_model="", screen_settle=False)
    np.random.seed(5 + 31 * rank)
    st2 = AL.run_alink_dfw(flags2, conv, bag2, nz2, student2, X_plain, X_dig, gen2, _LOOP_SIZE, col=0, verbose=0, group=group)
    out["adv_counts"] = np.array([st2.active_count, st2.un_size, st2.finetunes])
    out["adv_w"] = np.concatenate([w.ravel() for w in student2.siamese_net.get_weights()])
    # ---- (3) BASELINE configs[4] as worded: gradient-attack noise (FGSM / PGD: extensions, a random start keyed by the global
    # element) + a student fine-tuned in the head's bf16 compute mode
    convg = siamese.ArcFace(_LOOP_SIZE, "synthetic:r18:3", enable_grad=True, max_batch=64)
    student3 = siamese.SiameseNetwork((512,), "student", 0.1, seed=7, compute_dtype="bf16")
    ens3 = [siamese.SiameseNetwork((512,), "ens%d" % i, 0.1, seed=100 + i) for i in range(2)]
    nz3 = [noise.FGSM(student3, None, convg, eps=4.0, seed=60 + rank), noise.PGD(student3, None, convg, eps=4.0, alpha=1.5, steps=2, seed=61 + 3 * rank)]
    bag3 = committee.Bagging(ens3, nz3)
    feats3 = [convg.process(p) for p in X_plain]
    gen3 = pairs.getGenerator(pairs.getNormalGenerator(feats3, 8), pairs.getNormalGenerator(feats3, 8),
                              pairs.getImposterGenerator(feats3, feats3, 8), 8)
    flags3 = AL.Flags(alink_bs=3, batch_send=6, disparity_ratio=0.6, eps=0.0005, ft_epochs=1, mixture_ratio=2, out_model="", screen_settle=False)
    np.random.seed(9 + 13 * rank)
    st3 = AL.run_alink_dfw(flags3, convg, bag3, nz3, student3, X_plain, X_dig, gen3, _LOOP_SIZE, col=0, verbose=0, group=group)
    out["pgd_counts"] = np.array([st3.active_count, st3.un_size, st3.finetunes])
    out["pgd_w"] = np.concatenate([w.ravel() for w in student3.siamese_net.get_weights()])
    return out


def _mtp_worker(rank, world, port, path, tmp):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        import a_link_amd  # noqa: F401
        np.savez(path % rank, **_mtp_and_attack_run(dist.group.WORLD, rank, tmp))
    finally:
        dist.destroy_process_group()


def test_multirank_mtp_loop_and_few_pixel_attack_equal_single_process(gpu, tmp_path):
    """Two ranks on one card: the Multi-PIE loop (code/ALINK_MTP.py:150-266 — SmallRes trained end to end, every rank running
    the same train steps on rank 0's dropout masks and shuffles), an A2-LINK iteration whose noises include the few-pixel
    attack (code/attack.py:91-103: the searches split by pair), and BASELINE configs[4] as worded — FGSM / PGD noise
    (extensions) with a student fine-tuned in the bf16 compute mode.  Counts and student weights equal the single-process
    runs bit for bit on both ranks."""
    import socket
    import torch.multiprocessing as mp
    want = _mtp_and_attack_run(None, 0, str(tmp_path))
    assert want["mtp_counts"][3] >= 1 and want["adv_counts"][0] > 0 and want["pgd_counts"][0] > 0
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    path = str(tmp_path / "mtp_rank%d.npz")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_mtp_worker, args=(r, 2, port, path, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    for r in range(2):
        z = np.load(path % r)
        for k in want:
            assert np.array_equal(z[k], want[k]), (r, k, z[k][:4], want[k][:4])


def test_bench_preflight_under_a_launcher_with_one_rank(gpu):
    """`bench.py --dry-ranks`: under torch.distributed.run with one rank every collective shape of the timed
    legs runs on tiny known data over RCCL — all-reduce, merge_topk, the RowShards gathers, a calibration broadcast — each
    under a watchdog, one line per rank, exit code 0, in seconds."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    t = time.perf_counter()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29547", os.path.join(root, "bench.py"), "--gpus", "1", "--dry-ranks"],
                       cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-800:]
    assert "[preflight] rank 0 of 1" in r.stdout and "all_reduce, merge_topk, row_shards, broadcast_calibration ok" in r.stdout, r.stdout[-400:]
    assert time.perf_counter() - t < 120


def test_settled_pool_pass_survives_a_self_recalibration(gpu):
    """the exact handles re-calibrate themselves when a batch leaves the split-precision range, which changes the last
    bits of everything embedded afterwards — inside committee_pool_topk_settled the gallery (embedded first) and the rows settled
    later would mix two sets of scales.  Here the exact members are calibrated on ordinary images and the pool is 64x BRIGHTER: the screening
    view's pool pass (or the first exact batch) leaves the range, the scales drop mid-pass; the function must notice
    (info["recalibrated"]), run the pass again under the new scales, and return what the all-exact pass returns under those scales."""
    from a_link_amd import distributed as D
    from a_link_amd.backbone import IRBackbone
    from a_link_amd.head import DenseHead
    pool, gallery, params = _settle_case()
    exa, heads = [], []
    for m, p in enumerate(params[:2]):
        e = IRBackbone(p, image_size=(32, 32), max_batch=64, dtype="f16x2")
        e.calibrate(pool[:32])                                      # ordinary pixels; the pool below is 64x brighter: its activations leave these scales' range
        exa.append(e)
        h = DenseHead(512, lr=0.1, seed=10 + m)
        ws = h.get_weights()
        ws[4] = ws[4] * np.float32(40.0)
        h.set_weights(ws)
        heads.append(h)
    before = [e.state()["scale_exponents"] for e in exa]
    pd, gd = torch.from_numpy(pool).cuda().float() * 64.0, torch.from_numpy(gallery).cuda().float()
    info = {}
    v, i = D.committee_pool_topk_settled([e.screening_view() for e in exa], exa, heads, pd, gd, 48, shard_offset=0, info=info, min_sample=8, audit=8)
    after = [e.state()["scale_exponents"] for e in exa]
    assert info["recalibrated"] is True and after != before and all(a <= b for aa, bb in zip(after, before) for a, b in zip(aa, bb))
    want_v, want_i = D.committee_pool_topk(exa, heads, pd, gd, 48, shard_offset=0)           # all-exact under the scales it ended with
    assert [e.state()["scale_exponents"] for e in exa] == after
    assert torch.equal(i, want_i) and torch.equal(v, want_v)


# ---- SmallRes data-parallel step ("all-reduce(sum) of head / SmallRes gradients") -----------------------------
def _smallres_case():
    rs = np.random.RandomState(3)
    n = 9
    L = ((rs.randint(0, 256, (n, 32, 32, 3)) - 128.0) / 128.0).astype(np.float32)
    R = ((rs.randint(0, 256, (n, 32, 32, 3)) - 128.0) / 128.0).astype(np.float32)
    y = np.eye(2, dtype=np.float32)[rs.randint(0, 2, n)]
    sw = np.ones(n, np.float32)
    sw[2] = 0.0
    sw[5] = 2.5
    return L, R, y, sw


def _smallres_worker(rank, world, port, path):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        import a_link_amd  # noqa: F401
        from a_link_amd import distributed as D
        from a_link_amd.smallres import SmallResNet
        L, R, y, sw = _smallres_case()
        out = {}
        for mode in ("sharded", "replicated", "sharded_nodrop"):
            net = SmallResNet((32, 32, 3), 256, lr=0.1, seed=5, device=0)
            net.training_dropout = mode != "sharded_nodrop"
            np.random.seed(11)                                    # every rank holds the same host random state (alink_loop keeps it so)
            ms = []
            for step in range(3):
                ms.append(D.dp_train_on_batch(net, [L, R], y, sample_weight=sw, mode=mode.split("_")[0], exchange="host"))
            ms.append(D.dp_train_on_batch(net, [L[:1], R[:1]], y[:1], mode=mode.split("_")[0], exchange="host"))     # rank 1's shard is empty
            out[mode + "_w"] = np.concatenate([w.ravel() for w in net.get_weights()])
            out[mode + "_m"] = np.asarray(ms, np.float64)
        np.savez(path % rank, **out)
    finally:
        dist.destroy_process_group()


def test_smallres_sharded_step_world_size_2(gpu, tmp_path):
    """distributed.dp_train_on_batch on the end-to-end SmallRes student (reference code/siamese.py:134-170, trained by
    code/ALINK_MTP.py:121,255) with TWO ranks sharing cuda:0 (gloo carries the 2.4 MB exchange at feat = 256): each rank
    runs forward + backward on ITS rows with the global normaliser and the dropout masks of its GLOBAL rows, the flat
    [tower | head] gradient buffers are summed, both ranks apply the same Adadelta update.  9 pairs (rank 0: 5, rank 1: 4)
    with a zero and a 2.5 sample weight, three steps, then a 1-pair batch (rank 1's shard empty).  Sharded = the
    single-process steps to f32 summation order (weights 2e-6, metrics 2e-6) with dropout ON — which proves the masks of a
    slice are the whole batch's masks for those rows; replicated = the single-process steps bit for bit; both ranks end
    with the same bits."""
    import socket
    import torch.multiprocessing as mp
    from a_link_amd.smallres import SmallResNet
    L, R, y, sw = _smallres_case()
    want = {}
    for drop in (True, False):
        net = SmallResNet((32, 32, 3), 256, lr=0.1, seed=5, device=0)
        net.training_dropout = drop
        np.random.seed(11)
        ms = [net.train_on_batch([L, R], y, sample_weight=sw) for _ in range(3)]
        ms.append(net.train_on_batch([L[:1], R[:1]], y[:1]))
        want[drop] = (np.concatenate([w.ravel() for w in net.get_weights()]), np.asarray(ms, np.float64))
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    path = str(tmp_path / "sr%d.npz")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_smallres_worker, args=(r, 2, port, path)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    z = [np.load(path % r) for r in range(2)]
    for mode, drop, tol in (("sharded", True, 2e-6), ("replicated", True, 0.0), ("sharded_nodrop", False, 2e-6)):
        assert np.array_equal(z[0][mode + "_w"], z[1][mode + "_w"]), mode                  # the ranks agree to the bit
        for r in range(2):
            dw = np.abs(z[r][mode + "_w"] - want[drop][0]).max()
            dm = np.abs(z[r][mode + "_m"] - want[drop][1]).max()
            assert dw <= tol and dm <= max(tol, 0.0 if tol == 0.0 else 2e-6), (mode, r, dw, dm)


def test_smallres_masks_of_a_slice_are_the_batch_masks_of_its_rows(gpu):
    """SmallResNet._masks_of_rows (alink_keep_masks_at): the keep-masks a rank draws for pair rows lo : hi of an n-pair batch
    are, element for element, what train_on_batch's one stream of 2n(e1 + e2) masks holds for those rows — for slices that
    start at any row, an empty one, and a first element that is no multiple of 4 (the Philox block size)."""
    from a_link_amd.smallres import SmallResNet
    net = SmallResNet((20, 20, 3), 64, lr=0.1, seed=1, device=0)          # e1 = 9*9*32, e2 = 3*3*64: odd multiples
    e1, e2 = net.mask_sizes
    lib = gpu.load()
    n, seed = 7, 12345
    whole = torch.empty(2 * n * (e1 + e2), dtype=torch.uint8, device="cuda")
    gpu.check(lib.alink_keep_masks(gpu.ptr(whole), whole.numel(), 0.75, seed, None))
    w = whole.cpu().numpy()
    m1, m2 = w[:2 * n * e1].reshape(2 * n, e1), w[2 * n * e1:].reshape(2 * n, e2)
    assert 0.70 < w.mean() < 0.80
    for lo, hi in ((0, 7), (0, 3), (3, 7), (2, 3), (6, 7)):
        got = net._masks_of_rows(lo, hi, n, seed).cpu().numpy()
        k = hi - lo
        want = np.concatenate([m1[lo:hi].ravel(), m1[n + lo:n + hi].ravel(), m2[lo:hi].ravel(), m2[n + lo:n + hi].ravel()])
        assert got.shape == want.shape == (2 * k * (e1 + e2),) and np.array_equal(got, want), (lo, hi)
    # an unaligned first element on its own
    part = torch.empty(1001, dtype=torch.uint8, device="cuda")
    gpu.check(lib.alink_keep_masks_at(gpu.ptr(part), 1001, 0.75, seed, 4321 + 2, None))
    assert np.array_equal(part.cpu().numpy(), w[4323:4323 + 1001])
