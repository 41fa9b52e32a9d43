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
