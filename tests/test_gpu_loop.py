"""GPU: the A-LINK framework loop as library code (a-link_amd/alink_loop.py) against a literal,
reference-shaped restatement of code/ALINK_arc.py:142-254 on the same models, data and noise seeds:
identical oracle-query counts, identical fine-tune sets, identical student weights afterwards."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SIZE = (32, 32)


def _people(n, seed, lo=2, hi=3):
    rng = np.random.RandomState(seed)
    return [rng.randint(0, 256, (rng.randint(lo, hi + 1),) + SIZE + (3,)).astype(np.float32) for _ in range(n)]


def _build(seed, noises=("gaussian", "speckle")):
    from a_link_amd import committee, noise, siamese
    conv = siamese.ArcFace(SIZE, "synthetic:r18:3")
    student = siamese.SiameseNetwork((512,), "student", 0.1, seed=seed)
    ens = [siamese.SiameseNetwork((512,), "ens%d" % i, 0.1, seed=100 + i) for i in range(2)]
    nz = [noise.get_relevant_noise(n)(model=student, sess=None, feature_model=conv) for n in noises]
    for i, z in enumerate(nz):
        z._seed, z._calls = 1000 + i, 0
    return conv, student, ens, committee.Bagging(ens, nz), nz


def _literal_loop(flags, conv, bag, nz, student, X_plain_raw, X_dig_post, dataGen, col):
    """code/ALINK_arc.py:142-254, line for line (np.argsort made stable; Set -> sorted set)."""
    from a_link_amd import helpers, pairs
    tl, tr, ty = np.array([]), np.array([]), np.array([])
    ACTIVE, UN, sets, fts = 0, 0, [], 0
    for ii in range(0, len(X_dig_post), flags.alink_bs):
        batch_x, batch_y = pairs.createMiniBatch(X_plain_raw[ii:ii + flags.alink_bs], X_dig_post[ii:ii + flags.alink_bs])
        UN += len(batch_x[0])
        feats = [conv.process(p) for p in batch_x]                       # every pair occurrence embedded
        ens = bag.predict(feats)
        noisy = bag.attackModel(batch_x, SIZE, np.argmax(ens, axis=1))
        noisy = [[conv.process(p) for p in part] for part in noisy]
        dps = [student.predict([noisy[0][j], noisy[1][j]]) for j in range(len(nz))]
        mis = []
        for dp in dps:
            d = [-np.absolute(dp[j][col] - ens[j][col]) for j in range(len(dp))]
            mis.append(np.argsort(d, kind="stable")[:int(len(d) * flags.disparity_ratio)])
        works = set(mis[0].tolist())
        for m in mis[1:]:
            works &= set(m.tolist())
        q = []
        for j in sorted(works):
            e = ens[j][col]
            if e <= 0.5 - flags.eps or e >= 0.5 + flags.eps:
                ACTIVE += 1
                if (e >= 0.5) == (batch_y[j][0] >= 0.5):
                    q.append(j)
        sets.append(list(q))
        if not q:
            continue
        inter = np.array([ens[i][col] for i in q])
        mp = int(len(inter) / float(len(nz)))
        parts_l = [noisy[0][i][q[i * mp:(i + 1) * mp]] for i in range(len(nz))]
        parts_r = [noisy[1][i][q[i * mp:(i + 1) * mp]] for i in range(len(nz))]
        parts_y = [helpers.roundoff(inter)[i * mp:(i + 1) * mp] for i in range(len(nz))]
        tl = np.concatenate(([tl] if ty.shape[0] > 0 else []) + parts_l)
        tr = np.concatenate(([tr] if ty.shape[0] > 0 else []) + parts_r)
        ty = np.concatenate(([ty] if ty.shape[0] > 0 else []) + parts_y)
        if ty.shape[0] >= flags.batch_send:
            (ol, orr), oy = next(dataGen)
            for _ in range(flags.mixture_ratio - 1):
                t, yy = next(dataGen)
                ol, orr, oy = np.concatenate((ol, t[0])), np.concatenate((orr, t[1])), np.concatenate((oy, yy))
            tl = np.concatenate((tl, feats[0][q], ol))
            tr = np.concatenate((tr, feats[1][q], orr))
            ty = np.concatenate((ty, helpers.roundoff(inter), oy))
            student.finetune([tl, tr], ty, flags.ft_epochs, 16, 0)
            fts += 1
            tl, tr, ty = np.array([]), np.array([]), np.array([])
        if int(flags.active_ratio * UN) <= ACTIVE:
            break
    return ACTIVE, UN, sets, fts


@pytest.mark.parametrize("col", [0, 1])
def test_loop_equals_reference_shaped_loop(gpu, col, tmp_path):
    """the library loop with DEFAULT flags (every embedding exact: the reference's semantics, Flags.screen_settle = False since
    round 6) against a literal restatement of code/ALINK_arc.py:142-254"""
    from a_link_amd import alink_loop as AL, pairs
    flags = AL.Flags(alink_bs=3, batch_send=6, disparity_ratio=0.6, eps=0.0005, ft_epochs=2, mixture_ratio=2,
                     out_model=str(tmp_path / "post"))
    assert flags.screen_settle is False
    X_plain, X_dig = _people(6, 1), _people(6, 2)
    results = []
    for which in ("library", "literal"):
        conv, student, ens, bag, nz = _build(7)
        feats_plain = [conv.process(p) for p in X_plain]
        gen = pairs.getGenerator(pairs.getNormalGenerator(feats_plain, 8), pairs.getNormalGenerator(feats_plain, 8),
                                 pairs.getImposterGenerator(feats_plain, feats_plain, 8), 8)
        np.random.seed(5)                                      # balanced sampling + fit() shuffles
        if which == "library":
            # (a default-built ArcFace carries a screening form, but the loop only uses it under flags.screen_settle: the spy on
            # select_queries_settled must stay silent here)
            from a_link_amd import settle
            sets = []
            orig, orig_s = AL.selection.select_queries, settle.select_queries_settled

            def spy(*a, **k):
                r = orig(*a, **k)
                sets.append(list(r[0]))
                return r

            def spy_s(*a, **k):
                r = orig_s(*a, **k)
                sets.append(list(r[0]))
                return r
            AL.selection.select_queries, settle.select_queries_settled = spy, spy_s
            try:
                st = AL.run_alink_dfw(flags, conv, bag, nz, student, X_plain, X_dig, gen, SIZE, col=col, verbose=0)
            finally:
                AL.selection.select_queries, settle.select_queries_settled = orig, orig_s
            assert not st.settle_info, "default flags must take the all-exact path"
            results.append((st.active_count, st.un_size, sets, st.finetunes, student.siamese_net.get_weights()))
            assert (tmp_path / "post.h5").exists()
        else:
            a, u, sets, fts = _literal_loop(flags, conv, bag, nz, student, X_plain, X_dig, gen, col)
            results.append((a, u, sets, fts, student.siamese_net.get_weights()))
    lib, lit = results
    assert lib[0] == lit[0] and lib[1] == lit[1] and lib[3] == lit[3]
    assert lib[2] == lit[2]
    assert lib[3] >= 1, "test data must trigger at least one fine-tune"
    for a, b in zip(lib[4], lit[4]):
        assert np.array_equal(a, b)


def test_mtp_loop_runs_and_trains_smallres(gpu, tmp_path):
    """ALINK_MTP shape (config 1 scaled): high-res teacher features, SmallRes student on low-res pixels."""
    from a_link_amd import alink_loop as AL, committee, noise, pairs, siamese
    conv = siamese.ArcFace(SIZE, "synthetic:r18:3")                      # stands in for RESNET50(224) (test_gpu_resnet50 covers it)
    low = (16, 16)
    student = siamese.SmallRes(low + (3,), (64,), str(tmp_path / "lowres"), 0.1, seed=2)
    ens = [siamese.SiameseNetwork((512,), "e%d" % i, 0.1, seed=50 + i) for i in range(2)]
    nz = [noise.Gaussian(seed=1), noise.Noise()]
    bag = committee.Bagging(ens, nz)
    rng = np.random.RandomState(3)
    people = [rng.randint(0, 256, (2, 40, 40, 3)).astype(np.float32) for _ in range(6)]
    gen = pairs.getGeneratorMTP(pairs.getNormalGenerator(people, 16), 8, resize_res=low)
    flags = AL.Flags(alink_bs=3, batch_send=4, disparity_ratio=1.0, eps=0.0, ft_epochs=1, active_ratio=2.0,
                     out_model=str(tmp_path / "post"))
    w0 = student.siamese_net.get_weights()
    np.random.seed(0)
    st = AL.run_alink_mtp(flags, conv, bag, nz, student, people, gen, SIZE, low, verbose=0)
    assert st.iterations == 2 and st.un_size == 72 and st.active_count > 0
    assert st.finetunes >= 1 and any(not np.array_equal(a, b) for a, b in zip(w0, student.siamese_net.get_weights()))
    test_people = [np.asarray(noise.resize_images(p, low)) for p in people]
    acc = AL.top1_identification(student, test_people)
    assert 0.0 <= acc <= 1.0
    # the batched form against the reference's one-probe-per-call loop (code/ALINK_MTP.py:274-289), argmax quirk included
    gal = np.array([x[0] for x in test_people])
    hits = total = 0
    for i, person in enumerate(test_people):
        for x in person:
            sc = np.squeeze(student.predict([np.repeat(x[None], len(gal), axis=0), gal]))
            hits += int(np.argmax(sc) == i)
            total += 1
    assert acc == hits / float(total) and AL.top1_identification(student, test_people, chunk_pairs=7) == acc


def test_flags_match_reference_defaults():
    import argparse
    from a_link_amd import alink_loop as AL
    f = AL.add_flags(argparse.ArgumentParser()).parse_args([])
    assert (f.ft_epochs, f.batch_size, f.batch_send, f.mixture_ratio, f.alink_bs) == (3, 16, 64, 2, 16)
    assert (f.active_ratio, f.split_ratio, f.disparity_ratio, f.eps) == (1.0, 0.5, 0.25, 0.05)
    assert f.noise == 'gaussian,saltpepper,poisson,perlin,speckle,adversarial' and f.blind_strategy is False
    with pytest.raises(AttributeError):
        AL.Flags(nope=1)


@pytest.mark.parametrize("screen", ["bf16", "auto"])
def test_loop_with_screen_then_settle_equals_all_exact_loop(gpu, capsys, screen, tmp_path):
    """run_alink_dfw with a feature model that carries a 16-bit screening handle (ArcFace(screen_dtype=...)): the noisy
    pair occurrences — the bulk of an iteration's embeddings — go through the screening mode and only pairs near a cut
    (and the selected ones) through the exact mode.  Oracle-query count, every iteration's query list, the number of
    fine-tunes and the student's weights afterwards must equal the all-exact loop's, bit for bit."""
    from a_link_amd import alink_loop as AL, pairs, settle
    X_plain, X_dig = _people(6, 1), _people(6, 2)
    results = []
    for use in (True, False):
        flags = AL.Flags(alink_bs=3, batch_send=6, disparity_ratio=0.6, eps=0.0005, ft_epochs=2, mixture_ratio=2,
                         out_model=str(tmp_path / "post"), screen_settle=use)
        from a_link_amd import committee, noise, siamese
        conv = siamese.ArcFace(SIZE, "synthetic:r18:3", screen_dtype=screen)
        assert conv.backbones()[0] is not None and conv.backbones()[0].dtype in ("bf16", "f16")
        student = siamese.SiameseNetwork((512,), "student", 0.1, seed=7)
        ens = [siamese.SiameseNetwork((512,), "ens%d" % i, 0.1, seed=100 + i) for i in range(2)]
        nz = [noise.get_relevant_noise(n)(model=student, sess=None, feature_model=conv) for n in ("gaussian", "speckle")]
        for i, z in enumerate(nz):
            z._seed, z._calls = 1000 + i, 0
        bag = committee.Bagging(ens, nz)
        feats_plain = [conv.process(p) for p in X_plain]
        gen = pairs.getGenerator(pairs.getNormalGenerator(feats_plain, 8), pairs.getNormalGenerator(feats_plain, 8),
                                 pairs.getImposterGenerator(feats_plain, feats_plain, 8), 8)
        np.random.seed(5)
        sets = []
        o1, o2 = AL.selection.select_queries, settle.select_queries_settled
        AL.selection.select_queries = lambda *a, **k: (lambda r: (sets.append(list(r[0])), r)[1])(o1(*a, **k))
        settle.select_queries_settled = lambda *a, **k: (lambda r: (sets.append(list(r[0])), r)[1])(o2(*a, **k))
        try:
            st = AL.run_alink_dfw(flags, conv, bag, nz, student, X_plain, X_dig, gen, SIZE, col=0, verbose=0)
        finally:
            AL.selection.select_queries, settle.select_queries_settled = o1, o2
        results.append((st.active_count, st.un_size, sets, st.finetunes, student.siamese_net.get_weights(), st.settle_info))
    a, b = results
    assert len(a[5]) == len(a[2]) >= 2 and not b[5]
    with capsys.disabled():
        print("\n[loop, screening %s] (pair, noise) rows settled per iteration: %s" % (screen, ["%.0f %%" % (100 * i["fraction_settled"]) for i in a[5]]))
    assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2] and a[3] == b[3] >= 1
    for x, y in zip(a[4], b[4]):
        assert np.array_equal(x, y)


def test_mtp_generator_transforms_its_table_once(gpu):
    """readMTP.getGenerator (code/readMTP.py:80-113) resizes (and featurizes) every source batch of every round; over this
    package's own source the table is transformed ONCE and rows are gathered from it (both are per-image maps).  The batches
    must equal, bit for bit, the round-by-round transform of the same source behind an opaque generator — resize alone, and
    resize + a feature model."""
    from a_link_amd import pairs, siamese
    rng = np.random.RandomState(0)
    people = [rng.randint(0, 256, (rng.randint(2, 5), 40, 40, 3)).astype(np.float32) for _ in range(9)]
    fm = siamese.ArcFace((32, 32), "synthetic:r18:3", dtype="bf16", screen_dtype=None)

    def foreign(g):
        while True:
            yield next(g)
    for kw in (dict(resize_res=(32, 32)), dict(resize_res=(32, 32), featurize=fm)):
        own = pairs.getGeneratorMTP(pairs.getNormalGenerator(people, 16), 8, **kw)
        leg = pairs.getGeneratorMTP(foreign(pairs.getNormalGenerator(people, 16)), 8, **kw)
        assert own.indexable and not leg.indexable
        np.random.seed(4)
        a = [next(own) for _ in range(6)]
        np.random.seed(4)
        b = [next(leg) for _ in range(6)]
        for (xa, ya), (xb, yb) in zip(a, b):
            assert np.array_equal(ya, yb)
            for s in (0, 1):
                assert np.asarray(xa[s]).shape == np.asarray(xb[s]).shape and np.array_equal(np.asarray(xa[s]), np.asarray(xb[s])), kw.keys()
