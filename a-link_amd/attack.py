"""attack — drop-in for reference code/attack.py: the black-box few-pixel attack of A2-LINK.

    perturb_image(xs, img)                                   code/attack.py:5-29
    PixelAttacker(model).predict_classes / attack_success / attack / attack_all   code/attack.py:32-103

The search is the reference's population-batched differential evolution (differential_evolution.py).
What the reference does per generation — write k pixels into `popsize` copies of the stacked pair
image on the host, slice every copy in two, embed 2 x popsize images ONE AT A TIME through MXNet and
score the pairs through Keras (code/noise.py:158-168, code/siamese.py:232-234) — is here one
perturb kernel that writes both halves as two contiguous device batches (alink_perturb_images,
split = 1), two batched backbone launch chains and one fused pair-scoring kernel; only the
population (popsize x 5k doubles) goes up and popsize energies come back.

The fast path needs `model` to be a noise.PredictionWrappedModel over an ArcFace-style feature model
(`.model.model.embed_device`) and a DenseHead-backed pair model; any other duck-typed model takes the
reference's generic route (perturb on device, hand host arrays to model.predict).

attack_all advances the searches of a pair batch in LOCK-STEP (_LockstepEngine): 400 x K images per launch instead of
400, the success test read off the scores the generation already produced, the solvers' host bookkeeping hidden
behind the other lane's launch.  Per-pair results are the same bit for bit as the one-after-another form.
"""
import numpy as np

from . import _abi
from .differential_evolution import DifferentialEvolutionSolver, differential_evolution


def _perturb_device(xs, img_dev, split, device=None):
    """xs (n, 5k) float64 host, img_dev (Hc, W, 3) f32 device -> device tensor
    (n, Hc, W, 3) or, split, (2, n, Hc/2, W, 3)."""
    import torch
    device = _abi.resolve_device(device)
    lib = _abi.init(device)
    xs = np.ascontiguousarray(np.asarray(xs, dtype=np.float64))
    n, m = xs.shape
    if m % 5:
        raise ValueError("array split does not result in an equal division")       # np.split (code/attack.py:22)
    k = m // 5
    Hc, W, Cc = img_dev.shape
    assert Cc == 3
    xs_d = torch.from_numpy(xs).to(img_dev.device)
    shape = (2, n, Hc // 2, W, 3) if split else (n, Hc, W, 3)
    out = torch.empty(shape, dtype=torch.float32, device=img_dev.device)
    _abi.check(lib.alink_perturb_images(_abi.ptr(img_dev), _abi.ptr(xs_d), n, k, Hc, W, 1 if split else 0,
                                        _abi.ptr(out), _abi.current_stream()), "alink_perturb_images")
    return out


def perturb_image(xs, img, device=None):
    """n perturbation vectors [x, y, r, g, b] * k -> n perturbed copies of img (code/attack.py:5-29).
    NumPy in -> NumPy out (float32); CUDA tensor image in -> CUDA tensor out."""
    import torch
    xs = np.asarray(xs)
    if xs.ndim < 2:
        xs = np.array([xs])
    as_torch = isinstance(img, torch.Tensor)
    device = _abi.resolve_device(device)
    img_d = img.to("cuda:%d" % device, torch.float32).contiguous() if as_torch else \
        torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32)).to("cuda:%d" % device)
    r = xs.astype(int)[:, 0::5] if xs.shape[1] >= 5 else np.zeros((len(xs), 0), int)
    c = xs.astype(int)[:, 1::5] if xs.shape[1] >= 5 else np.zeros((len(xs), 0), int)
    if r.size and (r.min() < -img_d.shape[0] or r.max() >= img_d.shape[0] or c.min() < -img_d.shape[1]
                   or c.max() >= img_d.shape[1]):
        raise IndexError("perturbation pixel outside the %d x %d image" % (img_d.shape[0], img_d.shape[1]))
    out = _perturb_device(xs, img_d, False, device)
    return out if as_torch else out.cpu().numpy()


def _device_parts(wrapped, search):
    """(backbone, DenseHead) of the fused objective, or TypeError when `wrapped` has no device fast path"""
    from .head import DenseHead
    fm = getattr(wrapped, "feature_model", None)
    bb = getattr(getattr(fm, "model", None), "model", None)           # ArcFace -> FaceModel -> IRBackbone
    if search == "screen" and getattr(fm, "screen", None) is not None:
        bb = fm.screen.model
    elif search == "bf16" and hasattr(fm, "search_handle"):
        bb = fm.search_handle("bf16").model
    elif search not in ("exact", "screen", "bf16"):
        raise ValueError("search must be exact, screen or bf16")
    head = getattr(getattr(wrapped, "model", None), "siamese_net", None)
    if bb is None or not hasattr(bb, "embed_device") or not isinstance(head, DenseHead):
        raise TypeError("no device fast path for this model")
    if not getattr(wrapped.model, "_identity_preprocess", False):
        raise TypeError("pair model preprocesses its inputs")
    return bb, head


class _DevicePairScorer(object):
    """The fused objective: population -> P(class) of the perturbed pair, all on device.
    search="screen": the candidates of the search are embedded by the feature model's 16-bit SCREENING form where it has one
    (ArcFace.screen: 44 k embeddings/s against the exact mode's 15 k — the search is 20,400 backbone forwards per pair, 1.4 s in
    the exact mode).  The reference's search is a random one (differential evolution, unseeded: code/attack.py:81-83), so which
    arithmetic ranks its candidates is not contractual; the image it RETURNS is embedded like any other by whoever calls the
    attack.  search="bf16": a bfloat16 handle on the same checkpoint (ArcFace.search_handle), the fastest form.  Default "exact"."""

    def __init__(self, wrapped, image, device=None, search="exact"):
        import torch
        self.bb, self.head = _device_parts(wrapped, search)
        bb = self.bb
        image = np.asarray(image, dtype=np.float32)
        Hc, W, _ = image.shape
        if Hc % 2 or (Hc // 2, W) != tuple(bb.image_size):
            raise TypeError("stacked pair image %s does not match the backbone input" % (image.shape,))
        device = _abi.resolve_device(device)
        self.img = torch.from_numpy(np.ascontiguousarray(image)).to("cuda:%d" % device)
        self.device = device

    def predict(self, xs):
        halves = _perturb_device(xs, self.img, True, self.device)
        n = halves.shape[1]
        emb = self.bb.embed_device(halves.reshape(2 * n, *halves.shape[2:]))
        return self.head.predict_device(emb[:n], emb[n:]).cpu().numpy()


class _Lane(object):
    """Buffers of one launch of the lock-step search: the candidates of up to `slots` searches (group rows each)."""

    def __init__(self, eng, slots):
        import torch
        cap, dev = slots * eng.group, eng.dev
        self.slots = slots
        self.xs_host = torch.empty((cap, 5 * eng.k), dtype=torch.float64).pin_memory()
        self.xs_np = self.xs_host.numpy()
        self.xs_dev = torch.empty((cap, 5 * eng.k), dtype=torch.float64, device=dev)
        self.of_host = torch.empty(slots, dtype=torch.int32).pin_memory()
        self.of_np = self.of_host.numpy()
        self.of_dev = torch.empty(slots, dtype=torch.int32, device=dev)
        self.halves = torch.empty((2 * cap, eng.Hc // 2, eng.W, 3), dtype=torch.float32, device=dev)
        self.emb = torch.empty((2 * cap, eng.bb.emb), dtype=torch.float32, device=dev)
        self.pred = torch.empty((cap, eng.head.out_dim), dtype=torch.float32, device=dev)
        self.pred_host = torch.empty((cap, eng.head.out_dim), dtype=torch.float32).pin_memory()
        self.pred_np = self.pred_host.numpy()
        self.done = torch.cuda.Event()
        self.active = []                         # the searches whose candidates are in flight, in row order


class _Search(object):
    __slots__ = ("idx", "solver", "target_class", "targeted", "minimize", "nit")


class _LockstepEngine(object):
    """K independent differential-evolution searches advanced together (PixelAttacker.attack_all, code/attack.py:91-103).

    The reference attacks the pairs of a batch one after another, and so did this package until round 5: a 400-image launch
    per generation, a host synchronisation, the solver's bookkeeping, then a 2-image forward for the success test — 13.7 ms per
    generation where the launch itself is 9 ms.  Searches are independent (each has its own random stream), so here
      * the candidates of all the searches of a LANE go through one perturb launch, one backbone launch chain of
        2 x group x S images and one pair-scoring launch; two lanes are in flight, so the host-side bookkeeping of one
        (tell / ask of S solvers) runs while the device works on the other;
      * the success test (code/attack.py:47-63: argmax of the best member's two class scores) costs no forward at all: the
        best member IS one of the candidates just scored, and its score row rides along with it through the solver's
        selection (DifferentialEvolutionSolver.tell(aux=)).  An image embeds to the same bits whatever batch it arrives in,
        so the row equals what the reference's extra predict() would return;
      * a search that stops (success, or maxiter) frees its place in the lane for the next pair.
    Every search sees exactly the energies it would see alone, so the attacked images do not depend on K
    (tests/test_gpu_noise.py::test_lockstep_attack_equals_the_sequential_attack)."""

    def __init__(self, wrapped, images, search, device=None):
        import torch
        self.torch = torch
        self.bb, self.head = _device_parts(wrapped, search)
        self.device = _abi.resolve_device(device)
        self.dev = "cuda:%d" % self.device
        self.lib = _abi.init(self.device)
        first = images[0]
        self.Hc, self.W = int(first.shape[0]), int(first.shape[1])
        if len(first.shape) != 3 or first.shape[2] != 3 or self.Hc % 2 or (self.Hc // 2, self.W) != tuple(self.bb.image_size):
            raise TypeError("stacked pair image %s does not match the backbone input" % (tuple(first.shape),))
        # the stacked pair images of the whole call stay resident (301 KB each at 112 x 112: 3,840 pairs = 1.2 GB of 288)
        if isinstance(images, torch.Tensor):                     # (n, 2H, W, 3) already on a device: no trip through the host
            self.imgs = images.to(self.dev, torch.float32).contiguous()
        else:
            host = np.ascontiguousarray(np.stack([np.asarray(im, dtype=np.float32) for im in images]))
            self.imgs = torch.from_numpy(host).to(self.dev)
        self.ub = getattr(self.bb, "bb", self.bb)            # the IRBackbone under a one-product screening view

    # -- one launch ----------------------------------------------------------------------------------------------------
    def _enqueue(self, lane):
        """the device side of a step: candidates up, perturb, embed both halves, score, scores down (all asynchronous)"""
        S = len(lane.active)
        n = S * self.group
        st = _abi.current_stream(self.device)
        lane.xs_dev[:n].copy_(lane.xs_host[:n], non_blocking=True)
        lane.of_dev[:S].copy_(lane.of_host[:S], non_blocking=True)
        _abi.check(self.lib.alink_perturb_images_multi(_abi.ptr(self.imgs), _abi.ptr(lane.of_dev), self.group, _abi.ptr(lane.xs_dev),
                                                       n, self.k, self.Hc, self.W, 1, _abi.ptr(lane.halves), st),
                   "alink_perturb_images_multi")
        self.bb.embed_device(lane.halves[:2 * n], out=lane.emb[:2 * n])
        self.head.predict_device(lane.emb[:n], lane.emb[n:2 * n], out=lane.pred[:n])
        lane.pred_host[:n].copy_(lane.pred[:n], non_blocking=True)
        lane.done.record(self.torch.cuda.current_stream(self.device))

    def _launch(self, lane):
        for j, s in enumerate(lane.active):
            xs = s.solver.ask()
            if xs.shape != (self.group, 5 * self.k):
                raise RuntimeError("lock-step search: a solver asked for %s candidates, not %d" % (xs.shape, self.group))
            lane.xs_np[j * self.group:(j + 1) * self.group] = xs
            lane.of_np[j] = s.idx
        try:
            self._enqueue(lane)
        except _abi.AlinkError:
            # the other lane's forward raised the range flag while this one was being enqueued (the backbone reads it at
            # the start of a call when its check is deferred)
            if self.ub.dtype != "f16x2":
                raise
            self._recalibrate()

    def _range_left(self):
        """16-bit storage only: did a forward of the lanes in flight leave the float16 range?  (The backbone checks this
        itself after every call by synchronising the device — here the flag is read once per step, after the step's
        own event, so that the other lane keeps the device busy.)  Split precision re-calibrates on the images in flight
        (scales only go down), as IRBackbone._checked does, and the step is run again; plain f16 raises."""
        ub = self.ub
        if ub.dtype in ("f16", "f16x2") and ub.range_left(reset=False):
            if ub.dtype != "f16x2":
                self.torch.cuda.synchronize(self.device)
                ub.range_left()
                raise _abi.AlinkError("activations exceeded the float16 range in this network: build the backbone with dtype='bf16'")
            self._recalibrate()

    def _recalibrate(self):
        ub = self.ub
        self.torch.cuda.synchronize(self.device)
        ub.range_left()
        for lane in self.lanes:
            if lane.active:
                ub.calibrate(lane.halves[:2 * len(lane.active) * self.group], merge=True)
        for lane in self.lanes:
            if lane.active:
                self._enqueue(lane)
        self.torch.cuda.synchronize(self.device)
        if ub.range_left():
            raise _abi.AlinkError("activations exceeded the float16 range again after re-calibration")

    # -- the whole call ----------------------------------------------------------------------------------------------------
    def run(self, make_solver, targets, maxiter, lockstep, verbose=False, early_stop=True):
        """make_solver(i) -> DifferentialEvolutionSolver of pair i; targets[i] = (target_class, targeted, minimize).
        Returns the list of OptimizeResult, one per pair."""
        n_pairs = self.imgs.shape[0]
        results = [None] * n_pairs
        probe = make_solver(0)
        self.group, self.k = int(probe.num_population_members), probe.parameter_count // 5
        n_lanes = 2 if lockstep >= 2 and n_pairs >= 2 else 1
        per_lane = max(1, min(lockstep // n_lanes, (n_pairs + n_lanes - 1) // n_lanes))
        lanes = self.lanes = [_Lane(self, per_lane) for _ in range(n_lanes)]
        nxt = [0]

        def refill(lane):
            while len(lane.active) < lane.slots and nxt[0] < n_pairs:
                i = nxt[0]
                nxt[0] += 1
                s = _Search()
                s.idx, s.solver, s.nit = i, (probe if i == 0 else make_solver(i)), 0
                s.target_class, s.targeted, s.minimize = targets[i]
                lane.active.append(s)

        ub = self.ub
        saved_lazy = getattr(ub, "lazy_range_check", False)
        if ub.dtype in ("f16", "f16x2"):
            self.torch.cuda.synchronize(self.device)
            ub.range_left()
            ub.lazy_range_check = True
        try:
            for lane in lanes:
                refill(lane)
                if lane.active:
                    self._launch(lane)
            turn = 0
            while any(lane.active for lane in lanes):
                lane = lanes[turn % n_lanes]
                turn += 1
                if not lane.active:
                    continue
                lane.done.synchronize()
                self._range_left()
                keep = []
                for j, s in enumerate(lane.active):
                    rows = lane.pred_np[j * self.group:(j + 1) * self.group]
                    p = rows[:, s.target_class]
                    was_init = s.solver._pending[0] == "init"
                    s.solver.tell(p if s.minimize else 1 - p, aux=rows)
                    stop = None
                    if not was_init:
                        s.nit += 1
                        # attack_success (code/attack.py:47-63) on the best member's own score row
                        says = _success(s.solver.aux[0], s.target_class, s.targeted, verbose) if early_stop else None
                        stop = s.solver.after_generation(callback_says=says)
                    if stop is None and s.nit >= maxiter:
                        stop = "maxiter"
                    if stop:
                        results[s.idx] = s.solver.result(s.nit, stop)
                    else:
                        keep.append(s)
                lane.active = keep
                refill(lane)
                if lane.active:
                    self._launch(lane)
        finally:
            ub.lazy_range_check = saved_lazy
        return results

    def attacked_images(self, results, as_device=False):
        """perturb_image(result.x, image)[0] for every pair: one launch, one copy back (or none: as_device)"""
        torch = self.torch
        n = len(results)
        xs = torch.from_numpy(np.ascontiguousarray(np.stack([np.asarray(r.x, dtype=np.float64) for r in results]))).to(self.dev)
        of = torch.arange(n, dtype=torch.int32, device=self.dev)
        out = torch.empty((n, self.Hc, self.W, 3), dtype=torch.float32, device=self.dev)
        _abi.check(self.lib.alink_perturb_images_multi(_abi.ptr(self.imgs), _abi.ptr(of), 1, _abi.ptr(xs), n, self.k, self.Hc, self.W, 0,
                                                       _abi.ptr(out), _abi.current_stream(self.device)), "alink_perturb_images_multi")
        return out if as_device else out.cpu().numpy()


def _success(confidence, target_class, targeted_attack, verbose=False):
    """the test of PixelAttacker.attack_success (code/attack.py:47-63) on a score row"""
    predicted_class = np.argmax(confidence)
    if verbose:
        print('Confidence:', confidence[target_class])
    if ((targeted_attack and predicted_class == target_class) or
            (not targeted_attack and predicted_class != target_class)):
        return True


class PixelAttacker:
    def __init__(self, model, rng_compat=False, seed=None, search="exact", lockstep=32):
        self.model = model
        self.rng_compat = rng_compat
        self.seed = seed
        self.search = search                      # "exact" | "screen": _DevicePairScorer
        self.lockstep = int(lockstep)             # searches attack_all advances together (0: one pair after another)

    def _scorer(self, img):
        try:
            return _DevicePairScorer(self.model, img, search=self.search)
        except TypeError:
            return None

    def predict_classes(self, xs, img, target_class, minimize=True, _scorer=None):
        if _scorer is not None:
            predictions = _scorer.predict(np.atleast_2d(xs))[:, target_class]
        else:
            imgs_perturbed = perturb_image(xs, img)
            predictions = self.model.predict(imgs_perturbed)[:, target_class]
        return predictions if minimize else 1 - predictions

    def attack_success(self, x, img, target_class, targeted_attack=False, verbose=False, _scorer=None):
        if _scorer is not None:
            confidence = _scorer.predict(np.atleast_2d(x))[0]
        else:
            attack_image = perturb_image(x, img)
            confidence = self.model.predict(attack_image)[0]
        return _success(confidence, target_class, targeted_attack, verbose)

    def attack(self, image, actual_class, target, pixel_count, dimensions, maxiter=75, popsize=400, verbose=False,
               seed=None):
        """seed (not in the reference, which seeds nothing: code/attack.py:81-83): this search's own random stream,
        instead of the attacker's `self.seed` — what makes a pair's search independent of the pairs attacked before it"""
        seed = self.seed if seed is None else seed
        targeted_attack = target is not None
        target_class = target if targeted_attack else actual_class
        dim_x, dim_y = dimensions
        bounds = [(0, dim_x), (0, dim_y), (0, 256), (0, 256), (0, 256)] * pixel_count
        popmul = max(1, popsize // len(bounds))
        scorer = self._scorer(image)

        def predict_fn(xs):
            return self.predict_classes(xs, image, target_class, target is None, _scorer=scorer)

        def callback_fn(x, convergence):
            return self.attack_success(x, image, target_class, targeted_attack, verbose, _scorer=scorer)

        attack_result = differential_evolution(predict_fn, bounds, maxiter=maxiter, popsize=popmul, recombination=1,
                                               atol=-1, callback=callback_fn, polish=False, seed=seed,
                                               rng_compat=self.rng_compat)
        self.last_result = attack_result
        attack_image = perturb_image(attack_result.x, image)[0]
        return attack_image

    def _solver(self, target_class, pixel_count, dimensions, maxiter, popsize, seed):
        """the solver attack() runs (code/attack.py:65-83), without objective or callback: the lock-step engine feeds it"""
        dim_x, dim_y = dimensions
        bounds = [(0, dim_x), (0, dim_y), (0, 256), (0, 256), (0, 256)] * pixel_count
        popmul = max(1, popsize // len(bounds))
        return DifferentialEvolutionSolver(None, bounds, maxiter=maxiter, popsize=popmul, recombination=1, atol=-1, polish=False,
                                           seed=seed, rng_compat=self.rng_compat)

    def attack_all(self, input_data, targets, dimensions, pixel_count=40, maxiter=50, popsize=250, verbose=False,
                   seeds=None, lockstep=None, early_stop=True):
        """code/attack.py:91-103.  Not in the reference's signature:
        seeds (one per image): see attack() — a rank that attacks rows lo : hi of a pair batch with the seeds of those rows
            finds what the whole-batch call finds for them;
        lockstep: how many searches advance together on the device (_LockstepEngine; default: this attacker's `lockstep`,
            32).  0 = the reference's shape, one pair after another.  With `seeds` the attacked images are the same bit for bit
            whatever the value; without, the lock-step form first draws one 32-bit seed per pair from this attacker's stream
            (the sequential form lets the searches share the stream — the reference seeds nothing, only distributions are
            contractual);
        early_stop=False: run every search to maxiter (a timing aid: the cost of an attack that does not succeed)."""
        lockstep = self.lockstep if lockstep is None else int(lockstep)
        n = len(input_data)
        eng = None
        if lockstep > 0 and n > 0:
            try:
                eng = _LockstepEngine(self.model, input_data, self.search)
            except TypeError:
                eng = None                                 # no device fast path for this model: the generic route below
        if eng is None:
            if hasattr(input_data, "detach"):                    # the generic route works on host arrays
                input_data = input_data.detach().cpu().numpy()
            X = []
            for i, img in enumerate(input_data):
                target_class = np.argmax(targets[i])
                result = self.attack(img, 1 - target_class, target_class, pixel_count, dimensions, maxiter=maxiter,
                                     popsize=popsize, verbose=verbose, seed=None if seeds is None else seeds[i])
                X.append(result)
            return X
        if seeds is None:
            from .differential_evolution import _rng_of
            seeds = [int(v) for v in _rng_of(self.seed).randint(0, 2 ** 32, size=n, dtype=np.uint64)]
        tcs = [int(np.argmax(targets[i])) for i in range(n)]
        # attack(img, actual_class = 1 - target_class, target = target_class): targeted, energies 1 - P[target]
        results = eng.run(lambda i: self._solver(tcs[i], pixel_count, dimensions, maxiter, popsize, seeds[i]),
                          [(tc, True, False) for tc in tcs], maxiter, lockstep, verbose=verbose, early_stop=early_stop)
        self.last_results = results
        self.last_result = results[-1]
        if hasattr(input_data, "detach"):                        # device tensor in -> device tensor out
            return eng.attacked_images(results, as_device=True)
        return list(eng.attacked_images(results))
