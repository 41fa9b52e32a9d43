"""oracle/de.py — CPU restatement of the reference's population-batched differential evolution and of
the few-pixel attack built on it.  TEST INFRASTRUCTURE.

Follows reference code/differential_evolution.py (SciPy's solver at commit 70e61de, patched so the
objective receives the whole population: :630-645, :692-715) and code/attack.py:32-103, drawing from
the RandomState in the same order, so that a run with the same seed retraces the reference's
trajectory exactly.  Pinned by tests/golden/de.npz (tests/golden/make_golden_noise.py runs the
reference's own differential_evolution / PixelAttacker).
"""
import numpy as np

_EPS = np.finfo(np.float64).eps
_BIN = {"best1bin": "best1", "randtobest1bin": "randtobest1", "currenttobest1bin": "currenttobest1",
        "best2bin": "best2", "rand2bin": "rand2", "rand1bin": "rand1"}
_EXP = {k[:-3] + "exp": v for k, v in _BIN.items()}


class Result(dict):
    __getattr__ = dict.__getitem__


class Solver(object):
    def __init__(self, func, bounds, strategy="best1bin", maxiter=1000, popsize=15, tol=0.01, mutation=(0.5, 1),
                 recombination=0.7, seed=None, callback=None, atol=0, maxfun=np.inf):
        if strategy not in _BIN and strategy not in _EXP:
            raise ValueError("Please select a valid mutation strategy")
        self.strategy, self.func, self.callback = strategy, func, callback
        self.tol, self.atol, self.maxiter, self.maxfun = tol, atol, maxiter, maxfun
        self.scale = mutation
        self.dither = sorted([mutation[0], mutation[1]]) if hasattr(mutation, "__iter__") and len(mutation) > 1 else None
        self.cr = recombination
        lim = np.array(bounds, dtype="float").T                      # :386
        self.mid, self.span = 0.5 * (lim[0] + lim[1]), np.fabs(lim[0] - lim[1])
        self.npar = lim.shape[1]
        self.rng = seed if isinstance(seed, np.random.RandomState) else (
            np.random.mtrand._rand if seed is None else np.random.RandomState(seed))
        self.npop = max(5, popsize * self.npar)                      # :415
        self.nfev = 0
        self._init_lhs()

    # :433-469
    def _init_lhs(self):
        seg = 1.0 / self.npop
        samples = seg * self.rng.random_sample((self.npop, self.npar)) + \
            np.linspace(0., 1., self.npop, endpoint=False)[:, np.newaxis]
        self.pop = np.zeros_like(samples)
        for j in range(self.npar):
            order = self.rng.permutation(range(self.npop))
            self.pop[:, j] = samples[order, j]
        self.energies = np.ones(self.npop) * np.inf

    def to_params(self, t):                                           # :763-767
        return self.mid + (t - 0.5) * self.span

    @property
    def x(self):
        return self.to_params(self.pop[0])

    # :630-668
    def _first_energies(self):
        n = int(max(0, min(len(self.pop), self.maxfun - self.nfev + 1)))
        self.energies = self.func(np.array([self.to_params(c) for c in self.pop[:n]]))
        self.nfev += n
        b = np.argmin(self.energies)
        low = self.energies[b]
        self.energies[b] = self.energies[0]
        self.energies[0] = low
        self.pop[[0, b], :] = self.pop[[b, 0], :]

    def _samples(self, cand, k):                                      # :879-888
        idxs = list(range(self.npop))
        idxs.remove(cand)
        self.rng.shuffle(idxs)
        return idxs[:k]

    def _bprime(self, kind, cand, s):                                 # :820-877
        P, F = self.pop, self.scale
        if kind == "best1":
            return P[0] + F * (P[s[0]] - P[s[1]])
        if kind == "rand1":
            return P[s[0]] + F * (P[s[1]] - P[s[2]])
        if kind == "randtobest1":
            b = np.copy(P[s[0]])
            b += F * (P[0] - b)
            b += F * (P[s[1]] - P[s[2]])
            return b
        if kind == "currenttobest1":
            return P[cand] + F * (P[0] - P[cand] + P[s[0]] - P[s[1]])
        if kind == "best2":
            return P[0] + F * (P[s[0]] + P[s[1]] - P[s[2]] - P[s[3]])
        return P[s[0]] + F * (P[s[1]] + P[s[2]] - P[s[3]] - P[s[4]])

    def _mutate(self, cand):                                          # :782-818
        trial = np.copy(self.pop[cand])
        fill = self.rng.randint(0, self.npar)
        binom = self.strategy in _BIN
        b = self._bprime((_BIN if binom else _EXP)[self.strategy], cand, self._samples(cand, 5))
        if binom:
            cross = self.rng.rand(self.npar) < self.cr
            cross[fill] = True
            return np.where(cross, b, trial)
        i = 0
        while i < self.npar and self.rng.rand() < self.cr:
            trial[fill] = b[fill]
            fill = (fill + 1) % self.npar
            i += 1
        return trial

    def step(self):                                                   # :673-748
        if np.all(np.isinf(self.energies)):
            self._first_energies()
        if self.dither is not None:
            self.scale = self.rng.rand() * (self.dither[1] - self.dither[0]) + self.dither[0]
        n = int(max(0, min(self.npop, self.maxfun - self.nfev + 1)))
        trials = np.array([self._mutate(c) for c in range(n)])
        for t in trials:                                              # :775-780
            for i in np.where((t < 0) | (t > 1))[0]:
                t[i] = self.rng.rand()
        e = self.func(np.array([self.to_params(t) for t in trials]))
        self.nfev += n
        for c, (ec, t) in enumerate(zip(e, trials)):
            if ec < self.energies[c]:
                self.pop[c] = t
                self.energies[c] = ec
                if ec < self.energies[0]:
                    self.energies[0] = ec
                    self.pop[0] = t

    def solve(self):                                                  # :540-628 (polish=False)
        nit, stopped, msg = 0, False, "Optimization terminated successfully."
        if np.all(np.isinf(self.energies)):
            self._first_energies()
        for nit in range(1, self.maxiter + 1):
            self.step()
            conv = np.std(self.energies) / np.abs(np.mean(self.energies) + _EPS)
            if self.callback and self.callback(self.to_params(self.pop[0]), convergence=self.tol / conv) is True:
                stopped, msg = True, "callback function requested stop early by returning True"
                break
            if np.std(self.energies) <= self.atol + self.tol * np.abs(np.mean(self.energies)):
                break
        else:
            msg, stopped = "Maximum number of iterations has been exceeded.", True
        return Result(x=self.x, fun=self.energies[0], nfev=self.nfev, nit=nit, message=msg, success=not stopped)


def differential_evolution(func, bounds, **kw):
    return Solver(func, bounds, **kw).solve()


# ---------------------------------------------------------------------------------------------------
# code/attack.py:32-103
# ---------------------------------------------------------------------------------------------------
class PixelAttacker(object):
    def __init__(self, model):
        self.model = model

    def attack(self, image, actual_class, target, pixel_count, dimensions, maxiter=75, popsize=400, seed=None):
        from .noise import perturb_image
        targeted = target is not None
        tclass = target if targeted else actual_class
        dim_x, dim_y = dimensions
        bounds = [(0, dim_x), (0, dim_y), (0, 256), (0, 256), (0, 256)] * pixel_count
        popmul = max(1, popsize // len(bounds))

        def predict_fn(xs):
            p = self.model.predict(perturb_image(xs, image))[:, tclass]
            return p if target is None else 1 - p                      # :59-65 with minimize = (target is None)

        def callback_fn(x, convergence):
            conf = self.model.predict(perturb_image(x, image))[0]
            pred = np.argmax(conf)
            if (targeted and pred == tclass) or (not targeted and pred != tclass):
                return True

        res = differential_evolution(predict_fn, bounds, maxiter=maxiter, popsize=popmul, recombination=1, atol=-1,
                                     callback=callback_fn, seed=seed)
        return perturb_image(res.x, image)[0]

    def attack_all(self, input_data, targets, dimensions, pixel_count=40, maxiter=50, popsize=250, seed=None):
        out = []
        for i, img in enumerate(input_data):
            tclass = np.argmax(targets[i])
            out.append(self.attack(img, 1 - tclass, tclass, pixel_count, dimensions, maxiter=maxiter, popsize=popsize,
                                   seed=seed))
        return out
