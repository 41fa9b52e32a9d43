"""differential_evolution — drop-in for reference code/differential_evolution.py: SciPy's differential
evolution with the reference's one change — the objective receives the WHOLE population
(code/differential_evolution.py:630-645, 692-715), which is what lets the few-pixel attack evaluate
a generation as one batched backbone launch on the GPU.

    differential_evolution(func, bounds, args=(), strategy='best1bin', maxiter=1000, popsize=15,
                           tol=0.01, mutation=(0.5, 1), recombination=0.7, seed=None, callback=None,
                           disp=False, polish=True, init='latinhypercube', atol=0)

Host NumPy like the reference's (a population is 200 x 200 doubles; the cost is in `func`).  Two ways
of drawing the random numbers of a generation:
  * rng_compat=True — candidate by candidate in the reference's order (randint for the fill point, a
    shuffle of the other members' indices, the crossover uniforms; constraint repairs afterwards), so
    a seeded run retraces the reference's trajectory member for member (tests compare with golden
    trajectories recorded from the reference);
  * rng_compat=False (default) — the same distributions drawn for all candidates at once, ~50x less
    host time per generation.  The reference seeds nothing (seed=None at code/attack.py:81-83), so no
    caller can observe the difference.
"""
import numpy as np

_EPS = np.finfo(np.float64).eps
# strategy name -> (base vector kind, number of difference samples, crossover)
_STRATEGIES = {}
for _kind, _k in (("best1", 2), ("rand1", 3), ("randtobest1", 3), ("currenttobest1", 2), ("best2", 4), ("rand2", 5)):
    _STRATEGIES[_kind + "bin"] = (_kind, _k, "bin")
    _STRATEGIES[_kind + "exp"] = (_kind, _k, "exp")

MESSAGES = {"success": "Optimization terminated successfully.",
            "maxfev": "Maximum number of function evaluations has been exceeded.",
            "maxiter": "Maximum number of iterations has been exceeded.",
            "callback": "callback function requested stop early by returning True"}


class OptimizeResult(dict):
    """Attribute-style result (x, fun, nfev, nit, message, success), like scipy.optimize.OptimizeResult."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def _rng_of(seed):
    if seed is None:
        return np.random.mtrand._rand
    if isinstance(seed, np.random.RandomState):
        return seed
    return np.random.RandomState(seed)


class DifferentialEvolutionSolver(object):
    def __init__(self, func, bounds, args=(), strategy="best1bin", maxiter=1000, popsize=15, tol=0.01,
                 mutation=(0.5, 1), recombination=0.7, seed=None, maxfun=np.inf, callback=None, disp=False,
                 polish=True, init="latinhypercube", atol=0, rng_compat=False):
        if strategy not in _STRATEGIES:
            raise ValueError("Please select a valid mutation strategy")
        self.strategy, self.callback, self.polish, self.disp = strategy, callback, polish, disp
        self.tol, self.atol = tol, atol
        m = np.atleast_1d(np.asarray(mutation, dtype=float))
        if not np.all(np.isfinite(m)) or np.any(m >= 2) or np.any(m < 0):
            raise ValueError("The mutation constant must be a float in U[0, 2), or specified as a tuple(min, max)"
                             " where min < max and min, max are in U[0, 2).")
        self.scale = mutation
        self.dither = sorted(float(v) for v in m[:2]) if m.size > 1 else None
        self.cross_over_probability = recombination
        self.func, self.args = func, args
        limits = np.array(bounds, dtype="float").T
        if limits.shape[0] != 2 or not np.all(np.isfinite(limits)):
            raise ValueError("bounds should be a sequence containing real valued (min, max) pairs for each value in x")
        self.limits = limits
        self.maxiter = 1000 if maxiter is None else maxiter
        self.maxfun = np.inf if maxfun is None else maxfun
        self._centre = 0.5 * (limits[0] + limits[1])
        self._width = np.fabs(limits[0] - limits[1])
        self.parameter_count = limits.shape[1]
        self.random_number_generator = _rng_of(seed)
        self.rng_compat = bool(rng_compat)
        self.num_population_members = max(5, popsize * self.parameter_count)
        self._nfev = 0
        if isinstance(init, str):
            if init == "latinhypercube":
                self._init_lhs()
            elif init == "random":
                self.population = self.random_number_generator.random_sample(
                    (self.num_population_members, self.parameter_count))
            else:
                raise ValueError("The population initialization method must be one of 'latinhypercube' or 'random', "
                                 "or an array of shape (M, N) where N is the number of parameters and M>5")
        else:
            pop = np.asarray(init, dtype=np.float64)
            if pop.ndim != 2 or pop.shape[0] < 5 or pop.shape[1] != self.parameter_count:
                raise ValueError("The population supplied needs to have shape (M, len(x)), where M > 4.")
            self.population = np.clip((pop - self._centre) / self._width + 0.5, 0, 1)
            self.num_population_members = pop.shape[0]
        self.population_energies = np.full(self.num_population_members, np.inf)

    # latin hypercube: one stratum per member and parameter, strata permuted per parameter
    def _init_lhs(self):
        rng, n, d = self.random_number_generator, self.num_population_members, self.parameter_count
        seg = 1.0 / n
        samples = seg * rng.random_sample((n, d)) + np.linspace(0., 1., n, endpoint=False)[:, None]
        pop = np.empty_like(samples)
        for j in range(d):
            pop[:, j] = samples[rng.permutation(range(n)), j]
        self.population = pop

    # ---- scaling --------------------------------------------------------------------------------
    def _scale_parameters(self, trial):
        return self._centre + (trial - 0.5) * self._width

    def _unscale_parameters(self, parameters):
        return (parameters - self._centre) / self._width + 0.5

    @property
    def x(self):
        return self._scale_parameters(self.population[0])

    @property
    def convergence(self):
        return np.std(self.population_energies) / np.abs(np.mean(self.population_energies) + _EPS)

    # ---- energies ---------------------------------------------------------------------------------
    def _budget(self, n):
        return int(max(0, min(n, self.maxfun - self._nfev + 1)))

    # ---- ask / tell: the two halves of an evaluation --------------------------------------------------
    # The reference's loop is func(whole population) inside the solver (code/differential_evolution.py:630-645, 692-715).
    # ask() returns the scaled candidates the solver wants evaluated next — the initial population first, then one
    # generation's trials — and tell() takes their energies and does the bookkeeping.  __next__ / solve() are
    # tell(func(ask())); attack.PixelAttacker.attack_all advances K solvers with ONE batched objective launch per step
    # through the same two calls, so a search's trajectory does not depend on how many searches share the launch.
    # aux (optional, one row per candidate): carried along with the members exactly like their parameters — aux[0] is
    # always the row that belongs to the best member (what the attack's success test needs: the member's two class scores).
    def ask(self):
        if np.all(np.isinf(self.population_energies)):
            n = self._budget(len(self.population))
            self._pending = ("init", n, None)
            return self._scale_parameters(self.population[:n])
        if self.dither is not None:
            self.scale = self.random_number_generator.rand() * (self.dither[1] - self.dither[0]) + self.dither[0]
        n = self._budget(self.num_population_members)
        trials = self._trials_compat(n) if self.rng_compat else self._trials_fast(n)
        self._pending = ("gen", n, trials)
        return self._scale_parameters(trials)

    def tell(self, energies, aux=None):
        kind, n, trials = self._pending
        self._pending = None
        self._nfev += n
        if kind == "init":
            self.population_energies = np.array(energies)
            best = np.argmin(self.population_energies)
            e = self.population_energies
            e[0], e[best] = e[best], e[0]
            self.population[[0, best], :] = self.population[[best, 0], :]
            if aux is not None:
                self.aux = np.array(aux)
                self.aux[[0, best]] = self.aux[[best, 0]]
            return
        energies = np.asarray(energies)
        # member-wise greedy selection; position 0 also tracks the best trial seen so far in this sweep
        e = self.population_energies
        for c in np.nonzero(energies < e[:n])[0]:
            if energies[c] < e[c]:                     # e[0] may have dropped since the vector compare
                self.population[c] = trials[c]
                e[c] = energies[c]
                if aux is not None:
                    self.aux[c] = aux[c]
                if energies[c] < e[0]:
                    e[0] = energies[c]
                    self.population[0] = trials[c]
                    if aux is not None:
                        self.aux[0] = aux[c]

    def _calculate_population_energies(self):
        self.tell(self.func(self.ask(), *self.args))

    # ---- one generation ---------------------------------------------------------------------------
    def _base_and_diff(self, kind, cand, s):
        """bprime for candidates `cand` (array) with sample index matrix s (len(cand), k)."""
        P, F = self.population, self.scale
        if kind == "best1":
            return P[0] + F * (P[s[:, 0]] - P[s[:, 1]])
        if kind == "rand1":
            return P[s[:, 0]] + F * (P[s[:, 1]] - P[s[:, 2]])
        if kind == "randtobest1":
            b = np.copy(P[s[:, 0]])
            b += F * (P[0] - b)
            b += F * (P[s[:, 1]] - P[s[:, 2]])
            return b
        if kind == "currenttobest1":
            return P[cand] + F * (P[0] - P[cand] + P[s[:, 0]] - P[s[:, 1]])
        if kind == "best2":
            return P[0] + F * (P[s[:, 0]] + P[s[:, 1]] - P[s[:, 2]] - P[s[:, 3]])
        return P[s[:, 0]] + F * (P[s[:, 1]] + P[s[:, 2]] - P[s[:, 3]] - P[s[:, 4]])

    def _trials_compat(self, n):
        rng, d, npop = self.random_number_generator, self.parameter_count, self.num_population_members
        kind, _, xover = _STRATEGIES[self.strategy]
        trials = np.empty((n, d))
        for c in range(n):
            fill = rng.randint(0, d)
            others = list(range(npop))
            others.remove(c)
            rng.shuffle(others)
            s = np.asarray(others[:5])[None, :]
            b = self._base_and_diff(kind, np.array([c]), s)[0]
            t = np.copy(self.population[c])
            if xover == "bin":
                take = rng.rand(d) < self.cross_over_probability
                take[fill] = True
                t = np.where(take, b, t)
            else:
                i = 0
                while i < d and rng.rand() < self.cross_over_probability:
                    t[fill] = b[fill]
                    fill = (fill + 1) % d
                    i += 1
            trials[c] = t
        for t in trials:
            for i in np.where((t < 0) | (t > 1))[0]:
                t[i] = rng.rand()
        return trials

    def _trials_fast(self, n):
        rng, d, npop = self.random_number_generator, self.parameter_count, self.num_population_members
        kind, k, xover = _STRATEGIES[self.strategy]
        cand = np.arange(n)
        # k distinct members != candidate: draw from the shrinking remainder, then step over the
        # already excluded indices in ascending order
        s = np.empty((n, k), dtype=np.int64)
        excl = cand[:, None].copy()
        for j in range(k):
            r = rng.randint(0, npop - 1 - j, n)
            for col in range(excl.shape[1]):
                r = r + (r >= excl[:, col])
            s[:, j] = r
            excl = np.sort(np.concatenate([excl, r[:, None]], axis=1), axis=1)
        b = self._base_and_diff(kind, cand, s)
        fill = rng.randint(0, d, n)
        if xover == "bin":
            take = rng.rand(n, d) < self.cross_over_probability
            take[cand, fill] = True
        else:
            # run of L consecutive parameters from the fill point, L = number of leading successes
            u = rng.rand(n, d) < self.cross_over_probability
            run = np.where(u.all(axis=1), d, np.argmin(u, axis=1))
            pos = (np.arange(d)[None, :] - fill[:, None]) % d
            take = pos < run[:, None]
        trials = np.where(take, b, self.population[:n])
        bad = (trials < 0) | (trials > 1)
        nbad = int(bad.sum())
        if nbad:
            trials[bad] = rng.rand(nbad)
        return trials

    def __iter__(self):
        return self

    def __next__(self):
        if np.all(np.isinf(self.population_energies)):
            self._calculate_population_energies()
        self.tell(self.func(self.ask(), *self.args))
        return self.x, self.population_energies[0]

    next = __next__

    def after_generation(self, callback_says=None):
        """The reference's tests after a generation (code/differential_evolution.py:563-585), in its order: the callback's
        verdict (callback_says: what callback(best x, convergence=) returned, when the caller has evaluated it itself),
        then the spread of the energies.  Returns None to go on, else the stop message key."""
        if callback_says is None and self.callback:
            with np.errstate(divide="ignore"):
                conv = self.tol / self.convergence
            callback_says = self.callback(self._scale_parameters(self.population[0]), convergence=conv)
        if callback_says is True:
            return "callback"
        if np.std(self.population_energies) <= self.atol + self.tol * np.abs(np.mean(self.population_energies)):
            return "success"
        return None

    def result(self, nit, stop):
        """stop: "callback" | "success" | "maxiter" (what ended the search)"""
        return OptimizeResult(x=self.x, fun=self.population_energies[0], nfev=self._nfev, nit=nit,
                              message=MESSAGES[stop], success=(stop == "success"))

    def solve(self):
        nit, stop = 0, "success"
        if np.all(np.isinf(self.population_energies)):
            self._calculate_population_energies()
        for nit in range(1, self.maxiter + 1):
            next(self)
            if self.disp:
                print("differential_evolution step %d: f(x)= %g" % (nit, self.population_energies[0]))
            stop = self.after_generation()
            if stop:
                break
        else:
            stop = "maxiter"
        result = self.result(nit, stop)
        if self.polish:
            from scipy.optimize import minimize
            one = (lambda x, *a: float(np.asarray(self.func(np.asarray(x)[None, :], *a)).ravel()[0]))
            r = minimize(one, np.copy(result.x), method="L-BFGS-B", bounds=self.limits.T, args=self.args)
            self._nfev += r.nfev
            result.nfev = self._nfev
            if r.fun < result.fun:
                result.fun, result.x, result.jac = r.fun, r.x, r.jac
                self.population_energies[0] = r.fun
                self.population[0] = self._unscale_parameters(r.x)
        return result


def differential_evolution(func, bounds, args=(), strategy="best1bin", maxiter=1000, popsize=15, tol=0.01,
                           mutation=(0.5, 1), recombination=0.7, seed=None, callback=None, disp=False, polish=True,
                           init="latinhypercube", atol=0, rng_compat=False):
    solver = DifferentialEvolutionSolver(func, bounds, args=args, strategy=strategy, maxiter=maxiter, popsize=popsize,
                                         tol=tol, mutation=mutation, recombination=recombination, seed=seed,
                                         polish=polish, callback=callback, disp=disp, init=init, atol=atol,
                                         rng_compat=rng_compat)
    return solver.solve()
