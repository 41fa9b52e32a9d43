"""oracle/noise.py — CPU restatement of the A2-LINK perturbation stage.  TEST INFRASTRUCTURE.

Two layers:

1. `ref_*` — the reference's arithmetic drawing from `np.random` exactly as it does, so that with the
   same `np.random.seed` the outputs equal those of the reference's own classes.  Pinned by
   tests/golden/noise.npz (tests/golden/make_golden_noise.py imports reference code/noise.py and
   code/attack.py):   Gaussian code/noise.py:33-45, Poisson :68-76, Speckle :79-88, Perlin :91-150,
   perturb_image code/attack.py:5-29.
   SaltPepper (:48-65) is NOT pinned: the reference indexes with a *list* of index arrays, which
   NumPy < 1.23 (the reference's era, requirements.txt) read as a tuple (per-element writes) and the
   NumPy in this container reads as one fancy index on axis 0; the restatement uses the tuple
   meaning.  cv2.resize (code/committee.py:22-26) is third-party (OpenCV, absent
   here): `resize_bilinear` restates OpenCV's INTER_LINEAR sampling rule, unpinned.

2. `philox_*` — the same transformations fed from the Philox4x32-10 counter stream the device
   kernels use (a-link_amd/csrc/noise.hip), so device outputs can be compared element by element.
"""
import numpy as np

# ------------------------------------------------------------------------------------------------
# layer 1: reference-faithful, np.random stream
# ------------------------------------------------------------------------------------------------


def ref_gaussian(image, mean=10, var=10):
    row, col, ch = image.shape
    sigma = var ** 0.5
    gauss = np.random.normal(mean, sigma, (row, col, ch)).reshape(row, col, ch)
    return image + gauss


def ref_speckle(image):
    row, col, ch = image.shape
    gauss = (np.random.randn(row, col, ch) / 15).reshape(row, col, ch)
    return image + image * gauss


def ref_poisson(image):
    vals = len(np.unique(image))
    vals = 2 ** np.ceil(np.log2(vals))
    return np.random.poisson(image * vals) / float(vals)


def salt_pepper_counts(shape, s_vs_p=0.5, amount=0.004):
    size = int(np.prod(shape))
    return int(np.ceil(amount * size * s_vs_p)), int(np.ceil(amount * size * (1. - s_vs_p)))


def ref_saltpepper(image, s_vs_p=0.5, amount=0.004):
    out = np.copy(image)
    n_salt, n_pepper = salt_pepper_counts(image.shape, s_vs_p, amount)
    coords = [np.random.randint(0, i - 1, n_salt) for i in image.shape]
    out[tuple(coords)] = 1          # tuple: NumPy < 1.23 meaning of the reference's list index
    coords = [np.random.randint(0, i - 1, n_pepper) for i in image.shape]
    out[tuple(coords)] = 0
    return out


def quintic(t):
    return t * t * t * (t * (t * 6 - 15) + 10)


def perlin_octave(size, ns, phi):
    """One octave (code/noise.py:95-136) from the (size/ns + 1)^2 grid of angles `phi`:
    pixel (y, x) = (i ns + a, j ns + b) mixes the four surrounding nodes' <offset, unit vector>
    with quintic weights."""
    nc = int(size / ns)
    if nc * ns != size:
        raise ValueError("cannot reshape array of size %d into shape (%d,%d,%d,%d)" % (size * size, nc, ns, nc, ns))
    gs = int(size / ns + 1)
    phi = np.asarray(phi, dtype=np.float64).reshape(gs, gs)
    vx, vy = np.cos(phi), np.sin(phi)
    a = np.arange(ns, dtype=np.float64)
    q = quintic(a / ns)
    m = np.zeros((size, size))
    t = m.reshape(nc, ns, nc, ns)
    A, B = a[:, None], a[None, :]
    qa, qb = q[:, None], q[None, :]
    for i in range(nc):
        for j in range(nc):
            acc = np.zeros((ns, ns))
            for r in (0, 1):
                for s in (0, 1):
                    d = (B - s * ns) * vx[i + r, j + s] + (A - r * ns) * vy[i + r, j + s]
                    acc += (qa if r else 1 - qa) * (qb if s else 1 - qb) * d
            t[i, :, j, :] = acc
    return m


def perlin_octaves(row):
    return [56, 32, 16] if row % 56 == 0 else [50, 30, 15]


def ref_perlin(image):
    row, col, ch = image.shape
    assert row == col
    total = np.zeros((row, row))
    for ns in perlin_octaves(row):
        gs = int(row / ns + 1)
        phi = np.random.uniform(0, 2 * np.pi, (gs, gs))
        total = total + perlin_octave(row, ns, phi)
    return image + np.repeat(np.expand_dims(total, 2), 3, 2)


def perturb_image(xs, img):
    """code/attack.py:5-29."""
    xs = np.asarray(xs)
    if xs.ndim < 2:
        xs = np.array([xs])
    imgs = np.tile(img, [len(xs)] + [1] * (xs.ndim + 1))
    xs = xs.astype(int)
    for x, im in zip(xs, imgs):
        for px in x.reshape(-1, 5):
            im[px[0], px[1]] = px[2:5]
    return imgs


def resize_bilinear(images, new_size):
    """cv2.resize(image, new_size) for float images, INTER_LINEAR: new_size = (width, height);
    source coordinate of a destination pixel centre, floor + fraction, edges clamped."""
    images = np.asarray(images)
    n, H, W, C = images.shape
    Wo, Ho = int(new_size[0]), int(new_size[1])

    def coords(dst, src):
        f = ((np.arange(dst) + 0.5) * (float(src) / dst) - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = f - s.astype(np.float32)
        lo = s < 0
        s[lo], f[lo] = 0, 0
        hi = s >= src - 1
        s[hi], f[hi] = src - 1, 0
        return s, np.minimum(s + 1, src - 1), f.astype(np.float32)
    x0, x1, fx = coords(Wo, W)
    y0, y1, fy = coords(Ho, H)
    im = images.astype(np.float32)
    fx_ = fx[None, None, :, None]
    rows0 = im[:, y0][:, :, x0] * (1 - fx_) + im[:, y0][:, :, x1] * fx_
    rows1 = im[:, y1][:, :, x0] * (1 - fx_) + im[:, y1][:, :, x1] * fx_
    fy_ = fy[None, :, None, None]
    return rows0 * (1 - fy_) + rows1 * fy_


# ------------------------------------------------------------------------------------------------
# layer 2: the device's Philox4x32-10 stream (noise.hip)
# ------------------------------------------------------------------------------------------------
ST_NORMAL, ST_POISSON, ST_SALTPEPPER, ST_PERLIN = 0, 1, 2, 3
_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(idx, sub, st, seed):
    """Vectorised over `idx` (uint64 array): counter (idx_lo, idx_hi, sub, st), key = seed halves.
    Returns four uint32 arrays."""
    idx = np.asarray(idx, dtype=np.uint64)
    c0, c1 = idx & _MASK, idx >> np.uint64(32)
    c2 = np.broadcast_to(np.asarray(sub, dtype=np.uint64), idx.shape).copy()
    c3 = np.full(idx.shape, st, dtype=np.uint64)
    k0, k1 = int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = _M0 * c0, _M1 * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)
        n1 = p1 & _MASK
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)
        n3 = p0 & _MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0, k1 = (k0 + _W0) & 0xFFFFFFFF, (k1 + _W1) & 0xFFFFFFFF
    return [c.astype(np.uint32) for c in (c0, c1, c2, c3)]


def u01(x):
    return (x >> np.uint32(8)).astype(np.float32) * np.float32(5.9604644775390625e-8) + np.float32(2.98023223876953125e-8)


def u01d(hi, lo):
    v = ((hi >> np.uint32(5)).astype(np.uint64) << np.uint64(26)) | (lo >> np.uint32(6)).astype(np.uint64)
    return (v.astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def philox_normals(count, seed, offset=0):
    """The float32 standard normals element e = offset + i receives (Box-Muller on Philox block e>>2)."""
    assert offset % 4 == 0
    g = (offset // 4) + np.arange((count + 3) // 4, dtype=np.uint64)
    r = philox4x32_10(g, 0, ST_NORMAL, seed)
    z = np.empty((len(g), 4), dtype=np.float32)
    for h in (0, 1):
        rad = np.sqrt(np.float32(-2) * np.log(u01(r[2 * h])))
        th = np.float32(6.283185307179586) * u01(r[2 * h + 1])
        z[:, 2 * h] = rad * np.cos(th)
        z[:, 2 * h + 1] = rad * np.sin(th)
    return z.ravel()[:count]


def philox_gaussian(images, seed, mean=10.0, sigma=10 ** 0.5, offset=0):
    x = np.asarray(images, dtype=np.float32)
    z = philox_normals(x.size, seed, offset).reshape(x.shape)
    return x + (np.float32(mean) + np.float32(sigma) * z)


def philox_speckle(images, seed, divisor=15.0, offset=0):
    x = np.asarray(images, dtype=np.float32)
    z = philox_normals(x.size, seed, offset).reshape(x.shape)
    return x + x * (z / np.float32(divisor))


def _bounded(x, rng):
    return ((x.astype(np.uint64) * np.uint64(rng)) >> np.uint64(32)).astype(np.int64)


def philox_saltpepper(images, seed, s_vs_p=0.5, amount=0.004):
    x = np.array(images, dtype=np.float32)
    n, H, W, C = x.shape
    n_salt, n_pepper = salt_pepper_counts((H, W, C), s_vs_p, amount)
    for img in range(n):
        for phase, cnt, val in ((0, n_salt, 1.0), (1, n_pepper, 0.0)):
            k = np.arange(cnt, dtype=np.uint64)
            r = philox4x32_10((np.uint64(img) << np.uint64(32)) | k, phase, ST_SALTPEPPER, seed)
            x[img, _bounded(r[0], H - 1), _bounded(r[1], W - 1), _bounded(r[2], C - 1)] = val
    return x


def philox_perlin_vectors(n_images, nodes_total, seed):
    n = n_images * nodes_total
    g = np.arange((n + 3) // 4, dtype=np.uint64)
    r = philox4x32_10(g, 0, ST_PERLIN, seed)
    u = np.stack([u01(w) for w in r], axis=1).ravel()[:n]
    phi = np.float32(6.283185307179586) * u
    return np.stack([np.cos(phi), np.sin(phi)], axis=1).astype(np.float32).reshape(n_images, nodes_total, 2)


def perlin_from_vectors(images, vec, octaves=None):
    """Perlin noise from explicit unit vectors [n][nodes][2] laid out octave after octave."""
    x = np.asarray(images, dtype=np.float64)
    n, size = x.shape[0], x.shape[1]
    octaves = octaves or perlin_octaves(size)
    out = np.empty_like(x)
    for img in range(n):
        total, off = np.zeros((size, size)), 0
        for ns in octaves:
            gs = int(size / ns + 1)
            v = np.asarray(vec[img, off:off + gs * gs], dtype=np.float64)
            total = total + perlin_octave(size, ns, np.arctan2(v[:, 1], v[:, 0]))
            off += gs * gs
        out[img] = x[img] + total[:, :, None]
    return out


def poisson_vals(image):
    return 2 ** np.ceil(np.log2(len(np.unique(image))))


_LOGFACT8 = np.array([0.0, 0.0, 0.6931471806, 1.791759469, 3.178053830, 4.787491743, 6.579251212, 8.525161361], np.float32)


def _ptrs_f32(L, elem, seed):
    """noise.hip::poisson_ptrs_f32 in NumPy float32, operation for operation (setup and proposal are single correctly
    rounded operations on both sides; the acceptance test's log / log1p differ from the device's by an ulp or so, which
    flips a decision only when the two sides agree to ~1e-6).  L: float32 array, 10 <= L < 2^24.  Returns float64 counts."""
    f = np.float32
    L = np.asarray(L, f)
    slam = np.sqrt(L)
    b = f(0.931) + f(2.53) * slam
    a = f(-0.059) + f(0.02483) * b
    invalpha = f(1.1239) + f(1.1328) / (b - f(3.4))
    vr = f(0.9277) - f(3.6224) / (b - f(2.0))
    Li = np.floor(L)
    Lf043 = (L - Li) + f(0.43)
    a2 = f(2.0) * a
    inv_lam = f(1.0) / L
    res, done = np.zeros(len(L), np.float64), np.zeros(len(L), bool)
    sub = 0
    while not done.all():
        r = philox4x32_10(elem, sub, ST_POISSON, seed)
        sub += 1
        for wu, wv in ((r[0], r[1]), (r[2], r[3])):
            U = u01(wu) - f(0.5)
            V = u01(wv)
            us = f(0.5) - np.abs(U)
            kf = np.floor((a2 / us + b) * U + Lf043)
            acc1 = (us >= f(0.07)) & (V <= vr)
            k = Li + kf
            rej = (k < 0) | ((us < f(0.013)) & (V > us))
            us2 = us * us
            with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
                lhs = V * invalpha * us2 / (a + b * us2)
                small = k < 8
                ks = np.clip(k, 0, 7).astype(int)
                logp_small = -L + k * np.log(L) - _LOGFACT8[ks]
                d = kf - (L - Li)
                x = d * inv_lam
                ser = f(-1.0 / 110.0)
                for c in (1.0 / 90, -1.0 / 72, 1.0 / 56, -1.0 / 42, 1.0 / 30, -1.0 / 20, 1.0 / 12, -1.0 / 6, 0.5):
                    ser = (ser.astype(np.float64) * x.astype(np.float64) + np.float64(f(c))).astype(f)      # fmaf
                h = np.where(np.abs(x) < f(0.125), ser * x * x, (f(1.0) + x) * np.log1p(x) - x).astype(f)
                rk = f(1.0) / k
                logp_big = -L * h - f(0.5) * np.log(f(6.283185307179586) * k) - rk * f(1.0 / 12.0) + rk * rk * rk * f(1.0 / 360.0)
                logp = np.where(small, logp_small, logp_big)
                acc2 = np.log(lhs) <= logp
            take = ~done & (acc1 | (~rej & acc2))
            res = np.where(take, Li.astype(np.float64) + kf.astype(np.float64), res)
            done |= take
    return res


def philox_poisson(images, seed, first_image=0):
    """Same sampler as noise.hip (poisson_kernel / poisson_rest_kernel), element by element (slow: test sizes only):
    lam = x * vals in float32; 0 -> 0; 10 <= lam < 2^24 -> the float32 PTRS above; anything else the float64 forms
    (product of uniforms below 10, PTRS from 2^24)."""
    from scipy.special import gammaln
    x = np.asarray(images, dtype=np.float32)
    n = x.shape[0]
    per = x[0].size
    out = np.empty(x.shape, dtype=np.float32)
    flat_in, flat_out = x.reshape(n, per), out.reshape(n, per)
    for img in range(n):
        vals = float(poisson_vals(flat_in[img]))
        lamf = flat_in[img] * np.float32(vals)
        lam = flat_in[img].astype(np.float64) * vals
        elem = np.uint64((first_image + img) * per) + np.arange(per, dtype=np.uint64)
        k = np.full(per, np.nan)
        k[lam == 0] = 0.0
        mid = np.where((lamf >= np.float32(10)) & (lamf < np.float32(16777216)))[0]
        if len(mid):
            k[mid] = _ptrs_f32(lamf[mid], elem[mid], seed)
        # --- lam < 10: product of uniforms
        small = np.where((lam > 0) & (lam < 10))[0]
        if len(small):
            enlam = np.exp(-lam[small])
            prod, cnt, done = np.ones(len(small)), np.zeros(len(small)), np.zeros(len(small), bool)
            sub = 0
            while not done.all():
                r = philox4x32_10(elem[small], sub, ST_POISSON, seed)
                sub += 1
                for hi, lo in ((r[0], r[1]), (r[2], r[3])):
                    act = ~done
                    prod = np.where(act, prod * u01d(hi, lo), prod)
                    hit = act & (prod <= enlam)
                    done |= hit
                    cnt = np.where(act & ~hit, cnt + 1, cnt)
            k[small] = cnt
        # --- lam >= 2^24: PTRS in float64
        big = np.where(lamf >= np.float32(16777216))[0]
        if len(big):
            L = lam[big]
            slam, loglam = np.sqrt(L), np.log(L)
            b = 0.931 + 2.53 * slam
            a = -0.059 + 0.02483 * b
            invalpha = 1.1239 + 1.1328 / (b - 3.4)
            vr = 0.9277 - 3.6224 / (b - 2.0)
            res, done = np.zeros(len(big)), np.zeros(len(big), bool)
            sub = 0
            while not done.all():
                r = philox4x32_10(elem[big], sub, ST_POISSON, seed)
                sub += 1
                U = u01d(r[0], r[1]) - 0.5
                V = u01d(r[2], r[3])
                us = 0.5 - np.abs(U)
                kk = np.floor((2.0 * a / us + b) * U + L + 0.43)
                acc1 = (us >= 0.07) & (V <= vr)
                rej = (kk < 0) | ((us < 0.013) & (V > us))
                with np.errstate(invalid="ignore", divide="ignore"):
                    acc2 = (np.log(V) + np.log(invalpha) - np.log(a / (us * us) + b)) <= (-L + kk * loglam - gammaln(kk + 1.0))
                take = ~done & (acc1 | (~rej & acc2))
                res = np.where(take, kk, res)
                done |= take
            k[big] = res
        flat_out[img] = (k / vals).astype(np.float32)
    return out
