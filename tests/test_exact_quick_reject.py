"""The all-pairs kernel (RT_TRACE_EXACT, traverse_exact_kernel) does not run the shader's divide on every
pair: from the determinant a and the numerator x = dot(s, q) of u, both formed exactly as trace.glsl:7-12
forms them, it settles a pair as rejected when

        x * (x - a)  >  fma(a * a, 2^-20, 2^-100)            (all in binary32)

and sends every other pair through the shader's whole function.  This test holds the claim behind that --
"the comparison is true  =>  the shader rejects the pair" -- against the shader's own arithmetic
(f = 1 / a; u = f * x; reject if |a| < 1e-20 or u < 0 or u > 1) on values chosen to sit on every edge
the proof has: u next to 0 and next to 1 within a few ulp, tiny and huge a and x, subnormals, zeros,
infinities, NaN.  numpy's binary32 multiply, subtract and divide are correctly rounded, as the GPU's are."""
import numpy as np

f32 = np.float32


def shader_rejects(a, x):
    with np.errstate(all="ignore"):
        near_zero = (a > f32(-1e-20)) & (a < f32(1e-20))
        f = f32(1.0) / a
        u = f * x
        return near_zero | (u < f32(0.0)) | (u > f32(1.0))


def settled(a, x):
    """The kernel's comparison.  The fused multiply-add is formed in binary64 and rounded once more, which can
    land one ulp from the fused result: the test below also accepts the threshold one ulp lower, so whatever the
    hardware's fma yields is covered."""
    with np.errstate(all="ignore"):
        z = x * (x - a)
        a2 = a * a
        thr = (a2.astype(np.float64) * 2.0 ** -20 + 2.0 ** -100).astype(f32)
        # (an infinite threshold is infinite however the fma is formed)
        thr_lo = np.where(np.isfinite(thr), np.nextafter(thr, f32(-np.inf)), thr)
        return z > thr_lo, z > thr


def check(a, x):
    a = np.asarray(a, dtype=f32)
    x = np.asarray(x, dtype=f32)
    loose, tight = settled(a, x)
    rej = shader_rejects(a, x)
    bad = loose & ~rej
    assert not bad.any(), (a[bad][:5], x[bad][:5])
    return int(tight.sum()), int(rej.sum())


def test_random_magnitudes():
    rng = np.random.default_rng(7)
    n = 4_000_000
    a = (rng.choice([-1.0, 1.0], n) * np.exp2(rng.uniform(-140, 127, n))).astype(f32)
    x = (rng.choice([-1.0, 1.0], n) * np.exp2(rng.uniform(-140, 127, n))).astype(f32)
    check(a, x)   # (over this range a * a often overflows or underflows, and the comparison then settles nothing)
    # magnitudes a scene has: the comparison settles nearly every pair the shader rejects
    a = (rng.choice([-1.0, 1.0], n) * np.exp2(rng.uniform(-20, 20, n))).astype(f32)
    x = (a.astype(np.float64) * rng.uniform(-3.0, 4.0, n)).astype(f32)   # u spread evenly over (-3, 4)
    settled_n, rejected_n = check(a, x)
    assert settled_n > 0.9999 * rejected_n > 0


def test_u_next_to_its_bounds():
    rng = np.random.default_rng(8)
    n = 2_000_000
    a = (rng.choice([-1.0, 1.0], n) * np.exp2(rng.uniform(-70, 64, n))).astype(f32)
    # x = a * (1 + k ulp), x = a * k ulp, for small signed k: u within a few ulp of 1 and of 0
    k = rng.integers(-64, 65, n)
    x1 = a.copy()
    for _ in range(64):
        step = np.where(k > 0, 1, np.where(k < 0, -1, 0))
        x1 = np.where(step > 0, np.nextafter(x1, f32(np.inf) * np.sign(a)),
                      np.where(step < 0, np.nextafter(x1, f32(0)), x1)).astype(f32)
        k = k - step
    check(a, x1)
    # ratios 2^-30 .. 2^-18 either side of zero: around the margin the comparison keeps
    r = (rng.choice([-1.0, 1.0], n) * np.exp2(rng.uniform(-30, -18, n)))
    check(a, (a.astype(np.float64) * r).astype(f32))
    # ratios 1 +- 2^-26 .. 1 +- 2^-18
    r = 1.0 + rng.choice([-1.0, 1.0], n) * np.exp2(rng.uniform(-26, -18, n))
    check(a, (a.astype(np.float64) * r).astype(f32))


def test_special_values():
    tiny = np.nextafter(f32(0), f32(1))
    vals = np.array([0.0, -0.0, tiny, -tiny, 1e-45, 1e-38, -1e-38, 1.1754944e-38, 1e-30, 1e-21, 1e-20, -1e-20,
                     1.0000001e-20, 9.999999e-21, 1e-10, 0.5, 1.0, -1.0, 2.0, 1e10, 1.8e19, 1.9e19, 1e20, -1e20, 3e38,
                     -3e38, np.inf, -np.inf, np.nan], dtype=f32)
    a, x = np.meshgrid(vals, vals)
    check(a.ravel(), x.ravel())
    # every value against its neighbours
    nb = np.concatenate([vals, np.nextafter(vals, f32(np.inf)), np.nextafter(vals, f32(-np.inf))])
    a, x = np.meshgrid(nb, nb)
    check(a.ravel(), x.ravel())
