"""A second, independent restatement of the reference's three compute shaders, in numpy float32, written
from the shader text and not from oracle/rt_oracle.c.  Vectorised over rays, a Python loop over
triangles: for small scenes only.  tests/test_oracle_numpy.py holds it against the C oracle bit for
bit, so that a slip in either restatement (an operand order, a forgotten mutation, an off-by-one in the
kill rule) shows up as a difference between two texts that were written separately.

Follows "OpenGL Raytracer/raygen_cs.glsl":1-22, rayintersect_cs.glsl:1-34, trace.glsl:1-22, 57-78,
121-184 and raycol_cs.glsl:3-16, 18-128, statement by statement.  Built-ins whose result the GLSL
specification leaves to the implementation are fixed as the oracle's header fixes them:
normalize(v) = v * (1 / sqrt(v.v)); pow(x, 30) by squaring (x2 x4 x8 x16; x16 x8 x4 x2);
max(x, 0) = x < 0 ? 0 : x; reads outside a buffer or image give 0; ivec2() truncates; an rgba8 store
rounds clamp(x, 0, 1) * 255 to nearest, ties to even.  The ray index is x + y * width.
"""
import numpy as np

f32 = np.float32
NEAR_ZERO = f32(1e-20)


def V(x, y, z):
    return (x, y, z)


def dot(a, b):
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def cross(a, b):
    return (a[1] * b[2] - b[1] * a[2], a[2] * b[0] - b[2] * a[0], a[0] * b[1] - b[0] * a[1])


def sub(a, b):
    return (a[0] - b[0], a[1] - b[1], a[2] - b[2])


def add(a, b):
    return (a[0] + b[0], a[1] + b[1], a[2] + b[2])


def mul(a, s):
    return (a[0] * s, a[1] * s, a[2] * s)


def normalize(a):
    inv = f32(1.0) / np.sqrt(dot(a, a))
    return mul(a, inv)


def where3(m, a, b):
    return tuple(np.where(m, x, y) for x, y in zip(a, b))


def c3(v, n):
    """a constant vec3 as three float32 arrays of n elements"""
    return tuple(np.full(n, f32(x), dtype=f32) for x in v)


SPHERES = [((0.0, -0.7, 2.0), 0.5), ((-2.0, -0.0, 0.0), 0.2)]


def ray_vs_triangle(o, d, X, Y, Z):
    """trace.glsl:1-22 -> (hit mask, t)"""
    e1 = sub(Y, X)
    e2 = sub(Z, X)
    q = cross(d, e2)
    a = dot(e1, q)
    ok = ~((a > -NEAR_ZERO) & (a < NEAR_ZERO))
    f = f32(1.0) / a
    s = sub(o, X)
    u = f * dot(s, q)
    ok &= ~((u < f32(0.0)) | (u > f32(1.0)))
    r = cross(s, e1)
    v = f * dot(d, r)
    ok &= ~((v < f32(0.0)) | (u + v > f32(1.0)))
    t = f * dot(e2, r)
    return ok, t


def ray_sphere(o, d, c, radius):
    """trace.glsl:57-78 -> (hit mask, t)"""
    l = sub(c, o)
    s = dot(l, d)
    l_sqr = dot(l, l)
    r_sqr = f32(radius) * f32(radius)
    outside = l_sqr > r_sqr
    ok = ~((s < f32(0.0)) & outside)
    m_sqr = l_sqr - s * s
    ok &= ~(m_sqr > r_sqr)
    q = np.sqrt(r_sqr - m_sqr)
    t = np.where(outside, s - q, s + q)
    return ok, t


def trace(o, d, n_in, color_in, verts):
    """trace.glsl:121-184 for rays (o, d) whose n / color hold n_in / color_in -> t, id, n, color"""
    cnt = o[0].shape[0]
    t_best = np.full(cnt, np.inf, dtype=f32)
    pid = np.zeros(cnt, dtype=np.int32)
    n, color = n_in, color_in
    num_tris = verts.shape[0] // 3
    primitive = 0
    for k in range(num_tris):
        X, Y, Z = (c3(verts["pos"][3 * k + j], cnt) for j in range(3))
        hit, t = ray_vs_triangle(o, d, X, Y, Z)
        better = hit & (t > f32(0.0)) & (t < t_best)
        t_best = np.where(better, t, t_best)
        color = where3(better, c3((0.0, 0.0, 0.0), cnt), color)
        pid = np.where(better, np.int32(primitive), pid)
        nn = normalize(cross(sub(X, Z), sub(Y, Z)))
        n = where3(better, nn, n)
        primitive += 1
    for centre, radius in SPHERES:
        c = c3(centre, cnt)
        hit, t = ray_sphere(o, d, c, radius)
        better = hit & (t > f32(0.0)) & (t < t_best)
        t_best = np.where(better, t, t_best)
        color = where3(better, c3((0.0, 0.0, 1.0), cnt), color)
        pid = np.where(better, np.int32(primitive), pid)
        p = add(o, mul(d, t_best))
        nn = normalize(sub(p, c))
        n = where3(better, nn, n)
        primitive += 1
    return t_best, pid, n, color


class Frame:
    """The ray buffer and the output image, stage by stage."""

    def __init__(self, width, height):
        self.w, self.h = int(width), int(height)
        n = self.w * self.h
        z = lambda: tuple(np.zeros(n, dtype=f32) for _ in range(3))
        self.origin, self.dir, self.color, self.n = z(), z(), z(), z()
        self.t = np.zeros(n, dtype=f32)
        self.id = np.zeros(n, dtype=np.int32)
        self.pixels = np.zeros((self.h, self.w, 4), dtype=np.uint8)

    def raygen(self, cam):
        """raygen_cs.glsl:1-22"""
        pos, cdir, up, right = (tuple(f32(x) for x in np.asarray(v, dtype=f32)) for v in cam)
        n = self.w * self.h
        x = np.tile(np.arange(self.w, dtype=np.int32), self.h)
        y = np.repeat(np.arange(self.h, dtype=np.int32), self.w)
        hw = f32(self.w) / f32(2.0)
        hh = f32(self.h) / f32(2.0)
        dx = (x.astype(f32) - hw) / hw
        dy = (y.astype(f32) - hh) / hh
        P, D, U, R = (c3(v, n) for v in (pos, cdir, up, right))
        s = add(add(add(P, mul(D, f32(1.0))), mul(R, dx)), mul(U, dy))
        self.origin = P
        self.dir = normalize(sub(s, P))
        self.color = c3((1.0, 0.0, 0.0), n)
        self.t = np.full(n, f32(-1.0), dtype=f32)
        self.id = np.full(n, -1, dtype=np.int32)
        self.n = c3((0.0, 0.0, 0.0), n)

    def intersect(self, verts):
        """rayintersect_cs.glsl:1-34 (rays with primitiveID == -2 are left as they are)"""
        with np.errstate(all="ignore"):
            live = self.id != -2
            o = add(self.origin, mul(self.dir, f32(0.001)))
            t, pid, n, color = trace(o, self.dir, self.n, self.color, verts)
            o = sub(o, mul(self.dir, f32(0.001)))
            pid = np.where(np.isinf(t), np.int32(-2), pid)
            o = add(o, mul(self.dir, t))
            d = sub(self.dir, mul(n, f32(2.0) * dot(self.dir, n)))
            self.origin = where3(live, o, self.origin)
            self.dir = where3(live, d, self.dir)
            self.n = where3(live, n, self.n)
            self.color = where3(live, color, self.color)
            self.t = np.where(live, t, self.t)
            self.id = np.where(live, pid, self.id)

    def color_stage(self, verts, lights, num_lights, cam_pos, tex):
        """raycol_cs.glsl:18-128 (every ray, dead ones too)"""
        with np.errstate(all="ignore"):
            cnt = self.w * self.h
            final = c3((0.1, 0.1, 0.1), cnt)
            shadowed = np.ones(cnt, dtype=bool)
            light = c3((0.0, 0.0, 0.0), cnt)
            o = self.origin
            cam = c3(np.asarray(cam_pos, dtype=f32), cnt)
            zero = np.zeros(cnt, dtype=f32)
            for a in range(num_lights):
                lp = c3(lights["pos"][a], cnt)
                lc = c3(lights["color"][a], cnt)
                ld = normalize(sub(lp, o))
                ts, _, _, _ = trace(add(o, mul(ld, f32(0.0000001))), ld, c3((0, 0, 0), cnt), self.color, verts)
                o = sub(o, mul(ld, f32(0.0000001)))
                dist = np.sqrt(dot(sub(lp, o), sub(lp, o)))
                lit = ts > dist
                shadowed &= ~lit
                ndl = dot(self.n, ld)
                diffuse = np.where(ndl < f32(0.0), zero, ndl)
                v = normalize(sub(cam, o))
                I = mul(ld, f32(-1.0))
                refl = normalize(sub(I, mul(self.n, f32(2.0) * dot(self.n, I))))
                x = dot(v, refl)
                x = np.where(x < f32(0.0), zero, x)
                x2 = x * x; x4 = x2 * x2; x8 = x4 * x4; x16 = x8 * x8
                k = np.where(ndl > f32(0.0), ((x16 * x8) * x4) * x2, zero)
                att = f32(1.0) / (f32(0.5) * dist)
                contrib = mul(mul(lc, diffuse + k), att)
                light = where3(lit, add(light, contrib), light)
            # texture, final colour, kill rule
            nv = verts.shape[0]
            pos = np.zeros((cnt, 3, 3), dtype=f32)
            tc = np.zeros((cnt, 3, 2), dtype=f32)
            for j in range(3):
                idx = self.id.astype(np.int64) * 3 + j
                inside = (idx >= 0) & (idx < nv)
                safe = np.where(inside, idx, 0)
                pos[:, j] = np.where(inside[:, None], verts["pos"][safe], f32(0.0)) if nv else f32(0.0)
                tc[:, j] = np.where(inside[:, None], verts["texCoord"][safe], f32(0.0)) if nv else f32(0.0)
            A, B, C = (tuple(pos[:, j, q] for q in range(3)) for j in range(3))
            v0, v1, v2 = sub(B, A), sub(C, A), sub(o, A)
            d00, d01, d11 = dot(v0, v0), dot(v0, v1), dot(v1, v1)
            d20, d21 = dot(v2, v0), dot(v2, v1)
            denom = d00 * d11 - d01 * d01
            bv = (d11 * d20 - d01 * d21) / denom
            bw = (d00 * d21 - d01 * d20) / denom
            bu = f32(1.0) - bv - bw
            uvx = bu * tc[:, 0, 0] + bv * tc[:, 1, 0] + bw * tc[:, 2, 0]
            uvy = bu * tc[:, 0, 1] + bv * tc[:, 1, 1] + bw * tc[:, 2, 1]
            texc = c3((0.0, 0.0, 0.0), cnt)
            if tex is not None:
                th, tw = tex.shape[0], tex.shape[1]
                fx, fy = uvx * f32(tw), uvy * f32(th)
                ok = (fx > f32(-1.0)) & (fx < f32(tw)) & (fy > f32(-1.0)) & (fy < f32(th))  # trunc() lands in [0, size)
                px = np.where(ok, np.trunc(np.where(ok, fx, 0)), 0).astype(np.int64)
                py = np.where(ok, np.trunc(np.where(ok, fy, 0)), 0).astype(np.int64)
                texel = tex[py, px].astype(f32) / f32(255.0)
                texc = tuple(np.where(ok, texel[:, q], f32(0.0)) for q in range(3))
            lit_any = ~shadowed
            shaded = add(texc, mul(light, f32(0.3)))
            final = where3(lit_any, shaded, final)
            kill = lit_any & (self.id + 3 * 5 - 2 < nv // 3)
            self.id = np.where(kill, np.int32(-2), self.id)
            self.origin = where3(kill, o, self.origin)   # rays[i] = r stores the mutated origin
            rgb = [np.rint(np.clip(ch, f32(0.0), f32(1.0)) * f32(255.0)) for ch in final]
            rgb = [np.where(np.isnan(ch), 0, ch).astype(np.uint8) for ch in rgb]
            img = np.stack(rgb + [np.full(cnt, 255, dtype=np.uint8)], axis=1)
            self.pixels = img.reshape(self.h, self.w, 4)

    def render(self, cam, verts, lights, num_lights, tex, passes):
        """main.cpp:476-529: raygen, then passes x (intersect, color)"""
        self.raygen(cam)
        for _ in range(passes):
            self.intersect(verts)
            self.color_stage(verts, lights, num_lights, cam[0], tex)
