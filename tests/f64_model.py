"""An INDEPENDENT numeric model of the reference's path tracer, for tests only.

Written from test/ClKernels/GenerateColors.cl (the OpenCL source) -- not from oracle/pt_oracle.c and not from the HIP
kernels -- in DIFFERENT arithmetic: numpy float64 and libm (np.sin / np.cos / np.sqrt / np.tan, plain "/" and "**"),
plain dot / cross without fused multiply-add, vectorised over paths.  The oracle and the HIP kernels share one arithmetic
contract (PTSPEC: binary32, FMA forms of dot and cross, own sin / cos); a restatement error that both inherited from the
builder's reading of the kernel would pass every bit-exact comparison between them.  This model shares neither code nor
arithmetic with them: where it and the oracle take the same decisions (which triangle each bounce hits, how the path
ends) their radiances must agree to float32 rounding, and they can only take different decisions next to a decision
boundary, which the model knows the distance to.

What is kept in binary32 on purpose, because the reference's TYPES say so (these are values, not arithmetic):
the RNG's `(float)(*seed) * 2.3283064365386963e-10f` (GenerateColors.cl:70: a u32 -> float conversion), the literals with
an `f` suffix (TWO_PI, INV_PI, 0.01f, 0.45f, 1e-8f, 0.001f ...), `fov = (60.0f * M_PI) / 180.0f` narrowed to float (:267),
and the scene's float32 vertex / material data.

Margins.  Every comparison the kernel makes is recorded with its distance to the boundary:
  det vs 1e-8 (:100), u, v, u + v against 0 / 1 (:109,:117), t > 0 (:125) -- each as the distance of the NUMERATOR from
  its boundary in units of its terms' magnitude, which is what a rounding error scales with --, the closest hit against
  the runner-up (relative difference of t), fabs(n.x) > 0.001 (:166,:186), dot(n, dir) < 0 (:243),
  dot(wi, n) dot(wo, n) < 0 (:211) and pdf <= 0 (:251).
`margin` is the smallest of them over the whole path.  `sens` sums, over the path's GGX bounces, tan(incidence) / sin(theta):
binary32 forms sin(theta) = sqrt(1 - cos^2(theta)) (:185) with an absolute error of ~2^-24 in the radicand, i.e. a relative
error of ~2^-25 / sin^2(theta) in the half vector's tilt, which the throughput 2 albedo (wo.wh) / ((wo.n) cos(theta)) sees
multiplied by tan(incidence) sin(theta): the one place where two correct evaluations of the reference's formulas differ by
far more than 2^-24 (roughness 0.008: sin(theta) ~ 0.008, so a grazing reflection moves the weight by up to ~1e-3).
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
U32 = 2.0 ** -24   # unit roundoff of binary32
TWO_PI = float(F32(6.28318530718))
INV_PI = float(F32(0.31830988618))
BOUNCES = 16


def _f(x):
    """a float32 literal / datum as the float64 value it denotes"""
    return float(F32(x))


def hash_u32(x):
    """GenerateColors.cl:57 (the #else branch)"""
    return (np.uint64(1103515245) * np.asarray(x, np.uint64) + np.uint64(12345)) & np.uint64(0xFFFFFFFF)


def random_float(seed):
    """GenerateColors.cl:61-71.  seed: uint64 array holding u32 values; returns (new seed, float64 value of the float)"""
    M = np.uint64(0xFFFFFFFF)
    s = seed
    s = (s ^ np.uint64(61)) ^ (s >> np.uint64(16))
    s = (s + (s << np.uint64(3))) & M
    s = s ^ (s >> np.uint64(4))
    s = (s * np.uint64(0x27D4EB2D)) & M
    s = s ^ (s >> np.uint64(15))
    s = (np.uint64(1103515245) * s + np.uint64(12345)) & M
    # (float)(*seed): u32 -> binary32, round to nearest even; times the float literal 2^-32 (exact scaling)
    v = s.astype(np.float32).astype(np.float64) * 2.3283064365386963e-10
    return s, v


def _dot(a, b):
    return a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1] + a[..., 2] * b[..., 2]


def _cross(a, b):
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                     a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], axis=-1)


def _normalize(a):
    return a / np.sqrt(_dot(a, a))[..., None]


def generate_ray(xc, yc, W, H, seed):
    """GenerateColors.cl:263-288"""
    inv_w, inv_h = 1.0 / W, 1.0 / H
    aspect = W / H
    fov = _f((60.0 * np.pi) / 180.0)            # :267, evaluated in double, narrowed to float
    angle = np.tan(0.5 * fov)
    eye = np.array([0.0, 2.75, 4.0])
    center = eye + np.array([0.0, 0.0, -1.0])
    up = np.array([0.0, 1.0, 0.0])
    view = _normalize(center - eye)
    hol = _normalize(_cross(view, up))
    upd = _normalize(_cross(hol, view))
    seed, r1 = random_float(seed)
    seed, r2 = random_float(seed)
    x = xc + r1 - 0.5
    y = yc + r2 - 0.5
    x = (2.0 * ((x + 0.5) * inv_w) - 1.0) * angle * aspect
    y = -(1.0 - 2.0 * ((y + 0.5) * inv_h)) * angle
    d = _normalize(x[:, None] * hol + (-1.0 * y)[:, None] * upd + view)
    aimed = eye + 4.0 * d
    org = np.broadcast_to(eye, d.shape).copy()
    return org, _normalize(_normalize(aimed - eye)), seed   # :287, then getRay's own normalize (:75)


def intersect_world(o, d, P1, E1, E2, scale):
    """GenerateColors.cl:89-154 for N rays against T triangles.  Returns (hit index or -1, t, u, v, margin)."""
    pvec = _cross(d[:, None, :], E2[None, :, :])                 # N x T x 3
    det = _dot(E1[None, :, :], pvec)
    keep = ~((det < 1e-8) | (-det > 1e-8))                       # :100 (1e-8f as a value: 1e-8 within 2^-24 relative, immaterial here)
    area = np.sqrt(_dot(E1, E1) * _dot(E2, E2))[None, :] + 1e-300
    m_det = np.abs(det - _f(1e-8)) / area
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        inv = 1.0 / det
        tvec = o[:, None, :] - P1[None, :, :]
        u = _dot(tvec, pvec) * inv
        qvec = _cross(tvec, E1[None, :, :])
        v = _dot(d[:, None, :], qvec) * inv
        t = _dot(E2[None, :, :], qvec) * inv
    in_u = keep & ~((u < 0.0) | (u > 1.0))                      # :109
    in_v = in_u & ~((v < 0.0) | (u + v > 1.0))                  # :117
    ok = in_v & (t > 0.0) & (t < 1e20)                          # :125 against the initial tmax
    big = np.inf
    # Distances to the boundaries in units of the OPERANDS' magnitude (what a rounding error scales with): a comparison
    # x < y of two binary32 results whose terms have magnitude S can only come out differently in another arithmetic if
    # |x - y| is within a modest multiple of 2^-24 S.  u = Nu / det etc.: the boundaries are Nu = 0, Nu = det, Nv = 0,
    # Nu + Nv = det, Nt = 0.  Only decisions that are really taken count (a triangle culled at :100 decides nothing more).
    nt, npv, nq = np.sqrt(_dot(tvec, tvec)), np.sqrt(_dot(pvec, pvec)), np.sqrt(_dot(qvec, qvec))
    ne1, ne2 = np.sqrt(_dot(E1, E1))[None, :], np.sqrt(_dot(E2, E2))[None, :]
    Nu, Nv, Nt = _dot(tvec, pvec), _dot(d[:, None, :], qvec), _dot(E2[None, :, :], qvec)
    s_det = ne1 * npv + 1e-300
    m = np.abs(det - _f(1e-8)) / s_det
    m = np.where(np.isfinite(m), m, 0.0)
    s_u = nt * npv + 1e-300
    mu = np.where(keep, np.minimum(np.abs(Nu) / s_u, np.abs(Nu - det) / (s_u + s_det)), big)
    s_v = nq + 1e-300                                            # |d| = 1
    mv = np.where(in_u, np.minimum(np.abs(Nv) / s_v, np.abs(Nu + Nv - det) / (s_u + s_v + s_det)), big)
    mt = np.where(in_v, np.abs(Nt) / (ne2 * nq + 1e-300), big)
    margin = np.minimum(np.minimum(m, mu), np.minimum(mv, mt)).min(axis=1)
    tt = np.where(ok, t, np.inf)
    # the ascending loop with the strict t < tmax keeps the first of equal t: argmin takes the first minimum too
    idx = np.argmin(tt, axis=1)
    tbest = tt[np.arange(len(o)), idx]
    hit = np.isfinite(tbest)
    second = np.partition(tt, 1, axis=1)[:, 1] if tt.shape[1] > 1 else np.full(len(o), np.inf)
    with np.errstate(invalid="ignore"):
        m_close = np.where(hit & np.isfinite(second), (second - tbest) / np.maximum(second, 1e-300), big)
    margin = np.minimum(margin, m_close)
    n = np.arange(len(o))
    return np.where(hit, idx, -1), tbest, u[n, idx], v[n, idx], margin


def trace(tris, mats, gids, frames, W, H, max_bounces=BOUNCES):
    """Paths (gid[k], frame[k]) of GenerateColors (:302-312 + traceRays :223-261), no accumulation.

    Returns radiance (n, 3) float64, hits (n, max_bounces + 1) int32 in the oracle's convention (triangle per bounce, then
    -1 = missed / -2 = pdf <= 0, -3 beyond the end), margin (n,): the path's smallest distance to any decision boundary,
    sens (n,): the sum over the path's GGX bounces of tan(incidence) / sin(theta) + 1 / (xi (r^2 - 1) + 1) (module docstring),
    slack (n,): min over the path's decisions of margin / (estimated rounding error of a binary32 evaluation at that decision:
    a first-order propagation of origin / direction errors along the path) -- a binary32 evaluation that follows the reference
    can only take another decision where `slack` is of order 1.
    """
    P1 = tris["p1"][:, :3].astype(np.float64)
    E1 = tris["p2"][:, :3].astype(np.float64) - P1               # :92
    E2 = tris["p3"][:, :3].astype(np.float64) - P1               # :93
    NORM = _cross(E2, E1)                                        # :123
    tri_id = tris["id"].astype(np.int64)
    alb = mats["albedo"][:, :3].astype(np.float64)
    emi = mats["emissive"][:, :3].astype(np.float64)
    rough = mats["roughness"].astype(np.float64)
    mtype = mats["type"].astype(np.int64)
    scale = float(np.abs(np.concatenate([P1, P1 + E1, P1 + E2])).max()) + 1.0

    gids = np.asarray(gids, np.int64)
    n = gids.size
    seed = (gids.astype(np.uint64) + hash_u32(np.asarray(frames, np.int64).astype(np.uint64))) & np.uint64(0xFFFFFFFF)  # :308
    o, d, seed = generate_ray((gids % W).astype(np.float64), (gids // W).astype(np.float64), W, H, seed)

    rad = np.zeros((n, 3))
    mask = np.ones((n, 3))
    hits = np.full((n, max_bounces + 1), -3, np.int32)
    margin = np.full(n, np.inf)
    sens = np.zeros(n)
    e_p = np.zeros(n)             # estimated absolute error of a binary32 evaluation's ray origin / direction (see `slack`)
    e_d = np.full(n, 8.0 * U32)
    slack = np.full(n, np.inf)    # min over the path's decisions of margin / (rounding-error estimate at that decision)
    live = np.ones(n, bool)
    for b in range(max_bounces):
        L = np.nonzero(live)[0]
        if L.size == 0:
            break
        oo, dd = o[L], d[L]
        idx, t, u, v, mg = intersect_world(oo, dd, P1, E1, E2, scale)
        margin[L] = np.minimum(margin[L], mg)
        # the ray a binary32 evaluation holds at this bounce differs from this one by ~e_p in the origin and ~e_d in the
        # direction: its decisions can differ where a (normalised) margin is within a few times (e_p + e_d S) / S + 8 u
        err_here = 8.0 * U32 + (e_p[L] + e_d[L] * scale) / scale
        slack[L] = np.minimum(slack[L], mg / err_here)
        miss = idx < 0
        # :233-237
        rad[L[miss]] += mask[L[miss]] * _f(0.45)
        hits[L[miss], b] = -1
        live[L[miss]] = False
        H_ = ~miss
        L, oo, dd, idx, t, u, v = L[H_], oo[H_], dd[H_], idx[H_], t[H_], u[H_], v[H_]
        if L.size == 0:
            continue
        hits[L, b] = idx
        mat = tri_id[idx]                                                     # :239
        rad[L] += mask[L] * emi[mat] * 3.0                                    # :241
        p = oo + dd * t[:, None]                                              # :128
        Nn = NORM[idx]
        nrm = _normalize(u[:, None] * Nn + v[:, None] * Nn + (1.0 - u - v)[:, None] * Nn)   # :130
        dn = _dot(nrm, dd)
        margin[L] = np.minimum(margin[L], np.abs(dn))
        err_here = err_here[H_]
        slack[L] = np.minimum(slack[L], np.abs(dn) / err_here)
        # the hit point: p = o + d t, t itself known to (e_p + e_d t) / |cos(incidence)|
        e_p[L] = (e_p[L] + e_d[L] * t) * (1.0 + 1.0 / np.maximum(np.abs(dn), 1e-12)) + 4.0 * U32 * scale
        nrm = np.where((dn < 0.0)[:, None], nrm, -nrm)                        # :243
        wo = -dd
        s_ = seed[L]
        s_, r1 = random_float(s_)                                             # phi's draw first (:163, :182)
        s_, r2 = random_float(s_)
        seed[L] = s_
        phi = TWO_PI * r1
        spec = mtype[mat] == 2
        r2m1 = rough[mat] * rough[mat] - 1.0
        with np.errstate(divide="ignore", invalid="ignore"):
            den = r2 * r2m1 + 1.0
            cos_t = np.where(spec, np.sqrt((1.0 - r2) / den), np.sqrt(1.0 - r2))            # :184 | :171
            sin_t = np.where(spec, np.sqrt(np.maximum(0.0, 1.0 - cos_t * cos_t)), np.sqrt(r2))  # :185 | :165
            tan_a = np.sqrt(np.maximum(0.0, 1.0 - dn * dn)) / np.maximum(np.abs(dn), 1e-300)
            sens[L] += np.where(spec, tan_a / np.maximum(sin_t, 1e-300) + 1.0 / np.abs(den), 0.0)
            # the next direction: a cosine sample depends on the normal and the draws only (fresh rounding); a GGX reflection
            # carries the incoming direction's error (twice: reflect) plus the half vector's tilt error 2^-25 / sin(theta)
            e_d[L] = np.where(spec, 2.0 * e_d[L] + 2.0 * U32 / np.maximum(sin_t, 1e-300) + 8.0 * U32, 8.0 * U32)
        # (fabs(n.x) > 0.001 of a unit normal that is known to ~1e-7: in units of the threshold, so that an axis-aligned wall
        # -- n.x = 0 exactly, as far from this boundary as a normal can be -- does not cap every path's margin at 1e-3)
        margin[L] = np.minimum(margin[L], np.abs(np.abs(nrm[:, 0]) - _f(0.001)) / _f(0.001))
        slack[L] = np.minimum(slack[L], np.abs(np.abs(nrm[:, 0]) - _f(0.001)) / (8.0 * U32))
        axis = np.where((np.abs(nrm[:, 0]) > _f(0.001))[:, None], np.array([0.0, 1.0, 0.0]), np.array([1.0, 0.0, 0.0]))
        tv = _normalize(_cross(axis, nrm))
        sv = _cross(nrm, tv)
        sd = _normalize(sv * (np.cos(phi) * sin_t)[:, None] + tv * (np.sin(phi) * sin_t)[:, None] + nrm * cos_t[:, None])
        # Brdf (:195-221)
        wi = np.where(spec[:, None], -wo + 2.0 * _dot(wo, sd)[:, None] * sd, sd)           # reflect (:156-159)
        dwin = _dot(wi, nrm)
        dwon = _dot(wo, nrm)
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            r2_ = rough[mat] * rough[mat]
            D = r2_ * INV_PI / (cos_t * cos_t * (r2_ - 1.0) + 1.0) ** 2                     # :174-178
            pdf_s = D * cos_t / (4.0 * _dot(wo, sd))
            col_s = (D / (4.0 * dwin * dwon))[:, None] * alb[mat] * 2.0
        early = spec & (dwin * dwon < 0.0)                                                  # :211: pdf stays 0
        margin[L] = np.minimum(margin[L], np.where(spec, np.abs(dwin * dwon), np.inf))
        slack[L] = np.minimum(slack[L], np.where(spec, np.abs(dwin * dwon), np.inf) / (err_here + e_d[L]))
        pdf = np.where(spec, np.where(early, 0.0, pdf_s), dwin * INV_PI)
        col = np.where(spec[:, None], np.where(early[:, None], 0.0, col_s), alb[mat] * INV_PI)
        margin[L] = np.minimum(margin[L], np.where(early, np.inf, np.abs(pdf)))
        slack[L] = np.minimum(slack[L], np.where(early | spec, np.inf, np.abs(dwin)) / (8.0 * U32))
        ends = pdf <= 0.0                                                                   # :251 (a NaN pdf does not end the path)
        hits[L[ends], b + 1] = -2
        live[L[ends]] = False
        go = ~ends
        Lg = L[go]
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            mask[Lg] = mask[Lg] * (col[go] * dwin[go][:, None] / pdf[go][:, None])          # :253-255
        o[Lg] = p[go] + wi[go] * _f(0.01)                                                   # :257
        d[Lg] = _normalize(wi[go])
    rad = np.where(rad < 0.0, 0.0, rad)   # max(radiance, 0.0f) = (a < b) ? b : a  (:260; a NaN stays)
    return rad, hits, margin, sens, slack
