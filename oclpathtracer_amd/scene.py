"""Scene ingestion for the path-tracing hot path.

Mirrors the host half of the reference harness:

* ``load_model``      -- the ``cornellbox.bin`` reader + material table + quad split of
                         ``test/RaytraceTest.cpp:87-198`` (``loadModel``).
* ``TRIANGLE_DTYPE`` / ``MATERIAL_DTYPE`` -- the packed 64-byte device records of
                         ``test/RaytraceTest.cpp:50-76`` / ``test/ClKernels/GenerateColors.cl:12-28``.
* ``make_soup``       -- the synthetic 1M-triangle scene of BASELINE.json configs[4]
                         (definition: SURVEY.md S8d, "C5").

Fields the reference leaves uninitialised (``Material.roughness`` of diffuse surfaces, all
padding) are zero here.
"""
from __future__ import annotations

import os
import struct

import numpy as np

DIFFUSE = 1
SPECULAR = 2

TRIANGLE_DTYPE = np.dtype(
    [("p1", "<f4", 4), ("p2", "<f4", 4), ("p3", "<f4", 4), ("id", "<i4"), ("pad", "u1", 12)]
)
MATERIAL_DTYPE = np.dtype(
    [("albedo", "<f4", 4), ("emissive", "<f4", 4), ("roughness", "<f4"), ("type", "<i4"), ("pad", "u1", 24)]
)
assert TRIANGLE_DTYPE.itemsize == 64 and MATERIAL_DTYPE.itemsize == 64

DEFAULT_SCENE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "cornellbox.bin")


def parse_meshes(blob: bytes):
    """Decode the little-endian mesh stream (RaytraceTest.cpp:117-143).

    ``i32 nMesh`` then per mesh ``i32 nFaces, f32 albedoTag, nFaces x 4 i32, i32 nVerts,
    nVerts x 4 f32``.  Returns a list of ``(tag, idx[nFaces,4], vtx[nVerts,4])``.
    """
    off = 0

    def take(fmt):
        nonlocal off
        if off + struct.calcsize(fmt) > len(blob):
            raise ValueError("scene file truncated")
        (v,) = struct.unpack_from(fmt, blob, off)
        off += struct.calcsize(fmt)
        return v

    meshes = []
    n = take("<i")
    if n < 0:
        raise ValueError("negative mesh count")
    for _ in range(n):
        nf = take("<i")
        tag = np.float32(take("<f"))
        if nf < 0 or off + 16 * nf > len(blob):
            raise ValueError("scene file truncated (faces)")
        idx = np.frombuffer(blob, "<i4", nf * 4, off).reshape(nf, 4).copy()
        off += 16 * nf
        nv = take("<i")
        if nv < 0 or off + 16 * nv > len(blob):
            raise ValueError("scene file truncated (vertices)")
        vtx = np.frombuffer(blob, "<f4", nv * 4, off).reshape(nv, 4).copy()
        off += 16 * nv
        if nf and (idx.min() < 0 or idx.max() >= nv):
            raise ValueError("face index out of range")
        meshes.append((tag, idx, vtx))
    return meshes


def reference_material_table(n_meshes: int = 6, tags=None) -> list:
    """The per-mesh materials RaytraceTest.cpp:145-176 hard-codes by mesh index, as data: a list of
    ``{"albedo", "emissive", "roughness", "type"}`` dicts -- the format of a material side file."""
    table = []
    for i in range(n_meshes):
        tag = np.float32(0.5) if tags is None else np.float32(tags[i])
        m = {"albedo": [0.0, 0.0, 0.0, 0.0], "emissive": [0.0, 0.0, 0.0, 1.0], "roughness": 0.0, "type": "diffuse"}
        if tag != np.float32(0.5):
            m["emissive"] = [30.0, 30.0, 30.0, 1.0]
            m["albedo"] = [1.0, 1.0, 1.0, 1.0]
        if i in (0, 1, 2):
            m["albedo"] = [0.7, 0.7, 0.7, 1.0]
        if i == 3:
            m["albedo"] = [0.6, 0.0, 0.0, 1.0]
        if i == 4:
            m["albedo"] = [0.0, 0.6, 0.0, 1.0]
        if i == 5:
            m["albedo"] = [0.5, 0.35, 0.05, 0.0]
            m["roughness"] = 0.008
            m["type"] = "specular"
        table.append(m)
    return table


def _material_from_entry(e: dict) -> np.ndarray:
    m = np.zeros((), MATERIAL_DTYPE)
    kind = e.get("type", "diffuse")
    if kind not in ("diffuse", "specular"):
        raise ValueError("material type must be 'diffuse' or 'specular', got %r" % (kind,))
    for key in ("albedo", "emissive"):
        v = list(e.get(key, (0.0, 0.0, 0.0, 1.0)))
        if len(v) == 3:
            v.append(1.0)
        if len(v) != 4:
            raise ValueError("material %s needs 3 or 4 components" % key)
        m[key] = tuple(float(x) for x in v)
    m["roughness"] = float(e.get("roughness", 0.0))
    m["type"] = SPECULAR if kind == "specular" else DIFFUSE
    return m


def load_model(path: str | None = None, materials=None):
    """Return ``(triangles, materials)`` as structured arrays (36 / 18 for the Cornell box).

    ``materials``: None = the reference's hard-coded assignment (below); otherwise a per-mesh
    table -- a list of dicts as :func:`reference_material_table` returns, or the path of a JSON file
    holding such a list (materials from a side file instead of mesh-index ifs).

    Material assignment follows RaytraceTest.cpp:145-176: every mesh is DIFFUSE; a mesh whose
    tag != 0.5 is the light (emissive 30, albedo 1, then overridden to 0.7 for meshes 0-2);
    mesh 3 is red, mesh 4 green, mesh 5 the glossy boxes (SPECULAR, roughness 0.008, albedo
    (0.5, 0.35, 0.05, 0)).  Each quad (a,b,c,d) becomes triangles (a,b,c) and (c,d,a) sharing
    one material id (:186-193).
    """
    with open(path or DEFAULT_SCENE, "rb") as f:
        meshes = parse_meshes(f.read())
    if isinstance(materials, (str, os.PathLike)):
        import json

        with open(materials) as f:
            materials = json.load(f)
    if materials is not None and len(materials) != len(meshes):
        raise ValueError("material table has %d entries for %d meshes" % (len(materials), len(meshes)))
    tris = []
    mats = []
    mat_id = 0
    for i, (tag, idx, vtx) in enumerate(meshes):
        m = np.zeros((), MATERIAL_DTYPE)
        m["type"] = DIFFUSE
        if materials is not None:
            m = _material_from_entry(materials[i])
        elif tag != np.float32(0.5):
            m["emissive"] = (30.0, 30.0, 30.0, 1.0)
            m["albedo"] = (1.0, 1.0, 1.0, 1.0)
        else:
            m["emissive"] = (0.0, 0.0, 0.0, 1.0)
        if materials is None:
            if i in (0, 1, 2):
                m["albedo"] = (0.7, 0.7, 0.7, 1.0)
            if i == 3:
                m["albedo"] = (0.6, 0.0, 0.0, 1.0)
            if i == 4:
                m["albedo"] = (0.0, 0.6, 0.0, 1.0)
            if i == 5:
                m["albedo"] = (0.5, 0.35, 0.05, 0.0)
                m["roughness"] = 0.008
                m["type"] = SPECULAR
        for a, b, c, d in idx:
            p = [np.array([vtx[k][0], vtx[k][1], vtx[k][2], 0.0], np.float32) for k in (a, b, c, d)]
            for q in ((p[0], p[1], p[2]), (p[2], p[3], p[0])):
                t = np.zeros((), TRIANGLE_DTYPE)
                t["p1"], t["p2"], t["p3"] = q
                t["id"] = mat_id
                tris.append(t)
            mats.append(m.copy())
            mat_id += 1
    triangles = np.array(tris, TRIANGLE_DTYPE) if tris else np.zeros(0, TRIANGLE_DTYPE)
    materials = np.array(mats, MATERIAL_DTYPE) if mats else np.zeros(0, MATERIAL_DTYPE)
    if len(triangles) // 2 != len(materials):  # the reference's only check (:197)
        raise ValueError("triangle/material count mismatch")
    return triangles, materials


def _splitmix64(n: int, seed: int) -> np.ndarray:
    """n successive splitmix64 outputs (vectorised, two buffers reused in place: fresh pages are
    what costs time on a 10^7-element array)."""
    with np.errstate(over="ignore"):
        z = np.arange(1, n + 1, dtype=np.uint64)
        z *= np.uint64(0x9E3779B97F4A7C15)
        z += np.uint64(seed)
        t = np.empty_like(z)
        for shift, mul in ((30, 0xBF58476D1CE4E5B9), (27, 0x94D049BB133111EB)):
            np.right_shift(z, np.uint64(shift), out=t)
            z ^= t
            z *= np.uint64(mul)
        np.right_shift(z, np.uint64(31), out=t)
        z ^= t
        return z


def make_soup(ntri: int = 1_000_000, seed: int = 20261004, path: str | None = None):
    """Cornell box + synthetic small-triangle soup.

    Triangles 0..35 are the Cornell box; triangle i >= 36 has 24-bit uniforms
    u = (x >> 40) * 2^-24 drawn from splitmix64(seed): centre c, p1 = c,
    p2 = c + (2u-1)^3 * 0.02, p3 = c + (2u-1)^3 * 0.02, id = 18 + (i-36)//2;
    materials 18.. are DIFFUSE with albedo 0.2 + 0.6u, emissive 0.
    """
    base_t, base_m = load_model(path)
    if ntri < len(base_t):
        raise ValueError("ntri must be >= 36")
    extra = ntri - len(base_t)
    nm_extra = (extra + 1) // 2
    u = (_splitmix64(9 * extra + 3 * nm_extra, seed) >> np.uint64(40)).astype(np.float32) * np.float32(2.0**-24)
    ut = u[: 9 * extra].reshape(extra, 9)
    um = u[9 * extra :].reshape(nm_extra, 3)
    tris = np.zeros(ntri, TRIANGLE_DTYPE)
    tris[: len(base_t)] = base_t
    lo = np.array([-2.7, 0.05, -5.5], np.float32)
    span = np.array([5.4, 5.35, 5.4], np.float32)
    c = lo + span * ut[:, 0:3]
    d2 = (np.float32(2.0) * ut[:, 3:6] - np.float32(1.0)) * np.float32(0.02)
    d3 = (np.float32(2.0) * ut[:, 6:9] - np.float32(1.0)) * np.float32(0.02)
    t = tris[len(base_t) :]
    t["p1"][:, :3] = c
    t["p2"][:, :3] = c + d2
    t["p3"][:, :3] = c + d3
    t["id"] = 18 + np.arange(extra, dtype=np.int32) // 2
    mats = np.zeros(len(base_m) + nm_extra, MATERIAL_DTYPE)
    mats[: len(base_m)] = base_m
    m = mats[len(base_m) :]
    m["albedo"][:, :3] = np.float32(0.2) + np.float32(0.6) * um
    m["albedo"][:, 3] = 1.0
    m["emissive"][:, 3] = 1.0
    m["type"] = DIFFUSE
    return tris, mats


def f2c(v: np.ndarray) -> np.ndarray:
    """``f2c(sqrtf(v))`` of the reference's PPM writer (RaytraceTest.cpp:78-83, 280-285):
    ``a *= 255; min((int)a, 255)`` with the x86 float->int conversion (NaN / out-of-range give
    INT_MIN).  Returned as int32: the reference prints its u32 with %d."""
    with np.errstate(invalid="ignore"):
        a = np.sqrt(np.asarray(v, np.float32)) * np.float32(255)
        bad = np.isnan(a) | (a >= np.float32(2147483648.0)) | (a < np.float32(-2147483648.0))
        i = np.where(bad, 0.0, np.trunc(a)).astype(np.int64)
    i = np.where(bad, -2147483648, i)
    return np.minimum(i, 255).astype(np.int32)


def write_ppm(path: str, fb: np.ndarray, W: int, H: int) -> None:
    """Text 'P3' PPM exactly as RaytraceTest.cpp:277-287 writes it."""
    c = f2c(fb.reshape(H * W, 4)[:, :3])
    with open(path, "w") as f:
        f.write("P3\n%d %d\n%d\n" % (W, H, 255))
        f.write("".join("%d %d %d " % (r, g, b) for r, g, b in c))


def write_ppm_binary(path: str, rgb: np.ndarray, W: int, H: int) -> None:
    """Binary 'P6' PPM of the device tonemap's output (``pt_tonemap_ppm``: int32 r,g,b per pixel, the
    values the reference prints as text).  One byte per channel: the reference's INT_MIN for a NaN /
    overflowed pixel (RaytraceTest.cpp:78-83) has no byte representation and is written as 0."""
    c = np.asarray(rgb).reshape(H * W, 3)
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (W, H))
        f.write(np.clip(c, 0, 255).astype(np.uint8).tobytes())
