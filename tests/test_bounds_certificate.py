"""csrc/rpt_bounds_certify.hpp: the screen bounds the kernel culls with are PROVEN, not sampled.  Host code only (no GPU).

* every wrong proposal of rounds 1-3 (tests/golden/historic_wrong_bounds.json, rebuilt from the git history by
  tests/golden/make_historic_claims.py: the header as it stood before each fix, on the scenes of the soak finds) is REJECTED;
* soundness against the oracle: whatever claim the proof accepts — the proposer's, a shrunken or shifted copy of it, a random
  rectangle or octagon — contains every pixel the oracle (the kernel's float arithmetic, restated) hits;
* the proof is not vacuous: it accepts every proposal on the shipped scenes, and rejects claims that cut into an object;
* inputs the error model does not cover (non-finite matrices, an origin inside the shape, a boost whose directions do not
  cover the sphere) prove nothing."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_ffi  # noqa: F401  (builds the oracle)
import test_screen_bounds as tsb
from conftest import CONFIGS, load_config
from relativitypathtracer_amd import Scene, _ffi

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FULL = 3.0e38


def _root(scene, objs, i):
    if int(objs["type"][i]) != 2:
        return None
    n = scene.octrees()[int(objs["meshIndex"][i])]
    return (C.c_float * 6)(*n["min"][:3], *n["max"][:3])


def certify(raw, interval, root, bounds):
    st = (C.c_int * 4)()
    ok = _ffi.hip().rpt_certify_screen_bounds(raw.ctypes.data, interval, root, (C.c_float * 8)(*bounds), st)
    return bool(ok), tuple(st)


def proposed(raw, interval, root):
    b = (C.c_float * 8)()
    assert _ffi.hip().rpt_object_screen_bounds_proposed(raw.ctypes.data, interval, root, b) == 0
    return tuple(b)


def claims_nothing(b):
    return b[0] <= -FULL and b[1] <= -FULL and b[2] >= FULL and b[3] >= FULL and b[4] <= -FULL and b[5] >= FULL and b[6] <= -FULL and b[7] >= FULL


def test_every_historic_wrong_claim_is_rejected():
    with open(os.path.join(GOLDEN, "historic_wrong_bounds.json")) as f:
        claims = json.load(f)["claims"]
    assert len(claims) >= 15
    kinds = set()
    for c in claims:
        raw = np.frombuffer(bytes.fromhex(c["object_hex"]), dtype=np.uint8).copy()
        root = (C.c_float * 6)(*c["root_bounds"]) if c["root_bounds"] else None
        ok, st = certify(raw, c["interval"], root, c["bounds"])
        assert not ok, f"{c['find']} (object {c['object_index']}, header {c['header_commit']}): a claim that loses {c['lost_pixels']} hit pixels was 'proven' {st}"
        kinds.add(c["find"].split(":")[0])
        # what the library hands the kernel for the same object today: proven, or the full plane
        b = (C.c_float * 8)()
        assert _ffi.hip().rpt_object_screen_bounds(raw.ctypes.data, c["interval"], root, b) == 0
        if not claims_nothing(tuple(b)):
            assert certify(raw, c["interval"], root, tuple(b))[0]
    assert len(kinds) >= 6, kinds          # every class of find of rounds 1-3 is represented


def _grid(W, H):
    ys, xs = np.mgrid[0:H, 0:W]
    return (xs / W - 0.5) * (W / H), ys / H - 0.5


def _variants(rng, b):
    """Claims derived from a proposal: itself, each side pulled in, the whole thing shifted, random rectangles, octagon cuts."""
    u0, v0, u1, v1 = (max(-2.5, min(2.5, x)) for x in b[:4])
    out = [tuple(b)]
    if u0 > u1 or v0 > v1:                  # "not visible at all": that claim, and made-up rectangles
        u0, v0, u1, v1 = -0.3, -0.2, 0.3, 0.2
    w, h = max(u1 - u0, 1e-3), max(v1 - v0, 1e-3)
    for side in range(4):
        for frac in (0.02, 0.1, 0.3, 0.6):
            c = list(b)
            c[side] = c[side] + (frac * (w if side % 2 == 0 else h)) * (1 if side < 2 else -1)
            out.append(tuple(c))
    for _ in range(6):
        du, dv = rng.normal(scale=0.1 * w), rng.normal(scale=0.1 * h)
        out.append((b[0] + du, b[1] + dv, b[2] + du, b[3] + dv) + tuple(b[4:]))
    for _ in range(6):
        a, c_ = sorted(rng.uniform(-1.2, 1.2, size=2))
        d, e = sorted(rng.uniform(-0.6, 0.6, size=2))
        out.append((a, d, c_, e, -FULL, FULL, -FULL, FULL))
    for _ in range(4):          # diagonal cuts of the proposal's box
        p0, p1 = sorted(rng.uniform(u0 + v0, u1 + v1, size=2))
        m0, m1 = sorted(rng.uniform(u0 - v1, u1 - v0, size=2))
        out.append((b[0], b[1], b[2], b[3], p0, p1, m0, m1))
    out.append((FULL, FULL, -FULL, -FULL, -FULL, FULL, -FULL, FULL))       # "not visible at all"
    return out


def _soundness(scene, label, rng, frames=((192, 108), (150, 40))):
    objs = scene.objects()
    interval = scene.params["interval"]
    accepted = rejected = 0
    for i in range(min(len(objs), 64)):
        raw, root = objs[i:i + 1].copy(), _root(scene, objs, i)
        hits = [(tsb.hit_mask(scene, i, W, H),) + _grid(W, H) for (W, H) in frames]
        for claim in _variants(rng, proposed(raw, interval, root)):
            if claims_nothing(claim):
                continue
            ok, st = certify(raw, interval, root, claim)
            if not ok:
                rejected += 1
                continue
            accepted += 1
            for hit, u, v in hits:
                bad = hit & ~tsb.inside_bounds(claim, u, v)
                assert not bad.any(), f"{label}: object {i}: PROVEN claim {claim} leaves out {int(bad.sum())} hit pixels ({st})"
    return accepted, rejected


@pytest.mark.parametrize("name", list(CONFIGS))
def test_whatever_is_proven_contains_every_hit_shipped_scenes(name):
    rng = np.random.default_rng(31)
    acc, rej = _soundness(load_config(name), name, rng)
    assert acc > 0 and rej > 0, (acc, rej)


@pytest.mark.parametrize("seed", range(24))
def test_whatever_is_proven_contains_every_hit_generated_scenes(seed):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import verify_fuzz
    kind = ("random", "extreme", "close", "walls", "ellipsoids", "meshwalls")[seed % 6]
    scene, text = verify_fuzz.build(kind, 300 + seed)
    _soundness(scene, f"{kind} {300 + seed}\n{text}", np.random.default_rng(seed), frames=((160, 90),))


def test_every_proposal_on_the_shipped_scenes_is_proven():
    """The proof must not cost the cull: on the scenes the reference ships — at rest and from moving cameras — every proposal
    is accepted (a rejected one means the full plane for that object: correct, and slower)."""
    import math
    claims = proven = 0
    for name in ("cube", "arch", "bunny", "shadows", "cubes", "soccer", "rulers", "ladder_paradox"):
        s = Scene.from_file(name)
        for k in range(12):
            f = k / 11
            speed = 0.95 * f
            ang, el = 2.0 * math.pi * 3.0 * f, 0.6 * math.sin(2.0 * math.pi * 5.0 * f)
            s.set_camera((speed * math.cos(el) * math.sin(ang), speed * math.sin(el), speed * math.cos(el) * math.cos(ang)), 30.0 * f)
            s.update_objects()
            objs = s.objects()
            for i in range(min(len(objs), 64)):
                raw, root = objs[i:i + 1].copy(), _root(s, objs, i)
                b = proposed(raw, s.params["interval"], root)
                if claims_nothing(b):
                    continue
                claims += 1
                ok, st = certify(raw, s.params["interval"], root, b)
                proven += ok
                assert ok or st[0] in (6, 7), (name, k, i, st)          # (budget / no margin: legitimate; anything else here is a bug)
    assert claims > 500 and proven >= 0.995 * claims, (claims, proven)


def test_a_claim_that_cuts_into_the_object_is_rejected_with_a_reason():
    s = load_config("bunny")
    objs = s.objects()
    seen = set()
    for i in range(len(objs)):
        raw, root = objs[i:i + 1].copy(), _root(s, objs, i)
        b = proposed(raw, -1, root)
        if claims_nothing(b):
            continue
        mid_u, mid_v = 0.5 * (b[0] + b[2]), 0.5 * (b[1] + b[3])
        for claim in ((mid_u, b[1], b[2], b[3]), (b[0], b[1], b[2], mid_v), (FULL, FULL, -FULL, -FULL)):
            ok, st = certify(raw, -1, root, claim + (-FULL, FULL, -FULL, FULL))
            assert not ok
            seen.add(st[0])
    assert seen <= {5, 6, 7} and 7 in seen, seen          # witness outside the claim / a boundary point's ray meets the shape


def test_inputs_outside_the_error_model_prove_nothing():
    s = load_config("shadows")
    objs = s.objects()
    cube = next(i for i in range(len(objs)) if int(objs["type"][i]) == 1)
    ok_claim = proposed(objs[cube:cube + 1].copy(), -1, None)
    assert certify(objs[cube:cube + 1].copy(), -1, None, ok_claim)[0]
    for field, idx, value, reasons in (("InvM", (0, 0), np.nan, {1}), ("Lorentz", (1, 1), np.inf, {1}), ("Lorentz", (1, 0), 50.0, {2, 3, 5, 6, 7}),
                                       ("InvM", (0, 0), 1e20, {1})):
        o = objs[cube:cube + 1].copy()
        m = o[field][0].reshape(4, 4).copy()
        m[idx] = value
        o[field][0] = m.reshape(o[field][0].shape)
        ok, st = certify(o, -1, None, ok_claim)
        assert not ok and st[0] in reasons, (field, idx, value, st)
    # an origin inside the cube: nothing is claimed by the proposer and nothing is proven for a made-up claim
    o = objs[cube:cube + 1].copy()
    sc = o["stationaryCam"][0].copy()
    M = o["M"][0].reshape(4, 4)
    sc[1:4] = M[:3, 3]                       # the camera event's position = the cube's centre
    o["stationaryCam"][0] = sc
    assert claims_nothing(proposed(o, -1, None))
    ok, st = certify(o, -1, None, (-0.1, -0.1, 0.1, 0.1, -FULL, FULL, -FULL, FULL))
    assert not ok and st[0] == 4, st


def test_corrupted_object_records_never_yield_an_unsound_bound():
    """Object records with damaged matrices — single bit flips, entries scaled by 1e+-20, rows swapped, InvM no longer the inverse of M,
    a Lorentz matrix that is no boost — go through what rpt_set_objects runs (proposal + proof).  Whatever bound comes out is either
    the full plane or contains every pixel the oracle hits with the SAME damaged record: the proof reads the matrices the kernel
    reads and assumes nothing about where they came from."""
    rng = np.random.default_rng(99)
    lib = _ffi.hip()
    W, H = 128, 72
    u, v = _grid(W, H)
    used = kept = 0
    for trial in range(240):
        name = ("shadows", "cubes", "arch", "soccer", "bunny")[trial % 5]
        s = load_config(name)
        objs = s.objects().copy()
        i = int(rng.integers(0, min(len(objs), 8)))
        field = ("M", "InvM", "Lorentz", "InvLorentz", "stationaryCam")[int(rng.integers(0, 5))]
        raw = objs[field][i].reshape(-1).copy()
        kind = int(rng.integers(0, 5))
        k = int(rng.integers(0, raw.size))
        if kind == 0:
            bits = raw.view(np.uint32).copy()
            bits[k] ^= np.uint32(1) << np.uint32(rng.integers(0, 32))
            raw = bits.view(np.float32)
        elif kind == 1:
            raw[k] *= np.float32(10.0 ** rng.choice([-20, -6, -2, 2, 6, 20]))
        elif kind == 2 and raw.size == 16:
            m = raw.reshape(4, 4).copy()
            a, b = rng.choice(4, size=2, replace=False)
            m[[a, b]] = m[[b, a]]
            raw = m.reshape(-1)
        elif kind == 3:
            raw[k] = np.float32(rng.normal(scale=3.0))
        else:
            raw[k] = -raw[k]
        objs[field][i] = raw.reshape(objs[field][i].shape)
        rec = objs[i:i + 1].copy()
        root = _root(s, objs, i)
        b = (C.c_float * 8)()
        assert lib.rpt_object_screen_bounds(rec.ctypes.data, s.params["interval"], root, b) == 0       # (must return, whatever the record holds)
        b = tuple(b)
        used += 1
        if claims_nothing(b):
            continue
        kept += 1
        only = rec.copy()
        only["light"] = 0
        only["textureIndex"] = -1
        only["flashPeriod"] = 0
        only["color"] = (1.0, 0.5, 0.25, 0.0)
        _, rgb, _ = oracle_ffi.render(s, W, H, objects=only)
        hit = ~((rgb == tsb._background(s, W, H)).all(axis=2))          # (a NaN colour is a hit too: it is not the background)
        bad = hit & ~tsb.inside_bounds(b, u, v)
        assert not bad.any(), f"trial {trial} ({name} object {i}, {field}[{k}] kind {kind}): a proven bound {b} leaves out {int(bad.sum())} hit pixels"
    assert used == 240 and kept > 40, (used, kept)
