"""Host-side scene front-end: DSL parser, OBJ import, octree builder,
texture pool, per-frame Lorentz refresh.

Known answers come from SURVEY.md §8(a), which recorded them from the reference's own loader and
octree builder: node / octreeTris / leaf counts, nodes per depth, triangles per leaf.
"""
import hashlib
import os

import numpy as np
import pytest

from relativitypathtracer_amd import Scene, SceneError


def depth_histogram(oct, root):
    hist, level = [], [root]
    while level:
        hist.append(len(level))
        nxt = []
        for i in level:
            if oct["children"][i][0] != -1:
                nxt.extend(int(c) for c in oct["children"][i])
        level = nxt
    return hist


# scene sizes and octree shape measured from the reference's own builder
SURVEY = {
    "bunny": dict(objects=2, verts=2503, tri_words=44712, nodes=15281, octree_tris=78519, leaves=13371, empty=5227,
                  max_leaf=11, leaf_refs=32764, per_depth=[1, 8, 64, 360, 1640, 5784, 7424], tex=28311552),
    "shadows": dict(objects=5, verts=1426, tri_words=25632, nodes=5913, octree_tris=37794, leaves=5174, empty=2088,
                    max_leaf=47, leaf_refs=14440, per_depth=[1, 8, 64, 344, 1352, 2296, 1848], tex=0, uvs=1602),
}


@pytest.mark.parametrize("name", list(SURVEY))
def test_octree_matches_reference_builder_counts(name):
    k = SURVEY[name]
    s = Scene.from_file(name)
    d = s.desc()
    assert d.object_count == k["objects"] and d.vertex_count == k["verts"] and d.normal_count == k["verts"]
    assert d.triangle_words == k["tri_words"] and d.octree_count == k["nodes"] and d.octree_tri_count == k["octree_tris"]
    assert d.texture_bytes == k["tex"]
    if "uvs" in k:
        assert d.uv_count == k["uvs"]
    oct = s.octrees()
    leaves = oct[oct["children"][:, 0] == -1]
    assert len(leaves) == k["leaves"] and int((leaves["trisCount"] == 0).sum()) == k["empty"]
    assert int(leaves["trisCount"].max()) == k["max_leaf"] and int(leaves["trisCount"].sum()) == k["leaf_refs"]
    assert depth_histogram(oct, s.mesh_roots()[0]) == k["per_depth"]


def test_octree_structure_invariants():
    s = Scene.from_file("bunny")
    oct = s.octrees()
    inner = oct[oct["children"][:, 0] != -1]
    # the eight children are consecutive (what the derived device layout relies on) and come after the parent
    assert np.all(inner["children"] == inner["children"][:, :1] + np.arange(8))
    idx = np.nonzero(oct["children"][:, 0] != -1)[0]
    assert np.all(inner["children"][:, 0] > idx)
    # neighbour links are symmetric in extent: a neighbour on side s is at least as large as the node
    root = oct[s.mesh_roots()[0]]
    assert np.all(root["neighbors"] == -1)
    # children tile the parent: child min/max inside parent bounds
    for p in idx[:200]:
        ch = oct[oct["children"][p]]
        assert np.all(ch["min"][:, :3] >= oct["min"][p][:3] - 1e-6) and np.all(ch["max"][:, :3] <= oct["max"][p][:3] + 1e-5)
    # vt-less mesh: every uv index is 0 and the uvs array was padded to one (0,0) entry
    b = s.buffers()
    assert b["uvs"].shape == (1, 2) and not b["uvs"].any()
    assert np.all(b["triangles"][1::3] == 0)
    assert np.all(b["vertices"][:, 3] == 0)


def test_buffer_hashes_are_stable():
    """Regression pin of the exact bytes the front-end produces (self-generated; see SURVEY counts above)."""
    s = Scene.from_file("bunny")
    b = s.buffers()
    h = hashlib.sha256()
    for k in ("vertices", "normals", "triangles", "octrees", "octreeTris"):
        h.update(np.ascontiguousarray(b[k]).tobytes())
    digest = h.hexdigest()
    path = os.path.join(os.path.dirname(__file__), "golden", "bunny_buffers.sha256")
    if not os.path.exists(path):
        pytest.skip("golden hash missing")
    assert digest == open(path).read().strip()


def test_scene_dsl_semantics():
    # t0 may precede its T line (Scenes/cube.txt); I sets interval 0; defaults white point 1, ambient 1
    s = Scene.from_file("cube")
    o = s.objects()
    assert len(o) == 1 and o["type"][0] == 1 and s.params == {"white_point": [1.0, 1.0, 1.0], "ambient": 1.0, "interval": 0}
    assert o["textureIndex"][0] == 0 and (o["textureWidth"][0], o["textureHeight"][0]) == (224, 225)
    assert s.desc().texture_bytes == 224 * 225 * 3
    np.testing.assert_array_equal(o["M"][0], np.array([[1, 0, 0, 0], [0, 1, 0, -2], [0, 0, 1, 4], [0, 0, 0, 1]], np.float32))
    # light flag, colour, ambient, velocity
    s = Scene.from_file("shadows")
    o = s.objects()
    assert list(o["type"]) == [0, 1, 0, 1, 2] and list(o["light"]) == [1, 0, 0, 0, 0]
    np.testing.assert_allclose(o["color"][0][:3], [10, 10, 10])
    np.testing.assert_allclose(s.velocities()[0][:3], [0.95, 0, 0])
    assert s.params["white_point"] == [10.0, 10.0, 10.0] and abs(s.params["ambient"] - 0.2) < 1e-7
    assert o["meshIndex"][4] == s.mesh_roots()[0] and o["textureIndex"][4] == -1
    # flash period/duration
    o = Scene.from_file("rulers").objects()
    assert list(o["flashPeriod"]) == [2.0, 2.0] and list(o["flashDuration"]) == [1.0, 1.0]
    # 34 cubes, second texture-less line ordering variants (t before p)
    assert len(Scene.from_file("cubes").objects()) == 34


def test_scene_errors_are_reported_not_fatal():
    s = Scene()
    with pytest.raises(SceneError, match="out of range"):
        s.inputScene("Os\n t3\nR\n")
    s = Scene()
    with pytest.raises(SceneError):
        s.inputScene("TTextures/does_not_exist.jpg\nR\n")
    s = Scene()
    with pytest.raises(SceneError, match="Mesh index"):
        s.inputScene("Om0\nR\n")
    s = Scene()
    diag = s.inputScene("p1,2,3,0,0,1,0,1,1,1\nZ\nOs\nR\n")      # command before any object, unknown command: diagnostics only
    assert "Object must be defined" in diag and "Unrecognized" in diag and len(s.objects()) == 1
    s = Scene()
    s.inputScene("Os\nOc\n")                                      # no R: input ends at EOF
    assert len(s.objects()) == 2


def test_case_insensitive_and_aliased_assets():
    s = Scene()
    s.inputScene("MModels/PEAR.OBJ\nOm0\nR\n".replace("PEAR.OBJ", "Pear.obj"))
    assert s.desc().vertex_count == 1426
    s = Scene()
    s.inputScene("MModels/StanfordBunny.obj\nOm0\nR\n")          # aliased to Models/bunny.obj
    assert s.desc().vertex_count == 2503


def test_obj_import_with_vt_vn_and_second_mesh_quirk():
    s = Scene()
    s.inputScene("MModels/cube.obj\nMModels/triangle.obj\nOm0\nOm1\nR\n")
    b = s.buffers()
    d = s.desc()
    assert d.vertex_count == 8 + 3 and d.triangle_words == 9 * 13
    tri = b["triangles"].reshape(-1, 3, 3)
    np.testing.assert_array_equal(tri[0], [[0, 0, 0], [1, 1, 0], [2, 2, 0]])      # f 1/1/1 2/2/1 3/3/1
    # the second mesh's indices are offset by the first mesh's array sizes
    assert tri[12][:, 0].min() >= 8
    roots = s.mesh_roots()
    oct = s.octrees()
    # reference quirk (Mesh.cpp:16-19): the second root lists ALL triangles imported so far
    assert oct["trisCount"][roots[0]] == 12 and oct["trisCount"][roots[1]] == 13
    # vn normals are normalised on import
    n = b["normals"][:, :3]
    np.testing.assert_allclose(np.linalg.norm(n, axis=1), 1.0, rtol=1e-6)


def test_smooth_normals_for_vn_less_mesh():
    s = Scene.from_file("bunny")
    b = s.buffers()
    n = b["normals"][:, :3]
    assert n.shape[0] == 2503
    np.testing.assert_allclose(np.linalg.norm(n, axis=1), 1.0, rtol=2e-6)
    # normal k belongs to vertex k (ascending vertex order) and every corner points at its vertex's normal
    tri = b["triangles"].reshape(-1, 3, 3)
    np.testing.assert_array_equal(tri[:, :, 2], tri[:, :, 0])
    # area-weighted average of face normals, in float64, agrees to fp32 accuracy
    v = b["vertices"][:, :3].astype(np.float64)
    fn = np.cross(v[tri[:, 1, 0]] - v[tri[:, 0, 0]], v[tri[:, 2, 0]] - v[tri[:, 0, 0]])
    acc = np.zeros_like(v)
    for c in range(3):
        np.add.at(acc, tri[:, c, 0], fn)
    acc /= np.linalg.norm(acc, axis=1, keepdims=True)
    assert np.abs(acc - n).max() < 1e-4


def test_lorentz_update_properties():
    s = Scene.from_file("shadows")
    s.set_camera((0.3, -0.2, 0.5), 7.5)
    s.update_objects()
    o = s.objects()
    eta = np.diag([-1.0, 1, 1, 1])
    for k in range(len(o)):
        L, Li = o["Lorentz"][k].astype(np.float64), o["InvLorentz"][k].astype(np.float64)
        np.testing.assert_allclose(L @ Li, np.eye(4), atol=2e-5)                 # inverse pair
        np.testing.assert_allclose(L.T @ eta @ L, eta, atol=5e-5)                # preserves the Minkowski metric
        cam = np.array([7.5, 0, 0, 0])
        np.testing.assert_allclose(o["stationaryCam"][k], L @ cam, rtol=1e-5, atol=1e-5)
    # stationary object, stationary camera: identity (bit exact), camera event = (t,0,0,0)
    s.set_camera((0, 0, 0), 3.0)
    s.update_objects()
    o = s.objects()
    np.testing.assert_array_equal(o["Lorentz"][1], np.eye(4, dtype=np.float32))
    np.testing.assert_array_equal(o["stationaryCam"][1], np.array([3, 0, 0, 0], np.float32))
    # boost along x with v = 0.95: gamma and the time row
    g = 1.0 / np.sqrt(1 - 0.95 ** 2)
    np.testing.assert_allclose(o["Lorentz"][0][0], [g, -0.95 * g, 0, 0], rtol=1e-6, atol=1e-7)


def test_trs_inverse_and_velocity_addition():
    s = Scene()
    s.inputScene("Oc\n p1,-2,3,0.7,1,2,3,2,0.5,4\nR\n")
    o = s.objects()
    M, Mi = o["M"][0].astype(np.float64), o["InvM"][0].astype(np.float64)
    np.testing.assert_allclose(M @ Mi, np.eye(4), atol=1e-6)
    np.testing.assert_allclose(M[:3, 3], [1, -2, 3])
    np.testing.assert_allclose(np.linalg.norm(M[:3, :3], axis=0), [2, 0.5, 4], rtol=1e-6)
    # relativistic velocity addition never reaches c; collinear case matches (u+v)/(1+uv)
    for _ in range(400):
        s.accelerate((0, 0, 1), 50)
    v, _ = s.get_camera()
    assert 0.99 < v[2] < 1.0 and v[0] == 0 and v[1] == 0
    s.reset_velocity()
    s.accelerate((1, 0, 0), 1000)
    s.accelerate((1, 0, 0), 1000)
    v, _ = s.get_camera()
    a = np.tanh(0.2)
    assert abs(v[0] - (2 * a) / (1 + a * a)) < 1e-6
    # time advances only when unpaused, and only in the t slot
    s.advance_time(500)
    assert s.get_camera()[1] == [0, 0, 0, 0]
    s.set_paused(False)
    s.advance_time(500)
    assert s.get_camera()[1] == [0.5, 0, 0, 0]
    s.toggle_interval()
    assert s.params["interval"] == 0
    s.toggle_interval()
    assert s.params["interval"] == -1


def test_write_ppm_flips_rows(tmp_path):
    from relativitypathtracer_amd import write_ppm
    from relativitypathtracer_amd.renderer import PIXEL_DTYPE
    W, H = 5, 3
    px = np.zeros(W * H, dtype=PIXEL_DTYPE)
    px["rgba"][:, 0] = np.repeat(np.arange(H), W) * 10      # red = 10 * row (row 0 = bottom)
    p = tmp_path / "f.ppm"
    write_ppm(str(p), px, W, H)
    data = p.read_bytes()
    assert data.startswith(b"P6\n5 3\n255\n")
    body = np.frombuffer(data[len(b"P6\n5 3\n255\n"):], np.uint8).reshape(H, W, 3)
    assert list(body[:, 0, 0]) == [20, 10, 0]                # top row of the file is the last framebuffer row


def test_write_png_decodes_to_the_ppm_image(tmp_path):
    """rpt_write_png (stored zlib stream, own CRC-32 / Adler-32): an independent decoder must read back exactly the
    image rpt_write_ppm writes — on a frame wider than one 65 535-byte deflate block per few rows."""
    from PIL import Image
    from relativitypathtracer_amd import write_png, write_ppm
    from relativitypathtracer_amd.renderer import PIXEL_DTYPE
    rng = np.random.default_rng(5)
    for W, H in [(5, 3), (1, 1), (700, 123)]:
        px = np.zeros(W * H, dtype=PIXEL_DTYPE)
        px["rgba"] = rng.integers(0, 256, size=(W * H, 4), dtype=np.uint8)
        write_ppm(str(tmp_path / "f.ppm"), px, W, H)
        write_png(str(tmp_path / "f.png"), px, W, H)
        a = np.asarray(Image.open(tmp_path / "f.ppm").convert("RGB"))
        with Image.open(tmp_path / "f.png") as im:
            assert im.mode == "RGB" and im.size == (W, H)
            im.verify()                                   # checks every chunk CRC
        b = np.asarray(Image.open(tmp_path / "f.png"))
        assert np.array_equal(a, b)
