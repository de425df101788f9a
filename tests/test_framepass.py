"""GPU parity of the frame pass (x265hip_framepass_*): every output bit-exact against the C restatement
(oracle/x265_oracle_frame.c), from a 200x136 picture with 8x8 TU strips up to the BASELINE 1080p configuration, and a
two-frame chain where the second frame searches the first frame's border-extended reconstruction."""
import os
import sys

import numpy as np
import pytest

from frame_oracle import make_scene, make_scene_yuv, oracle_frame_pass, same_results

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fpmod():
    from x265_amd import hipprim as hp, framepass
    assert hp.lib().x265hip_device_count() > 0
    hp.check(hp.lib().x265hip_init(0))
    return framepass


@pytest.mark.parametrize("depth,method,subme,qp", [(8, 1, 2, 28), (10, 1, 2, 30), (8, 0, 3, 22), (8, 1, 5, 35), (10, 1, 7, 24), (8, 3, 3, 28), (10, 3, 2, 26), (8, 2, 2, 28), (10, 2, 3, 30)])
def test_small_frame_pass_is_bit_exact(fpmod, depth, method, subme, qp):
    sc = make_scene(200, 136, depth=depth, seed=11 + qp, tile=48, sigma=3.0 * (1 if depth == 8 else 4))
    fp = fpmod.FramePass(200, 136, depth=depth, qp=qp, merange=57, method=method, subme=subme)
    got = fp.run_host(sc["src"], sc["ref"])
    want = oracle_frame_pass(sc["src"], sc["ref"], depth=depth, qp=qp, merange=57, method=method, subme=subme)
    assert same_results(got, want) == []
    assert sum(int(x.sum()) for x in want["numSig"]) > 0          # the transform path was exercised


def test_1080p_frame_pass_is_bit_exact(fpmod):
    """BASELINE.json configs[1]: 1920x1080, --me hex, merange 57, subme 2 (preset medium)."""
    sc = make_scene(1920, 1080, depth=8, seed=4321)
    fp = fpmod.FramePass(1920, 1080, depth=8, qp=28)
    got = fp.run_host(sc["src"], sc["ref"])
    want = oracle_frame_pass(sc["src"], sc["ref"], depth=8, qp=28)
    assert same_results(got, want) == []


def test_two_frame_chain_uses_recon_as_reference(fpmod):
    from x265_amd.framepass import Plane
    w, h, depth, qp = 328, 200, 8, 26
    a = make_scene(w, h, depth=depth, seed=5, tile=64)
    b = make_scene(w, h, depth=depth, seed=6, tile=64)
    fp = fpmod.FramePass(w, h, depth=depth, qp=qp)
    s1, r0 = Plane(w, h, depth, a["src"]), Plane(w, h, depth, a["ref"])
    s2 = Plane(w, h, depth, b["src"])
    p, rec1, rec2 = Plane(w, h, depth), Plane(w, h, depth), Plane(w, h, depth)
    fp.run(s1, r0, p, rec1)
    fp.run(s2, rec1, p, rec2)                                     # frame 2 references frame 1's recon, all on device
    got = fp.results()
    got["pred"], got["recon"] = p.get(), rec2.get(with_margins=True)
    o1 = oracle_frame_pass(a["src"], a["ref"], depth=depth, qp=qp)
    m = 96
    rec1_host = np.ascontiguousarray(o1["recon"][m:m + h, m:m + w])
    # the oracle pads by edge replication, which is what extend_border produced on the device
    assert np.array_equal(o1["recon"], np.pad(rec1_host, m, mode="edge"))
    want = oracle_frame_pass(b["src"], rec1_host, depth=depth, qp=qp)
    assert same_results(got, want) == []


def test_mvcost_table_matches_oracle(fpmod):
    """host BitCost::setQP restatement in the product == pinned oracle table (also covered on CPU in test_host_logic)."""
    from x265_amd import hipprim as hp
    from oracle import pyoracle as po
    t = np.zeros(4 * 32768 + 1, np.uint16)
    for qp in (0, 17, 28, 51):
        for depth in (8, 10):
            hp.check(hp.lib().x265hip_mvcost_table(qp, depth, t.ctypes.data, 2 * 32768))
            assert np.array_equal(t, po.mvcost_table(qp, depth))


@pytest.mark.parametrize("depth", [8, 10])
def test_subpel_planes_match_reference_filters(fpmod, depth):
    """x265hip_build_subpel_planes: plane[yF*4+xF] == luma_hpp / luma_vpp / luma_hvpp of the oracle on the whole padded picture."""
    from backends import Orc
    from x265_amd import hipprim as hp
    from x265_amd.hipprim import DevBuf, check
    w, h, m = 136, 72, 96
    rng = np.random.default_rng(9 + depth)
    pic = rng.integers(0, 1 << depth, size=(h + 2 * m, w + 2 * m)).astype(hp.pix_dtype(depth))
    S = w + 2 * m
    d = DevBuf(pic)
    planes = DevBuf.zeros((16, h + 2 * m, S), pic.dtype)
    org = (m * S + m) * pic.itemsize
    check(hp.lib().x265hip_build_subpel_planes(depth, d.ptr + org, S, w, h, m, m, planes.ptr + org, (h + 2 * m) * S, None))
    got = planes.get()
    o = Orc(depth)
    lo = 4                                                   # computed region: everything but the outermost 4 rows / columns
    H, W = h + 2 * m, w + 2 * m
    bad = []
    for yf in range(4):
        for xf in range(4):
            want = np.zeros((H, W), pic.dtype)
            for y0 in range(lo, H - lo, 64):
                for x0 in range(lo, W - lo, 64):
                    bh, bw = min(64, H - lo - y0), min(64, W - lo - x0)
                    bw -= bw % 4
                    if not (xf | yf):
                        blk = pic[y0:y0 + bh, x0:x0 + bw]
                    elif not yf:
                        blk = o.interp("hpp", 0, bw, bh, pic, (y0, x0), xf)
                    elif not xf:
                        blk = o.interp("vpp", 0, bw, bh, pic, (y0, x0), yf)
                    else:
                        blk = o.interp("hvpp", 0, bw, bh, pic, (y0, x0), xf, yf)
                    want[y0:y0 + bh, x0:x0 + bw] = blk
            if not np.array_equal(got[yf * 4 + xf][lo:H - lo, lo:W - lo], want[lo:H - lo, lo:W - lo]):
                bad.append((xf, yf, int((got[yf * 4 + xf][lo:H - lo, lo:W - lo] != want[lo:H - lo, lo:W - lo]).sum())))
    assert not bad, bad


@pytest.mark.parametrize("depth,qp", [(8, 30), (10, 32)])
def test_4k_frame_pass_is_bit_exact(fpmod, depth, qp):
    """BASELINE.json configs[2]/[3] picture size (3840x2160, 8-bit and Main10): every output against the CPU restatement."""
    sc = make_scene(3840, 2160, depth=depth, seed=77 + depth, sigma=3.0 * (1 if depth == 8 else 4))
    fp = fpmod.FramePass(3840, 2160, depth=depth, qp=qp)
    got = fp.run_host(sc["src"], sc["ref"])
    want = oracle_frame_pass(sc["src"], sc["ref"], depth=depth, qp=qp)
    assert same_results(got, want) == []


def test_8k_frame_pass_properties(fpmod):
    """BASELINE.json configs[4] picture size (7680x4320): size-independent properties instead of the (slow) CPU pass:
    a static scene gives zero vectors, zero levels and recon == reference; a globally shifted scene is found exactly; the
    recon margins are edge replicas; running twice is deterministic."""
    w, h, depth = 7680, 4320, 8
    rng = np.random.default_rng(8)
    base = make_scene(1920 + 64, 1080 + 64, depth=depth, seed=8, sigma=0.0)["ref"]
    big = np.tile(base, (5, 5))[: h + 64, : w + 64]
    ref = np.ascontiguousarray(big[32:32 + h, 32:32 + w])
    fp = fpmod.FramePass(w, h, depth=depth, qp=28)
    still = fp.run_host(ref, ref)
    assert all(not x.any() for x in still["mv"]) and all(not x.any() for x in still["numSig"])
    m = 96
    assert np.array_equal(still["recon"][m:-m, m:-m], ref)
    assert np.array_equal(still["recon"], np.pad(ref, m, mode="edge"))
    # global shift by (+3, -2) full pels: every interior 16x16 PU must find (12, -8) quarter-pels with zero SAD residual cost
    src = np.ascontiguousarray(big[32 - 2:32 - 2 + h, 32 + 3:32 + 3 + w])
    a = fp.run_host(src, ref)
    b = fp.run_host(src, ref)
    assert all(np.array_equal(x, y) for x, y in zip(a["mv"], b["mv"])) and np.array_equal(a["recon"], b["recon"])
    mv16 = a["mv"][2].reshape(h // 16, w // 16, 2)
    inner = mv16[4:-4, 4:-4].reshape(-1, 2)
    assert (inner == np.array([12, -8])).all(axis=1).mean() > 0.999


@pytest.mark.parametrize("w,h,depth,qp,subme", [(200, 136, 8, 28, 2), (200, 136, 10, 33, 3), (328, 200, 8, 40, 2), (1920, 1080, 8, 28, 2)])
def test_yuv_frame_pass_is_bit_exact(fpmod, w, h, depth, qp, subme):
    """4:2:0 pass: chroma prediction (predInterChromaPixel), chroma residual chain (16x16 / 4x4 TUs, chroma QpParam), chroma recon
    borders — on top of the luma pass, every output against the C restatement."""
    sc = make_scene_yuv(w, h, depth=depth, seed=21 + qp, tile=48 if w < 1000 else 96, sigma=3.0 * (1 if depth == 8 else 4))
    fp = fpmod.FramePass(w, h, depth=depth, qp=qp, subme=subme)
    got = fp.run_host_yuv(sc)
    want = oracle_frame_pass(sc["src"], sc["ref"], depth=depth, qp=qp, subme=subme, src_c=(sc["src_cb"], sc["src_cr"]), ref_c=(sc["ref_cb"], sc["ref_cr"]))
    assert same_results(got, want) == []
    assert sum(int(x.sum()) for x in want["cnumSig"]) > 0


@pytest.mark.parametrize("w,h,depth,qp,method,subme", [(200, 136, 8, 28, 3, 3), (328, 200, 10, 30, 1, 4), (640, 360, 8, 26, 3, 3), (200, 136, 8, 28, 2, 3)])
def test_yuv_frame_pass_with_chroma_satd_search(fpmod, w, h, depth, qp, method, subme):
    """BASELINE configs[2]/[3] shape of the search: STAR / HEX at subme 3-4 on a 4:2:0 picture, where every sub-pel comparison of
    motionEstimate carries the chroma SATD term — the whole pass against the C restatement."""
    sc = make_scene_yuv(w, h, depth=depth, seed=77 + qp, tile=48, sigma=3.0 * (1 if depth == 8 else 4))
    fp = fpmod.FramePass(w, h, depth=depth, qp=qp, method=method, subme=subme)
    got = fp.run_host_yuv(sc)
    want = oracle_frame_pass(sc["src"], sc["ref"], depth=depth, qp=qp, method=method, subme=subme, src_c=(sc["src_cb"], sc["src_cr"]),
                             ref_c=(sc["ref_cb"], sc["ref_cr"]))
    assert same_results(got, want) == []
    luma_only = oracle_frame_pass(sc["src"], sc["ref"], depth=depth, qp=qp, method=method, subme=subme)
    assert any(not np.array_equal(a, b) for a, b in zip(want["mv"], luma_only["mv"]))      # the chroma term does steer the search


@pytest.mark.parametrize("w,h,depth,qp,method,subme", [(200, 136, 8, 28, 1, 2), (328, 200, 10, 30, 1, 3), (200, 136, 8, 35, 3, 2), (1920, 1080, 8, 28, 1, 2)])
def test_b_frame_pass_is_bit_exact(fpmod, w, h, depth, qp, method, subme):
    """The B variant: both lists searched at every level (list 1 against the next picture), bi-predictive prediction of luma and chroma from
    the two lists' 8x8 vectors (two 14-bit predictions + addAvg), then the usual chains — every output incl. list 1's vectors / costs."""
    sc = make_scene_yuv(w, h, depth=depth, seed=91 + qp, tile=48 if w < 1000 else 96, sigma=3.0 * (1 if depth == 8 else 4))
    nxt = make_scene_yuv(w, h, depth=depth, seed=91 + qp, tile=48 if w < 1000 else 96, sigma=5.0 * (1 if depth == 8 else 4), vmax=5)
    # the "future" reference: the same content generator with other motion / noise (the source of that scene)
    r1 = (nxt["src"], nxt["src_cb"], nxt["src_cr"])
    fp = fpmod.FramePass(w, h, depth=depth, qp=qp, method=method, subme=subme)
    got = fp.run_host_yuv_b(sc, *r1)
    want = oracle_frame_pass(sc["src"], sc["ref"], depth=depth, qp=qp, method=method, subme=subme, src_c=(sc["src_cb"], sc["src_cr"]),
                             ref_c=(sc["ref_cb"], sc["ref_cr"]), ref1=r1[0], ref1_c=(r1[1], r1[2]))
    assert same_results(got, want) == []
    assert any(int(np.abs(m).sum()) for m in want["mv1"])


@pytest.mark.parametrize("depth", [8, 10])
def test_pred_inter_chroma_matches_oracle(fpmod, depth):
    from oracle import pyoracle as po
    from x265_amd import hipprim as hp
    from x265_amd.hipprim import DevBuf, check, dev_i32
    L, O = hp.lib(), po.oracle()
    rng = np.random.default_rng(31 + depth)
    H, W = 120, 160                                             # chroma plane incl. margins; luma coordinates are twice that
    refs = [rng.integers(0, 1 << depth, size=(H, W)).astype(hp.pix_dtype(depth)) for _ in range(2)]
    fn = getattr(O, "orc_pred_inter_chroma_%s" % po.sfx(depth))
    fn.restype, fn.argtypes = None, [po.vp, po.ip, po.vp, po.ip] + [po.i32] * 7
    for (lw, lh) in [(8, 8), (16, 16), (32, 32), (64, 64), (16, 8), (8, 16), (32, 24), (24, 32), (64, 48), (8, 32)]:
        n = 19
        bx = (rng.integers(12, W - 12 - lw // 2, n)) * 2
        by = (rng.integers(12, H - 12 - lh // 2, n)) * 2
        mv = rng.integers(-60, 61, size=(n, 2)).astype(np.int32)
        bad = []
        for i in range(n):
            dref = [DevBuf(r) for r in refs]
            dd = [DevBuf.zeros((H, W), refs[0].dtype) for _ in range(2)]
            xy, q = dev_i32([int(bx[i]), int(by[i])]), dev_i32(mv[i])
            check(L.x265hip_pred_inter_chroma_batch(depth, lw, lh, dref[0].ptr, dref[1].ptr, W, dd[0].ptr, dd[1].ptr, W, xy.ptr, q.ptr, 1, None))
            for pl in range(2):
                want = np.zeros((H, W), refs[0].dtype)
                fn(po.ptr(refs[pl]), W, po.ptr(want), W, int(bx[i]), int(by[i]), lw, lh, int(mv[i, 0]), int(mv[i, 1]), depth)
                if not np.array_equal(dd[pl].get(), want):
                    bad.append((lw, lh, i, pl))
        assert not bad, bad[:5]


def test_graph_replay_matches_plain_launches(fpmod):
    """X265HIP_GRAPH=1: the frame pass captured as a hipGraph (after the first plain run) and replayed must leave the same outputs as plain
    launches — a chain of four passes alternating two reconstruction buffers, in a subprocess because the switch is read once per process."""
    import subprocess, sys, os, hashlib
    code = r"""
import ctypes as C, hashlib, sys
import numpy as np
sys.path.insert(0, %r)
from x265_amd import hipprim as hp
from x265_amd.framepass import FramePass, Picture
from x265_amd.synth import make_scene_yuv
L = hp.lib(); hp.check(L.x265hip_init(0))
w, h = 328, 200
sc = make_scene_yuv(w, h, depth=8, seed=5, tile=48)
src = Picture(w, h, 8, sc["src"], sc["src_cb"], sc["src_cr"]); ref = Picture(w, h, 8, sc["ref"], sc["ref_cb"], sc["ref_cr"])
pred = Picture(w, h, 8); rec = [Picture(w, h, 8), Picture(w, h, 8)]
fp = FramePass(w, h, depth=8, qp=30)
st = C.c_void_p(); hp.check(L.x265hip_stream_create(C.byref(st)))
cur = ref
hsh = hashlib.sha256()
for k in range(6):
    fp.run_yuv(src, cur, pred, rec[k & 1], st)
    hp.check(L.x265hip_stream_sync(st))
    cur = rec[k & 1]
    hsh.update(cur.y.get(True).tobytes()); hsh.update(cur.cb.get(True).tobytes()); hsh.update(fp.fetch(1, 3).tobytes())
print(hsh.hexdigest())
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for graph in (False, True):
        env = dict(os.environ)
        env.pop("X265HIP_GRAPH", None)
        if graph:
            env["X265HIP_GRAPH"] = "1"
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr[-1500:]
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1] and len(outs[0]) == 64


def _golden_cases():
    import json
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "framepass_configs_golden.json")
    return json.load(open(p))["cases"]


@pytest.mark.parametrize("name", sorted(_golden_cases()))
def test_baseline_configs_match_golden(fpmod, name):
    """BASELINE.json configs[2], [3] and [4] AS CONFIGURED — 3840x2160 STAR merange 57 subme 3 with the chroma SATD term, 3840x2160 Main10 STAR
    subme 4, 7680x4320 HEX — plus the 1080p B pass and two Main12 cases: the HIP frame pass on the seeded scene, every output hashed and compared
    with the digests the pinned CPU oracle produced (tests/golden/make_framepass_golden.py; the oracle needs minutes per case, the GPU test only hashes)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_framepass_golden as mg
    case = _golden_cases()[name]
    p = case["params"]
    key = tuple(p[k] for k in ("width", "height", "depth", "qp", "method", "merange", "subme", "seed", "pass"))
    assert mg.CASES[name] == key, "golden file and generator disagree: regenerate"
    sc, nxt = mg.scene(key)
    fp = fpmod.FramePass(p["width"], p["height"], depth=p["depth"], qp=p["qp"], merange=p["merange"], method=p["method"], subme=p["subme"])
    got = fp.run_host_yuv_b(sc, *nxt) if nxt else fp.run_host_yuv(sc)
    dg = mg.digest(got)
    bad = sorted(k for k in case["digests"] if dg.get(k) != case["digests"][k])
    assert not bad, bad
    assert set(dg) == set(case["digests"])
    fp.close()
