"""SURVEY.md §8b "Errors": "a failing GPU call must fall back internally to the C implementation — never abort, never return garbage".
Every seam of the bound encoder (x265_amd/host/*.cpp) has the reference's own host function one branch away; this test makes each device entry
point the seams use fail — from its first call on, and from a later call on (a device that dies in the middle of the encode) — through the
emulated ABI's failure injection (tests/support/la_emul.c, X265HIP_EMUL_FAIL) and checks that the encoder (a) keeps running, (b) says once what
happened, (c) still produces the unmodified reference's bytes; and that X265HIP=require turns the same event into a fatal error (what bench.py,
the tools and the GPU tests run with: a number measured on a fallback is worthless)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
sys.path.insert(0, ROOT)

W, H, FRAMES = 640, 384, 12
ARGS = ["--input-res", "%dx%d" % (W, H), "--input-depth", "8", "--fps", "30", "--frames", str(FRAMES), "--pools", "4", "-F", "2", "--hash", "1", "--preset", "medium",
        "--bframes", "2", "--keyint", "8", "--min-keyint", "8"]

# entry point -> (module that reports it, call numbers from which it fails)
POINTS = {
    "la_create": ("lookahead", [1]), "la_set_frame": ("lookahead", [1, 9]), "la_weights": ("lookahead", [1, 3]), "la_put_vectors": ("lookahead", [1]),
    "la_estimate": ("lookahead", [1, 4]),
    "refpic_create": ("refplanes", [1, 3]), "refpic_reset": ("refplanes", [1, 4]), "rows_final": ("refplanes", [1, 17]),
    "source_energy": ("srcplanes", [1, 8]), "srcpic_create": ("srcplanes", [1, 3]), "srcpic_upload": ("srcplanes", [1, 5]),
    "sadsurf_attach": ("sadplanes", [1, 4]),
    "cuserve_open": ("cuserve", [1]), "cuserve_submit": ("cuserve", [1, 50]), "cuserve_job": ("cuserve", [1, 37]),
    "cuserve_submit_sao": ("saostats", [1, 20]), "saostats_job": ("saostats", [1, 11]),
    "cuserve_submit_intra": ("intrascan", [1, 15]), "intrascan_job": ("intrascan", [1, 9]),
}
OPTIONAL = {"la_put_vectors", "la_weights"}          # not reached by every clip: the byte comparison still counts, the message is not demanded


def _need(name):
    p = os.path.join(REF, name)
    if not os.path.exists(p):
        pytest.skip("%s not built (needs /root/reference at build time: make -C oracle ref emul)" % name)
    return p


@pytest.fixture(scope="module")
def clip_and_reference(tmp_path_factory):
    ref = _need("x265_8bit")
    _need("x265_emul_8bit")
    from x265_amd.synth import make_clip
    d = tmp_path_factory.mktemp("fallback")
    yuv = str(d / "clip.yuv")
    make_clip(yuv, W, H, FRAMES, seed=31, fade=True)
    out = str(d / "ref.hevc")
    r = subprocess.run([ref, "--input", yuv] + ARGS + ["-o", out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-600:]
    return yuv, open(out, "rb").read(), d


def _encode(yuv, d, env_extra, tag):
    out = str(d / (tag + ".hevc"))
    env = dict(os.environ, X265HIP_VERBOSE="1", **env_extra)
    env.pop("X265HIP", None)
    if "X265HIP" in env_extra:
        env["X265HIP"] = env_extra["X265HIP"]
    r = subprocess.run([os.path.join(REF, "x265_emul_8bit"), "--input", yuv] + ARGS + ["-o", out], capture_output=True, text=True, timeout=600, env=env)
    return r, (open(out, "rb").read() if os.path.exists(out) else b"")


def test_without_failures_the_seams_serve(clip_and_reference):
    yuv, want, d = clip_and_reference
    r, got = _encode(yuv, d, {}, "plain")
    assert r.returncode == 0 and got == want, r.stderr[-600:]
    for word in ("lookahead:", "refplanes:", "srcplanes:", "sadplanes:", "cuserve:", "saostats:", "intrascan:"):
        assert any(l.startswith("x265hip: " + word) for l in r.stderr.splitlines()), (word, r.stderr[-1200:])
    assert "OFF from here on" not in r.stderr


@pytest.mark.parametrize("point", sorted(POINTS))
def test_a_failing_device_call_falls_back_to_the_reference_code(clip_and_reference, point):
    yuv, want, d = clip_and_reference
    module, froms = POINTS[point]
    for n in froms:
        r, got = _encode(yuv, d, {"X265HIP_EMUL_FAIL": "%s:%d" % (point, n)}, "%s_%d" % (point, n))
        assert r.returncode == 0, (point, n, r.stderr[-800:])
        assert got == want, "bitstream differs after %s failed from call %d on" % (point, n)
        # (a service that has lost a job is dead for both kinds of jobs that share it — CU jobs and SAO statistics jobs: the other module says so too)
        said = [l for l in r.stderr.splitlines() if "OFF from here on" in l and ("x265hip: %s:" % module) in l]
        if point not in OPTIONAL or said:
            assert len(said) == 1 and "emulated failure of " + point in said[0], (point, n, r.stderr[-800:])


def test_require_makes_a_device_failure_fatal(clip_and_reference):
    yuv, want, d = clip_and_reference
    r, got = _encode(yuv, d, {"X265HIP_EMUL_FAIL": "rows_final:5", "X265HIP": "require"}, "require")
    assert r.returncode != 0
    assert "X265HIP=require: fatal" in r.stderr
