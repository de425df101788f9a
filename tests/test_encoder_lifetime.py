"""Encoders opened and closed one after another in ONE process through the x265 API (tests/support/two_encoders.cpp: five sessions, four picture
sizes, the first geometry twice): the bindings key their state by addresses of the encoder's objects — reference-picture buffers, source-picture
buffers, the Lookahead and its Lowres frames — and malloc hands those addresses out again once an encoder is closed.  The PicYuv::destroy and
Lookahead::destroy seams retire that state; without them the second session already differs from the reference and a later one crashes.
Every session's bitstream must equal the one of the same program linked against the reference's objects only."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
SESSIONS = 5


def _need(name):
    p = os.path.join(REF, name)
    if not os.path.exists(p):
        pytest.skip("%s not built (needs /root/reference at build time: make -C oracle emul hip)" % name)
    return p


def _run(exe, prefix, env=None, par=False):
    r = subprocess.run([exe, prefix] + (["par"] if par else []), capture_output=True, text=True, timeout=900, env=dict(os.environ, X265HIP_VERBOSE="1", X265HIP="require", **(env or {})))
    assert r.returncode == 0, "%s: rc %d\n%s" % (os.path.basename(exe), r.returncode, r.stderr[-600:])
    return [open("%s_%d.hevc" % (prefix, k), "rb").read() for k in range(SESSIONS)], r.stderr


def _compare(bound, tmp_path, par=False):
    ref, _ = _run(_need("two_encoders_ref8"), str(tmp_path / "ref"))
    got, log = _run(bound, str(tmp_path / "bound"), par=par)
    assert all(len(b) > 1000 for b in ref)
    for k in range(SESSIONS):
        assert ref[k] == got[k], "session %d differs from the reference" % k
    for what in ("lookahead:", "srcplanes:", "refplanes:"):
        line = [l for l in log.splitlines() if l.startswith("x265hip: " + what)]
        assert line, log[-600:]
    served = [l for l in log.splitlines() if l.startswith("x265hip: lookahead:")][0]
    assert int(served.split()[2]) > 100, served                      # every session's estimates counted, not only the last one's


def test_sequential_encoders_in_one_process_with_emulated_abi(tmp_path):
    _compare(_need("two_encoders_emul8"), tmp_path)


def test_concurrent_encoders_in_one_process_with_emulated_abi(tmp_path):
    """two, then three encoders alive at the same time (an ABR ladder's shape): one lookahead session per live encoder, mirrors and source
    entries per buffer; the outputs equal the sequential reference run's"""
    _compare(_need("two_encoders_emul8"), tmp_path, par=True)


@pytest.mark.gpu
def test_sequential_encoders_in_one_process_on_gpu(tmp_path):
    _compare(_need("two_encoders_hip8"), tmp_path)


@pytest.mark.gpu
def test_concurrent_encoders_in_one_process_on_gpu(tmp_path):
    _compare(_need("two_encoders_hip8"), tmp_path, par=True)
