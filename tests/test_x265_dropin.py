"""The drop-in proof on a real GPU: the REFERENCE's own binaries, linked with x265_amd/host/x265_hip_primitives.cpp
(our setupAssemblyPrimitives) and libx265hip.so, built into oracle/_ref by `make -C oracle hip` where /root/reference
exists (the binaries travel to the GPU box; the reference sources do not).

 1. the reference TestBench (source/test/testbench.cpp): C table vs our table on its own random / min / max inputs,
    bit-exact or it aborts — run until its correctness phase ends;
 2. BASELINE.json configs[0]: QCIF all-intra ultrafast encode — the bitstream produced with the GPU primitives behind the
    EncoderPrimitives table must be byte-identical to the C-primitive build's (the reference's regression criterion,
    source/test/regression-tests.txt:3-7)."""
import os
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")


def _need(name):
    # the product's integration builds (reference objects + the binding + libx265hip.so) live in integration/_build, the reference alone and the
    # emulated-ABI test binaries in oracle/_ref
    p = os.path.join(ROOT, "integration", "_build", name) if "_hip" in name else os.path.join(REF, name)
    if not os.path.exists(p):
        pytest.skip("%s not built (needs /root/reference at build time: make -C integration hip)" % name)
    return p


@pytest.mark.parametrize("bits", [8, 10, 12])
def test_reference_testbench_passes_with_gpu_primitives(bits):
    exe = _need("TestBench_hip%d" % bits)
    env = dict(os.environ, X265HIP_VERBOSE="1", X265HIP_TABLE="percall")
    p = subprocess.Popen([exe, "--cpuid", "SSE2"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
    lines, t0, done = [], time.time(), False
    try:
        for line in p.stdout:
            lines.append(line.rstrip())
            if "Test performance improvement" in line:
                done = True
                break
            if time.time() - t0 > 400:
                break
    finally:
        p.kill()
    out = "\n".join(lines)
    assert "has failed" not in out and "failed" not in out.lower(), out[-2000:]
    assert done, "TestBench did not finish its correctness phase: " + out[-1500:]
    assert out.count("Testing primitives:") >= 1


ENCODES = {
    # BASELINE.json configs[0]: QCIF all-intra, preset ultrafast, 8 frames
    "intra-ultrafast": (8, 8, ["--preset", "ultrafast", "--keyint", "1"]),
    # P frames: motion search, interpolation, residual coding of inter CUs
    "p-ultrafast": (8, 6, ["--preset", "ultrafast", "--bframes", "0", "--keyint", "6"]),
    # the shape of configs[1] on a small picture: preset medium (hex, subme 2, rd 3, RDOQ off … weightp, B frames with the lookahead)
    "b-medium": (8, 6, ["--preset", "medium", "--keyint", "6", "--rc-lookahead", "4", "--bframes", "2"]),
    # Main10 build of the same
    "b-medium-main10": (10, 6, ["--preset", "medium", "--keyint", "6", "--rc-lookahead", "4", "--bframes", "2"]),
    # the shape of configs[2]: preset slow --me star --merange 57 (star search, subme 3 with the chroma SATD term, rect / amp PUs, RDOQ)
    "slow-star": (8, 4, ["--preset", "slow", "--me", "star", "--merange", "57", "--keyint", "4", "--rc-lookahead", "3", "--bframes", "1"]),
    # the shape of configs[3]: Main10 preset slower --rd 6 (RDOQ level 2 and its coefficient-scan cost helpers, psy-rdoq, subme 4)
    "slower-rd6-main10": (10, 4, ["--preset", "slower", "--rd", "6", "--keyint", "4", "--rc-lookahead", "3", "--bframes", "1"]),
    # Main12 build
    "b-medium-main12": (12, 4, ["--preset", "medium", "--keyint", "4", "--rc-lookahead", "3", "--bframes", "1"]),
    # uneven multi-hexagon search
    "umh-medium": (8, 4, ["--preset", "medium", "--me", "umh", "--keyint", "4", "--rc-lookahead", "3", "--bframes", "1"]),
}


@pytest.mark.parametrize("name", sorted(ENCODES))
def test_qcif_bitstream_identical_to_c_primitives(tmp_path, name):
    """The reference encoder with the GPU table vs with its own C table: the same bytes out (the reference's regression
    criterion), for intra, P and B / lookahead encodes, 8-bit and Main10."""
    bits, frames, extra = ENCODES[name]
    c_exe, g_exe = _need("x265_%dbit" % bits), _need("x265_hip_%dbit" % bits)
    sys.path.insert(0, ROOT)
    from x265_amd.synth import make_clip
    yuv = str(tmp_path / "qcif.yuv")
    make_clip(yuv, 176, 144, frames, seed=99, tile=48)
    args = ["--input", yuv, "--input-res", "176x144", "--input-depth", "8", "--fps", "30", "--frames", str(frames), "--pools", "none", "-F", "1",
            "--hash", "1"] + extra
    outs = {}
    for tag, exe in (("c", c_exe), ("gpu", g_exe)):
        o = str(tmp_path / (tag + ".hevc"))
        env = dict(os.environ, X265HIP_VERBOSE="1", X265HIP_TABLE="percall")
        r = subprocess.run([exe] + args + ["-o", o], capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stderr[-800:]
        outs[tag] = (open(o, "rb").read(), r.stderr)
    assert len(outs["c"][0]) > 1000
    assert outs["c"][0] == outs["gpu"][0], "bitstreams differ"
    served = [l for l in outs["gpu"][1].splitlines() if "primitive calls served by the GPU" in l]
    assert served and int(served[0].split()[1]) > 1000, outs["gpu"][1][-600:]


@pytest.mark.parametrize("case", ["1080p-medium-hex", "720p-main10-fade"])
def test_lookahead_seam_on_gpu_is_byte_identical(case):
    """BASELINE.json configs[1] through the drop-in as it is meant to be used: the reference encoder + x265_hip_primitives.cpp +
    x265_hip_lookahead.cpp + libx265hip.so, default settings (C slots stay, the lookahead's batched cost estimates run on the GPU —
    INTEGRATION.md §5), all host cores.  Same bytes as the unmodified encoder, and the estimates were really served by the device."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import encode_fps
    if case == "1080p-medium-hex":
        r = encode_fps.measure(frames=24, width=1920, height=1080, bits=8, preset="medium", extra=("--me", "hex"))
    else:
        from x265_amd.synth import make_clip
        clip = "/tmp/x265hip_fade_%d.yuv" % os.getpid()
        make_clip(clip, 1280, 720, 16, seed=77, fade=True)
        try:
            r = encode_fps.measure(frames=16, width=1280, height=720, bits=10, preset="medium", extra=("--bframes", "2"), clip=clip)
        finally:
            os.remove(clip)
    if "error" in r and "not built" in r["error"]:
        pytest.skip(r["error"])
    assert "error" not in r, r
    assert r["byte_identical"], r
    served = [l for l in r["gpu"]["served"] if "frame-cost estimates" in l]
    assert served and int(served[0].split()[2]) > 10, r
    planes = [l for l in r["gpu"]["served"] if "refplanes:" in l]
    assert planes and int(planes[0].split()[2]) > 10000, r
    psy = [l for l in r["gpu"]["served"] if "srcplanes:" in l]
    assert psy and int(psy[0].split()[5]) > 10000, r


BASELINE_ENCODES = {
    # BASELINE.json configs[2]: 3840x2160 preset slow --me star --merange 57
    "configs[2] 4K slow star": (8, 3840, 2160, 8, "slow", ("--me", "star", "--merange", "57")),
    # configs[3]: 3840x2160 Main10 preset slower --rd 6
    "configs[3] 4K main10 slower rd6": (10, 3840, 2160, 6, "slower", ("--rd", "6")),
    # configs[4]: 7680x4320 preset medium (one encoder; the 8-GPU form is N of these)
    "configs[4] 8K medium": (8, 7680, 4320, 6, "medium", ()),
}


@pytest.mark.parametrize("name", sorted(BASELINE_ENCODES))
def test_baseline_configs_encode_byte_identical_through_the_seams(name, monkeypatch):
    """BASELINE.json configs[2], [3], [4] as real encodes: the reference encoder with every bound module on the GPU (4K: 32 400 lowres blocks per estimate,
    8.7 M-sample planes; 8K: 130 k blocks, 35 M-sample planes) against the unmodified encoder — same bytes, with X265HIP_VERIFY=1 (every served value is
    recomputed by the reference's function beside it: a mismatch aborts the encoder), and every module that applies to the configuration really served:
    the lookahead session, the reference-picture planes, the CU jobs (coefficient mode at the RDOQ presets of configs[2] / [3]: forward transform units
    served), SAO statistics jobs (8-bit builds), intra scan jobs (rd levels 2-4)."""
    import re
    bits, w, h, frames, preset, extra = BASELINE_ENCODES[name]
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import encode_fps
    monkeypatch.setenv("X265HIP_VERIFY", "1")
    # the Main10 configuration is fed 10-bit samples (BASELINE.json configs[3] is a 10-bit encode of 10-bit material)
    r = encode_fps.measure(frames=frames, width=w, height=h, bits=bits, preset=preset, extra=extra, seed=31, input_depth=10 if bits == 10 else 8)
    if "error" in r and "not built" in r["error"]:
        pytest.skip(r["error"])
    assert "error" not in r, r
    assert r["byte_identical"], r
    served = [l for l in r["gpu"]["served"] if "frame-cost estimates" in l]
    assert served and int(served[0].split()[2]) >= 3, r
    planes = [l for l in r["gpu"]["served"] if "refplanes:" in l]
    assert planes and int(planes[0].split()[2]) > 1000, r
    text = "\n".join(r["gpu"]["served"])
    m = re.search(r"cuserve: (\d+) CU residual quad-trees .*?: (\d+) forward transform\+quant units and (\d+) inverse units served", text)
    assert m and int(m.group(1)) > 100 and int(m.group(2)) > 100, text            # (configs[2] / [3]: rdoQuant on the host, coefficient-mode jobs: forward units only)
    if name.startswith("configs[4]"):
        assert int(m.group(3)) > 100, text                                        # preset medium: the whole chain, inverse units too
    m = re.search(r"saostats: SAO statistics of (\d+) CTU planes", text)
    if bits == 8:
        assert m and int(m.group(1)) > 100, text
    m = re.search(r"intrascan: the 35-mode sa8d scans of (\d+) blocks", text)
    if not name.startswith("configs[3]"):                                         # (--rd 6: checkIntraInInter is not on the path)
        assert m and int(m.group(1)) > 10, text


def test_8k_encode_with_two_places_is_byte_identical(monkeypatch):
    """BASELINE.json configs[4] names the multi-GPU form: 7680x4320 preset medium, frame-parallel, reconstructed reference pictures exchanged between
    the GPUs.  Inside one encoder that is X265HIP_DEVICES: mirrors and source pictures take the places in turn, SAD surfaces are built
    from replicas fed device to device.  Here with two places on the one GPU of the box — every line of the path except the xGMI hop itself — against
    the unmodified encoder: same bytes, and the exchange happened."""
    import re
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import encode_fps
    monkeypatch.setenv("X265HIP_DEVICES", "0,0")
    r = encode_fps.measure(frames=5, width=7680, height=4320, bits=8, preset="medium", extra=(), seed=33)
    if "error" in r and "not built" in r["error"]:
        pytest.skip(r["error"])
    assert "error" not in r, r
    assert r["byte_identical"], r
    ex = [l for l in r["gpu"]["served"] if "x265hip: places:" in l]
    m = ex and re.search(r"(\d+) replicas of reference pictures at other places, (\d+) bands of reconstructed rows \(([\d.]+) MB\)", ex[0])
    assert m and int(m.group(1)) > 0 and int(m.group(2)) > 0 and float(m.group(3)) > 10.0, r["gpu"]["served"]
    sad = [l for l in r["gpu"]["served"] if "x265hip: sadplanes:" in l]
    assert sad and int(sad[0].split()[2]) > 10000, r["gpu"]["served"]
