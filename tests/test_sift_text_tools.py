"""CPU: the two conversions of the reference's converter program (test_main.cpp:30-60) through the C ABI and
the command-line tool.  Expected text is restated here from FeaturePoints.cpp:281-325 (`ofstream << float`
prints six significant digits, i.e. %g; every value is followed by a blank); the C++ reference classes
are not buildable in this tree (they need Windows headers), so these bytes are parity unpinned beyond
that restatement."""
import os
import subprocess

import numpy as np

import visual_slam_amd as vsm
from visual_slam_amd import host, synth

TOOL = os.path.join(os.path.dirname(vsm.lib_path()), "vsm_sift_tool")


def _expected_rows(loc, des, what):
    rows = []
    for i in range(len(des)):
        r = ""
        if what != "descriptors":
            r += "".join("%g " % float(v) for v in loc[i])
        if what != "locations":
            r += "".join("%d " % int(v) for v in des[i])
        rows.append(r + "\n")
    return "".join(rows)


def test_sift2txt_writes_the_rows_of_sift2txt_and_write_out_data(tmp_path):
    rng = np.random.default_rng(4)
    des = synth.sift_like(37, rng)
    loc = np.column_stack([rng.random(37) * 1600, rng.random(37) * 1200, np.zeros(37), rng.random(37) * 7.5 + 1e-3,
                           rng.random(37) * 6.28 - 3.14]).astype(np.float32)
    loc[0] = [0.0, 1234567.0, 1e-7, -0.5, 3.0]      # exponent forms, integers, negatives
    sift = str(tmp_path / "a.sift")
    host.write_sift(sift, loc, des)
    for what in ("both", "locations", "descriptors"):
        txt = str(tmp_path / ("a_%s.txt" % what))
        assert host.sift2txt(sift, txt, what) == 37
        assert open(txt).read() == _expected_rows(loc, des, what)


def test_txt2sift_gives_placeholder_descriptors(tmp_path):
    xy = np.array([[10.9, 20.1], [0.0, 5.0], [639.0, 479.0]], np.float32)
    txt = str(tmp_path / "k.txt")
    host.save_feature_locations(xy, txt)             # main.py:133-140 writes truncated integers
    sift = str(tmp_path / "k.sift")
    assert host.txt2sift(txt, sift) == 3
    loc, des = host.read_sift(sift)
    np.testing.assert_array_equal(loc, np.column_stack([np.trunc(xy), np.zeros((3, 3))]).astype(np.float32))
    np.testing.assert_array_equal(des, np.ones((3, 128), np.uint8))
    assert os.path.getsize(sift) == 20 + 3 * 5 * 4 + 3 * 128 + 4


def test_missing_and_malformed_inputs(tmp_path):
    import pytest
    with pytest.raises(vsm.VsmError):
        host.sift2txt(str(tmp_path / "nope.sift"), str(tmp_path / "o.txt"))
    bad = tmp_path / "bad.txt"
    bad.write_text("3 0\n1 2\n")                     # announces three rows, holds one
    with pytest.raises(vsm.VsmError):
        host.txt2sift(str(bad), str(tmp_path / "o.sift"))


def test_command_line_tool_converts_a_directory_both_ways(tmp_path):
    a, b, c = tmp_path / "txt", tmp_path / "sift", tmp_path / "back"
    for d in (a, b, c):
        d.mkdir()
    rng = np.random.default_rng(6)
    pts = [rng.integers(0, 640, (n, 2)).astype(np.float32) for n in (5, 1, 12)]
    for i, xy in enumerate(pts):
        host.save_feature_locations(xy, str(a / ("%d.txt" % i)))
    r = subprocess.run([TOOL, "txt2sift", str(a), str(b), "3"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([TOOL, "sift2txt", str(b), str(c), "3", "locations"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    for i, xy in enumerate(pts):
        assert open(c / ("%d.txt" % i)).read() == "".join("%g %g 0 0 0 \n" % (x, y) for x, y in xy)
    assert subprocess.run([TOOL, "sift2txt", str(b), str(c), "4"], capture_output=True, timeout=60).returncode == 1
