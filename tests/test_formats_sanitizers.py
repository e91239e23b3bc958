"""The file-format code (csrc/formats.cpp: .sift, match txt, keypoint txt, .mat putative and inlier
records with directory regrow and block moves, cached writer, readers with short buffers) compiled for the
host with AddressSanitizer and UndefinedBehaviorSanitizer and driven through its C ABI."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_format_code_is_clean_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "format_driver")
    cmd = ["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-static-libasan", "-static-libubsan",
           "-fno-omit-frame-pointer", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "visual_slam_amd", "csrc", "formats.cpp"), os.path.join(HERE, "san", "format_driver.cpp"),
           "-o", exe, "-pthread"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if res.returncode != 0 and ("cannot find" in res.stderr and "san" in res.stderr):
        pytest.skip("static sanitizer runtimes are not installed")
    assert res.returncode == 0, res.stderr
    res = subprocess.run([exe, str(tmp_path / "work")], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "sanitizer driver done" in res.stdout
