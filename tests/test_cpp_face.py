"""The C++ face of the drop-in boundary (include/hip_streaming_upsampler.h), compiled the way a caller of the
reference class would compile it (tests/cpp/test_hip_upsampler.cpp uses the totton::vulkan names) and linked
against lib/libmi_upsampler.so. CPU: builds, links, error strings, copy/move of an unloaded object. GPU: the
reference's three known-answer cases (tests/cpp/test_vulkan_upsampler.cpp:124-195) through the class."""
from __future__ import annotations

import subprocess

import pytest

from conftest import ROOT

SRC = ROOT / "tests" / "cpp" / "test_hip_upsampler.cpp"
LIBDIR = ROOT / "totton-rasp-gpu-dsp_amd" / "lib"


@pytest.fixture(scope="module")
def face_binary(tmp_path_factory):
    out = tmp_path_factory.mktemp("cppface") / "test_hip_upsampler"
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", f"-I{ROOT / 'include'}", str(SRC), "-o", str(out),
           f"-L{LIBDIR}", "-lmi_upsampler", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath-link,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    return out


def test_cpp_face_builds_and_host_side_semantics(face_binary, tmp_path):
    r = subprocess.run([str(face_binary), str(tmp_path), "--no-gpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "OK", r.stderr


@pytest.mark.gpu
def test_cpp_face_known_answers_on_gpu(face_binary, tmp_path, gpu):
    r = subprocess.run([str(face_binary), str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "OK", r.stderr + r.stdout
