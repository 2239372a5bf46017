"""The KeyFrame / MapPoint side of the C++ host mirror (fishbird_host.hpp): every remaining ORBmatcher entry point with the
reference's argument meaning and map mutations, driven on two real views of one scene (tests/cpp/kf_matchers_test.cpp).
The kernels behind them are parity-tested against the oracle through the C-ABI elsewhere; this checks the gather /
scatter code and the Fuse / Replace bookkeeping end to end: matches must agree with the known image shift."""
import os
import re
import subprocess
import tempfile

import pytest

import fishbirdeyevisualslam_amd as fb
from fishbirdeyevisualslam_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(d):
    exe = os.path.join(d, "kf_matchers_test")
    libdir = os.path.dirname(fb.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", os.path.join(ROOT, "tests", "cpp", "kf_matchers_test.cpp"), "-o", exe,
                           "-L", libdir, "-lfishbird_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_kf_host_header_compiles_on_cpu():
    fb.lib()
    _build(tempfile.mkdtemp())


@pytest.mark.gpu
def test_keyframe_matchers_end_to_end():
    d = tempfile.mkdtemp()
    exe = _build(d)
    synth.synth_image(1000, 640, 480).tofile(os.path.join(d, "img.raw"))
    out = subprocess.check_output([exe, os.path.join(d, "img.raw"), "640", "480"]).decode()
    print(out)
    v = {}
    for line in out.splitlines():
        head = line.split("=")[0]
        for k, x in re.findall(r"(\w+)=(\d+)", line):
            v[k if k in ("N1", "N2", "NP", "lvl0") or k == head else head + "_" + k] = int(x)
    N1, N2, NP = v["N1"], v["N2"], v["NP"]
    assert N1 > 900 and N2 > 900 and NP == N1 - N1 // 4
    # initialisation: level-0 key points only, nearly all re-found at the shifted position; vbPrevMatched updated
    assert v["init"] > 0.6 * v["lvl0"] and v["init_ok"] > 0.95 * v["init"]
    # BoW-gated KF -> Frame, relocalisation projection, Sim3 projection: each finds most of the map points, at the shift
    for k in ("bowF", "reloc", "projsim3"):
        assert v[k] > 0.5 * NP, (k, v[k])
        assert v[k + "_ok"] > 0.95 * v[k], (k, v[k], v[k + "_ok"])
    # triangulation: only key points of K1 WITHOUT a map point are matched; bOnlyStereo finds nothing on monocular key frames
    assert v["tri"] > 0.4 * (N1 // 4) and v["tri_ok"] > 0.9 * v["tri"] and v["tri_free"] == v["tri"] and v["tri_stereo"] == 0
    # Fuse: observations added to K2 (2 observations each, index consistent), a second Fuse finds nothing new
    # (two map points that pick the same K2 feature collide: the later one goes through Replace, so inK2 <= fuse)
    assert v["fuse"] > 0.6 * NP and 0.9 * v["fuse"] < v["fuse_inK2"] <= v["fuse"] and v["fuse_ok"] > 0.95 * v["fuse_inK2"] and v["fuse_again"] == 0
    # KF <-> KF by BoW and by Sim3 now return the shared MapPoints; a repeated SearchBySim3 has nothing left to match
    assert v["bowKK"] > 0.5 * v["fuse"] and v["bowKK_ok"] > 0.93 * v["bowKK"]
    assert v["sim3"] > 0.6 * v["fuse"] and v["sim3_ok"] > 0.97 * v["sim3"] and v["sim3_again"] == 0
    # duplicates: Fuse-Sim3 names the map point to keep; plain Fuse replaces the duplicate (fewer observations) by it
    assert v["fusesim3"] > 0.6 * NP and v["fusesim3_ok"] > 0.97 * v["fusesim3"]
    assert v["fusedup"] > 0.6 * NP and v["fusedup_bad"] == v["fusedup"] and v["fusedup_replaced"] > 0.97 * v["fusedup"]
    # bird: octave-0 reference keys within the 10 px window
    assert v["bird"] > 0.4 * v["lvl0"] and v["bird_ok"] > 0.95 * v["bird_dmatches"]
