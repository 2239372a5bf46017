"""GPU test of the C++ host mirror (fishbirdeyevisualslam_amd/host/fishbird_host.hpp): a plain g++ program that
links only libfishbird_hip.so, driven like Tracking would, cross-checked against the oracle."""
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

import fishbirdeyevisualslam_amd as fb
import oracle_lib as O
from fishbirdeyevisualslam_amd import cabi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(d):
    exe = os.path.join(d, "host_test")
    libdir = os.path.dirname(fb.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "cpp", "host_test.cpp"), "-o", exe, "-L", libdir,
                           "-lfishbird_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_host_header_compiles_on_cpu():
    fb.lib()  # make sure the .so exists
    _build(tempfile.mkdtemp())


@pytest.mark.gpu
def test_host_mirror_end_to_end():
    d = tempfile.mkdtemp()
    exe = _build(d)
    img = synth.synth_image(1000, 640, 480)
    img.tofile(os.path.join(d, "img.raw"))
    out = subprocess.check_output([exe, os.path.join(d, "img.raw"), "640", "480", os.path.join(d, "out.bin")]).decode()
    m = re.search(r"N=(\d+) nmatches=(\d+) self=(\d+) inliers=(\d+) tx=([-\d.]+) dist00=(\d+)", out)
    assert m, out
    N, nmatches, self_, inl, tx, d00 = int(m[1]), int(m[2]), int(m[3]), int(m[4]), float(m[5]), int(m[6])
    raw = open(os.path.join(d, "out.bin"), "rb").read()
    k = np.frombuffer(raw[4: 4 + 24 * N], cabi.KP_DTYPE)
    dsc = np.frombuffer(raw[4 + 24 * N: 4 + 56 * N], np.uint8).reshape(N, 32)
    kun = np.frombuffer(raw[4 + 56 * N:], cabi.KP_DTYPE)
    ko, do = O.orb_extract(O.orb_params(nfeatures=1000), img)
    assert np.array_equal(k, ko) and np.array_equal(dsc, do)
    assert d00 == 0
    assert nmatches > 0.9 * N and self_ > 0.9 * nmatches      # a frame matched against itself
    assert inl > 0.9 * nmatches and abs(tx) < 2e-3            # the 2 cm perturbation is optimised away
    ms = re.search(r"sizes_ok=(\d+) bird_inliers=(\d+)", out)
    assert ms and int(ms[1]) == 1 and int(ms[2]) == 60, out   # outlier vectors keep their sizes and contents across modes
    m2 = re.search(r"inview=(\d+) local=(\d+) selflocal=(\d+)", out)
    assert m2, out
    in_view, n_local, self_local = int(m2[1]), int(m2[2]), int(m2[3])
    assert in_view > 0.95 * N                                  # isInFrustum: every point of the frame itself is visible
    assert n_local > 0.8 * N and self_local > 0.9 * n_local     # TrackLocalMap re-finds the frame's own points
    # Frame::UndistortKeyPoints / ComputeImageBounds through the mirror == the oracle on the same key points
    from fishbirdeyevisualslam_amd import more_problems as M
    mu = re.search(r"undist same=(\d+) bounds=([-\d.]+),([-\d.]+),([-\d.]+),([-\d.]+)", out)
    assert mu and int(mu[1]) == N
    K4 = np.array([650.0, 648.0, 640.0, 360.0], np.float32)
    D4 = np.array([-0.02, 0.004, -0.001, 0.0002], np.float32)
    np.testing.assert_array_equal(kun, M.undistort(O.lib(), "orc_", k.copy(), K4=K4, D4=D4))
    np.testing.assert_allclose([float(mu[i]) for i in range(2, 6)], M.image_bounds(O.lib(), "orc_", 1280, 720, K4=K4, D4=D4), rtol=0, atol=2e-6 * 1280)
