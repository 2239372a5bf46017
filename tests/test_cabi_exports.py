"""CPU tests of the drop-in boundary: the shared library loads without a GPU, exports every symbol that
include/fishbird.h declares, the ctypes mirrors have the C sizes, and compute calls fail loudly (no CPU
fallback) when there is no device."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import numpy as np

import fishbirdeyevisualslam_amd as fb
from fishbirdeyevisualslam_amd import cabi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "fishbird.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(fb_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported():
    lib = fb.lib()
    decl = _declared()
    assert len(decl) >= 28
    missing = [s for s in decl if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(cabi.EXPORTS) == decl
    lib.fb_abi_version.restype = C.c_int
    assert lib.fb_abi_version() == 1


def test_ctypes_mirrors_match_the_header_layout():
    names = {"fb_keypoint": cabi.KP_DTYPE.itemsize, "fb_orb_params": C.sizeof(cabi.OrbParams), "fb_orb_tables": C.sizeof(cabi.OrbTables),
             "fb_grid_geom": C.sizeof(cabi.GridGeom), "fb_camera": C.sizeof(cabi.Camera), "fb_proj_frame_args": C.sizeof(cabi.ProjFrameArgs),
             "fb_bird_mp_args": C.sizeof(cabi.BirdMpArgs), "fb_proj_points_args": C.sizeof(cabi.ProjPointsArgs),
             "fb_birdview_args": C.sizeof(cabi.BirdviewArgs), "fb_pose_opt_args": C.sizeof(cabi.PoseOptArgs),
             "fb_local_ba_args": C.sizeof(cabi.LocalBAArgs), "fb_prof_entry": C.sizeof(cabi.ProfEntry),
             "fb_feature_vector": C.sizeof(cabi.FeatureVector), "fb_bow_args": C.sizeof(cabi.BowArgs),
             "fb_triangulation_args": C.sizeof(cabi.TriangulationArgs), "fb_proj_kf_args": C.sizeof(cabi.ProjKfArgs),
             "fb_bow_kf_args": C.sizeof(cabi.BowKfArgs), "fb_kf_target": C.sizeof(cabi.KfTarget), "fb_mp_list": C.sizeof(cabi.MpList),
             "fb_fuse_args": C.sizeof(cabi.FuseArgs), "fb_proj_sim3_args": C.sizeof(cabi.ProjSim3Args),
             "fb_sim3_args": C.sizeof(cabi.Sim3Args), "fb_init_match_args": C.sizeof(cabi.InitMatchArgs),
             "fb_frustum_args": C.sizeof(cabi.FrustumArgs), "fb_bird_filter_args": C.sizeof(cabi.BirdFilterArgs),
             "fb_frame_params": C.sizeof(cabi.FrameParams), "fb_map_points": C.sizeof(cabi.MapPoints),
             "fb_map_points_bird": C.sizeof(cabi.MapPointsBird), "fb_track_args": C.sizeof(cabi.TrackArgs),
             "fb_frame_view": C.sizeof(cabi.FrameView),
             "fb_vocabulary": C.sizeof(cabi.Vocabulary), "fb_bow_transform_args": C.sizeof(cabi.BowTransformArgs)}
    src = '#include <stdio.h>\n#include "fishbird.h"\nint main(void){\n' + "".join(
        'printf("%s %%zu\\n", sizeof(%s));\n' % (n, n) for n in names) + "return 0;}\n"
    d = tempfile.mkdtemp()
    open(os.path.join(d, "s.c"), "w").write(src)
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")])
    got = dict(l.split() for l in subprocess.check_output([os.path.join(d, "s")]).decode().splitlines())
    for n, sz in names.items():
        assert int(got[n]) == sz, (n, got[n], sz)


def test_header_is_plain_c():
    d = tempfile.mkdtemp()
    open(os.path.join(d, "h.c"), "w").write('#include "fishbird.h"\nint main(void){return FB_ABI_VERSION-1;}\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), os.path.join(d, "h.c"), "-o", os.path.join(d, "h")])


def test_no_cpu_fallback_without_device():
    lib = fb.lib()
    if lib.fb_device_count() > 0:
        return  # on the GPU box this is covered by the gpu tests
    a = np.zeros((4, 32), np.uint8)
    out = np.zeros(4, np.int32)
    rc = lib.fb_descriptor_distance(C.c_void_p(a.ctypes.data), C.c_void_p(a.ctypes.data), 4, C.c_void_p(out.ctypes.data))
    assert rc == cabi.FB_ERR_NODEVICE
    assert b"no CPU fallback" in lib.fb_last_error()
    p = cabi.OrbParams(2000, 1.2, 8, 15, 5)
    h = C.c_void_p()
    assert lib.fb_orb_create(C.byref(p), C.byref(h)) == 0   # host-only: tables
    t = cabi.OrbTables()
    assert lib.fb_orb_get_tables(h, C.byref(t)) == 0 and list(t.features_per_level)[:2] == [434, 362]
    img = np.zeros((64, 64), np.uint8)
    n = C.c_int32(0)
    assert lib.fb_orb_extract(h, C.c_void_p(img.ctypes.data), 64, 64, 64, None, None, C.byref(n)) == cabi.FB_ERR_NODEVICE
    lib.fb_orb_destroy(h)
    assert lib.fb_shutdown() == 0   # nothing to release, still fine


def test_product_never_touches_the_oracle():
    """The package (product path) must not import, link or open anything under oracle/."""
    pkg = os.path.join(ROOT, "fishbirdeyevisualslam_amd")
    for dirpath, _, files in os.walk(pkg):
        if os.path.basename(dirpath) == "build":
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".inc")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in txt and "pyoracle" not in txt and "orc_" not in txt, os.path.join(dirpath, f)
                assert not re.search(r'#include\s+"\.\./\.\./oracle', txt), f
    needed = subprocess.check_output(["readelf", "-d", fb.LIB_PATH]).decode()
    assert "oracle" not in needed
