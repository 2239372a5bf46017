"""Thin ctypes callers of the HIP C-ABI used by the GPU parity tests (host-pointer entry points)."""
import ctypes as C

import numpy as np

import fishbirdeyevisualslam_amd as fb
from fishbirdeyevisualslam_amd import cabi


def call(name, args):
    fb.check(getattr(fb.lib(), name)(C.byref(args)), name)


class Orb:
    def __init__(self, params):
        self.params = params
        self.h = C.c_void_p()
        fb.check(fb.lib().fb_orb_create(C.byref(params), C.byref(self.h)), "fb_orb_create")
        fb.lib().fb_orb_capacity.restype = C.c_int
        self.cap = fb.lib().fb_orb_capacity(C.byref(params))

    def close(self):
        if self.h:
            fb.lib().fb_orb_destroy(self.h)
            self.h = C.c_void_p()

    def tables(self):
        t = cabi.OrbTables()
        fb.check(fb.lib().fb_orb_get_tables(self.h, C.byref(t)), "fb_orb_get_tables")
        return t

    def extract(self, img):
        img = np.ascontiguousarray(img)
        h, w = img.shape
        kps = np.zeros(self.cap, cabi.KP_DTYPE)
        desc = np.zeros((self.cap, 32), np.uint8)
        n = C.c_int32(0)
        fb.check(fb.lib().fb_orb_extract(self.h, C.c_void_p(img.ctypes.data), w, h, w, C.c_void_p(kps.ctypes.data),
                                         C.c_void_p(desc.ctypes.data), C.byref(n)), "fb_orb_extract")
        return kps[: n.value].copy(), desc[: n.value].copy()

    def level(self, b, level, maxpix):
        buf = np.zeros(maxpix, np.uint8)
        w, h = C.c_int(0), C.c_int(0)
        fb.check(fb.lib().fb_orb_get_level(self.h, b, level, C.c_void_p(buf.ctypes.data), C.byref(w), C.byref(h)),
                 "fb_orb_get_level")
        return buf[: w.value * h.value].reshape(h.value, w.value).copy()


def grid_build(kps, n, batch, stride, geom, cs, ci):
    """fb_grid_build_batch_dev through torch device buffers."""
    import torch
    dev = torch.device("cuda:0")
    dk = torch.from_numpy(kps.view(np.uint8).reshape(-1)).to(dev)
    dn = torch.from_numpy(n).to(dev)
    dcs = torch.zeros(cs.size, dtype=torch.int32, device=dev)
    dci = torch.zeros(ci.size, dtype=torch.int32, device=dev)
    fb.check(fb.lib().fb_grid_build_batch_dev(C.c_void_p(dk.data_ptr()), C.c_void_p(dn.data_ptr()), batch, stride,
                                              C.byref(geom), C.c_void_p(dcs.data_ptr()), C.c_void_p(dci.data_ptr()),
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream)), "fb_grid_build")
    torch.cuda.synchronize()
    cs[...] = dcs.cpu().numpy().reshape(cs.shape)
    ci[...] = dci.cpu().numpy().reshape(ci.shape)
