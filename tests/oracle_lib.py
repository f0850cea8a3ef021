"""ctypes view of oracle/liborb_oracle.so -- the CPU restatement of the reference, test use only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liborb_oracle.so")

KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                           ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
BEST2_DTYPE = np.dtype([("best", "<i4"), ("best_idx", "<i4"), ("second", "<i4"), ("pad", "<i4")])


class Params(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32), ("blur_rounding", C.c_int32), ("trig_mode", C.c_int32)]


def _build_if_needed():
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("orb_oracle_extract.c", "orb_oracle_match.c", "orb_oracle_bow.c", "orb_oracle.h")]
    if not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])


_build_if_needed()
lib = C.CDLL(LIB_PATH)
_vp, _i, _f, _d = C.c_void_p, C.c_int, C.c_float, C.c_double
_pd = C.c_ssize_t
lib.orc_cv_round.restype = _i; lib.orc_cv_round.argtypes = [_d]
lib.orc_fast_atan2.restype = _f; lib.orc_fast_atan2.argtypes = [_f, _f]
lib.orc_fast_atan2_array.restype = None; lib.orc_fast_atan2_array.argtypes = [_vp, _vp, _i, _vp]
lib.orc_resize_linear_u8.restype = None; lib.orc_resize_linear_u8.argtypes = [_vp, _i, _i, _pd, _vp, _i, _i, _pd]
lib.orc_border101.restype = None; lib.orc_border101.argtypes = [_vp, _i, _i, _pd, _i]
lib.orc_fast9.restype = _i; lib.orc_fast9.argtypes = [_vp, _i, _i, _pd, _i, _i, _vp, _vp, _vp, _i]
lib.orc_fast9_score.restype = _i; lib.orc_fast9_score.argtypes = [_vp, _pd, _i]
lib.orc_gauss7.restype = None; lib.orc_gauss7.argtypes = [_vp, _i, _i, _pd, _vp, _pd, _i]
lib.orc_gauss_taps.restype = None; lib.orc_gauss_taps.argtypes = [_vp]
lib.orc_extractor_create.restype = _vp; lib.orc_extractor_create.argtypes = [C.POINTER(Params)]
lib.orc_extractor_destroy.restype = None; lib.orc_extractor_destroy.argtypes = [_vp]
lib.orc_extractor_tables.restype = _i; lib.orc_extractor_tables.argtypes = [_vp] * 7
lib.orc_extract.restype = _i; lib.orc_extract.argtypes = [_vp, _vp, _i, _i, _pd, _vp, _vp, _i, C.POINTER(_i)]
lib.orc_level_size.restype = _i; lib.orc_level_size.argtypes = [_vp, _i, C.POINTER(_i), C.POINTER(_i)]
lib.orc_level_padded.restype = _vp; lib.orc_level_padded.argtypes = [_vp, _i, C.POINTER(_i), C.POINTER(_i)]
lib.orc_level_blurred.restype = _vp; lib.orc_level_blurred.argtypes = [_vp, _i]
lib.orc_level_candidates.restype = _i; lib.orc_level_candidates.argtypes = [_vp, _i, _vp, _vp, _vp, _i]
lib.orc_level_keypoints.restype = _i; lib.orc_level_keypoints.argtypes = [_vp, _i, _vp, _i]
lib.orc_distribute_octree.restype = _i; lib.orc_distribute_octree.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i]
lib.orc_descriptor_distance.restype = _i; lib.orc_descriptor_distance.argtypes = [_vp, _vp]
lib.orc_hamming_best2.restype = None; lib.orc_hamming_best2.argtypes = [_vp, _i, _vp, C.c_int64, _vp]
lib.orc_compute_three_maxima.restype = None; lib.orc_compute_three_maxima.argtypes = [_vp, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]
lib.orc_grid_create.restype = _vp; lib.orc_grid_create.argtypes = [_vp, _i, _f, _f, _f, _f]
lib.orc_grid_destroy.restype = None; lib.orc_grid_destroy.argtypes = [_vp]
lib.orc_grid_features_in_area.restype = _i; lib.orc_grid_features_in_area.argtypes = [_vp, _f, _f, _f, _i, _i, _vp, _i]
lib.orc_search_for_initialization.restype = _i
lib.orc_search_for_initialization.argtypes = [_vp, _vp, _i, _vp, _vp, _i, _f, _f, _f, _f, _vp, _i, _f, _i, _vp]
lib.orc_search_by_projection_mappoints.restype = _i
lib.orc_search_by_projection_mappoints.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _f, _f, _f, _f, _f, _vp, _vp]
lib.orc_search_by_projection_lastframe.restype = _i
lib.orc_search_by_projection_lastframe.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _f, _f, _f, _f, _i, _i, _i, _vp, _vp]
lib.orc_stereo_refine.restype = None
lib.orc_stereo_refine.argtypes = [_vp, _vp, _vp, _i, _vp, _vp, _vp, _f, _f, _f, _vp, _vp, _vp]
lib.orc_stereo_match.restype = None
lib.orc_stereo_match.argtypes = [_vp, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _f, _f, _vp, _vp]


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def make_params(nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th_fast=20, min_th_fast=7, blur_rounding=0, trig_mode=0):
    return Params(nfeatures, scale_factor, nlevels, ini_th_fast, min_th_fast, blur_rounding, trig_mode)


def cv_round(v):
    return lib.orc_cv_round(float(v))


def fast_atan2(y, x):
    return lib.orc_fast_atan2(float(y), float(x))


def fast_atan2_array(y, x):
    y = np.ascontiguousarray(y, np.float32); x = np.ascontiguousarray(x, np.float32)
    out = np.empty(len(y), np.float32)
    lib.orc_fast_atan2_array(_ptr(y), _ptr(x), len(y), _ptr(out))
    return out


def resize_linear(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    lib.orc_resize_linear_u8(_ptr(src), src.shape[1], src.shape[0], src.strides[0], _ptr(dst), dw, dh, dst.strides[0])
    return dst


def border101(img, pad):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    buf = np.zeros((h + 2 * pad, w + 2 * pad), np.uint8)
    buf[pad:pad + h, pad:pad + w] = img
    lib.orc_border101(buf.ctypes.data + pad * buf.strides[0] + pad, w, h, buf.strides[0], pad)
    return buf


def fast9(img, threshold, nonmax=True):
    img = np.ascontiguousarray(img, np.uint8)
    cap = img.size
    ox = np.zeros(cap, np.int16); oy = np.zeros(cap, np.int16); os_ = np.zeros(cap, np.uint8)
    n = lib.orc_fast9(_ptr(img), img.shape[1], img.shape[0], img.strides[0], threshold, int(nonmax), _ptr(ox), _ptr(oy), _ptr(os_), cap)
    return ox[:n].copy(), oy[:n].copy(), os_[:n].copy()


def fast9_score(img, x, y, threshold):
    img = np.ascontiguousarray(img, np.uint8)
    return lib.orc_fast9_score(img.ctypes.data + y * img.strides[0] + x, img.strides[0], threshold)


def gauss7(img, rounding=0):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.zeros_like(img)
    lib.orc_gauss7(_ptr(img), img.shape[1], img.shape[0], img.strides[0], _ptr(out), out.strides[0], rounding)
    return out


def gauss_taps():
    t = np.zeros(7, np.int32)
    lib.orc_gauss_taps(_ptr(t))
    return t


class Extractor:
    def __init__(self, params):
        self.params = params
        self.h = lib.orc_extractor_create(C.byref(params))
        if not self.h:
            raise ValueError("bad oracle params")

    def close(self):
        if getattr(self, "h", None):
            lib.orc_extractor_destroy(self.h)
            self.h = None

    __del__ = close

    def tables(self):
        n = self.params.nlevels
        out = {k: np.zeros(n, np.float32) for k in ("scale", "inv_scale", "sigma2", "inv_sigma2")}
        quota = np.zeros(n, np.int32); umax = np.zeros(16, np.int32)
        lib.orc_extractor_tables(self.h, _ptr(out["scale"]), _ptr(out["inv_scale"]), _ptr(out["sigma2"]), _ptr(out["inv_sigma2"]), _ptr(quota), _ptr(umax))
        out["quota"] = quota; out["umax"] = umax
        return out

    def extract(self, img, cap=None):
        img = np.ascontiguousarray(img, np.uint8)
        cap = cap or (self.params.nfeatures + 8 * self.params.nlevels + 64)
        kps = np.zeros(cap, KEYPOINT_DTYPE); desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int()
        rc = lib.orc_extract(self.h, _ptr(img), img.shape[1], img.shape[0], img.strides[0], _ptr(kps), _ptr(desc), cap, C.byref(n))
        if rc:
            raise RuntimeError(f"oracle extract status {rc}")
        return kps[:n.value].copy(), desc[:n.value].copy()

    def level_size(self, level):
        w, h = C.c_int(), C.c_int()
        lib.orc_level_size(self.h, level, C.byref(w), C.byref(h))
        return w.value, h.value

    def level_padded(self, level):
        pw, ph = C.c_int(), C.c_int()
        p = lib.orc_level_padded(self.h, level, C.byref(pw), C.byref(ph))
        buf = (C.c_uint8 * (pw.value * ph.value)).from_address(p)
        return np.frombuffer(buf, np.uint8).reshape(ph.value, pw.value).copy()

    def level_blurred(self, level):
        w, h = self.level_size(level)
        p = lib.orc_level_blurred(self.h, level)
        if not p:
            return None
        buf = (C.c_uint8 * (w * h)).from_address(p)
        return np.frombuffer(buf, np.uint8).reshape(h, w).copy()

    def level_candidates(self, level):
        w, h = self.level_size(level)
        cap = w * h // 4 + 64
        x = np.zeros(cap, np.int16); y = np.zeros(cap, np.int16); s = np.zeros(cap, np.uint8)
        n = lib.orc_level_candidates(self.h, level, _ptr(x), _ptr(y), _ptr(s), cap)
        return x[:n].copy(), y[:n].copy(), s[:n].copy()

    def level_keypoints(self, level):
        cap = self.params.nfeatures + 64
        k = np.zeros(cap, KEYPOINT_DTYPE)
        n = lib.orc_level_keypoints(self.h, level, _ptr(k), cap)
        return k[:n].copy()


def distribute_octree(x, y, s, minX, maxX, minY, maxY, N):
    x = np.ascontiguousarray(x, np.int16); y = np.ascontiguousarray(y, np.int16); s = np.ascontiguousarray(s, np.uint8)
    cap = len(x) + 16
    out = np.zeros(cap, np.int32)
    n = lib.orc_distribute_octree(_ptr(x), _ptr(y), _ptr(s), len(x), minX, maxX, minY, maxY, N, _ptr(out), cap)
    if n < 0:
        raise ValueError("unsupported geometry")
    return out[:n].copy()


ANGLE_BITS_END = int(np.float32(360.0).view(np.uint32))     # float angles in [0, 360): bit patterns 0 .. this - 1
lib.orc_trig_range.restype = None; lib.orc_trig_range.argtypes = [_i, C.c_uint32, C.c_uint32, _vp, _vp]
lib.orc_trig_count_differences.restype = C.c_uint64; lib.orc_trig_count_differences.argtypes = [_i, _i, C.c_uint32, C.c_uint32]


def _threads():
    return max(1, min(16, (os.cpu_count() or 1)))


def trig_range(mode, first_bits, count, threads=None):
    """(cos, sin) of angle * (float)(pi/180) under trig_mode `mode` for the float angles with bit patterns
    first_bits .. first_bits + count - 1; split over host threads (ctypes releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    cs = np.empty(count, np.float32); sn = np.empty(count, np.float32)
    step = 1 << 20
    def run(o):
        n = min(step, count - o)
        lib.orc_trig_range(mode, first_bits + o, n, cs[o:].ctypes.data, sn[o:].ctypes.data)
    with ThreadPoolExecutor(threads or _threads()) as ex:
        list(ex.map(run, range(0, count, step)))
    return cs, sn


def trig_count_differences(mode_a, mode_b, first_bits=0, count=None, threads=None):
    from concurrent.futures import ThreadPoolExecutor
    count = ANGLE_BITS_END - first_bits if count is None else count
    step = 1 << 22
    def run(o):
        return lib.orc_trig_count_differences(mode_a, mode_b, first_bits + o, min(step, count - o))
    with ThreadPoolExecutor(threads or _threads()) as ex:
        return sum(ex.map(run, range(0, count, step)))


def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    return lib.orc_descriptor_distance(_ptr(a), _ptr(b))


def hamming_best2(q, t):
    q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
    out = np.zeros(q.shape[0], BEST2_DTYPE)
    lib.orc_hamming_best2(_ptr(q), q.shape[0], _ptr(t), t.shape[0], _ptr(out))
    return out


def compute_three_maxima(sizes):
    sizes = np.ascontiguousarray(sizes, np.int32)
    i1, i2, i3 = C.c_int(-1), C.c_int(-1), C.c_int(-1)
    lib.orc_compute_three_maxima(_ptr(sizes), len(sizes), C.byref(i1), C.byref(i2), C.byref(i3))
    return i1.value, i2.value, i3.value


class FrameGrid:
    def __init__(self, kps, bounds):
        self.kps = np.ascontiguousarray(kps, KEYPOINT_DTYPE)
        self.h = lib.orc_grid_create(_ptr(self.kps), len(self.kps), *[float(b) for b in bounds])

    def close(self):
        if getattr(self, "h", None):
            lib.orc_grid_destroy(self.h)
            self.h = None

    __del__ = close

    def features_in_area(self, x, y, r, min_level=-1, max_level=-1):
        out = np.zeros(max(len(self.kps), 1), np.int32)
        n = lib.orc_grid_features_in_area(self.h, x, y, r, min_level, max_level, _ptr(out), len(out))
        return out[:n].copy()


def search_for_initialization(k1, d1, k2, d2, bounds, prev_matched, window=100, nnratio=0.9, check_ori=True):
    k1 = np.ascontiguousarray(k1, KEYPOINT_DTYPE); k2 = np.ascontiguousarray(k2, KEYPOINT_DTYPE)
    d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
    prev = np.ascontiguousarray(prev_matched, np.float32).copy()
    m12 = np.full(len(k1), -1, np.int32)
    n = lib.orc_search_for_initialization(_ptr(k1), _ptr(d1), len(k1), _ptr(k2), _ptr(d2), len(k2),
                                          bounds[0], bounds[1], bounds[2], bounds[3], _ptr(prev), window, nnratio, int(check_ori), _ptr(m12))
    return m12, n, prev


def search_by_projection_keyframe(qu, qv, qradius, qlevel, qdesc, k2, d2, bounds, claimed):
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    qu, qv, qradius = f32(qu), f32(qv), f32(qradius)
    qlevel = np.ascontiguousarray(qlevel, np.int32)
    qdesc = np.ascontiguousarray(qdesc, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
    k2 = np.ascontiguousarray(k2, KEYPOINT_DTYPE)
    claimed = np.ascontiguousarray(claimed, np.int32).copy()
    lib.orc_search_by_projection_keyframe.restype = C.c_int
    lib.orc_search_by_projection_keyframe.argtypes = [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _f, _f, _f, _f, _vp]
    n = lib.orc_search_by_projection_keyframe(_ptr(qu), _ptr(qv), _ptr(qradius), _ptr(qlevel), _ptr(qdesc), len(qu), _ptr(k2), _ptr(d2),
                                              len(k2), bounds[0], bounds[1], bounds[2], bounds[3], _ptr(claimed))
    return claimed, n


class FeatureVectorC(C.Structure):
    _fields_ = [("node_ids", _vp), ("node_ptr", _vp), ("features", _vp), ("n_nodes", C.c_int32)]


def feature_vector(node_of_feature):
    node_of_feature = np.asarray(node_of_feature, np.int64)
    ids = np.unique(node_of_feature).astype(np.uint32)
    order = np.argsort(node_of_feature, kind="stable").astype(np.int32)
    counts = np.array([np.count_nonzero(node_of_feature == i) for i in ids], np.int64)
    ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    fv = FeatureVectorC(_ptr(ids) if len(ids) else None, _ptr(ptr), _ptr(order) if len(order) else None, len(ids))
    return fv, (ids, ptr, order)


def search_by_bow(d1, angle1, eligible1, nodes1, d2, angle2, eligible2, nodes2, nnratio=0.7, check_ori=True, keyframe_form=0):
    d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
    a1 = np.ascontiguousarray(angle1, np.float32); a2 = np.ascontiguousarray(angle2, np.float32)
    e1 = np.ascontiguousarray(eligible1, np.uint8)
    e2 = None if eligible2 is None else np.ascontiguousarray(eligible2, np.uint8)
    fv1, keep1 = feature_vector(nodes1); fv2, keep2 = feature_vector(nodes2)
    m12 = np.full(len(d1), -1, np.int32); m21 = np.full(len(d2), -1, np.int32)
    lib.orc_search_by_bow.restype = C.c_int
    lib.orc_search_by_bow.argtypes = [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _f, _i, _i, _vp, _vp]
    n = lib.orc_search_by_bow(_ptr(d1), _ptr(a1), _ptr(e1), len(d1), C.byref(fv1), _ptr(d2), _ptr(a2), None if e2 is None else _ptr(e2),
                              len(d2), C.byref(fv2), nnratio, int(check_ori), keyframe_form, _ptr(m12), _ptr(m21))
    return m12, m21, n


def search_for_triangulation(k1, d1, has_mp1, uright1, nodes1, k2, d2, has_mp2, uright2, nodes2, F12, ex, ey, scale2, sigma2_2,
                             only_stereo=False, check_ori=True):
    k1 = np.ascontiguousarray(k1, KEYPOINT_DTYPE); k2 = np.ascontiguousarray(k2, KEYPOINT_DTYPE)
    d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
    h1 = np.ascontiguousarray(has_mp1, np.uint8); h2 = np.ascontiguousarray(has_mp2, np.uint8)
    u1 = np.ascontiguousarray(uright1, np.float32); u2 = np.ascontiguousarray(uright2, np.float32)
    F = np.ascontiguousarray(F12, np.float32).reshape(9)
    sc = np.ascontiguousarray(scale2, np.float32); sg = np.ascontiguousarray(sigma2_2, np.float32)
    fv1, keep1 = feature_vector(nodes1); fv2, keep2 = feature_vector(nodes2)
    m12 = np.full(len(k1), -1, np.int32)
    lib.orc_search_for_triangulation.restype = C.c_int
    lib.orc_search_for_triangulation.argtypes = [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _f, _f, _vp, _vp, _i, _i, _vp]
    n = lib.orc_search_for_triangulation(_ptr(k1), _ptr(d1), _ptr(h1), _ptr(u1), len(k1), C.byref(fv1), _ptr(k2), _ptr(d2), _ptr(h2), _ptr(u2),
                                         len(k2), C.byref(fv2), _ptr(F), ex, ey, _ptr(sc), _ptr(sg), int(only_stereo), int(check_ori), _ptr(m12))
    return m12, n


def search_by_projection_mappoints(qu, qv, qr, qur, qlevel, qdesc, k2, d2, uright2, bounds, nnratio, claimed, transient=None):
    transient = None if transient is None else np.ascontiguousarray(transient, np.uint8)
    f32 = lambda a: None if a is None else np.ascontiguousarray(a, np.float32)
    qu, qv, qr, qur, uright2 = f32(qu), f32(qv), f32(qr), f32(qur), f32(uright2)
    qlevel = np.ascontiguousarray(qlevel, np.int32)
    qdesc = np.ascontiguousarray(qdesc, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
    k2 = np.ascontiguousarray(k2, KEYPOINT_DTYPE)
    claimed = np.ascontiguousarray(claimed, np.int32).copy()
    n = lib.orc_search_by_projection_mappoints(_ptr(qu), _ptr(qv), _ptr(qr), _ptr(qur), _ptr(qlevel), _ptr(qdesc), len(qu),
                                               _ptr(k2), _ptr(d2), _ptr(uright2), len(k2), bounds[0], bounds[1], bounds[2], bounds[3],
                                               nnratio, _ptr(transient), _ptr(claimed))
    return claimed, n


def search_by_projection_lastframe(qu, qv, qradius, qur, qoctave, qangle, qdesc, k2, d2, uright2, bounds, mode, th_dist, check_ori, claimed, transient=None):
    transient = None if transient is None else np.ascontiguousarray(transient, np.uint8)
    f32 = lambda a: None if a is None else np.ascontiguousarray(a, np.float32)
    qu, qv, qradius, qur, qangle, uright2 = f32(qu), f32(qv), f32(qradius), f32(qur), f32(qangle), f32(uright2)
    qoctave = np.ascontiguousarray(qoctave, np.int32)
    qdesc = np.ascontiguousarray(qdesc, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
    k2 = np.ascontiguousarray(k2, KEYPOINT_DTYPE)
    claimed = np.ascontiguousarray(claimed, np.int32).copy()
    n = lib.orc_search_by_projection_lastframe(_ptr(qu), _ptr(qv), _ptr(qradius), _ptr(qur), _ptr(qoctave), _ptr(qangle), _ptr(qdesc), len(qu),
                                               _ptr(k2), _ptr(d2), _ptr(uright2), len(k2), bounds[0], bounds[1], bounds[2], bounds[3],
                                               mode, th_dist, int(check_ori), _ptr(transient), _ptr(claimed))
    return claimed, n


def stereo_refine(left, right, kl, kr, best_dist, best_idx, bf, min_d, max_d):
    """left/right: oracle Extractors that have just extracted the pair."""
    kl = np.ascontiguousarray(kl); kr = np.ascontiguousarray(kr)
    bd = np.ascontiguousarray(best_dist, np.int32); bi = np.ascontiguousarray(best_idx, np.int32)
    ur = np.zeros(len(kl), np.float32); dp = np.zeros(len(kl), np.float32); sad = np.zeros(len(kl), np.int32)
    lib.orc_stereo_refine(left.h, right.h, _ptr(kl), len(kl), _ptr(kr), _ptr(bd), _ptr(bi), bf, min_d, max_d, _ptr(ur), _ptr(dp), _ptr(sad))
    return ur, dp, sad


def stereo_match(kl, dl, kr, dr, scale_factors, n_rows, min_d, max_d):
    kl = np.ascontiguousarray(kl, KEYPOINT_DTYPE); kr = np.ascontiguousarray(kr, KEYPOINT_DTYPE)
    dl = np.ascontiguousarray(dl, np.uint8); dr = np.ascontiguousarray(dr, np.uint8)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    bd = np.zeros(len(kl), np.int32); bi = np.zeros(len(kl), np.int32)
    lib.orc_stereo_match(_ptr(kl), _ptr(dl), len(kl), _ptr(kr), _ptr(dr), len(kr), _ptr(sf), len(sf), n_rows, min_d, max_d, _ptr(bd), _ptr(bi))
    return bd, bi


class VocabularyC(C.Structure):
    _fields_ = [("k", C.c_int32), ("L", C.c_int32), ("scoring", C.c_int32), ("weighting", C.c_int32), ("n_nodes", C.c_int32),
                ("parent", _vp), ("is_leaf", _vp), ("descriptors", _vp), ("weight", _vp)]


lib.orc_distinctive_descriptor.restype = _i; lib.orc_distinctive_descriptor.argtypes = [_vp, _i]
lib.orc_vocabulary_transform.restype = _i
lib.orc_vocabulary_transform.argtypes = [C.POINTER(VocabularyC), _vp, _i, _i, _vp, _vp, _vp, _vp, C.POINTER(C.c_int32), _vp, _vp, _vp,
                                         C.POINTER(C.c_int32)]


def distinctive_descriptor(desc):
    desc = np.ascontiguousarray(desc, np.uint8)
    return lib.orc_distinctive_descriptor(_ptr(desc) if len(desc) else None, len(desc))


def vocabulary_transform(k, L, parent, is_leaf, descriptors, weight, desc, levelsup=4, scoring=0, weighting=0):
    """Returns (word, node, bow_ids, bow_values, fv_node_ids, fv_node_ptr, fv_features)."""
    parent = np.ascontiguousarray(parent, np.int32); is_leaf = np.ascontiguousarray(is_leaf, np.uint8)
    descriptors = np.ascontiguousarray(descriptors, np.uint8); weight = np.ascontiguousarray(weight, np.float64)
    desc = np.ascontiguousarray(desc, np.uint8)
    n = desc.shape[0]
    voc = VocabularyC(k, L, scoring, weighting, len(parent), _ptr(parent), _ptr(is_leaf), _ptr(descriptors), _ptr(weight))
    word = np.zeros(max(n, 1), np.int32); node = np.zeros(max(n, 1), np.int32)
    ids = np.zeros(max(n, 1), np.uint32); vals = np.zeros(max(n, 1), np.float64)
    fn = np.zeros(max(n, 1), np.uint32); fp = np.zeros(n + 1, np.int32); ff = np.zeros(max(n, 1), np.int32)
    nw = C.c_int32(0); nn = C.c_int32(0)
    rc = lib.orc_vocabulary_transform(C.byref(voc), _ptr(desc), n, levelsup, _ptr(word), _ptr(node), _ptr(ids), _ptr(vals), C.byref(nw),
                                      _ptr(fn), _ptr(fp), _ptr(ff), C.byref(nn))
    if rc:
        raise ValueError("malformed vocabulary tree")
    return word[:n], node[:n], ids[:nw.value], vals[:nw.value], fn[:nn.value], fp[:nn.value + 1], ff[:fp[nn.value]]


lib.orc_cvt_color_to_gray.restype = None; lib.orc_cvt_color_to_gray.argtypes = [_vp, _i, _i, _pd, _i, _i, _vp, _pd]


def cvt_color_to_gray(img, blue_first):
    img = np.ascontiguousarray(img, np.uint8)
    h, w, c = img.shape
    out = np.zeros((h, w), np.uint8)
    lib.orc_cvt_color_to_gray(_ptr(img), w, h, img.strides[0], c, 1 if blue_first else 0, _ptr(out), out.strides[0])
    return out


lib.orc_undistort_points.restype = None; lib.orc_undistort_points.argtypes = [_vp, _i] + [_f] * 9 + [_vp]


def undistort_points(xy, fx, fy, cx, cy, k1, k2, p1, p2, k3):
    xy = np.ascontiguousarray(xy, np.float32)
    out = np.zeros_like(xy)
    lib.orc_undistort_points(_ptr(xy), xy.shape[0], fx, fy, cx, cy, k1, k2, p1, p2, k3, _ptr(out))
    return out


lib.orc_remap_linear_u8.restype = None; lib.orc_remap_linear_u8.argtypes = [_vp, _i, _i, _pd, _vp, _vp, _pd, _vp, _i, _i, _pd]


def remap_linear_u8(src, map_x, map_y):
    src = np.ascontiguousarray(src, np.uint8); map_x = np.ascontiguousarray(map_x, np.float32); map_y = np.ascontiguousarray(map_y, np.float32)
    dh, dw = map_x.shape
    out = np.zeros((dh, dw), np.uint8)
    lib.orc_remap_linear_u8(_ptr(src), src.shape[1], src.shape[0], src.strides[0], _ptr(map_x), _ptr(map_y), map_x.strides[0] // 4,
                            _ptr(out), dw, dh, out.strides[0])
    return out


lib.orc_fuse_candidates.restype = None
lib.orc_fuse_candidates.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _f, _f, _f, _f, _i, _vp, _vp]
lib.orc_search_by_sim3.restype = _i
lib.orc_search_by_sim3.argtypes = [_vp] * 8 + [_i] + [_vp] * 8 + [_i, _f, _f, _f, _f, _vp]


def fuse_candidates(qu, qv, qradius, qlevel, qdesc, k2, d2, bounds, qur=None, uright2=None, inv_level_sigma2=None):
    f32 = lambda a: None if a is None else np.ascontiguousarray(a, np.float32)
    qu, qv, qradius, qur, uright2, inv_level_sigma2 = f32(qu), f32(qv), f32(qradius), f32(qur), f32(uright2), f32(inv_level_sigma2)
    qlevel = np.ascontiguousarray(qlevel, np.int32); qdesc = np.ascontiguousarray(qdesc, np.uint8)
    k2 = np.ascontiguousarray(k2, KEYPOINT_DTYPE); d2 = np.ascontiguousarray(d2, np.uint8)
    bi = np.zeros(len(qu), np.int32); bd = np.zeros(len(qu), np.int32)
    lib.orc_fuse_candidates(_ptr(qu), _ptr(qv), _ptr(qur), _ptr(qradius), _ptr(qlevel), _ptr(qdesc), len(qu), _ptr(k2), _ptr(d2), _ptr(uright2),
                            len(k2), _ptr(inv_level_sigma2), bounds[0], bounds[1], bounds[2], bounds[3], 0 if qur is None else 1, _ptr(bi), _ptr(bd))
    return bi, bd


def search_by_sim3(side1, k1, d1, side2, k2, d2, bounds):
    def prep(side):
        valid, u, v, r, lvl, desc = side
        return (np.ascontiguousarray(valid, np.uint8), np.ascontiguousarray(u, np.float32), np.ascontiguousarray(v, np.float32),
                np.ascontiguousarray(r, np.float32), np.ascontiguousarray(lvl, np.int32), np.ascontiguousarray(desc, np.uint8))
    a, b = prep(side1), prep(side2)
    k1 = np.ascontiguousarray(k1, KEYPOINT_DTYPE); k2 = np.ascontiguousarray(k2, KEYPOINT_DTYPE)
    d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
    m12 = np.zeros(len(k1), np.int32)
    n = lib.orc_search_by_sim3(*[_ptr(x) for x in a], _ptr(k1), _ptr(d1), len(k1), *[_ptr(x) for x in b], _ptr(k2), _ptr(d2), len(k2),
                               bounds[0], bounds[1], bounds[2], bounds[3], _ptr(m12))
    return m12, n
