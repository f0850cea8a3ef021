"""ctypes view of liborbx.so (C ABI in include/orbx.h) for tests and benchmarks.

The product is the shared library; this module only marshals numpy arrays / raw device
pointers into it.  Import with ``importlib.import_module("visual-slam_amd")`` (the directory
name is not a Python identifier).  Loading fails loudly when the library has not been built:
there is no CPU fallback.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ORBX_LIB_PATH") or os.path.join(_HERE, "csrc", "liborbx.so")   # the override lets a tool time two builds side by side

KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                           ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
BEST2_DTYPE = np.dtype([("best", "<i4"), ("best_idx", "<i4"), ("second", "<i4"), ("pad", "<i4")])
assert KEYPOINT_DTYPE.itemsize == 28 and BEST2_DTYPE.itemsize == 16

STAGES = ("pyramid", "blur", "fast", "octree", "describe")


class Params(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32), ("blur_rounding", C.c_int32), ("trig_mode", C.c_int32)]


class Camera(C.Structure):
    """orbx_camera: pinhole intrinsics and k1, k2, p1, p2, k3."""
    _fields_ = [(n, C.c_float) for n in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3")]


class OrbxError(RuntimeError):
    def __init__(self, status, text):
        super().__init__(f"orbx status {status}: {text}")
        self.status = status


def _load():
    # PyTorch bundles its own HIP runtime under the same soname as /opt/rocm's.  Whichever is mapped
    # first serves the whole process, and torch cannot initialise on top of the system one, so when
    # torch is installed let it load first; liborbx then binds to that runtime as well.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(make -C visual-slam_amd/csrc) first; there is no fallback path")
    return C.CDLL(LIB_PATH)


lib = _load()
_vp, _i, _f, _i64 = C.c_void_p, C.c_int, C.c_float, C.c_int64
_pd = C.c_ssize_t

_SIGS = {
    "orbx_status_string": (C.c_char_p, [_i]),
    "orbx_last_error": (C.c_char_p, []),
    "orbx_build_stamp": (C.c_char_p, []),
    "orbx_shard_range": (_i, [_i64, _i, _i, C.POINTER(_i64), C.POINTER(_i64)]),
    "orbx_extractor_create": (_i, [C.POINTER(Params), _i, _i, _i, _i, C.POINTER(_vp)]),
    "orbx_extractor_destroy": (None, [_vp]),
    "orbx_scale_tables": (_i, [C.POINTER(Params), _vp, _vp, _vp, _vp, _vp, _vp]),
    "orbx_level_size": (_i, [C.POINTER(Params), _i, _i, _i, C.POINTER(_i), C.POINTER(_i)]),
    "orbx_keypoint_capacity": (_i, [C.POINTER(Params), _i, _i]),
    "orbx_extractor_set_input_format": (_i, [_vp, _i]),
    "orbx_undistort_keypoints_batch_device": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "orbx_undistort_points": (_i, [_vp, _vp, _vp, _i, _vp]),
    "orbx_extract": (_i, [_vp, _vp, _pd, _vp, _vp, _i, C.POINTER(_i)]),
    "orbx_extract_batch": (_i, [_vp, _vp, _pd, _pd, _i, _vp, _vp, _vp, _i]),
    "orbx_extract_batch_device": (_i, [_vp, _vp, _pd, _pd, _i, _vp, _vp, _vp, _i, _vp]),
    "orbx_extract_batch_device_multi": (_i, [_vp, _i, _vp, _pd, _pd, _i, _vp, _vp, _vp, _i, _vp, _i]),
    "orbx_extract_multi_join": (_i, [_vp, _i, _vp]),
    "orbx_extractor_input_view": (_i, [_vp, C.POINTER(_vp), C.POINTER(_pd), C.POINTER(_pd)]),
    "orbx_pyramid_level": (_i, [_vp, _i, _i, _vp, _pd]),
    "orbx_pyramid_levels": (_i, [_vp, _i, _vp, _vp]),
    "orbx_pyramid_level_device": (_i, [_vp, _i, _i, C.POINTER(_vp), C.POINTER(_i), C.POINTER(_i), C.POINTER(_pd)]),
    "orbx_debug_blurred_level": (_i, [_vp, _i, _i, _vp, _pd]),
    "orbx_debug_trig_range": (_i, [_i, C.c_uint32, C.c_uint32, _vp, _vp, _vp]),
    "orbx_search_by_projection_keyframe": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _f, _f, _f, _f, _vp, C.POINTER(_i)]),
    "orbx_project_best1": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _f, _f, _f, _f, _vp, _vp, _vp, _i, _vp, _vp]),
    "orbx_search_by_sim3": (_i, [_vp] + [_vp] * 8 + [_i] + [_vp] * 8 + [_i, _f, _f, _f, _f, _vp, C.POINTER(_i)]),
    "orbx_search_by_bow": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _f, _i, _i, _vp, _vp, C.POINTER(_i)]),
    "orbx_search_for_triangulation": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _f, _f, _vp, _vp,
                                           _i, _i, _i, _vp, C.POINTER(_i)]),
    "orbx_stereo_refine_batch_device": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _f, _f, _f, _vp, _vp, _vp, _vp]),
    "orbx_stereo_refine": (_i, [_vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _f, _f, _f, _vp, _vp]),
    "orbx_debug_candidates": (_i, [_vp, _i, _i, _vp, _i, C.POINTER(_i)]),
    "orbx_debug_level_keypoints": (_i, [_vp, _i, _i, _vp, _i, C.POINTER(_i)]),
    "orbx_distribute_octree": (_i, [_i, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, C.POINTER(_i)]),
    "orbx_distribute_octree_ex": (_i, [_i, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, C.POINTER(_i), _i]),
    "orbx_extractor_set_profiling": (_i, [_vp, _i]),
    "orbx_extractor_stage_ms": (_i, [_vp, _vp]),
    "orbx_matcher_create": (_i, [_i, C.POINTER(_vp)]),
    "orbx_matcher_destroy": (None, [_vp]),
    "orbx_descriptor_distance": (_i, [_vp, _vp]),
    "orbx_compute_three_maxima": (None, [_vp, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "orbx_hamming_best2": (_i, [_vp, _vp, _i, _vp, _i64, _vp]),
    "orbx_hamming_best2_device": (_i, [_vp, _vp, _i, _vp, _i64, _i64, _vp, _vp]),
    "orbx_hamming_best2_device_static": (_i, [_vp, _vp, _i, _vp, _i64, _i64, _vp, _vp]),
    "orbx_debug_best2_tile_walks": (_i, [_vp, C.POINTER(C.c_uint64), _i]),
    "orbx_best2_merge_device": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "orbx_best2_merge": (_i, [_vp, _i, _i, _vp]),
    "orbx_hamming_best2_workspace_bytes": (_i64, [_i, _i64]),
    "orbx_hamming_csr": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _vp]),
    "orbx_grid_create": (_i, [_vp, _i, _f, _f, _f, _f, C.POINTER(_vp)]),
    "orbx_grid_destroy": (None, [_vp]),
    "orbx_grid_features_in_area": (_i, [_vp, _f, _f, _f, _i, _i, _vp, _i, C.POINTER(_i)]),
    "orbx_search_for_initialization": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _f, _f, _f, _f, _vp, _i, _f, _i, _vp, C.POINTER(_i)]),
    "orbx_search_for_initialization_batch_device": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _f, _f, _vp, _i, _f, _i, _vp, _vp, _vp]),
    "orbx_search_by_projection_batch_device": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _f, _f, _i, _i, _f, _i, _vp, _vp, _vp]),
    "orbx_search_by_projection_mappoints": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _f, _f, _f, _f, _f, _vp, _vp, C.POINTER(_i)]),
    "orbx_search_by_projection_lastframe": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _f, _f, _f, _f, _i, _i, _i, _vp, _vp, C.POINTER(_i)]),
    "orbx_stereo_match": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _f, _f, _vp, _vp]),
    "orbx_stereo_match_batch_device": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _f, _f, _vp, _vp, _vp]),
    "orbx_distinctive_descriptors": (_i, [_vp, _vp, _vp, _i, _vp]),
    "orbx_distinctive_descriptors_device": (_i, [_vp, _vp, _vp, _i, _i64, _vp, _vp]),
    "orbx_vocabulary_create": (_i, [_i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, C.POINTER(_vp)]),
    "orbx_vocabulary_destroy": (None, [_vp]),
    "orbx_vocabulary_words": (_i, [_vp]),
    "orbx_vocabulary_descend_device": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "orbx_vocabulary_transform": (_i, [_vp, _vp, _i, _i, _vp, _vp, C.POINTER(C.c_int32), _vp, _vp, _vp, C.POINTER(C.c_int32)]),
    "orbx_synth_vocabulary": (_i, [C.c_uint32, _i, _i, _vp, _vp, _vp, _vp]),
    "orbx_remap_create": (_i, [_i, _vp, _vp, _pd, _i, _i, _i, _i, C.POINTER(_vp)]),
    "orbx_remap_destroy": (None, [_vp]),
    "orbx_remap_batch_device": (_i, [_vp, _vp, _pd, _pd, _i, _vp, _pd, _pd, _vp]),
    "orbx_remap_image": (_i, [_vp, _vp, _pd, _vp, _pd]),
    "orbx_mat_record_bytes": (_i64, [_i, _i, _i]),
    "orbx_mat_record_write": (_i64, [_vp, _i, _i, _i, _i, _vp, _i64]),
    "orbx_mat_record_read": (_i64, [_vp, _i64, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_vp)]),
    "orbx_synth_frame": (None, [C.c_uint32, _i, _i, _vp, _pd]),
    "orbx_synth_stereo": (None, [C.c_uint32, _i, _i, _vp, _vp, _pd]),
    "orbx_synth_descriptors": (None, [C.c_uint32, _i64, _vp]),
    "orbx_synth_queries": (None, [C.c_uint32, _vp, _i64, _i, _vp, _vp]),
}
EXPORTED = tuple(_SIGS)
for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)
    _fn.restype = _res
    _fn.argtypes = _args


def _check(status):
    if status != 0:
        raise OrbxError(status, (lib.orbx_last_error() or b"").decode() or lib.orbx_status_string(status).decode())


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def make_params(nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th_fast=20, min_th_fast=7, blur_rounding=0, trig_mode=0):
    return Params(nfeatures, scale_factor, nlevels, ini_th_fast, min_th_fast, blur_rounding, trig_mode)


def build_stamp():
    return lib.orbx_build_stamp().decode()


def shard_range(total, rank, world):
    """[begin, end) of the units rank `rank` of `world` owns (orbx_shard_range)."""
    b, e = _i64(), _i64()
    _check(lib.orbx_shard_range(total, rank, world, C.byref(b), C.byref(e)))
    return b.value, e.value


def scale_tables(p):
    n = p.nlevels
    out = {k: np.zeros(n, np.float32) for k in ("scale", "inv_scale", "sigma2", "inv_sigma2")}
    quota = np.zeros(n, np.int32)
    umax = np.zeros(16, np.int32)
    _check(lib.orbx_scale_tables(C.byref(p), _ptr(out["scale"]), _ptr(out["inv_scale"]), _ptr(out["sigma2"]),
                                 _ptr(out["inv_sigma2"]), _ptr(quota), _ptr(umax)))
    out["quota"] = quota
    out["umax"] = umax
    return out


def level_size(p, width, height, level):
    w, h = C.c_int(), C.c_int()
    _check(lib.orbx_level_size(C.byref(p), width, height, level, C.byref(w), C.byref(h)))
    return w.value, h.value


def keypoint_capacity(p, width, height):
    n = lib.orbx_keypoint_capacity(C.byref(p), width, height)
    if n < 0:
        _check(-n)
    return n


def synth_frame(index, w, h):
    img = np.empty((h, w), np.uint8)
    lib.orbx_synth_frame(index, w, h, _ptr(img), w)
    return img


def synth_stereo(index, w, h):
    left, right = np.empty((h, w), np.uint8), np.empty((h, w), np.uint8)
    lib.orbx_synth_stereo(index, w, h, _ptr(left), _ptr(right), w)
    return left, right


def synth_descriptors(seed, n):
    out = np.empty((n, 32), np.uint8)
    lib.orbx_synth_descriptors(seed, n, _ptr(out))
    return out


def synth_queries(seed, table, nq):
    out = np.empty((nq, 32), np.uint8)
    src = np.empty(nq, np.int64)
    lib.orbx_synth_queries(seed, _ptr(table), table.shape[0], nq, _ptr(out), _ptr(src))
    return out, src


def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return lib.orbx_descriptor_distance(_ptr(a), _ptr(b))


def compute_three_maxima(sizes):
    sizes = np.ascontiguousarray(sizes, np.int32)
    i1, i2, i3 = C.c_int(-1), C.c_int(-1), C.c_int(-1)
    lib.orbx_compute_three_maxima(_ptr(sizes), len(sizes), C.byref(i1), C.byref(i2), C.byref(i3))
    return i1.value, i2.value, i3.value


def best2_merge(parts):
    """parts: [n_parts, nq] BEST2_DTYPE (best_idx global) -> [nq]"""
    parts = np.ascontiguousarray(parts, BEST2_DTYPE)
    out = np.zeros(parts.shape[1], BEST2_DTYPE)
    _check(lib.orbx_best2_merge(_ptr(parts), parts.shape[0], parts.shape[1], _ptr(out)))
    return out


def unpack_keys(packed):
    packed = np.asarray(packed, np.uint32)
    return (packed >> 20).astype(np.int32), ((packed >> 8) & 0xFFF).astype(np.int32), (packed & 0xFF).astype(np.int32)


def pack_keys(x, y, s):
    return ((np.asarray(x, np.uint32) << 20) | (np.asarray(y, np.uint32) << 8) | np.asarray(s, np.uint32)).astype(np.uint32)


def distribute_octree(packed, minX, maxX, minY, maxY, N, device=-1, key_storage=0):
    packed = np.ascontiguousarray(packed, np.uint32)
    cap = max(N + 16, 4 * 64 + 16)
    out = np.zeros(cap, np.uint32)
    n = C.c_int()
    _check(lib.orbx_distribute_octree_ex(device, _ptr(packed), len(packed), minX, maxX, minY, maxY, N, _ptr(out), cap, C.byref(n), key_storage))
    return out[:n.value]


class Extractor:
    """orbx_extractor context bound to one image size."""

    def __init__(self, params, width, height, max_batch=1, device=-1):
        self.params, self.width, self.height, self.max_batch = params, width, height, max_batch
        self.cap = keypoint_capacity(params, width, height)
        h = C.c_void_p()
        _check(lib.orbx_extractor_create(C.byref(params), width, height, max_batch, device, C.byref(h)))
        self.h = h
        self.channels = 1

    FORMATS = {"gray": (0, 1), "rgb": (1, 3), "bgr": (2, 3), "rgba": (3, 4), "bgra": (4, 4)}

    def set_input_format(self, name):
        """Input layout of the extract calls: "gray" (default) or interleaved "rgb" / "bgr" / "rgba" / "bgra"."""
        code, ch = self.FORMATS[name]
        _check(lib.orbx_extractor_set_input_format(self.h, code))
        self.channels = ch

    def _image_shape(self):
        return (self.height, self.width) if self.channels == 1 else (self.height, self.width, self.channels)

    def close(self):
        if getattr(self, "h", None) and lib is not None:      # lib is None while the interpreter shuts down
            lib.orbx_extractor_destroy(self.h)
            self.h = None

    __del__ = close

    def extract(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        assert img.shape == self._image_shape()
        kps = np.zeros(self.cap, KEYPOINT_DTYPE)
        desc = np.zeros((self.cap, 32), np.uint8)
        n = C.c_int()
        _check(lib.orbx_extract(self.h, _ptr(img), img.strides[0], _ptr(kps), _ptr(desc), self.cap, C.byref(n)))
        return kps[:n.value], desc[:n.value]

    def extract_batch(self, imgs):
        imgs = np.ascontiguousarray(imgs, np.uint8)
        b = imgs.shape[0]
        assert imgs.shape[1:] == self._image_shape()
        kps = np.zeros((b, self.cap), KEYPOINT_DTYPE)
        desc = np.zeros((b, self.cap, 32), np.uint8)
        counts = np.zeros(b, np.int32)
        _check(lib.orbx_extract_batch(self.h, _ptr(imgs), imgs.strides[1], imgs.strides[0], b, _ptr(kps), _ptr(desc), _ptr(counts), self.cap))
        return kps, desc, counts

    def extract_batch_device(self, d_images, row_stride, frame_stride, n_frames, d_kps, d_desc, d_counts, cap, stream=None):
        _check(lib.orbx_extract_batch_device(self.h, d_images, row_stride, frame_stride, n_frames, d_kps, d_desc, d_counts, cap, stream))

    def input_view(self):
        """(device pointer, row stride, frame stride) of the place where frames can be written to skip the input copy."""
        ptr, rs, fs = _vp(), _pd(), _pd()
        _check(lib.orbx_extractor_input_view(self.h, C.byref(ptr), C.byref(rs), C.byref(fs)))
        return ptr.value, rs.value, fs.value

    def input_view_array(self, n_frames=None):
        """The input view as an object with __cuda_array_interface__ (n_frames x height x width bytes, strided), e.g. for
        torch.as_tensor(ex.input_view_array(), device="cuda")."""
        ptr, rs, fs = self.input_view()

        class _View:
            pass
        v = _View()
        v.__cuda_array_interface__ = {"shape": (int(n_frames or self.max_batch), self.height, self.width), "strides": (fs, rs, 1),
                                      "typestr": "|u1", "data": (ptr, False), "version": 2}
        return v

    def undistort_points(self, cam, xy):
        xy = np.ascontiguousarray(xy, np.float32)
        out = np.zeros_like(xy)
        _check(lib.orbx_undistort_points(self.h, C.byref(cam), _ptr(xy), xy.shape[0], _ptr(out)))
        return out

    def undistort_keypoints_batch_device(self, cam, d_kps, d_counts, cap, n_frames, d_kps_un, stream=None):
        _check(lib.orbx_undistort_keypoints_batch_device(self.h, C.byref(cam), d_kps, d_counts, cap, n_frames, d_kps_un, stream))

    def pyramid_level(self, frame, level):
        w, h = level_size(self.params, self.width, self.height, level)
        out = np.zeros((h + 38, w + 38), np.uint8)
        _check(lib.orbx_pyramid_level(self.h, frame, level, _ptr(out), out.strides[0]))
        return out

    def pyramid_levels(self, frame):
        """Every padded level of one frame in one round trip."""
        outs = []
        for l in range(self.params.nlevels):
            w, h = level_size(self.params, self.width, self.height, l)
            outs.append(np.zeros((h + 38, w + 38), np.uint8))
        ptrs = (C.c_void_p * len(outs))(*[o.ctypes.data for o in outs])
        strides = (C.c_ssize_t * len(outs))(*[o.strides[0] for o in outs])
        _check(lib.orbx_pyramid_levels(self.h, frame, ptrs, strides))
        return outs

    def stereo_refine(self, right, kl, kr, best_dist, best_idx, bf, min_d, max_d):
        """Sub-pixel part of Frame::ComputeStereoMatches on this (left) and `right` extractor's last frame 0."""
        kl = np.ascontiguousarray(kl); kr = np.ascontiguousarray(kr)
        bd = np.ascontiguousarray(best_dist, np.int32); bi = np.ascontiguousarray(best_idx, np.int32)
        ur = np.zeros(len(kl), np.float32); dp = np.zeros(len(kl), np.float32)
        _check(lib.orbx_stereo_refine(self.h, right.h, _ptr(kl), len(kl), _ptr(kr), len(kr), _ptr(bd), _ptr(bi), bf, min_d, max_d, _ptr(ur), _ptr(dp)))
        return ur, dp

    def stereo_refine_batch_device(self, right, d_kl, d_nl, d_kr, cap, n_pairs, d_bd, d_bi, bf, min_d, max_d, d_ur, d_dp, d_sad, stream=None):
        _check(lib.orbx_stereo_refine_batch_device(self.h, right.h, d_kl, d_nl, d_kr, cap, n_pairs, d_bd, d_bi, bf, min_d, max_d, d_ur, d_dp, d_sad, stream))

    def debug_blurred_level(self, frame, level):
        w, h = level_size(self.params, self.width, self.height, level)
        out = np.zeros((h, w), np.uint8)
        _check(lib.orbx_debug_blurred_level(self.h, frame, level, _ptr(out), out.strides[0]))
        return out

    def debug_candidates(self, frame, level):
        cap = (self.width * self.height) // 3 + 1024
        out = np.zeros(cap, np.uint32)
        n = C.c_int()
        _check(lib.orbx_debug_candidates(self.h, frame, level, _ptr(out), cap, C.byref(n)))
        return out[:n.value]

    def debug_level_keypoints(self, frame, level):
        out = np.zeros(self.cap + 16, np.uint32)
        n = C.c_int()
        _check(lib.orbx_debug_level_keypoints(self.h, frame, level, _ptr(out), len(out), C.byref(n)))
        return out[:n.value]

    def set_profiling(self, on):
        _check(lib.orbx_extractor_set_profiling(self.h, int(on)))

    def stage_ms(self):
        ms = np.zeros(len(STAGES), np.float32)
        _check(lib.orbx_extractor_stage_ms(self.h, _ptr(ms)))
        return ms


def extract_batch_device_multi(extractors, d_images, row_stride, frame_stride, n_frames, d_kps, d_desc, d_counts, cap, stream=None, join=True):
    """One call over several contexts (orbx_extract_batch_device_multi); d_images None: every context reads its input view."""
    arr = (_vp * len(extractors))(*[e.h for e in extractors])
    _check(lib.orbx_extract_batch_device_multi(arr, len(extractors), d_images, row_stride, frame_stride, n_frames, d_kps, d_desc, d_counts, cap,
                                               stream, 0 if join else 1))


def extract_multi_join(extractors, stream=None):
    arr = (_vp * len(extractors))(*[e.h for e in extractors])
    _check(lib.orbx_extract_multi_join(arr, len(extractors), stream))


class FeatureVectorC(C.Structure):
    """orbx_feature_vector: a DBoW2::FeatureVector flattened (node ids ascending, CSR feature lists)."""
    _fields_ = [("node_ids", _vp), ("node_ptr", _vp), ("features", _vp), ("n_nodes", C.c_int32)]


def feature_vector(node_of_feature):
    """Builds the flattened feature vector of features 0..n-1 from the vocabulary node of each (features of a node in
    index order, as DBoW2 appends them).  Returns (struct, keepalive arrays)."""
    node_of_feature = np.asarray(node_of_feature, np.int64)
    ids = np.unique(node_of_feature).astype(np.uint32)
    order = np.argsort(node_of_feature, kind="stable").astype(np.int32)
    counts = np.array([np.count_nonzero(node_of_feature == i) for i in ids], np.int64)
    ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    fv = FeatureVectorC(_ptr(ids) if len(ids) else None, _ptr(ptr), _ptr(order) if len(order) else None, len(ids))
    return fv, (ids, ptr, order)


class Matcher:
    """orbx_matcher context."""

    def __init__(self, device=-1):
        h = C.c_void_p()
        _check(lib.orbx_matcher_create(device, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None) and lib is not None:
            lib.orbx_matcher_destroy(self.h)
            self.h = None

    __del__ = close

    def hamming_best2(self, q, t):
        q = np.ascontiguousarray(q, np.uint8)
        t = np.ascontiguousarray(t, np.uint8)
        out = np.zeros(q.shape[0], BEST2_DTYPE)
        _check(lib.orbx_hamming_best2(self.h, _ptr(q), q.shape[0], _ptr(t), t.shape[0], _ptr(out)))
        return out

    def hamming_best2_device(self, d_q, nq, d_t, nt, index_base, d_out, stream=None):
        _check(lib.orbx_hamming_best2_device(self.h, d_q, nq, d_t, nt, index_base, d_out, stream))

    def best2_tile_walks(self, reset=True):
        """Tiles of 16 rows x 16 queries whose keys the brute-force kernel walked on this GPU since the last reset."""
        v = C.c_uint64(0)
        _check(lib.orbx_debug_best2_tile_walks(self.h, C.byref(v), 1 if reset else 0))
        return int(v.value)

    def hamming_best2_device_static(self, d_q, nq, d_t, nt, index_base, d_out, stream=None):
        """The variant that allocates nothing (fixed device scratch)."""
        _check(lib.orbx_hamming_best2_device_static(self.h, d_q, nq, d_t, nt, index_base, d_out, stream))

    def best2_merge_device(self, d_parts, n_parts, nq, d_out, stream=None):
        _check(lib.orbx_best2_merge_device(self.h, d_parts, n_parts, nq, d_out, stream))

    def hamming_csr(self, q, t, rowptr, cand):
        q = np.ascontiguousarray(q, np.uint8)
        t = np.ascontiguousarray(t, np.uint8)
        rowptr = np.ascontiguousarray(rowptr, np.int32)
        cand = np.ascontiguousarray(cand, np.int32)
        dist = np.zeros(len(cand), np.int32)
        _check(lib.orbx_hamming_csr(self.h, _ptr(q), q.shape[0], _ptr(t), t.shape[0], _ptr(rowptr), _ptr(cand), _ptr(dist)))
        return dist

    def distinctive_descriptors(self, desc, group_ptr):
        desc = np.ascontiguousarray(desc, np.uint8); group_ptr = np.ascontiguousarray(group_ptr, np.int32)
        best = np.zeros(len(group_ptr) - 1, np.int32)
        _check(lib.orbx_distinctive_descriptors(self.h, _ptr(desc) if len(desc) else None, _ptr(group_ptr), len(best), _ptr(best)))
        return best

    def distinctive_descriptors_device(self, d_desc, d_group_ptr, n_groups, n_rows, d_best, stream=None):
        _check(lib.orbx_distinctive_descriptors_device(self.h, d_desc, d_group_ptr, n_groups, n_rows, d_best, stream))

    def search_for_initialization(self, k1, d1, k2, d2, bounds, prev_matched, window=100, nnratio=0.9, check_ori=True):
        k1 = np.ascontiguousarray(k1, KEYPOINT_DTYPE); k2 = np.ascontiguousarray(k2, KEYPOINT_DTYPE)
        d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
        prev = np.ascontiguousarray(prev_matched, np.float32).copy()
        m12 = np.full(len(k1), -1, np.int32)
        n = C.c_int()
        _check(lib.orbx_search_for_initialization(self.h, _ptr(k1), _ptr(d1), len(k1), _ptr(k2), _ptr(d2), len(k2),
                                                  bounds[0], bounds[1], bounds[2], bounds[3], _ptr(prev), window, nnratio,
                                                  int(check_ori), _ptr(m12), C.byref(n)))
        return m12, n.value, prev

    def search_by_projection_keyframe(self, qu, qv, qradius, qlevel, qdesc, k2, d2, bounds, claimed):
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        qu, qv, qradius = f32(qu), f32(qv), f32(qradius)
        qlevel = np.ascontiguousarray(qlevel, np.int32)
        qdesc = np.ascontiguousarray(qdesc, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
        k2 = np.ascontiguousarray(k2, KEYPOINT_DTYPE)
        claimed = np.ascontiguousarray(claimed, np.int32).copy()
        n = C.c_int()
        _check(lib.orbx_search_by_projection_keyframe(self.h, _ptr(qu), _ptr(qv), _ptr(qradius), _ptr(qlevel), _ptr(qdesc), len(qu),
                                                      _ptr(k2), _ptr(d2), len(k2), bounds[0], bounds[1], bounds[2], bounds[3],
                                                      _ptr(claimed), C.byref(n)))
        return claimed, n.value

    def project_best1(self, qu, qv, qradius, qlevel, qdesc, k2, d2, bounds, qur=None, uright2=None, inv_level_sigma2=None):
        """Candidate loop of ORBmatcher::Fuse; qur given = the reprojection test of the local-mapping form."""
        f32 = lambda a: None if a is None else np.ascontiguousarray(a, np.float32)
        qu, qv, qradius, qur, uright2, inv_level_sigma2 = f32(qu), f32(qv), f32(qradius), f32(qur), f32(uright2), f32(inv_level_sigma2)
        qlevel = np.ascontiguousarray(qlevel, np.int32); qdesc = np.ascontiguousarray(qdesc, np.uint8)
        k2 = np.ascontiguousarray(k2, KEYPOINT_DTYPE); d2 = np.ascontiguousarray(d2, np.uint8)
        bi = np.zeros(len(qu), np.int32); bd = np.zeros(len(qu), np.int32)
        _check(lib.orbx_project_best1(self.h, _ptr(qu), _ptr(qv), _ptr(qradius), _ptr(qlevel), _ptr(qdesc), len(qu), _ptr(k2), _ptr(d2), len(k2),
                                      bounds[0], bounds[1], bounds[2], bounds[3], _ptr(qur), _ptr(uright2), _ptr(inv_level_sigma2),
                                      0 if inv_level_sigma2 is None else len(inv_level_sigma2), _ptr(bi), _ptr(bd)))
        return bi, bd

    def search_by_sim3(self, side1, k1, d1, side2, k2, d2, bounds):
        """side = (valid, u, v, radius, level, map-point descriptors) of each key frame's features projected into the other."""
        def prep(side):
            valid, u, v, r, lvl, desc = side
            return (np.ascontiguousarray(valid, np.uint8), np.ascontiguousarray(u, np.float32), np.ascontiguousarray(v, np.float32),
                    np.ascontiguousarray(r, np.float32), np.ascontiguousarray(lvl, np.int32), np.ascontiguousarray(desc, np.uint8))
        a, b = prep(side1), prep(side2)
        k1 = np.ascontiguousarray(k1, KEYPOINT_DTYPE); k2 = np.ascontiguousarray(k2, KEYPOINT_DTYPE)
        d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
        m12 = np.zeros(len(k1), np.int32); n = C.c_int()
        _check(lib.orbx_search_by_sim3(self.h, *[_ptr(x) for x in a], _ptr(k1), _ptr(d1), len(k1), *[_ptr(x) for x in b], _ptr(k2), _ptr(d2),
                                       len(k2), bounds[0], bounds[1], bounds[2], bounds[3], _ptr(m12), C.byref(n)))
        return m12, n.value

    def search_by_bow(self, d1, angle1, eligible1, nodes1, d2, angle2, eligible2, nodes2, nnratio=0.7, check_ori=True, keyframe_form=0):
        d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
        a1 = np.ascontiguousarray(angle1, np.float32); a2 = np.ascontiguousarray(angle2, np.float32)
        e1 = np.ascontiguousarray(eligible1, np.uint8)
        e2 = None if eligible2 is None else np.ascontiguousarray(eligible2, np.uint8)
        fv1, keep1 = feature_vector(nodes1); fv2, keep2 = feature_vector(nodes2)
        m12 = np.full(len(d1), -1, np.int32); m21 = np.full(len(d2), -1, np.int32)
        n = C.c_int()
        _check(lib.orbx_search_by_bow(self.h, _ptr(d1), _ptr(a1), _ptr(e1), len(d1), C.byref(fv1), _ptr(d2), _ptr(a2),
                                      None if e2 is None else _ptr(e2), len(d2), C.byref(fv2), nnratio, int(check_ori), keyframe_form,
                                      _ptr(m12), _ptr(m21), C.byref(n)))
        return m12, m21, n.value

    def search_for_triangulation(self, k1, d1, has_mp1, uright1, nodes1, k2, d2, has_mp2, uright2, nodes2, F12, ex, ey,
                                 scale2, sigma2_2, only_stereo=False, check_ori=True):
        k1 = np.ascontiguousarray(k1, KEYPOINT_DTYPE); k2 = np.ascontiguousarray(k2, KEYPOINT_DTYPE)
        d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
        h1 = np.ascontiguousarray(has_mp1, np.uint8); h2 = np.ascontiguousarray(has_mp2, np.uint8)
        u1 = np.ascontiguousarray(uright1, np.float32); u2 = np.ascontiguousarray(uright2, np.float32)
        F = np.ascontiguousarray(F12, np.float32).reshape(9)
        sc = np.ascontiguousarray(scale2, np.float32); sg = np.ascontiguousarray(sigma2_2, np.float32)
        fv1, keep1 = feature_vector(nodes1); fv2, keep2 = feature_vector(nodes2)
        m12 = np.full(len(k1), -1, np.int32)
        n = C.c_int()
        _check(lib.orbx_search_for_triangulation(self.h, _ptr(k1), _ptr(d1), _ptr(h1), _ptr(u1), len(k1), C.byref(fv1),
                                                 _ptr(k2), _ptr(d2), _ptr(h2), _ptr(u2), len(k2), C.byref(fv2), _ptr(F), ex, ey,
                                                 _ptr(sc), _ptr(sg), len(sc), int(only_stereo), int(check_ori), _ptr(m12), C.byref(n)))
        return m12, n.value

    def search_for_initialization_batch_device(self, d_k1, d_d1, d_n1, d_k2, d_d2, d_n2, cap, n_pairs, bounds, d_prev, window, nnratio,
                                               check_ori, d_matches12, d_nmatches, stream):
        """All arguments starting with d_ are device pointers (ints)."""
        _check(lib.orbx_search_for_initialization_batch_device(self.h, d_k1, d_d1, d_n1, d_k2, d_d2, d_n2, cap, n_pairs, bounds[0], bounds[1],
                                                               bounds[2], bounds[3], d_prev, window, nnratio, int(check_ori), d_matches12,
                                                               d_nmatches, stream))

    def search_by_projection_batch_device(self, form, d_qu, d_qv, d_qr, d_qur, d_qlevel, d_qangle, d_qdesc, d_qtransient, d_nq, qcap,
                                          d_k2, d_d2, d_uright2, d_n2, cap, n_calls, bounds, mode, th_dist, nnratio, check_ori, d_claimed,
                                          d_nmatches, stream):
        """All arguments starting with d_ are device pointers (ints; None for the optional ones)."""
        _check(lib.orbx_search_by_projection_batch_device(self.h, form, d_qu, d_qv, d_qr, d_qur, d_qlevel, d_qangle, d_qdesc, d_qtransient, d_nq,
                                                          qcap, d_k2, d_d2, d_uright2, d_n2, cap, n_calls, bounds[0], bounds[1], bounds[2],
                                                          bounds[3], mode, th_dist, nnratio, int(check_ori), d_claimed, d_nmatches, stream))

    def search_by_projection_mappoints(self, qu, qv, qr, qur, qlevel, qdesc, k2, d2, uright2, bounds, nnratio, claimed, transient=None):
        transient = None if transient is None else np.ascontiguousarray(transient, np.uint8)
        f32 = lambda a: None if a is None else np.ascontiguousarray(a, np.float32)
        qu, qv, qr, qur, uright2 = f32(qu), f32(qv), f32(qr), f32(qur), f32(uright2)
        qlevel = np.ascontiguousarray(qlevel, np.int32)
        qdesc = np.ascontiguousarray(qdesc, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
        k2 = np.ascontiguousarray(k2, KEYPOINT_DTYPE)
        claimed = np.ascontiguousarray(claimed, np.int32).copy()
        n = C.c_int()
        _check(lib.orbx_search_by_projection_mappoints(self.h, _ptr(qu), _ptr(qv), _ptr(qr), _ptr(qur), _ptr(qlevel), _ptr(qdesc), len(qu),
                                                       _ptr(k2), _ptr(d2), _ptr(uright2), len(k2), bounds[0], bounds[1], bounds[2], bounds[3],
                                                       nnratio, _ptr(transient), _ptr(claimed), C.byref(n)))
        return claimed, n.value

    def search_by_projection_lastframe(self, qu, qv, qradius, qur, qoctave, qangle, qdesc, k2, d2, uright2, bounds, mode, th_dist,
                                       check_ori, claimed, transient=None):
        transient = None if transient is None else np.ascontiguousarray(transient, np.uint8)
        f32 = lambda a: None if a is None else np.ascontiguousarray(a, np.float32)
        qu, qv, qradius, qur, qangle, uright2 = f32(qu), f32(qv), f32(qradius), f32(qur), f32(qangle), f32(uright2)
        qoctave = np.ascontiguousarray(qoctave, np.int32)
        qdesc = np.ascontiguousarray(qdesc, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
        k2 = np.ascontiguousarray(k2, KEYPOINT_DTYPE)
        claimed = np.ascontiguousarray(claimed, np.int32).copy()
        n = C.c_int()
        _check(lib.orbx_search_by_projection_lastframe(self.h, _ptr(qu), _ptr(qv), _ptr(qradius), _ptr(qur), _ptr(qoctave), _ptr(qangle),
                                                       _ptr(qdesc), len(qu), _ptr(k2), _ptr(d2), _ptr(uright2), len(k2),
                                                       bounds[0], bounds[1], bounds[2], bounds[3], mode, th_dist, int(check_ori),
                                                       _ptr(transient), _ptr(claimed), C.byref(n)))
        return claimed, n.value

    def stereo_match(self, kl, dl, kr, dr, scale_factors, n_rows, min_d, max_d):
        kl = np.ascontiguousarray(kl, KEYPOINT_DTYPE); kr = np.ascontiguousarray(kr, KEYPOINT_DTYPE)
        dl = np.ascontiguousarray(dl, np.uint8); dr = np.ascontiguousarray(dr, np.uint8)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        bd = np.zeros(len(kl), np.int32); bi = np.zeros(len(kl), np.int32)
        _check(lib.orbx_stereo_match(self.h, _ptr(kl), _ptr(dl), len(kl), _ptr(kr), _ptr(dr), len(kr), _ptr(sf), len(sf), n_rows,
                                     min_d, max_d, _ptr(bd), _ptr(bi)))
        return bd, bi

    def stereo_match_batch_device(self, d_kl, d_dl, d_nl, d_kr, d_dr, d_nr, cap, n_pairs, scale_factors, n_rows, min_d, max_d,
                                  d_best_dist, d_best_idx, stream=None):
        sf = np.ascontiguousarray(scale_factors, np.float32)
        _check(lib.orbx_stereo_match_batch_device(self.h, d_kl, d_dl, d_nl, d_kr, d_dr, d_nr, cap, n_pairs, _ptr(sf), len(sf), n_rows,
                                                  min_d, max_d, d_best_dist, d_best_idx, stream))


def synth_vocabulary(seed, k, L):
    """orbx_synth_vocabulary: (parent, is_leaf, descriptors, weight) of a full k-ary tree of depth L."""
    n = lib.orbx_synth_vocabulary(seed, k, L, None, None, None, None)
    if n < 0:
        raise ValueError("bad k / L")
    parent = np.zeros(n, np.int32); leaf = np.zeros(n, np.uint8); desc = np.zeros((n, 32), np.uint8); weight = np.zeros(n, np.float64)
    lib.orbx_synth_vocabulary(seed, k, L, _ptr(parent), _ptr(leaf), _ptr(desc), _ptr(weight))
    return parent, leaf, desc, weight


def mat_record_write(mat, elem_type=0):
    """The cv::Mat payload of the reference's map archives for a 2-D array (elem_type: the OpenCV type code, CV_8U = 0)."""
    mat = np.ascontiguousarray(mat)
    rows, cols = mat.shape
    out = np.zeros(lib.orbx_mat_record_bytes(rows, cols, mat.itemsize), np.uint8)
    n = lib.orbx_mat_record_write(_ptr(mat.view(np.uint8)) if mat.size else None, rows, cols, mat.itemsize, elem_type, _ptr(out), len(out))
    if n != len(out):
        raise ValueError("record not written")
    return out.tobytes()


def mat_record_read(buf):
    """Returns (array of raw bytes rows x (cols*elem_size), elem_size, elem_type, record length)."""
    raw = np.frombuffer(buf, np.uint8)
    rows, cols, es, et = _i(), _i(), _i(), _i()
    data = C.c_void_p()
    n = lib.orbx_mat_record_read(_ptr(raw), len(raw), C.byref(rows), C.byref(cols), C.byref(es), C.byref(et), C.byref(data))
    if n < 0:
        raise ValueError("truncated or inconsistent record")
    return raw[24:n].reshape(rows.value, cols.value * es.value).copy(), es.value, et.value, n


class Remapper:
    """orbx_remapper: cv::remap(INTER_LINEAR) with fixed float maps (stereo rectification)."""

    def __init__(self, map_x, map_y, src_w, src_h, device=-1):
        map_x = np.ascontiguousarray(map_x, np.float32); map_y = np.ascontiguousarray(map_y, np.float32)
        assert map_x.shape == map_y.shape
        self.dh, self.dw = map_x.shape
        self.sw, self.sh = src_w, src_h
        h = C.c_void_p()
        _check(lib.orbx_remap_create(device, _ptr(map_x), _ptr(map_y), map_x.strides[0] // 4, self.dw, self.dh, src_w, src_h, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None) and lib is not None:
            lib.orbx_remap_destroy(self.h)
            self.h = None

    __del__ = close

    def remap(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        assert img.shape == (self.sh, self.sw)
        out = np.zeros((self.dh, self.dw), np.uint8)
        _check(lib.orbx_remap_image(self.h, _ptr(img), img.strides[0], _ptr(out), out.strides[0]))
        return out

    def remap_batch_device(self, d_src, src_row_stride, src_frame_stride, n_frames, d_dst, dst_row_stride, dst_frame_stride, stream=None):
        _check(lib.orbx_remap_batch_device(self.h, d_src, src_row_stride, src_frame_stride, n_frames, d_dst, dst_row_stride, dst_frame_stride, stream))


class Vocabulary:
    """orbx_vocabulary: a DBoW2 tree in device memory (scoring / weighting: the enum values of DBoW2's BowVector.h)."""

    def __init__(self, k, L, parent, is_leaf, descriptors, weight, scoring=0, weighting=0, device=-1):
        parent = np.ascontiguousarray(parent, np.int32); is_leaf = np.ascontiguousarray(is_leaf, np.uint8)
        descriptors = np.ascontiguousarray(descriptors, np.uint8); weight = np.ascontiguousarray(weight, np.float64)
        h = C.c_void_p()
        _check(lib.orbx_vocabulary_create(device, k, L, scoring, weighting, len(parent), _ptr(parent), _ptr(is_leaf), _ptr(descriptors),
                                          _ptr(weight), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None) and lib is not None:
            lib.orbx_vocabulary_destroy(self.h)
            self.h = None

    __del__ = close

    def words(self):
        return lib.orbx_vocabulary_words(self.h)

    def descend_device(self, d_desc, d_counts, cap, n_frames, levelsup, d_word, d_node, stream=None):
        _check(lib.orbx_vocabulary_descend_device(self.h, d_desc, d_counts, cap, n_frames, levelsup, d_word, d_node, stream))

    def transform(self, desc, levelsup=4):
        """Returns (bow_ids, bow_values, fv_node_ids, fv_node_ptr, fv_features)."""
        desc = np.ascontiguousarray(desc, np.uint8)
        n = desc.shape[0]
        ids = np.zeros(max(n, 1), np.uint32); vals = np.zeros(max(n, 1), np.float64)
        fn = np.zeros(max(n, 1), np.uint32); fp = np.zeros(n + 1, np.int32); ff = np.zeros(max(n, 1), np.int32)
        nw = C.c_int32(0); nn = C.c_int32(0)
        _check(lib.orbx_vocabulary_transform(self.h, _ptr(desc) if n else None, n, levelsup, _ptr(ids), _ptr(vals), C.byref(nw),
                                             _ptr(fn), _ptr(fp), _ptr(ff), C.byref(nn)))
        return ids[:nw.value], vals[:nw.value], fn[:nn.value], fp[:nn.value + 1], ff[:fp[nn.value]]


class FrameGrid:
    def __init__(self, kps, bounds):
        self.kps = np.ascontiguousarray(kps, KEYPOINT_DTYPE)
        h = C.c_void_p()
        _check(lib.orbx_grid_create(_ptr(self.kps), len(self.kps), bounds[0], bounds[1], bounds[2], bounds[3], C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None) and lib is not None:
            lib.orbx_grid_destroy(self.h)
            self.h = None

    __del__ = close

    def features_in_area(self, x, y, r, min_level=-1, max_level=-1):
        out = np.zeros(max(len(self.kps), 1), np.int32)
        n = C.c_int()
        _check(lib.orbx_grid_features_in_area(self.h, x, y, r, min_level, max_level, _ptr(out), len(out), C.byref(n)))
        return out[:n.value]
