"""
ctypes binding of libmtm_hip.so (C ABI: include/mtm_hip.h).

There is NO CPU fallback: if the library cannot be loaded, or no GPU is visible when a context is
requested, the call fails loudly.  The oracle under oracle/ is test infrastructure and is never
imported from here.
"""
import ctypes
import struct
import os
import threading
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MTM_LIB_PATH: a differently built libmtm_hip.so (kernel timing experiments, tools/probes); never a fallback
LIB_PATH = os.environ.get("MTM_LIB_PATH") or os.path.join(_HERE, "libmtm_hip.so")

MTM_U8, MTM_F32, MTM_U16 = 0, 1, 2
GROUP_EXCHANGE_HOST, GROUP_EXCHANGE_RCCL = 0, 1
PEAKS_LOCAL, PEAKS_GLOBAL = 0, 1
BORDER_CONSTANT, BORDER_NEAREST = 0, 1
KERNEL_AUTO, KERNEL_NAIVE, KERNEL_DOT4, KERNEL_MFMA = 0, 1, 2, 3
OPT_KERNEL, OPT_PEAK_BORDER, OPT_HIT_CAPACITY, OPT_DOT4_VARIANT, OPT_EXACT_DIV, OPT_HITS_ONLY, OPT_F32_MFMA = 1, 2, 3, 4, 5, 6, 7
ALL_OPTIONS = (OPT_KERNEL, OPT_PEAK_BORDER, OPT_HIT_CAPACITY, OPT_DOT4_VARIANT, OPT_EXACT_DIV, OPT_HITS_ONLY, OPT_F32_MFMA)
POISON_SCRATCH, POISON_LDS, POISON_ARENAS = 1, 2, 4
E_OVERFLOW = -5
E_HIP = -2
COMM_ID_BYTES = 128
ABI_VERSION = 9


class MtmTempl(ctypes.Structure):
    _fields_ = [("px", ctypes.c_void_p), ("mask", ctypes.c_void_p),
                ("rows", ctypes.c_int32), ("cols", ctypes.c_int32),
                ("chans", ctypes.c_int32), ("dtype", ctypes.c_int32),
                ("row_stride", ctypes.c_int64), ("mask_row_stride", ctypes.c_int64)]


class MtmHit(ctypes.Structure):
    _fields_ = [("templ_idx", ctypes.c_int32), ("x", ctypes.c_int32), ("y", ctypes.c_int32),
                ("w", ctypes.c_int32), ("h", ctypes.c_int32), ("score", ctypes.c_float)]


class MtmTiming(ctypes.Structure):
    _fields_ = [("total_ms", ctypes.c_float), ("score_ms", ctypes.c_float),
                ("peaks_ms", ctypes.c_float), ("ncc_kernel_ms", ctypes.c_float),
                ("ncc_launches", ctypes.c_int32), ("kernel_used", ctypes.c_int32),
                ("n_hits", ctypes.c_int64), ("hits_only", ctypes.c_int32), ("sclk_mhz", ctypes.c_float),
                ("ncc_sum_ms", ctypes.c_float), ("f32_route", ctypes.c_int32),
                ("sq_launches", ctypes.c_int32), ("masked_stat_ms", ctypes.c_float),
                ("f32_pieces", ctypes.c_int32)]


HIT_DTYPE = np.dtype([("templ_idx", "<i4"), ("x", "<i4"), ("y", "<i4"), ("w", "<i4"), ("h", "<i4"),
                      ("score", "<f4")])
assert HIT_DTYPE.itemsize == ctypes.sizeof(MtmHit) == 24
# mtm_templ as a numpy record: a whole template list is filled column-wise instead of field by field
TEMPL_DTYPE = np.dtype([("px", "<u8"), ("mask", "<u8"), ("rows", "<i4"), ("cols", "<i4"), ("chans", "<i4"),
                        ("dtype", "<i4"), ("row_stride", "<i8"), ("mask_row_stride", "<i8")])
assert TEMPL_DTYPE.itemsize == ctypes.sizeof(MtmTempl) == 48
# mtm_variant: one augmentation of a base template (mtm_set_templates_augmented)
VARIANT_DTYPE = np.dtype([("rot90", "<i4"), ("flip_lr", "<i4"), ("flip_ud", "<i4"), ("rows", "<i4"), ("cols", "<i4"),
                          ("down", "<i4")])

# every symbol include/mtm_hip.h declares: (restype, argtypes)
_P = ctypes.POINTER
SYMBOLS = {
    "mtm_abi_version": (ctypes.c_int, []),
    "mtm_device_count": (ctypes.c_int, []),
    "mtm_last_error": (ctypes.c_char_p, []),
    "mtm_ctx_create": (ctypes.c_int, [_P(ctypes.c_void_p), ctypes.c_int]),
    "mtm_ctx_destroy": (None, [ctypes.c_void_p]),
    "mtm_set_option": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64]),
    "mtm_get_option": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, _P(ctypes.c_int64)]),
    "mtm_debug_poison": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
    "mtm_debug_quotient_check": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, _P(ctypes.c_uint64)]),
    "mtm_set_image": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_int, ctypes.c_int64]),
    "mtm_set_image_downscaled": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_int64, ctypes.c_int]),
    "mtm_set_templates": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
    "mtm_set_templates_augmented": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                                   ctypes.c_int]),
    "mtm_score_map": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64]),
    "mtm_find_matches": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_void_p,
                                        ctypes.c_int64, _P(ctypes.c_int64)]),
    "mtm_find_matches_image": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_double, ctypes.c_void_p,
                                              ctypes.c_int64, _P(ctypes.c_int64)]),
    "mtm_find_matches_image_nms": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                  ctypes.c_int, ctypes.c_int64, ctypes.c_double, ctypes.c_double, ctypes.c_int64,
                                                  ctypes.c_void_p, ctypes.c_int64, _P(ctypes.c_int64)]),
    "mtm_find_matches_image_sharded_nms": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                          ctypes.c_int, ctypes.c_int64, ctypes.c_double, ctypes.c_double,
                                                          ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                                          ctypes.c_void_p, ctypes.c_int64, _P(ctypes.c_int64)]),
    "mtm_find_matches_next": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_void_p,
                                             ctypes.c_int64, _P(ctypes.c_int64), ctypes.c_void_p, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64]),
    "mtm_find_matches_async": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_double]),
    "mtm_find_matches_wait": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]),
    "mtm_last_hits": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, _P(ctypes.c_int64)]),
    "mtm_last_score_map": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64]),
    "mtm_get_timing": (ctypes.c_int, [ctypes.c_void_p, _P(MtmTiming)]),
    "mtm_nms": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_double, ctypes.c_int,
                               ctypes.c_int64, ctypes.c_double, ctypes.c_void_p, _P(ctypes.c_int64)]),
    "mtm_group_create": (ctypes.c_int, [_P(ctypes.c_void_p), _P(ctypes.c_int), ctypes.c_int]),
    "mtm_group_destroy": (None, [ctypes.c_void_p]),
    "mtm_group_size": (ctypes.c_int, [ctypes.c_void_p]),
    "mtm_group_ctx": (ctypes.c_void_p, [ctypes.c_void_p, ctypes.c_int]),
    "mtm_group_set_option": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64]),
    "mtm_group_shards": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_void_p]),
    "mtm_group_find_matches": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                              ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64,
                                              ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_int64,
                                              _P(ctypes.c_int64)]),
    "mtm_group_find_matches_nms": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                                  ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64,
                                                  ctypes.c_double, ctypes.c_double, ctypes.c_int64, ctypes.c_void_p,
                                                  ctypes.c_int64, _P(ctypes.c_int64)]),
    "mtm_group_last_hits": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, _P(ctypes.c_int64)]),
    "mtm_group_comm_init": (ctypes.c_int, [ctypes.c_void_p]),
    "mtm_group_comm_ranks": (ctypes.c_int, [ctypes.c_void_p]),
    "mtm_group_set_exchange": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "mtm_group_exchange_used": (ctypes.c_int, [ctypes.c_void_p]),
    "mtm_host_alloc": (ctypes.c_void_p, [ctypes.c_size_t]),
    "mtm_host_free": (None, [ctypes.c_void_p]),
    "mtm_comm_unique_id": (ctypes.c_int, [ctypes.c_void_p]),
    "mtm_comm_init": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
    "mtm_comm_allgather_hits": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                               ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                               _P(ctypes.c_int64)]),
    "mtm_comm_last_gather": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                            _P(ctypes.c_int64)]),
    "mtm_comm_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "mtm_comm_init_all": (ctypes.c_int, [_P(ctypes.c_void_p), ctypes.c_int]),
    "mtm_comm_count": (ctypes.c_int, [ctypes.c_void_p]),
    "mtm_comm_allgather_hits_all": (ctypes.c_int, [_P(ctypes.c_void_p), ctypes.c_int, _P(ctypes.c_void_p), _P(ctypes.c_int64),
                                                   ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, _P(ctypes.c_int64)]),
}

_lib = None
_lib_lock = threading.Lock()


class MtmError(RuntimeError):
    """A libmtm_hip call failed (message from mtm_last_error)."""


def load():
    """Load libmtm_hip.so and declare every prototype.  Raises if the library is missing."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise MtmError(
                "libmtm_hip.so is not built (%s). Build it with "
                "`python multitemplatematching-python_amd/build.py` (needs hipcc); this package has "
                "no CPU fallback." % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)       # ctypes releases the GIL during every call
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)       # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if lib.mtm_abi_version() != ABI_VERSION:
            raise MtmError("libmtm_hip.so ABI version mismatch")
        _lib = lib
        return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().mtm_last_error()
        raise MtmError("%s failed (%d): %s" % (what or "libmtm_hip call", rc, (msg or b"").decode()))


def _pixel_rows(a):
    """Return (array_kept_alive, pointer, row_stride_bytes) for a (rows, cols[, C]) array whose
    rows have contiguous pixels; anything else (e.g. a transposed view) is copied."""
    ai = a.__array_interface__
    if ai["strides"] is None:           # C-contiguous (the usual case): one dictionary look-up, no checks needed
        shp = ai["shape"]
        return a, ai["data"][0], (shp[1] * shp[2] if len(shp) == 3 else shp[1]) * a.itemsize
    item = a.itemsize
    ok = a.strides[-1] == item and (a.ndim == 2 or a.strides[1] == a.shape[2] * item)
    if a.ndim == 3 and a.shape[2] == 1:
        ok = ok or a.strides[1] == item
    if not ok or a.strides[0] < a.shape[1] * (a.shape[2] if a.ndim == 3 else 1) * item or not a.flags.aligned:
        a = np.ascontiguousarray(a)
    return a, a.ctypes.data, a.strides[0]


def _dtype_code(a):
    if a.dtype == np.uint8:
        return MTM_U8
    if a.dtype == np.float32:
        return MTM_F32
    if a.dtype == np.uint16:
        return MTM_U16
    raise MtmError("libmtm_hip takes uint8, uint16 or float32 pixels (got %s)" % a.dtype)


_DT_CODES = {np.dtype(np.uint8): MTM_U8, np.dtype(np.float32): MTM_F32, np.dtype(np.uint16): MTM_U16}


_PACK_TEMPL = struct.Struct("<QQiiiiqq").pack_into      # one mtm_templ record (TEMPL_DTYPE)
_TYPESTR_CODES = {"|u1": MTM_U8, "<u2": MTM_U16, "<f4": MTM_F32}


def templ_records(templates):
    """[(template, mask or None), ...] (pixel policy already applied) -> (mtm_templ records, arrays kept alive).
    On the path of every call whose template objects are new (32 templates: 40 us; 78 us as field-wise numpy assignments)."""
    n = len(templates)
    buf = bytearray(48 * max(n, 1))
    keep = []
    off = 0
    for t, m in templates:
        ai = t.__array_interface__
        if ai["strides"] is None:           # C-contiguous (the usual case)
            shp = ai["shape"]
            tp = ai["data"][0]
            c = shp[2] if len(shp) == 3 else 1
            ts = shp[1] * c * t.itemsize
            code = _TYPESTR_CODES.get(ai["typestr"])
        else:
            t, tp, ts = _pixel_rows(t)
            shp = t.shape
            c = shp[2] if len(shp) == 3 else 1
            code = _DT_CODES.get(t.dtype)
        if code is None:
            raise MtmError("libmtm_hip takes uint8, uint16 or float32 pixels (got %s)" % t.dtype)
        keep.append(t)
        if m is not None:
            m, mp, mst = _pixel_rows(m)
            keep.append(m)
        else:
            mp = mst = 0
        _PACK_TEMPL(buf, off, tp, mp, shp[0], shp[1], c, code, ts, mst)
        off += 48
    return np.frombuffer(buf, dtype=TEMPL_DTYPE), keep


class FrozenUnits(list):
    """A list of (template, mask) units that its owner promises never to mutate (MTM._ListMemo)."""
    __slots__ = ()


def _zero_copy(templates, keep):
    """True if templ_records handed the caller's own buffers to the library (no array had to be copied)."""
    it = iter(keep)
    for t, m in templates:
        if next(it) is not t or (m is not None and next(it) is not m):
            return False
    return True


class _PinnedBlock:
    """Owner of one mtm_host_alloc block (freed with the last numpy view of it)."""

    def __init__(self, nbytes):
        lib = load()
        self.ptr = lib.mtm_host_alloc(int(nbytes))
        if not self.ptr:
            check(E_HIP, "mtm_host_alloc")
        self.nbytes = int(nbytes)
        self.__array_interface__ = {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 3}

    def __del__(self):
        ptr, self.ptr = getattr(self, "ptr", None), None
        if ptr and _lib is not None:
            _lib.mtm_host_free(ptr)


def pinned_empty(shape, dtype=np.uint8):
    """An uninitialised numpy array in page-locked host memory (mtm_host_alloc): images kept in such arrays are
    uploaded by plain DMA transfers, without the staging copy pageable memory needs."""
    dtype = np.dtype(dtype)
    shape = (int(shape),) if np.isscalar(shape) else tuple(int(v) for v in shape)
    n = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    block = _PinnedBlock(max(n, 1))
    return np.asarray(block)[:n].view(dtype).reshape(shape)       # the views keep `block` alive


class _RecordMemo:
    """Marshalled template lists, memoised (shared by Context and Group)."""
    _rec_key = _rec = _rec_keep = _rec_src = None

    def _records(self, templates):
        """templ_records(templates), memoised on the identity (and shape) of the arrays: a caller that passes the same
        template objects call after call - the usual loop over images - pays for the marshalling once.  The arrays are
        kept referenced, so an id cannot be recycled; changed PIXELS are the library's business (it compares the bytes
        with the copy it packed from in every mtm_set_templates)."""
        # (a unit list MTM's own memo hands over again - never mutated, shapes re-checked by the caller in this very call)
        if type(templates) is FrozenUnits and templates is self._rec_src and self._rec_key is not None:
            return self._rec
        key = [(id(t), t.shape, id(m)) for t, m in templates]
        if key != self._rec_key:
            self._rec, keep = templ_records(templates)
            self._rec_keep = (keep, [t for t, _ in templates], [m for _, m in templates])
            # Only records that point at the caller's own buffers may be reused: a template that had to be copied
            # (np.rot90(base), base[:, ::-1], base.T ...) would otherwise be matched from the copy of the FIRST call
            # for ever, even after the caller edited the array in place - the reference re-reads it on every call.
            self._rec_key = key if _zero_copy(templates, keep) else None
        self._rec_src = templates
        return self._rec


_LIVE = weakref.WeakSet()          # contexts that exist right now (test support: live_contexts)


def live_contexts():
    return [c for c in list(_LIVE) if c._h]


class Context(_RecordMemo):
    """One GPU context (single caller: guarded by a lock)."""

    def __init__(self, device=None):
        lib = load()
        if device is None:
            device = int(os.environ.get("MTM_DEVICE", os.environ.get("LOCAL_RANK", "0")))
            n = lib.mtm_device_count()
            if n > 0:
                device %= n
        h = ctypes.c_void_p()
        check(lib.mtm_ctx_create(ctypes.byref(h), int(device)), "mtm_ctx_create")
        self._lib = lib
        self._h = h
        self.device = device
        self.lock = threading.RLock()
        self._keep = []
        self._rec_key, self._rec, self._rec_keep = None, None, None
        _LIVE.add(self)
        k = os.environ.get("MTM_KERNEL")
        if k:
            self.set_option(OPT_KERNEL, {"auto": 0, "naive": 1, "dot4": 2, "mfma": 3}[k.lower()])
        b = os.environ.get("MTM_PEAK_BORDER")
        if b:
            self.set_option(OPT_PEAK_BORDER, {"constant": 0, "nearest": 1}[b.lower()])

    def close(self):
        _LIVE.discard(self)
        if self._h:
            self._lib.mtm_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    def set_option(self, opt, value):
        check(self._lib.mtm_set_option(self._h, int(opt), int(value)), "mtm_set_option")

    def get_option(self, opt):
        v = ctypes.c_int64(0)
        check(self._lib.mtm_get_option(self._h, int(opt), ctypes.byref(v)), "mtm_get_option")
        return int(v.value)

    def options(self):
        """{option: value} of every MTM_OPT_* option (what a fresh context starts with is `DEFAULT_OPTIONS` after the
        environment switches have been applied: compare with Context(...).options())."""
        return {o: self.get_option(o) for o in ALL_OPTIONS}

    def debug_poison(self, pattern=0xFF, what=7):
        """Test support (mtm_debug_poison): a byte pattern into every wave slot's scratch memory (1), every CU's LDS (2) and
        the context's per-call work buffers (4) - memory no result may depend on."""
        check(self._lib.mtm_debug_poison(self._h, int(pattern), int(what)), "mtm_debug_poison")

    def debug_quotient_check(self, n_cases=1 << 28, seed=1):
        """Test support (mtm_debug_quotient_check): the epilogue's division-free quotient against the IEEE division on
        n_cases operand triples -> dict(cases, mismatches, took_division, max_ulp_distance)."""
        out = (ctypes.c_uint64 * 4)()
        check(self._lib.mtm_debug_quotient_check(self._h, int(n_cases), int(seed), out), "mtm_debug_quotient_check")
        return {"cases": int(out[0]), "mismatches": int(out[1]), "took_division": int(out[2]), "max_ulp_distance": int(out[3])}

    def set_image(self, image, downscale=1):
        """Upload the search image; `downscale` > 1 area-averages it by that integer factor on the
        device (mtm_set_image_downscaled)."""
        a, ptr, stride = _pixel_rows(image)
        chans = 1 if a.ndim == 2 else a.shape[2]
        check(self._lib.mtm_set_image_downscaled(self._h, ptr, a.shape[0], a.shape[1], chans, _dtype_code(a), stride,
                                                 int(downscale)), "mtm_set_image")


    def set_templates(self, templates, method):
        """templates: list of (array, mask_or_None) with identical dtype policy already applied."""
        rec = self._records(templates)
        check(self._lib.mtm_set_templates(self._h, rec.ctypes.data, len(templates), int(method)), "mtm_set_templates")

    def set_templates_augmented(self, bases, variants, method):
        """bases: list of (uint8 array, uint8 mask or None); variants: VARIANT_DTYPE records.  Units are
        base-major: unit index = base * len(variants) + variant (mtm_set_templates_augmented)."""
        rec, keep = templ_records(bases)
        var = np.ascontiguousarray(variants, dtype=VARIANT_DTYPE)
        check(self._lib.mtm_set_templates_augmented(self._h, rec.ctypes.data, len(bases), var.ctypes.data, len(var),
                                                    int(method)), "mtm_set_templates_augmented")

    def search(self, templates, image, method, mode, score_threshold):
        """One search = templates + image in, hit records out (the engine interface shared with Group)."""
        self.set_templates(templates, method)
        return self.find_matches_image(image, mode, score_threshold)

    def score_map(self, idx, shape):
        out = np.empty(shape, dtype=np.float32)
        check(self._lib.mtm_score_map(self._h, int(idx), out.ctypes.data, out.strides[0]), "mtm_score_map")
        return out

    def find_matches(self, mode, score_threshold, next_image=None):
        """Hits of the current image.  With `next_image`, that image is uploaded while the kernels run
        and is the current image when the call returns (mtm_find_matches_next)."""
        cap = 4096
        out = np.empty(cap, dtype=HIT_DTYPE)
        n = ctypes.c_int64(0)
        if next_image is None:
            rc = self._lib.mtm_find_matches(self._h, int(mode), float(score_threshold), out.ctypes.data, cap,
                                            ctypes.byref(n))
        else:
            a, ptr, stride = _pixel_rows(next_image)
            chans = 1 if a.ndim == 2 else a.shape[2]
            rc = self._lib.mtm_find_matches_next(self._h, int(mode), float(score_threshold), out.ctypes.data, cap,
                                                 ctypes.byref(n), ptr, a.shape[0], a.shape[1], chans,
                                                 _dtype_code(a), stride)
        if rc == E_OVERFLOW:        # the result stays in the context: fetch it, do not recompute
            cap = int(n.value)
            out = np.empty(cap, dtype=HIT_DTYPE)
            rc = self._lib.mtm_last_hits(self._h, out.ctypes.data, cap, ctypes.byref(n))
        check(rc, "mtm_find_matches")
        return out[:n.value]

    def find_matches_image(self, image, mode, score_threshold):
        """set_image + find_matches in one native call (mtm_find_matches_image): no round trip in between, the
        image crosses PCIe in row bands under the score kernel where the layout allows."""
        a, ptr, stride = _pixel_rows(image)
        chans = 1 if a.ndim == 2 else a.shape[2]
        cap = 4096
        out = np.empty(cap, dtype=HIT_DTYPE)
        n = ctypes.c_int64(0)
        rc = self._lib.mtm_find_matches_image(self._h, ptr, a.shape[0], a.shape[1], chans, _dtype_code(a), stride,
                                              int(mode), float(score_threshold), out.ctypes.data, cap, ctypes.byref(n))
        if rc == E_OVERFLOW:
            cap = int(n.value)
            out = np.empty(cap, dtype=HIT_DTYPE)
            rc = self._lib.mtm_last_hits(self._h, out.ctypes.data, cap, ctypes.byref(n))
        check(rc, "mtm_find_matches_image")
        return out[:n.value]

    def search_nms(self, templates, image, method, score_threshold, max_overlap, n_object=-1):
        """search() + MTM's non-maxima suppression in one native call (mtm_find_matches_image_nms): the kept hits, best
        first.  Dense images leave thousands of peaks on the device; most of the suppressed ones never leave it."""
        self.set_templates(templates, method)
        a, ptr, stride = _pixel_rows(image)
        chans = 1 if a.ndim == 2 else a.shape[2]
        cap = 4096
        out = np.empty(cap, dtype=HIT_DTYPE)
        n = ctypes.c_int64(0)
        rc = self._lib.mtm_find_matches_image_nms(self._h, ptr, a.shape[0], a.shape[1], chans, _dtype_code(a), stride,
                                                  float(score_threshold), float(max_overlap), int(n_object), out.ctypes.data,
                                                  cap, ctypes.byref(n))
        if rc == E_OVERFLOW:
            cap = int(n.value)
            out = np.empty(cap, dtype=HIT_DTYPE)
            rc = self._lib.mtm_last_hits(self._h, out.ctypes.data, cap, ctypes.byref(n))
        check(rc, "mtm_find_matches_image_nms")
        return out[:n.value]

    def search_sharded_nms(self, templates, image, method, score_threshold, max_overlap, n_object, global_idx):
        """This rank's step of a sharded matchTemplates in one native call (mtm_find_matches_image_sharded_nms): search
        `templates` (this rank's shard; global_idx = their positions in the whole list), all-gather the ranks' hits over
        the context's communicator, merge, suppress.  Collective: every rank calls it; every rank gets the same list."""
        gidx = np.ascontiguousarray(global_idx, dtype=np.int32)
        n_t = len(templates)
        if n_t:
            self.set_templates(templates, method)
            a, ptr, stride = _pixel_rows(image)
            chans = 1 if a.ndim == 2 else a.shape[2]
            shape, code = a.shape, _dtype_code(a)
        else:
            ptr, stride, chans, shape, code = None, 0, 1, (0, 0), MTM_U8
        cap = 4096
        out = np.empty(cap, dtype=HIT_DTYPE)
        n = ctypes.c_int64(0)
        rc = self._lib.mtm_find_matches_image_sharded_nms(self._h, ptr, shape[0], shape[1], chans, code, stride,
                                                          float(score_threshold), float(max_overlap), int(n_object),
                                                          int(method), gidx.ctypes.data, n_t, out.ctypes.data, cap, ctypes.byref(n))
        if rc == E_OVERFLOW:
            cap = int(n.value)
            out = np.empty(cap, dtype=HIT_DTYPE)
            rc = self._lib.mtm_last_hits(self._h, out.ctypes.data, cap, ctypes.byref(n))
        check(rc, "mtm_find_matches_image_sharded_nms")
        return out[:n.value]

    def find_matches_async(self, mode, score_threshold):
        """Start mtm_find_matches on the context's worker thread and return at once; collect the hits with
        find_matches_wait().  Nothing else may use the context in between."""
        check(self._lib.mtm_find_matches_async(self._h, int(mode), float(score_threshold)), "mtm_find_matches_async")

    def find_matches_wait(self):
        cap = 4096
        out = np.empty(cap, dtype=HIT_DTYPE)
        n = ctypes.c_int64(0)
        rc = self._lib.mtm_find_matches_wait(self._h, out.ctypes.data, cap, ctypes.byref(n))
        if rc == E_OVERFLOW:
            cap = int(n.value)
            out = np.empty(cap, dtype=HIT_DTYPE)
            rc = self._lib.mtm_last_hits(self._h, out.ctypes.data, cap, ctypes.byref(n))
        check(rc, "mtm_find_matches_wait")
        return out[:n.value]

    def last_score_map(self, idx, shape):
        """Score map of template `idx` as the last find_matches computed it (map mode only)."""
        out = np.empty(shape, dtype=np.float32)
        check(self._lib.mtm_last_score_map(self._h, int(idx), out.ctypes.data, out.strides[0]), "mtm_last_score_map")
        return out

    def timing(self):
        t = MtmTiming()
        check(self._lib.mtm_get_timing(self._h, ctypes.byref(t)), "mtm_get_timing")
        return {f: getattr(t, f) for f, _ in MtmTiming._fields_}

    # ---- RCCL hit exchange ------------------------------------------------------------------
    def comm_init(self, uid, n_ranks, rank):
        buf = (ctypes.c_char * COMM_ID_BYTES).from_buffer_copy(bytes(uid))
        check(self._lib.mtm_comm_init(self._h, buf, int(n_ranks), int(rank)), "mtm_comm_init")
        self.n_ranks = n_ranks

    def allgather_hits(self, local):
        """RCCL all-gather of hit records.  Collective: exactly ONE exchange per call on every rank - when the
        output buffer turns out too small (a local matter: the buffers are sized per rank) the gathered
        records are fetched from the context with mtm_comm_last_gather, the collective is not repeated."""
        local = np.ascontiguousarray(local, dtype=HIT_DTYPE)
        cap = 4096
        out = np.empty(cap, dtype=HIT_DTYPE)
        counts = np.zeros(self.n_ranks, dtype=np.int64)
        n = ctypes.c_int64(0)
        rc = self._lib.mtm_comm_allgather_hits(self._h, local.ctypes.data, len(local), out.ctypes.data, cap,
                                               counts.ctypes.data, ctypes.byref(n))
        if rc == E_OVERFLOW:
            cap = int(n.value)
            out = np.empty(cap, dtype=HIT_DTYPE)
            rc = self._lib.mtm_comm_last_gather(self._h, out.ctypes.data, cap, counts.ctypes.data, ctypes.byref(n))
        check(rc, "mtm_comm_allgather_hits")
        return out[:n.value], counts


class Group(_RecordMemo):
    """Several GPUs in one process (mtm_group): units sharded over the devices (LPT on their MAC cost), the image
    uploaded and searched on every device concurrently by native worker threads, hit lists merged on the host in
    template order.  Same ``search`` interface and results as a single Context."""

    def __init__(self, devices):
        lib = load()
        devices = [int(d) for d in devices]
        if not devices:
            raise MtmError("Group needs at least one device")
        arr = (ctypes.c_int * len(devices))(*devices)
        h = ctypes.c_void_p()
        check(lib.mtm_group_create(ctypes.byref(h), arr, len(devices)), "mtm_group_create")
        self._lib, self._h, self.devices = lib, h, devices
        self.lock = threading.RLock()
        for env, opt, table in (("MTM_KERNEL", OPT_KERNEL, {"auto": 0, "naive": 1, "dot4": 2, "mfma": 3}),
                                ("MTM_PEAK_BORDER", OPT_PEAK_BORDER, {"constant": 0, "nearest": 1})):
            v = os.environ.get(env)
            if v:
                self.set_option(opt, table[v.lower()])

    def close(self):
        if self._h:
            self._lib.mtm_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    def __len__(self):
        return len(self.devices)

    def set_option(self, opt, value):
        check(self._lib.mtm_group_set_option(self._h, int(opt), int(value)), "mtm_group_set_option")

    # ---- hit exchange: host merge (default) or the in-process RCCL all-gather ------
    def comm_init(self, strict=True):
        """ncclCommInitAll over the group's devices and the RCCL exchange for the searches that follow.  Returns the
        number of ranks of the communicator (ncclCommCount); with strict=False a failure (RCCL missing, a device listed
        twice) leaves the host merge in place and returns 0 instead of raising."""
        rc = self._lib.mtm_group_comm_init(self._h)
        if rc != 0:
            if strict:
                check(rc, "mtm_group_comm_init")
            return 0
        return self.comm_ranks()

    def comm_ranks(self):
        return int(self._lib.mtm_group_comm_ranks(self._h))

    def set_exchange(self, kind):
        """"host" | "rccl" """
        check(self._lib.mtm_group_set_exchange(self._h, {"host": GROUP_EXCHANGE_HOST, "rccl": GROUP_EXCHANGE_RCCL}[kind]),
              "mtm_group_set_exchange")

    def exchange_used(self):
        return {GROUP_EXCHANGE_HOST: "host", GROUP_EXCHANGE_RCCL: "rccl"}[int(self._lib.mtm_group_exchange_used(self._h))]

    def shards(self, templates, image_shape, method):
        """device index of every unit, as a search over an image of this shape would assign them"""
        rec, keep = templ_records(templates)
        dev = np.zeros(max(len(templates), 1), dtype=np.int32)
        check(self._lib.mtm_group_shards(self._h, rec.ctypes.data, len(templates), int(method), int(image_shape[0]),
                                         int(image_shape[1]), dev.ctypes.data), "mtm_group_shards")
        return dev[:len(templates)]

    def timing(self, i):
        t = MtmTiming()
        check(self._lib.mtm_get_timing(self._lib.mtm_group_ctx(self._h, int(i)), ctypes.byref(t)), "mtm_get_timing")
        return {f: getattr(t, f) for f, _ in MtmTiming._fields_}

    def search(self, templates, image, method, mode, score_threshold):
        rec = self._records(templates)
        a, ptr, stride = _pixel_rows(image)
        chans = 1 if a.ndim == 2 else a.shape[2]
        cap = 4096
        out = np.empty(cap, dtype=HIT_DTYPE)
        n = ctypes.c_int64(0)
        rc = self._lib.mtm_group_find_matches(self._h, rec.ctypes.data, len(templates), int(method), ptr, a.shape[0],
                                              a.shape[1], chans, _dtype_code(a), stride, int(mode), float(score_threshold),
                                              out.ctypes.data, cap, ctypes.byref(n))
        if rc == E_OVERFLOW:
            cap = int(n.value)
            out = np.empty(cap, dtype=HIT_DTYPE)
            rc = self._lib.mtm_group_last_hits(self._h, out.ctypes.data, cap, ctypes.byref(n))
        check(rc, "mtm_group_find_matches")
        return out[:n.value]


    def search_nms(self, templates, image, method, score_threshold, max_overlap, n_object=-1):
        """search() + MTM's non-maxima suppression on the merged list in one native call (mtm_group_find_matches_nms)."""
        rec = self._records(templates)
        a, ptr, stride = _pixel_rows(image)
        chans = 1 if a.ndim == 2 else a.shape[2]
        cap = 4096
        out = np.empty(cap, dtype=HIT_DTYPE)
        n = ctypes.c_int64(0)
        rc = self._lib.mtm_group_find_matches_nms(self._h, rec.ctypes.data, len(templates), int(method), ptr, a.shape[0],
                                                  a.shape[1], chans, _dtype_code(a), stride, float(score_threshold),
                                                  float(max_overlap), int(n_object), out.ctypes.data, cap, ctypes.byref(n))
        if rc == E_OVERFLOW:
            cap = int(n.value)
            out = np.empty(cap, dtype=HIT_DTYPE)
            rc = self._lib.mtm_group_last_hits(self._h, out.ctypes.data, cap, ctypes.byref(n))
        check(rc, "mtm_group_find_matches_nms")
        return out[:n.value]


def parse_devices(spec):
    """"all" | "0,1,2" | iterable of ints | int -> list of device ids (validated against the visible devices)."""
    n = load().mtm_device_count()
    if isinstance(spec, str):
        spec = spec.strip().lower()
        ids = list(range(n)) if spec == "all" else [int(x) for x in spec.split(",") if x.strip() != ""]
    elif isinstance(spec, int):
        ids = [spec]
    else:
        ids = [int(x) for x in spec]
    if not ids:
        raise MtmError("no HIP device visible (libmtm_hip has no CPU fallback)")
    for d in ids:
        if d < 0 or d >= n:
            raise MtmError("device %d is not visible (%d device(s))" % (d, n))
    return ids


_engines = {}


def engine_for(devices=None):
    """The search engine of a matchTemplates / findMatches call: the default single-GPU context, or a device
    group when several devices are asked for (argument, or the MTM_DEVICES environment variable: "all" or a
    comma-separated list).  Engines are created once per device list and reused."""
    if devices is None:
        devices = os.environ.get("MTM_DEVICES")
    if devices is None or devices == "":
        return default_context()
    ids = tuple(parse_devices(devices))
    if len(ids) == 1 and "MTM_DEVICES_FORCE_GROUP" not in os.environ:
        key = ("ctx", ids[0])
        with _default_lock:
            if key not in _engines:
                _engines[key] = Context(ids[0])
            return _engines[key]
    with _default_lock:
        if ids not in _engines:
            _engines[ids] = Group(ids)
        return _engines[ids]


def comm_unique_id():
    buf = (ctypes.c_char * COMM_ID_BYTES)()
    check(load().mtm_comm_unique_id(buf), "mtm_comm_unique_id")
    return bytes(buf)


def nms_indices(boxes, scores, score_threshold, max_overlap, ascending=False, n_object=-1):
    """cv2.dnn.NMSBoxes through the C ABI (host code, works without a GPU)."""
    n = len(boxes)
    hits = np.zeros(n, dtype=HIT_DTYPE)
    if n:
        b = np.asarray(boxes, dtype=np.int64).reshape(n, 4)
        hits["x"], hits["y"], hits["w"], hits["h"] = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
        hits["score"] = np.asarray(scores, dtype=np.float32)
    keep = np.empty(max(n, 1), dtype=np.int32)
    m = ctypes.c_int64(0)
    check(load().mtm_nms(hits.ctypes.data, n, float(score_threshold), int(bool(ascending)), int(n_object),
                         float(max_overlap), keep.ctypes.data, ctypes.byref(m)), "mtm_nms")
    return keep[:m.value]


def nms_hits(hits, score_threshold, max_overlap, ascending=False):
    """Same as nms_indices, on a structured hit array (boxes and float32 scores are taken from it)."""
    n = len(hits)
    if hits.dtype != HIT_DTYPE or not hits.flags.c_contiguous:
        hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
    keep = np.empty(max(n, 1), dtype=np.int32)
    m = ctypes.c_int64(0)
    check(load().mtm_nms(hits.ctypes.data, n, float(score_threshold), int(bool(ascending)), -1, float(max_overlap),
                         keep.ctypes.data, ctypes.byref(m)), "mtm_nms")
    return keep[:m.value]


_default_ctx = None
_default_lock = threading.Lock()


def default_context():
    global _default_ctx
    with _default_lock:
        if _default_ctx is None:
            _default_ctx = Context()
        return _default_ctx
