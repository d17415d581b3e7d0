"""markov-huffman-coding_amd — MI355X-native Markov-Huffman codec (ctypes face of libmhc.so).

The product is the C-ABI library (include/mh.h) built from csrc/ — hand-written HIP kernels for
gfx950 plus the C++ host model.  This module only binds it for tests and bench.py; it contains no
codec logic and never falls back to a CPU implementation: if libmhc.so is missing it raises, and
every compute call needs a GPU (MH_ERR_NO_DEVICE otherwise).

The directory name carries a hyphen, so import it through `__graft_entry__.load_package()`
(which registers it as module `mhc_amd`).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MH_LIB: an experimental build of the same library (csrc/Makefile `make exp TAG=x EXPFLAGS=-D...`), for A/B runs
LIB_PATH = os.environ.get("MH_LIB") or os.path.join(_HERE, "libmhc.so")

MH_OK = 0
MH_ERR_ARG, MH_ERR_NO_DEVICE, MH_ERR_HIP, MH_ERR_CORRUPT, MH_ERR_TYPE = -1, -2, -3, -4, -5
MH_ERR_BADTABLE, MH_ERR_CODE_TOO_LONG, MH_ERR_CAPACITY, MH_ERR_TIMEOUT, MH_ERR_NOMEM = -6, -7, -8, -9, -10
PREV0 = 0x20
CHUNK_DEFAULT = 1024
INDEX_BIT_MASK = 0x00FFFFFFFFFFFFFF
INDEX2_BIT_MASK = 0x0000FFFFFFFFFFFF      # order-2 models: two context bytes in bits 48..63

# every symbol include/mh.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "mh_strerror", "mh_last_hip_error", "mh_last_index_path", "mh_last_encode_retries", "mh_total_encode_retries", "mh_device_count", "mh_set_device",
    "mh_dev_malloc", "mh_dev_free", "mh_dev_upload", "mh_dev_download",
    "mh_model_from_counts", "mh_dev_model_from_counts", "mh_dev_model_workspace", "mh_dev_model_from_counts_ws",
    "mh_model_from_table_bits", "mh_model_write_table",
    "mh_model_type", "mh_model_max_code_len", "mh_model_min_code_len", "mh_model_get_code", "mh_model_get_lut", "mh_model_decode_layout", "mh_model_tile_layout",
    "mh_model_image", "mh_model_free",
    "mh_set_input_residency", "mh_histogram_o1", "mh_histogram_o0", "mh_histogram_o2", "mh_dev_histogram_o2", "mh_dev_histogram_o2_ws", "mh_dev_histogram_o2_workspace", "mh_encode", "mh_encode_bound", "mh_stream_header",
    "mh_stream_parse_header", "mh_decode",
    "mh_dev_histogram_workspace", "mh_dev_histogram_o1", "mh_dev_histogram_o0",
    "mh_decode_to", "mh_model_payload_bits", "mh_dev_encode_workspace", "mh_dev_encode", "mh_dev_payload_bits", "mh_dev_encode_at", "mh_dev_encode_ctx", "mh_dev_encode_hist", "mh_dev_decode_workspace", "mh_dev_decode", "mh_dev_decode_dn",
    "mh_dev_build_index_workspace", "mh_dev_build_index", "mh_dev_status",
    "mh_dev_model2_workspace", "mh_dev_model2_array", "mh_dev_model2_build_slice", "mh_dev_model2_finish",
    "mh_dev_encode_fine", "mh_dev_encode_ctx_fine", "mh_dev_decode_fine", "mh_dev_build_index_fine", "mh_dev_decode_stream_states", "mh_dev_decode_stream_emit", "mh_dev_index_path", "mh_dev_encode_path", "mh_dev_decode_path", "mh_dev_decode_variant",
]


class MhError(RuntimeError):
    def __init__(self, status, what=""):
        self.status = status
        msg = lib().mh_strerror(status).decode() if _lib is not None else str(status)
        super().__init__("%s: %s (%d)" % (what, msg, status))


_lib = None


def lib():
    """Loads libmhc.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libmhc.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "or `make -C markov-huffman-coding_amd/csrc`")
        l = C.CDLL(LIB_PATH)
        vp, sz, u8, u32, u64, i32 = C.c_void_p, C.c_size_t, C.c_uint8, C.c_uint32, C.c_uint64, C.c_int
        pi, pu64, psz = C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.POINTER(C.c_size_t)
        l.mh_strerror.restype = C.c_char_p
        l.mh_strerror.argtypes = [i32]
        l.mh_total_encode_retries.restype = u64
        l.mh_dev_malloc.argtypes = [C.POINTER(vp), sz]
        l.mh_dev_free.argtypes = [vp]
        l.mh_dev_upload.argtypes = [vp, vp, sz]
        l.mh_dev_download.argtypes = [vp, vp, sz]
        l.mh_model_from_counts.argtypes = [vp, i32, C.POINTER(vp)]
        l.mh_dev_model_from_counts.argtypes = [vp, i32, vp, C.POINTER(vp)]
        l.mh_dev_model_workspace.argtypes = [i32]
        l.mh_dev_model_workspace.restype = sz
        l.mh_dev_model_from_counts_ws.argtypes = [vp, i32, vp, sz, vp, C.POINTER(vp)]
        l.mh_model_from_table_bits.argtypes = [vp, sz, C.POINTER(vp)]
        l.mh_model_write_table.argtypes = [vp, vp, sz, psz]
        l.mh_model_type.argtypes = [vp]
        l.mh_model_max_code_len.argtypes = [vp]
        l.mh_model_min_code_len.argtypes = [vp]
        l.mh_model_get_code.argtypes = [vp, i32, i32, pi, pu64]
        l.mh_model_get_lut.argtypes = [vp, i32, i32, pi, pi, pi, pi]
        l.mh_model_decode_layout.argtypes = [vp, pi, pi, pi]
        l.mh_model_tile_layout.argtypes = [vp, pi, pi, pi]
        l.mh_model_image.argtypes = [vp, i32, vp, sz, psz]
        l.mh_model_free.argtypes = [vp]
        l.mh_model_free.restype = None
        l.mh_histogram_o1.argtypes = [vp, sz, u8, vp]
        l.mh_histogram_o0.argtypes = [vp, sz, vp]
        l.mh_histogram_o2.argtypes = [vp, sz, vp]
        l.mh_dev_histogram_o2.argtypes = [vp, sz, C.c_uint16, vp, vp]
        l.mh_dev_histogram_o2_ws.argtypes = [vp, sz, C.c_uint16, vp, vp, sz, vp]
        l.mh_dev_histogram_o2_workspace.argtypes = [sz]
        l.mh_dev_histogram_o2_workspace.restype = sz
        l.mh_encode.argtypes = [vp, vp, sz, u8, vp, sz, pu64, vp, u32]
        l.mh_encode_bound.argtypes = [vp, sz]
        l.mh_encode_bound.restype = sz
        l.mh_stream_header.argtypes = [vp, u64]
        l.mh_stream_header.restype = u8
        l.mh_stream_parse_header.argtypes = [vp, u8, u64, pu64]
        l.mh_decode.argtypes = [vp, vp, u64, u8, vp, sz, psz, vp, u32, u64]
        l.mh_dev_histogram_workspace.argtypes = [sz]
        l.mh_dev_histogram_workspace.restype = sz
        l.mh_dev_histogram_o1.argtypes = [vp, sz, u8, vp, vp, sz, vp]
        l.mh_dev_histogram_o0.argtypes = [vp, sz, vp, vp, sz, vp]
        l.mh_dev_encode_workspace.argtypes = [sz]
        l.mh_dev_encode_workspace.restype = sz
        l.mh_dev_encode.argtypes = [vp, vp, sz, u8, vp, sz, vp, vp, u32, vp, sz, vp]
        l.mh_dev_payload_bits.argtypes = [vp, vp, vp, vp]
        l.mh_dev_encode_at.argtypes = [vp, vp, sz, u8, vp, vp, sz, vp, vp, u32, vp, sz, vp]
        l.mh_dev_encode_ctx.argtypes = [vp, vp, sz, u32, vp, vp, sz, vp, vp, u32, vp, sz, vp]
        l.mh_dev_encode_hist.argtypes = [vp, vp, sz, u8, vp, vp, sz, vp, vp, u32, vp, sz, vp, sz, vp]
        l.mh_dev_decode_workspace.argtypes = [u64, u64, u32]
        l.mh_dev_decode_workspace.restype = sz
        l.mh_dev_decode.argtypes = [vp, vp, u64, vp, u64, vp, u32, vp, sz, vp]
        l.mh_dev_decode_dn.argtypes = [vp, vp, vp, u64, vp, u64, vp, u32, vp, sz, vp]
        l.mh_dev_build_index_workspace.argtypes = [u64]
        l.mh_dev_build_index_workspace.restype = sz
        l.mh_dev_build_index.argtypes = [vp, vp, u64, u8, vp, u64, u32, vp, vp, sz, vp]
        l.mh_dev_status.argtypes = [vp, vp]
        l.mh_dev_encode_fine.argtypes = [vp, vp, sz, u8, vp, vp, sz, vp, vp, u32, vp, vp, sz, vp, sz, vp]
        l.mh_dev_model2_workspace.argtypes = []
        l.mh_dev_model2_workspace.restype = sz
        l.mh_dev_model2_array.argtypes = [i32, psz, psz]
        l.mh_dev_model2_build_slice.argtypes = [vp, u32, u32, vp, sz, vp]
        l.mh_dev_model2_finish.argtypes = [vp, sz, vp, C.POINTER(vp)]
        l.mh_dev_encode_ctx_fine.argtypes = [vp, vp, sz, u32, vp, vp, sz, vp, vp, u32, vp, vp, sz, vp]
        l.mh_dev_decode_fine.argtypes = [vp, vp, u64, vp, vp, u64, vp, u32, vp, vp, sz, vp]
        l.mh_dev_build_index_fine.argtypes = [vp, vp, u64, u8, vp, u64, u32, vp, u64, vp, vp, sz, vp]
        l.mh_dev_decode_stream_states.argtypes = [vp, vp, u64, u8, vp, vp, sz, vp]
        l.mh_dev_decode_stream_emit.argtypes = [vp, vp, u64, u8, vp, u64, vp, sz, vp]
        l.mh_dev_index_path.argtypes = [vp, vp]
        l.mh_dev_decode_variant.argtypes = [vp, vp]
        l.mh_dev_encode_path.argtypes = [vp, vp]
        l.mh_dev_decode_path.argtypes = [vp, vp]
        _lib = l
    return _lib


def _check(status, what):
    if status != MH_OK:
        raise MhError(status, what)


def device_count():
    return lib().mh_device_count()


def _u8(data):
    if isinstance(data, (bytes, bytearray, memoryview)):
        return np.frombuffer(data, dtype=np.uint8)
    return np.ascontiguousarray(data, dtype=np.uint8)


def _ptr(a):
    return a.ctypes.data if a.size else None


class DeviceBuffer:
    """A hipMalloc'ed buffer (for tests that drive the mh_dev_* calls without torch)."""

    def __init__(self, nbytes, init=None):
        self.nbytes = nbytes
        self.ptr = C.c_void_p()
        _check(lib().mh_dev_malloc(C.byref(self.ptr), nbytes), "mh_dev_malloc")
        if init is not None:
            a = np.ascontiguousarray(init)
            _check(lib().mh_dev_upload(self.ptr, a.ctypes.data, a.nbytes), "mh_dev_upload")

    def download(self, dtype=np.uint8):
        out = np.zeros(self.nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        _check(lib().mh_dev_download(out.ctypes.data, self.ptr, out.nbytes), "mh_dev_download")
        return out

    def __del__(self):
        if getattr(self, "ptr", None) and _lib is not None:
            _lib.mh_dev_free(self.ptr)
            self.ptr = None


def histogram_o1(data, prev0=PREV0):
    a = _u8(data)
    out = np.zeros(65536, dtype=np.uint64)
    _check(lib().mh_histogram_o1(_ptr(a), a.size, prev0, out.ctypes.data), "mh_histogram_o1")
    return out


def histogram_o0(data):
    a = _u8(data)
    out = np.zeros(256, dtype=np.uint64)
    _check(lib().mh_histogram_o0(_ptr(a), a.size, out.ctypes.data), "mh_histogram_o0")
    return out


def histogram_o2(data):
    """Order-2 extension (parity unpinned): counts[ctx * 256 + sym], ctx = the two previous bytes."""
    a = _u8(data)
    out = np.zeros(1 << 24, dtype=np.uint64)
    _check(lib().mh_histogram_o2(_ptr(a), a.size, out.ctypes.data), "mh_histogram_o2")
    return out


class Model:
    """Owns an mh_model* (tables resident on the current device)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def from_counts(cls, counts, order):
        c = np.ascontiguousarray(counts, dtype=np.uint64)
        if c.size != {0: 256, 1: 65536, 2: 1 << 24}[order]:
            raise ValueError("counts size")
        h = C.c_void_p()
        _check(lib().mh_model_from_counts(c.ctypes.data, order, C.byref(h)), "mh_model_from_counts")
        return cls(h)

    @classmethod
    def from_device_counts(cls, d_counts_ptr, order, stream=None):
        h = C.c_void_p()
        _check(lib().mh_dev_model_from_counts(d_counts_ptr, order, stream, C.byref(h)), "mh_dev_model_from_counts")
        return cls(h)

    @classmethod
    def from_device_counts_ws(cls, d_counts_ptr, order, d_ws_ptr, ws_bytes, stream=None):
        """Model built into a caller workspace (no allocation inside, one stream sync); the caller keeps
        the workspace alive for as long as the model is used."""
        h = C.c_void_p()
        _check(lib().mh_dev_model_from_counts_ws(d_counts_ptr, order, d_ws_ptr, ws_bytes, stream, C.byref(h)),
               "mh_dev_model_from_counts_ws")
        return cls(h)

    @classmethod
    def from_data(cls, data, order=1):
        """Histogram on the GPU, then tree build (the `markovhuffman in -d table` path)."""
        return cls.from_counts({0: histogram_o0, 1: histogram_o1, 2: histogram_o2}[order](data), order)

    @classmethod
    def from_table(cls, table_bytes):
        a = _u8(table_bytes)
        h = C.c_void_p()
        _check(lib().mh_model_from_table_bits(_ptr(a), a.size, C.byref(h)), "mh_model_from_table_bits")
        return cls(h)

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.mh_model_free(self._h)
            self._h = None

    @property
    def handle(self):
        return self._h

    @property
    def type(self):
        return lib().mh_model_type(self._h)

    @property
    def max_code_len(self):
        return lib().mh_model_max_code_len(self._h)

    @property
    def min_code_len(self):
        return lib().mh_model_min_code_len(self._h)

    def decode_layout(self):
        """(primary_bits, secondary_entries, in_lds) of the device decode tables."""
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        _check(lib().mh_model_decode_layout(self._h, C.byref(a), C.byref(b), C.byref(c)), "mh_model_decode_layout")
        return a.value, b.value, bool(c.value)

    def tile_layout(self):
        """(primary_bits, secondary_bits, secondary_entries) of the tile decoder's tables; primary_bits 0 = none."""
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        _check(lib().mh_model_tile_layout(self._h, C.byref(a), C.byref(b), C.byref(c)), "mh_model_tile_layout")
        return a.value, b.value, c.value

    def image(self, which):
        """Device image `which` (see mh.h: 0 enc16 ... 7 walk tree) as bytes."""
        n = C.c_size_t(0)
        _check(lib().mh_model_image(self._h, which, None, 0, C.byref(n)), "mh_model_image")
        out = np.zeros(max(n.value, 1), dtype=np.uint8)
        _check(lib().mh_model_image(self._h, which, out.ctypes.data, n.value, C.byref(n)), "mh_model_image")
        return out[:n.value].tobytes()

    def table_bytes(self):
        n = C.c_size_t(0)
        _check(lib().mh_model_write_table(self._h, None, 0, C.byref(n)), "mh_model_write_table")
        out = np.zeros(max(n.value, 1), dtype=np.uint8)
        _check(lib().mh_model_write_table(self._h, out.ctypes.data, n.value, C.byref(n)), "mh_model_write_table")
        return out[:n.value].tobytes()

    def codes(self):
        """(len8[65536], code64[65536]) indexed prev*256+sym (get_encoding for every pair)."""
        lens = np.zeros(65536, dtype=np.uint8)
        codes = np.zeros(65536, dtype=np.uint64)
        l, c = C.c_int(), C.c_uint64()
        for p in range(256):
            for s in range(256):
                lib().mh_model_get_code(self._h, p, s, C.byref(l), C.byref(c))
                lens[p * 256 + s] = l.value
                codes[p * 256 + s] = c.value
        return lens, codes

    def codes_o2(self):
        """Order-2 model: (len8[1 << 24], code64[1 << 24]) indexed ctx*256+sym, straight from the device tables."""
        return (np.frombuffer(self.image(1), dtype=np.uint8), np.frombuffer(self.image(3), dtype=np.uint64))

    def lut(self, prev, w):
        p, i, v, d = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        lib().mh_model_get_lut(self._h, prev, w, C.byref(p), C.byref(i), C.byref(v), C.byref(d))
        return bool(p.value), bool(i.value), v.value, d.value

    # ---- host-buffer codec calls -------------------------------------------------------------
    def encode(self, data, prev0=PREV0, chunk_symbols=None):
        """Returns (payload bytes, nbits, index or None)."""
        a = _u8(data)
        cap = lib().mh_encode_bound(self._h, a.size)
        out = np.zeros(cap, dtype=np.uint8)
        nbits = C.c_uint64(0)
        idx = None
        if chunk_symbols:
            idx = np.zeros(max((a.size + chunk_symbols - 1) // chunk_symbols, 1), dtype=np.uint64)
        _check(lib().mh_encode(self._h, _ptr(a), a.size, prev0, out.ctypes.data, cap, C.byref(nbits),
                               idx.ctypes.data if idx is not None else None, chunk_symbols or 0), "mh_encode")
        nb = (nbits.value + 7) // 8
        if idx is not None:
            idx = idx[:(a.size + chunk_symbols - 1) // chunk_symbols]
        return out[:nb].tobytes(), nbits.value, idx

    def compress(self, data, chunk_symbols=None):
        """Whole compressed file (header byte + payload), as i_coding_provider::compress writes it."""
        payload, nbits, idx = self.encode(data, PREV0, chunk_symbols)
        return bytes([lib().mh_stream_header(self._h, nbits)]) + payload, nbits, idx

    def decode(self, payload, nbits, prev0=PREV0, index=None, chunk_symbols=0, n_symbols=0, cap=None):
        a = _u8(payload)
        if cap is None:
            cap = n_symbols if index is not None else nbits
        out = np.zeros(max(cap, 1), dtype=np.uint8)
        n = C.c_size_t(0)
        ip = None
        if index is not None:
            index = np.ascontiguousarray(index, dtype=np.uint64)
            ip = index.ctypes.data if index.size else None
            if ip is None and n_symbols == 0:
                ip = out.ctypes.data  # any non-null pointer: zero entries are read
        _check(lib().mh_decode(self._h, _ptr(a), nbits, prev0, out.ctypes.data, cap, C.byref(n), ip,
                               chunk_symbols, n_symbols), "mh_decode")
        return out[:n.value].tobytes()

    def decompress(self, blob, index=None, chunk_symbols=0, n_symbols=0):
        """Whole compressed file in, original bytes out (i_coding_provider::decompress)."""
        a = _u8(blob)
        nbits = C.c_uint64(0)
        if a.size < 1:
            raise MhError(MH_ERR_CORRUPT, "decompress")
        _check(lib().mh_stream_parse_header(self._h, int(a[0]), a.size, C.byref(nbits)), "mh_stream_parse_header")
        return self.decode(a[1:], nbits.value, PREV0, index, chunk_symbols, n_symbols)
