"""ctypes binding of libdexgpu.so (the C-ABI of include/dexgpu.h).

The HIP library is the product: there is no CPU fallback.  Loading fails loudly when the shared
object has not been built (`make lib` or `python -c 'import __graft_entry__ as g; g.build()'`).
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DEXGPU_LIB") or os.path.join(HERE, "libdexgpu.so")   # DEXGPU_LIB: A/B builds

DX_OK = 0
ERR_NAMES = {-1: "DX_E_ARG", -2: "DX_E_HIP", -3: "DX_E_FORMAT", -4: "DX_E_DEGENERATE",
             -5: "DX_E_UNSUPPORTED", -6: "DX_E_NOMEM", -7: "DX_E_MISMATCH", -8: "DX_E_SPACE", -9: "DX_E_IO"}

DX_ALPHA_BASES, DX_ALPHA_ARROW = 0, 1
DX_LETTERS_LOWER, DX_LETTERS_UPPER, DX_LETTERS_ARROW = 0, 1, 2
DX_DEL, DX_INS, DX_MRG, DX_SUB, DX_DRUN, DX_SRUN = range(6)
KERNELS = ["k_pack2_encode", "k_pack2_decode", "k_qv_prescan", "k_qv_hist", "k_qv_sizes", "k_scan",
           "k_qv_encode", "k_qv_decode", "k_synth", "k_index", "k_qv_compact", "k_qv_encode_text",
           "k_qv_decode_sub", "k_qv_decode_runs", "k_qv_decode_plain", "k_qv_decode_tags", "k_qv_walk"]
DECODE_KERNELS = ["k_qv_decode", "k_qv_decode_sub", "k_qv_decode_runs", "k_qv_decode_plain", "k_qv_decode_tags"]


class QVBatch(C.Structure):
    _fields_ = [("d_text", C.c_void_p), ("d_off", C.c_void_p), ("d_len", C.c_void_p),
                ("n", C.c_uint64), ("text_bytes", C.c_uint64), ("line_pad", C.c_uint32)]


class QVParams(C.Structure):
    _fields_ = [("delChar", C.c_int32), ("subChar", C.c_int32),
                ("del_first", C.c_int64), ("sub_first", C.c_int64)]


class Scheme(C.Structure):
    _fields_ = [("type", C.c_int32), ("bits", C.c_uint32 * 256), ("lens", C.c_int32 * 256)]


class QVCoding(C.Structure):
    _fields_ = [("s", Scheme * 6), ("delChar", C.c_int32), ("subChar", C.c_int32)]


HIST = (C.c_uint64 * 256) * 6


class OnepassInfo(C.Structure):
    _fields_ = [("groups", C.c_int32), ("direct", C.c_int32), ("tokens", C.c_int32), ("reserved", C.c_int32),
                ("region_bytes", C.c_uint64), ("scratch_bytes", C.c_uint64), ("avail_bytes", C.c_uint64),
                ("token_bytes", C.c_uint64), ("text_entries", C.c_uint64),
                ("chain_waits", C.c_uint64 * 3)]


class QVIndex(C.Structure):
    _fields_ = [("n", C.c_uint64), ("rec_off", C.POINTER(C.c_uint64)), ("hdr_off", C.POINTER(C.c_uint64)),
                ("seg", C.POINTER(C.c_uint32)), ("len", C.POINTER(C.c_uint32)), ("hdr4", C.POINTER(C.c_int32)),
                ("coding", QVCoding), ("prefix", C.c_char_p), ("newv", C.c_int), ("flip", C.c_int),
                ("gidx", C.POINTER(C.c_uint32)), ("gidx_off", C.POINTER(C.c_uint64)), ("gidx_words", C.c_uint64),
                ("gidx_none", C.c_uint64)]

class QVDIndex(C.Structure):                     # dx_qv_dindex
    _fields_ = [("n", C.c_uint64), ("d_rec_off", C.c_void_p), ("d_hdr_off", C.c_void_p), ("d_seg", C.c_void_p),
                ("d_len", C.c_void_p), ("d_hdr4", C.c_void_p), ("pieces", C.c_uint64), ("piece_bytes", C.c_uint64),
                ("d_gidx", C.c_void_p), ("d_gidx_off", C.c_void_p), ("gidx_words", C.c_uint64), ("gidx_none", C.c_uint64),
                ("gidx_nosync", C.c_uint64), ("sync_kinds", C.c_uint32)]

# name -> (restype, argtypes); every symbol include/dexgpu.h declares
_P = C.c_void_p
SINK_FN = C.CFUNCTYPE(C.c_int, _P, C.POINTER(C.c_uint8), C.c_size_t, C.c_size_t)     # dx_sink_fn
READ_FN = C.CFUNCTYPE(C.c_long, _P, _P, C.c_size_t)                                   # dx_read_fn
SIGNATURES = {
    "dx_device_count": (C.c_int, []),
    "dx_open": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "dx_close": (None, [_P]),
    "dx_last_error": (C.c_char_p, [_P]),
    "dx_set_stream": (C.c_int, [_P, _P]),
    "dx_reset_stream": (C.c_int, [_P]),
    "dx_sync": (C.c_int, [_P]),
    "dx_malloc": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "dx_free": (C.c_int, [_P, _P]),
    "dx_h2d": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "dx_d2h": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "dx_memset": (C.c_int, [_P, _P, C.c_int, C.c_size_t]),
    "dx_profile": (C.c_int, [_P, C.c_int]),
    "dx_profile_get": (C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "dx_kernel_name": (C.c_char_p, [C.c_int]),
    "dx_pack2_encode": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, C.c_uint64, _P, _P, _P, _P]),
    "dx_pack2_decode": (C.c_int, [_P, C.c_int, _P, _P, _P, C.c_uint64, C.c_uint32, _P, _P]),
    "dx_frame_bound": (C.c_size_t, [_P, C.c_uint64, C.c_int32, C.c_int]),
    "dx_frame_headers": (C.c_int, [_P, _P, C.c_uint64, C.c_int, C.POINTER(C.c_int32), _P, _P]),
    "dx_snr_to_cnr": (C.c_uint16, [C.c_float]),
    "dx_index_quiva": (C.c_int, [_P, C.c_size_t, C.c_uint64, _P, _P, _P, C.POINTER(C.c_uint64),
                                 C.POINTER(C.c_size_t), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    "dx_index_seq": (C.c_int, [C.c_int, _P, C.c_size_t, C.c_uint64, _P, _P, _P, _P, _P,
                               C.POINTER(C.c_uint64), C.POINTER(C.c_size_t), C.POINTER(C.c_uint64),
                               C.POINTER(C.c_int)]),
    "dx_index_quiva_device": (C.c_int, [_P, _P, C.c_uint64, C.POINTER(_P), C.POINTER(_P), C.POINTER(C.c_uint64),
                                        C.POINTER(_P), C.POINTER(C.c_size_t), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    "dx_parse_quiva_headers": (C.c_int, [_P, _P, C.c_uint64, _P, C.POINTER(C.c_size_t), C.POINTER(C.c_uint64)]),
    "dx_index_seq_device": (C.c_int, [_P, C.c_int, _P, C.c_uint64, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P),
                                      C.POINTER(C.c_uint64), C.POINTER(_P), C.POINTER(_P), C.POINTER(C.c_size_t),
                                      C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    "dx_parse_seq_headers": (C.c_int, [C.c_int, _P, _P, C.c_uint64, _P, _P, C.POINTER(C.c_size_t), C.POINTER(C.c_uint64)]),
    "dx_qv_prescan": (C.c_int, [_P, C.POINTER(QVBatch), C.c_uint64, C.POINTER(QVParams)]),
    "dx_qv_lossy_text": (C.c_int, [_P, C.POINTER(QVBatch)]),
    "dx_mem_info": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "dx_qv_hist": (C.c_int, [_P, C.POINTER(QVBatch), C.c_uint64, C.POINTER(QVParams), C.POINTER(HIST),
                             C.POINTER(C.c_uint64)]),
    "dx_qv_scan": (C.c_int, [_P, C.POINTER(QVBatch), C.c_uint64, C.POINTER(QVParams), C.POINTER(HIST),
                             C.POINTER(C.c_uint64)]),
    "dx_qv_build": (C.c_int, [C.POINTER(HIST), C.c_uint64, C.POINTER(QVParams), C.c_int, C.POINTER(QVCoding)]),
    "dx_qv_write_coding": (C.c_int, [C.POINTER(QVCoding), C.c_char_p, C.c_size_t, _P, C.c_size_t,
                                     C.POINTER(C.c_size_t)]),
    "dx_qv_read_coding": (C.c_int, [_P, C.c_size_t, C.POINTER(QVCoding), C.POINTER(C.c_int), _P, C.c_size_t,
                                    C.POINTER(C.c_size_t)]),
    "dx_qv_set_coding": (C.c_int, [_P, C.POINTER(QVCoding), C.c_int]),
    "dx_qv_sizes": (C.c_int, [_P, C.POINTER(QVBatch), _P, _P, _P, C.POINTER(C.c_uint64)]),
    "dx_qv_encode": (C.c_int, [_P, C.POINTER(QVBatch), _P, _P, _P, _P, _P]),
    "dx_qv_encode_onepass": (C.c_int, [_P, C.POINTER(QVBatch), _P, _P, _P, _P, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    "dx_qv_encode_onepass_begin": (C.c_int, [_P, C.POINTER(QVBatch), _P, _P, _P, _P, _P, C.c_uint64]),
    "dx_qv_encode_onepass_end": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "dx_qv_onepass_info": (C.c_int, [_P, C.POINTER(OnepassInfo)]),
    "dx_set_scratch_budget": (C.c_int, [_P, C.c_uint64]),
    "dx_trim": (C.c_int, [_P, C.c_int]),
    "dx_qv_out_bound": (C.c_uint64, [C.POINTER(HIST), C.c_uint64, C.POINTER(QVCoding), C.c_int]),
    "dx_qv_decode": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_uint64, C.c_int, _P, _P]),
    "dx_qv_walk": (C.c_int, [_P, C.c_size_t, _P]),
    "dx_qv_walk_indexed": (C.c_int, [_P, C.c_size_t, _P, C.c_int]),
    "dx_file_undexqv_plan_index": (C.c_int, [_P, _P]),
    "dx_file_undexqv_plan_on": (C.c_int, [_P, _P, C.c_size_t, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "dx_qv_walk_device": (C.c_int, [_P, _P, C.c_uint64, C.c_uint64, _P, C.c_int, C.c_int, _P]),
    "dx_qv_dindex_free": (None, [_P, _P]),
    "dx_qv_use_dindex": (C.c_int, [_P, _P, _P]),
    "dx_set_sink_threads": (C.c_int, [_P, C.c_int]),
    "dx_h2d_fd": (C.c_int, [_P, _P, C.c_int, C.c_uint64, C.c_size_t]),
    "dx_file_dexqv_fd_to": (C.c_int, [_P, C.c_int, C.c_size_t, C.c_int, SINK_FN, _P, C.POINTER(C.c_size_t), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    "dx_qv_use_index": (C.c_int, [_P, _P, _P, C.c_uint64, _P, _P, C.c_uint64]),
    "dx_qv_index_free": (None, [_P]),
    "dx_file_pack2": (C.c_int, [_P, C.c_int, _P, C.c_size_t, C.POINTER(_P), C.POINTER(C.c_size_t),
                                C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    "dx_file_unpack2": (C.c_int, [_P, C.c_int, _P, C.c_size_t, C.c_uint32, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "dx_file_dexqv": (C.c_int, [_P, _P, C.c_size_t, C.c_int, C.POINTER(_P), C.POINTER(C.c_size_t),
                                C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    "dx_file_dexqv_sharded": (C.c_int, [_P, C.c_int, _P, C.c_size_t, C.c_int, C.POINTER(_P), C.POINTER(C.c_size_t),
                                        C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    "dx_file_pack2_sharded": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_size_t, C.POINTER(_P), C.POINTER(C.c_size_t),
                                        C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    "dx_file_undexqv": (C.c_int, [_P, _P, C.c_size_t, C.c_int, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "dx_qv_subindex": (C.c_int, [_P, C.c_int]),
    "dx_file_pack2_stream": (C.c_int, [_P, C.c_int, READ_FN, _P, C.c_size_t, SINK_FN, _P, C.POINTER(C.c_size_t), C.POINTER(C.c_uint64),
                                       C.POINTER(C.c_int)]),
    "dx_file_unpack2_stream": (C.c_int, [_P, C.c_int, READ_FN, _P, C.c_size_t, C.c_uint32, SINK_FN, _P, C.POINTER(C.c_size_t)]),
    "dx_file_unpack2_to": (C.c_int, [_P, C.c_int, _P, C.c_size_t, C.c_uint32, SINK_FN, _P, C.POINTER(C.c_size_t)]),
    "dx_file_dexqv_to": (C.c_int, [_P, _P, C.c_size_t, C.c_int, SINK_FN, _P, C.POINTER(C.c_size_t),
                                   C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    "dx_file_undexqv_plan": (C.c_int, [_P, C.c_size_t, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "dx_file_undexqv_run": (C.c_int, [_P, _P, C.c_int, SINK_FN, _P]),
    "dx_file_undexqv_plan_free": (None, [_P]),
    "dx_d2h_stream": (C.c_int, [_P, _P, C.c_size_t, SINK_FN, _P]),
    "dx_file_free": (None, [_P]),
    "dx_entries_new": (_P, []),
    "dx_entries_free": (None, [_P]),
    "dx_entries_add": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P]),
    "dx_entries_compress": (C.c_int, [_P, _P, C.c_int, C.POINTER(QVCoding), C.POINTER(_P), C.POINTER(C.c_size_t),
                                      C.POINTER(_P)]),
    "dx_synth_quiva": (C.c_int, [_P, C.c_uint32, C.c_uint64, C.c_uint64, _P, _P, _P, _P, C.c_int,
                                 C.c_char_p, _P]),
}

_lib = None


class DexGPUError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


def load() -> C.CDLL:
    """Load libdexgpu.so and bind every C-ABI symbol; raises if the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP library is the product and there is no CPU fallback. "
            "Build it with `make lib` (hipcc --offload-arch=gfx950) or __graft_entry__.build().")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)            # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
