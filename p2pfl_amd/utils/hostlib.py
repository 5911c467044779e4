"""ctypes loader for ``libp2pfl_host.so`` (C++ host runtime: AES-NI link cipher, PARAMS framing,
message-id ring). Built on demand with g++ — it contains no GPU code, so it also runs in the
CPU-only test container."""
from __future__ import annotations

import ctypes
from ctypes import POINTER, c_char_p, c_int, c_size_t, c_uint64, c_void_p

_lib = None


def load():
    global _lib
    if _lib is None:
        from ..ops.build import build_host
        path = build_host()
        lib = ctypes.CDLL(str(path))
        lib.p2pfl_aes128_expand_key.argtypes = [c_char_p, c_char_p]
        lib.p2pfl_aes128_expand_key.restype = None
        lib.p2pfl_aes128_ecb_host.argtypes = [c_char_p, c_char_p, c_char_p, c_size_t, c_int]
        lib.p2pfl_aes128_ecb_host.restype = c_int
        lib.p2pfl_params_frame_count.argtypes = [c_size_t, c_size_t]
        lib.p2pfl_params_frame_count.restype = c_size_t
        lib.p2pfl_params_build.argtypes = [c_char_p, c_size_t, c_size_t, c_char_p, POINTER(c_size_t)]
        lib.p2pfl_params_build.restype = c_size_t
        lib.p2pfl_reasm_new.argtypes = [c_size_t]
        lib.p2pfl_reasm_new.restype = c_void_p
        lib.p2pfl_reasm_free.argtypes = [c_void_p]
        lib.p2pfl_reasm_free.restype = None
        lib.p2pfl_reasm_append.argtypes = [c_void_p, c_char_p, c_size_t]
        lib.p2pfl_reasm_append.restype = None
        lib.p2pfl_reasm_size.argtypes = [c_void_p]
        lib.p2pfl_reasm_size.restype = c_size_t
        lib.p2pfl_reasm_data.argtypes = [c_void_p]
        lib.p2pfl_reasm_data.restype = c_void_p
        lib.p2pfl_reasm_clear.argtypes = [c_void_p]
        lib.p2pfl_reasm_clear.restype = None
        lib.p2pfl_idring_new.argtypes = [c_size_t]
        lib.p2pfl_idring_new.restype = c_void_p
        lib.p2pfl_idring_free.argtypes = [c_void_p]
        lib.p2pfl_idring_free.restype = None
        lib.p2pfl_idring_check_and_add.argtypes = [c_void_p, c_uint64]
        lib.p2pfl_idring_check_and_add.restype = c_int
        lib.p2pfl_idring_contains.argtypes = [c_void_p, c_uint64]
        lib.p2pfl_idring_contains.restype = c_int
        lib.p2pfl_idring_size.argtypes = [c_void_p]
        lib.p2pfl_idring_size.restype = c_size_t
        _lib = lib
    return _lib
