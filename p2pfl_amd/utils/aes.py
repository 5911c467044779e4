"""AES-128 helpers shared by the host (AES-NI, C++) and device (csrc/aes.hip) link ciphers."""
from __future__ import annotations

import ctypes

from .hostlib import load


def expand_key(key16: bytes) -> bytes:
    """The 176-byte AES-128 key schedule (11 round keys) consumed by the GPU kernel."""
    if len(key16) != 16:
        raise ValueError("AES-128 needs a 16-byte key")
    out = ctypes.create_string_buffer(176)
    load().p2pfl_aes128_expand_key(key16, out)
    return out.raw


def ecb_host(key16: bytes, data: bytes, decrypt: bool) -> bytes:
    if len(key16) != 16:
        raise ValueError("AES-128 needs a 16-byte key")
    if len(data) % 16:
        raise ValueError("ECB input must be a multiple of 16 bytes")
    out = ctypes.create_string_buffer(len(data))
    rc = load().p2pfl_aes128_ecb_host(key16, bytes(data), out, len(data), int(decrypt))
    if rc != 0:
        raise ValueError("AES failure")
    return out.raw
