"""Link encryption (reference: fedstellar/encrypter.py:28-193).

* :class:`AESCipher` – AES-128, 16-byte random key, ECB, space padding to the block size: the
  reference's wire behaviour (encrypter.py:109-193), executed by the AES-NI host library
  (``csrc/host/aes_host.cpp``); the same key schedule drives the GPU kernel (``csrc/aes.hip``).
* :class:`RSACipher` – RSA-1024 with PKCS#1 OAEP (SHA-1, MGF1), base64(DER SubjectPublicKeyInfo)
  public-key exchange like the reference (encrypter.py:48-101). pycryptodome is not available on
  the target image, so keygen/OAEP are implemented here on Python integers.

The reference exchanges RSA keys but then sends the AES key in clear (base_node.py:256,354);
:meth:`RSACipher.encrypt` / :meth:`decrypt` are used by this build's handshake to wrap it.
"""
from __future__ import annotations

import base64
import hashlib
import os
import secrets

from .utils.aes import ecb_host, expand_key


class Encrypter:
    def encrypt(self, message: bytes) -> bytes:
        raise NotImplementedError

    def decrypt(self, message: bytes) -> bytes:
        raise NotImplementedError


# ---------------------------------------------------------------------------- AES
class AESCipher(Encrypter):
    BLOCK = 16

    def __init__(self, key: bytes = None):
        self.__key = os.urandom(16) if key is None else bytes(key)
        if len(self.__key) != 16:
            raise ValueError("AES-128 key must be 16 bytes")

    def encrypt(self, message: bytes) -> bytes:
        return ecb_host(self.__key, message, decrypt=False)

    def decrypt(self, message: bytes) -> bytes:
        return ecb_host(self.__key, message, decrypt=True)

    def add_padding(self, msg: bytes) -> bytes:
        """Pad with spaces to a multiple of 16 (text commands tolerate trailing whitespace)."""
        rest = len(msg) % self.BLOCK
        return msg if rest == 0 else msg + b" " * (self.BLOCK - rest)

    def get_block_size(self) -> int:
        return self.BLOCK

    def get_key(self) -> bytes:
        return self.__key

    def round_keys(self) -> bytes:
        """176-byte expanded key for the device kernel."""
        return expand_key(self.__key)

    @staticmethod
    def key_len() -> int:
        # the reference returns 256 here although its key is 16 bytes (encrypter.py:118,193)
        return 16


# ---------------------------------------------------------------------------- RSA
_SMALL_PRIMES = [3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97]


def _is_probable_prime(n: int, rounds: int = 24) -> bool:
    if n < 2:
        return False
    for p in [2] + _SMALL_PRIMES:
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for _ in range(rounds):
        a = secrets.randbelow(n - 3) + 2
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = pow(x, 2, n)
            if x == n - 1:
                break
        else:
            return False
    return True


def _gen_prime(bits: int) -> int:
    while True:
        c = secrets.randbits(bits) | (1 << (bits - 1)) | (1 << (bits - 2)) | 1
        if _is_probable_prime(c):
            return c


def _der_len(n: int) -> bytes:
    if n < 0x80:
        return bytes([n])
    b = n.to_bytes((n.bit_length() + 7) // 8, "big")
    return bytes([0x80 | len(b)]) + b


def _der_int(v: int) -> bytes:
    b = v.to_bytes(v.bit_length() // 8 + 1, "big")
    return b"\x02" + _der_len(len(b)) + b


def _der_seq(*parts: bytes) -> bytes:
    body = b"".join(parts)
    return b"\x30" + _der_len(len(body)) + body


_RSA_OID = bytes.fromhex("06092a864886f70d0101010500")   # rsaEncryption OID + NULL


def _encode_spki(n: int, e: int) -> bytes:
    rsa_pub = _der_seq(_der_int(n), _der_int(e))
    bitstr = b"\x03" + _der_len(len(rsa_pub) + 1) + b"\x00" + rsa_pub
    return _der_seq(_der_seq(_RSA_OID), bitstr)


def _read_tlv(buf: bytes, pos: int):
    tag = buf[pos]
    ln = buf[pos + 1]
    pos += 2
    if ln & 0x80:
        k = ln & 0x7F
        ln = int.from_bytes(buf[pos:pos + k], "big")
        pos += k
    return tag, buf[pos:pos + ln], pos + ln


def _decode_spki(der: bytes):
    _, outer, _ = _read_tlv(der, 0)
    _, _alg, p = _read_tlv(outer, 0)
    _, bits, _ = _read_tlv(outer, p)
    _, seq, _ = _read_tlv(bits[1:], 0)
    _, nb, p = _read_tlv(seq, 0)
    _, eb, _ = _read_tlv(seq, p)
    return int.from_bytes(nb, "big"), int.from_bytes(eb, "big")


def _mgf1(seed: bytes, length: int) -> bytes:
    out = b""
    for c in range((length + 19) // 20):
        out += hashlib.sha1(seed + c.to_bytes(4, "big")).digest()
    return out[:length]


class RSACipher(Encrypter):
    """RSA key pair + the peer's public key; OAEP(SHA-1) encrypt-to-peer / decrypt-own."""

    BITS = 1024
    _HLEN = 20

    def __init__(self):
        e = 65537
        while True:
            p, q = _gen_prime(self.BITS // 2), _gen_prime(self.BITS // 2)
            n = p * q
            phi = (p - 1) * (q - 1)
            if p != q and n.bit_length() == self.BITS and phi % e != 0:
                break
        self.__n, self.__e, self.__d = n, e, pow(e, -1, phi)
        self.__public_der = _encode_spki(n, e)
        self.__pair = None   # (n, e) of the peer

    def get_key(self) -> bytes:
        """base64(DER SubjectPublicKeyInfo) of the own public key (fixed length for a key size)."""
        return base64.b64encode(self.__public_der)

    def load_pair_public_key(self, key: bytes) -> None:
        self.__pair = _decode_spki(base64.b64decode(key))

    def encrypt(self, message: bytes) -> bytes:
        if self.__pair is None:
            raise ValueError("peer public key not loaded")
        n, e = self.__pair
        k = (n.bit_length() + 7) // 8
        if len(message) > k - 2 * self._HLEN - 2:
            raise ValueError("message too long for RSA-OAEP")
        lhash = hashlib.sha1(b"").digest()
        db = lhash + b"\x00" * (k - len(message) - 2 * self._HLEN - 2) + b"\x01" + message
        seed = os.urandom(self._HLEN)
        masked_db = bytes(a ^ b for a, b in zip(db, _mgf1(seed, len(db))))
        masked_seed = bytes(a ^ b for a, b in zip(seed, _mgf1(masked_db, self._HLEN)))
        em = b"\x00" + masked_seed + masked_db
        return pow(int.from_bytes(em, "big"), e, n).to_bytes(k, "big")

    def decrypt(self, message: bytes) -> bytes:
        k = (self.__n.bit_length() + 7) // 8
        em = pow(int.from_bytes(message, "big"), self.__d, self.__n).to_bytes(k, "big")
        masked_seed, masked_db = em[1:1 + self._HLEN], em[1 + self._HLEN:]
        seed = bytes(a ^ b for a, b in zip(masked_seed, _mgf1(masked_db, self._HLEN)))
        db = bytes(a ^ b for a, b in zip(masked_db, _mgf1(seed, len(masked_db))))
        if em[0] != 0 or db[:self._HLEN] != hashlib.sha1(b"").digest():
            raise ValueError("RSA-OAEP decryption error")
        idx = db.index(b"\x01", self._HLEN)
        return db[idx + 1:]
