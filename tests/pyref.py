"""Independent pure-Python references used to pin the C oracle (tests only).

Nothing here is derived from the reference tree: python ints for secp256k1, hashlib for
SHA-2, numpy's RandomState for MT19937.
"""
import hashlib

P = 2**256 - 2**32 - 977
N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
GX = 0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798
GY = 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8


def ec_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    if a[0] == b[0]:
        if (a[1] + b[1]) % P == 0:
            return None
        lam = 3 * a[0] * a[0] * pow(2 * a[1], -1, P) % P
    else:
        lam = (b[1] - a[1]) * pow(b[0] - a[0], -1, P) % P
    x = (lam * lam - a[0] - b[0]) % P
    return x, (lam * (a[0] - x) - a[1]) % P


def ec_mul(k):
    r, q = None, (GX, GY)
    while k:
        if k & 1:
            r = ec_add(r, q)
        q = ec_add(q, q)
        k >>= 1
    return r


def normalize(key: bytes) -> int:
    k = int.from_bytes(key, "big")
    if k == 0 or k >= N:
        k %= N
        if k == 0:
            k = 1
    return k


def pubkeys(k: int):
    x, y = ec_mul(k)
    xb, yb = x.to_bytes(32, "big"), y.to_bytes(32, "big")
    return bytes([2 | (y & 1)]) + xb, b"\x04" + xb + yb


B58 = "123456789ABCDEFGHJKLMNPQRSTUVWXYZabcdefghijkmnopqrstuvwxyz"


def b58check(payload: bytes) -> str:
    full = payload + hashlib.sha256(hashlib.sha256(payload).digest()).digest()[:4]
    n = int.from_bytes(full, "big")
    s = ""
    while n:
        n, r = divmod(n, 58)
        s = B58[r] + s
    return "1" * (len(full) - len(full.lstrip(b"\0"))) + s


def armory_key(seed: bytes):
    """Armory root-key chain in python ints + hashlib/hmac: chaincode = HMAC-SHA256(SHA256d(seed), "Derive Chaincode from
    Root Key"), then four times key <- key * (SHA256d(04||X||Y) xor chaincode) mod n.  None where a secret key / scalar
    would be rejected (zero or >= n)."""
    import hmac
    assert len(seed) == 32
    h2 = hashlib.sha256(hashlib.sha256(seed).digest()).digest()
    cc = hmac.new(h2, b"Derive Chaincode from Root Key", hashlib.sha256).digest()
    k = int.from_bytes(seed, "big")
    if not 0 < k < N:
        return None
    for _ in range(4):
        pub = pubkeys(k)[1]
        d = hashlib.sha256(hashlib.sha256(pub).digest()).digest()
        s = int.from_bytes(bytes(a ^ b for a, b in zip(d, cc)), "big")
        if not 0 < s < N:
            return None
        k = k * s % N
    return k.to_bytes(32, "big")


def bip32_ckd(k: bytes, c: bytes, index: int, hardened: bool):
    """CKDpriv in python ints + hmac (BIP-32): None where a libsecp256k1 caller would see an error."""
    import hmac
    ki = int.from_bytes(k, "big")
    if not 0 < ki < N:
        return None
    idx = (index | 0x80000000) if hardened else index
    data = (b"\0" + k if hardened else pubkeys(ki)[0]) + idx.to_bytes(4, "big")
    I = hmac.new(c, data, hashlib.sha512).digest()
    il = int.from_bytes(I[:32], "big")
    if il >= N or (il + ki) % N == 0:
        return None
    return ((il + ki) % N).to_bytes(32, "big"), I[32:]


def bip32_path(seed: bytes, comps):
    """Master key from a BIP-32 seed, then CKDpriv along comps = [(index, hardened)]."""
    import hmac
    I = hmac.new(b"Bitcoin seed", seed, hashlib.sha512).digest()
    k, c = I[:32], I[32:]
    for index, hardened in comps:
        r = bip32_ckd(k, c, index, hardened)
        if r is None:
            return None
        k, c = r
    return k


def bip39_mnemonic(words, entropy: bytes):
    bits = "".join(f"{b:08b}" for b in entropy) + "".join(f"{b:08b}" for b in hashlib.sha256(entropy).digest())[:len(entropy) // 4]
    return " ".join(words[int(bits[11 * i:11 * i + 11], 2)] for i in range(len(bits) // 11))


def bitimage_key(words, data: bytes, passphrase: bytes, comps):
    """file bytes -> base64 -> SHA-256 -> BIP-39 mnemonic -> PBKDF2 seed -> BIP-32 path, all from the standard library."""
    import base64
    entropy = hashlib.sha256(base64.b64encode(data)).digest()
    seed = hashlib.pbkdf2_hmac("sha512", bip39_mnemonic(words, entropy).encode(), b"mnemonic" + passphrase, 2048)
    return bip32_path(seed, comps)
