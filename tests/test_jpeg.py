"""cv::imdecode(GRAYSCALE) restatement (jackal_navigation_amd/csrc/jpeg.hip): the host half on CPU.

Third-party arithmetic (libjpeg via OpenCV) pinned by Pillow / libjpeg-turbo fixtures
(tests/golden/jpeg_cases.npz, tests/golden/make_jpeg_golden.py).  Here: marker parsing and Huffman decoding through the
host hook jn_host_jpeg_coefficients, with the inverse DCT restated in numpy (the IJG "slow integer" definition) so that
the whole chain can be compared with the fixtures without a GPU; tests/test_gpu_round2.py runs the kernel."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cases():
    z = np.load(os.path.join(ROOT, "tests", "golden", "jpeg_cases.npz"))
    names = sorted({k.split("__")[0] for k in z.files})
    return z, names


def islow_idct(block):
    """8x8 IJG slow-integer inverse DCT of one dequantised block (int64 numpy, exact), result before range limiting."""
    def one_d(v, shift):
        z2, z3 = v[2], v[6]
        z1 = (z2 + z3) * 4433
        tmp2 = z1 + z3 * (-15137); tmp3 = z1 + z2 * 6270
        tmp0 = (v[0] + v[4]) << 13; tmp1 = (v[0] - v[4]) << 13
        tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
        t0, t1, t2, t3 = v[7], v[5], v[3], v[1]
        z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2; z4 = t1 + t3
        z5 = (z3 + z4) * 9633
        t0 = t0 * 2446; t1 = t1 * 16819; t2 = t2 * 25172; t3 = t3 * 12299
        z1 = z1 * -7373; z2 = z2 * -20995; z3 = z3 * -16069 + z5; z4 = z4 * -3196 + z5
        t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4
        r = 1 << (shift - 1)
        return [(tmp10 + t3 + r) >> shift, (tmp11 + t2 + r) >> shift, (tmp12 + t1 + r) >> shift, (tmp13 + t0 + r) >> shift,
                (tmp13 - t0 + r) >> shift, (tmp12 - t1 + r) >> shift, (tmp11 - t2 + r) >> shift, (tmp10 - t3 + r) >> shift]
    b = block.astype(np.int64)
    ws = np.stack(one_d([b[r] for r in range(8)], 11))            # pass 1 works on columns: element r of every column at once
    out = np.stack(one_d([ws[:, k] for k in range(8)], 18), axis=1)
    return out


def range_limit(x):
    i = x & 1023
    return np.where(i < 128, 128 + i, np.where(i < 512, 255, np.where(i < 896, 0, i - 896))).astype(np.uint8)


def host_decode(jn, data):
    L = jn.load()
    buf = np.ascontiguousarray(data, np.uint8)
    q = (C.c_uint16 * 64)()
    w, h, bw, bh = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    n = L.jn_host_jpeg_coefficients(buf.ctypes.data, buf.size, None, 0, q, C.byref(w), C.byref(h), C.byref(bw), C.byref(bh))
    if n < 0:
        return -n, None
    coef = np.zeros(n, np.int16)
    assert L.jn_host_jpeg_coefficients(buf.ctypes.data, buf.size, coef.ctypes.data, n, q, C.byref(w), C.byref(h), C.byref(bw), C.byref(bh)) == n
    quant = np.array(list(q), np.int64).reshape(8, 8)
    blocks = coef.reshape(bh.value, bw.value, 8, 8).astype(np.int64) * quant
    img = np.zeros((bh.value * 8, bw.value * 8), np.uint8)
    for by in range(bh.value):
        for bx in range(bw.value):
            img[8 * by:8 * by + 8, 8 * bx:8 * bx + 8] = range_limit(islow_idct(blocks[by, bx]))
    return 0, img[:h.value, :w.value]


def test_entropy_decoder_and_idct_definition_reproduce_libjpeg(jn):
    z, names = cases()
    checked = 0
    for name in names:
        if name == "progressive":
            continue
        st, img = host_decode(jn, z[name + "__jpeg"])
        assert st == 0, name
        assert img.shape == tuple(z[name + "__shape"]), name
        assert hashlib.sha256(img.tobytes()).digest() == z[name + "__sha256"].tobytes(), name
        if name + "__gray" in z.files:
            assert np.array_equal(img, z[name + "__gray"])
        checked += 1
    assert checked >= 10


def test_jpeg_info_and_refusals(jn):
    from jackal_navigation_amd import _lib, node
    z, names = cases()
    assert node.jpeg_info(z["webcam_640x360_q80_420__jpeg"]) == (640, 360)
    assert node.jpeg_info(z["ragged_35x21_q95_422__jpeg"]) == (35, 21)
    with pytest.raises(_lib.JnError) as e:
        node.jpeg_info(z["progressive__jpeg"])
    assert e.value.status == _lib.JN_ERR_UNSUPPORTED
    st, _ = host_decode(jn, z["progressive__jpeg"])
    assert st == _lib.JN_ERR_UNSUPPORTED
    good = z["q90_422__jpeg"]
    for bad in (good[:200], np.zeros(64, np.uint8), good[2:]):
        st, _ = host_decode(jn, bad)
        assert st == _lib.JN_ERR_INVALID
    cut = good[: len(good) // 2].copy()                    # truncated scan: still decodes (zeros are fed past the end), like libjpeg's warning path
    st, img = host_decode(jn, cut)
    assert st in (0, _lib.JN_ERR_INVALID)


def _dht(counts, nsym=None, tc_th=0x00):
    counts = list(counts) + [0] * (16 - len(counts))
    n = sum(counts) if nsym is None else nsym
    body = bytes([tc_th]) + bytes(counts) + bytes(range(n % 257))[:n]
    return b"\xff\xc4" + (len(body) + 2).to_bytes(2, "big") + body


def test_damaged_huffman_tables_and_oversized_frames_are_refused(jn):
    """a DHT that is not a prefix code (255 codes of length 1) made the 8-bit look-up index run past the
    table — an out-of-bounds stack write from a 40-byte file.  libjpeg rejects such tables (JERR_BAD_HUFF_TABLE); so must we.
    (medium): SOF dimensions drive the coefficient buffer; frames beyond 8192x8192 are refused before any allocation."""
    from jackal_navigation_amd import _lib
    soi = b"\xff\xd8"
    for counts in ([255], [3], [2, 3], [1, 1, 1, 1, 1, 1, 1, 1, 2], [0] * 15 + [255, ]):
        st, _ = host_decode(jn, np.frombuffer(soi + _dht(counts) + b"\xff\xd9", np.uint8))
        assert st == _lib.JN_ERR_INVALID, counts
    # the all-ones code of a length is reserved (T.81 C.2): 2 codes of length 1 use it
    st, _ = host_decode(jn, np.frombuffer(soi + _dht([2]) + b"\xff\xd9", np.uint8))
    assert st == _lib.JN_ERR_INVALID
    z, _ = cases()
    good = bytes(z["q90_422__jpeg"])
    i = good.index(b"\xff\xc0")
    huge = bytearray(good); huge[i + 5:i + 9] = b"\xff\xff\xff\xff"          # 65535 x 65535
    st, _ = host_decode(jn, np.frombuffer(bytes(huge), np.uint8))
    assert st == _lib.JN_ERR_UNSUPPORTED
    big = bytearray(good); big[i + 5:i + 9] = (8200).to_bytes(2, "big") + (16).to_bytes(2, "big")
    st, _ = host_decode(jn, np.frombuffer(bytes(big), np.uint8))
    assert st == _lib.JN_ERR_UNSUPPORTED


def test_fuzzed_files_never_crash_the_host_decoder(jn):
    """Small fuzz corpus: byte flips, truncations, spliced segments and random DHT/DQT/SOF/SOS bodies over three valid files.
    Every outcome must be a status (or a decoded image), never a crash; test_sanitizers.py repeats this under ASan/UBSan."""
    z, names = cases()
    rng = np.random.default_rng(2026)
    seeds = [bytes(z[n + "__jpeg"]) for n in ("ragged_35x21_q95_422", "q90_422", "restart_blocks_q80_444")]
    outcomes = {}
    for it in range(600):
        b = bytearray(seeds[it % len(seeds)])
        kind = it % 5
        if kind == 0:                                                       # flips in the headers
            for _ in range(int(rng.integers(1, 8))):
                b[int(rng.integers(2, min(len(b), 700)))] = int(rng.integers(0, 256))
        elif kind == 1:                                                     # flips anywhere
            for _ in range(int(rng.integers(1, 30))):
                b[int(rng.integers(2, len(b)))] = int(rng.integers(0, 256))
        elif kind == 2:
            b = b[:int(rng.integers(2, len(b)))]
        elif kind == 3:                                                     # a random table segment right after SOI
            m = [0xC4, 0xDB, 0xC0, 0xDA, 0xDD][int(rng.integers(0, 5))]
            body = bytes(rng.integers(0, 256, int(rng.integers(0, 300))).astype(np.uint8))
            b = bytearray(b[:2] + bytes([0xFF, m]) + (len(body) + 2).to_bytes(2, "big") + body + bytes(b[2:]))
        else:                                                               # a random prefix-violating DHT
            b = bytearray(b[:2] + _dht(list(rng.integers(0, 256, 16)), nsym=int(rng.integers(0, 257))) + bytes(b[2:]))
        st, img = host_decode(jn, np.frombuffer(bytes(b), np.uint8))
        outcomes[st] = outcomes.get(st, 0) + 1
    assert set(outcomes) <= {0, 2, 3, 5}, outcomes
    assert outcomes.get(3, 0) > 50, outcomes
