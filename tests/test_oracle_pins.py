"""Pins of the CPU oracle (not gpu): everything the reference itself provides to check a restatement against.

  * Philox4x32-10 known-answer vectors (Random123's kat_vectors) for the RNG seam;
  * Morton keys against the reference's own "morton code.cpp" (compiled where it lies into oracle/_ref, plus the
    committed vectors generated from it: tests/golden/morton_vectors.npz);
  * PNG bytes against the reference's own svpng.inc (oracle/_ref + committed files);
  * the integrator as a whole against the renders the reference publishes (result/*.png block means): statistical,
    because those runs are time-seeded -- bitwise parity of traversal/shading is UNPINNED.
"""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT, SCENES

GOLD = os.path.join(ROOT, "tests", "golden")


def test_philox_known_answers(oracle):
    assert oracle.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert oracle.philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert oracle.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_uniform_is_53_bit_unit_interval(oracle):
    u = np.array([oracle.uniform(7, p, k, d, s) for p in range(4) for k in range(4) for d in range(3) for s in range(9)])
    assert (u >= 0).all() and (u < 1).all()
    assert len(np.unique(u)) == len(u)
    assert np.all(u * 2.0 ** 53 == np.floor(u * 2.0 ** 53))
    # slots 2b and 2b+1 come from one Philox block: words (0,1) and (2,3)
    o = oracle.philox([3, 5, (2 << 16) | 4, 0x4D435054], [9, 0])
    assert oracle.uniform(9, 3, 5, 2, 8) == ((o[0] << 32 | o[1]) >> 11) * 2.0 ** -53
    assert oracle.uniform(9, 3, 5, 2, 9) == ((o[2] << 32 | o[3]) >> 11) * 2.0 ** -53


def test_morton_against_committed_reference_vectors(oracle):
    g = np.load(os.path.join(GOLD, "morton_vectors.npz"))
    mine = np.array([oracle.morton(float(a), float(b), float(c)) for a, b, c in g["xyz"]], dtype=np.uint32)
    assert np.array_equal(mine, g["code"])


def test_morton_against_reference_binary(oracle):
    so = os.path.join(ROOT, "oracle", "_ref", "libref_morton.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    ref = C.CDLL(so)._Z13getMortonCodefff
    ref.restype = C.c_uint32
    ref.argtypes = [C.c_float] * 3
    rng = np.random.default_rng(99)
    pts = rng.uniform(-3, 6, size=(5000, 3)).astype(np.float32)
    for a, b, c in pts:
        assert oracle.morton(float(a), float(b), float(c)) == ref(C.c_float(a), C.c_float(b), C.c_float(c))


def _golden_image(w, h):
    y, x = np.mgrid[0:h, 0:w]
    return np.stack([(x * 7 + y * 3) % 256, (x * x + y) % 256, (x ^ (y * 5)) % 256], axis=-1).astype(np.uint8)


@pytest.mark.parametrize("w,h", [(37, 23), (256, 5)])
def test_png_bytes_equal_reference_svpng(oracle, w, h):
    want = open(os.path.join(GOLD, "svpng_%dx%d.png" % (w, h)), "rb").read()
    assert oracle.png_encode(_golden_image(w, h)) == want


def test_png_against_reference_binary(oracle, tmp_path):
    so = os.path.join(ROOT, "oracle", "_ref", "libref_svpng.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref not built")
    ref = C.CDLL(so)
    ref.ref_svpng_write.argtypes = [C.c_char_p, C.c_uint, C.c_uint, C.c_char_p]
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, size=(31, 77, 3), dtype=np.uint8)
    out = str(tmp_path / "r.png")
    assert ref.ref_svpng_write(out.encode(), 77, 31, img.tobytes()) == 0
    assert oracle.png_encode(img) == open(out, "rb").read()
    w, h, rows = oracle.read_stored_png(out)
    assert (w, h) == (77, 31) and np.array_equal(rows, img)


def test_quantize_is_truncating_clamp(oracle):
    v = np.array([-1.0, 0.0, 0.5 / 255, 0.999 / 255, 1.0 / 255, 0.5, 1.0, 1.5, 254.999 / 255, np.inf])
    assert oracle.quantize(v).tolist() == [0, 0, 0, 0, 1, 127, 255, 255, 254, 255]


def _blocks(img8, b):
    h, w, _ = img8.shape
    hh, ww = (h // b) * b, (w // b) * b
    return img8[:hh, :ww].astype(np.float32).reshape(hh // b, b, ww // b, b, 3).mean(axis=(1, 3))


def test_cornell_matches_published_render_statistically(oracle):
    """The shipped main() renders cornell-box at SPP 25 (MTPC/MTPC.cpp:78); result/cornell-box-SPP25.png is that run.
    The oracle at a quarter of the resolution (same field of view) must reproduce its 16x16-block means."""
    pub = np.load(os.path.join(GOLD, "published_renders.npz"))["cornell_spp25"]      # [64,64,3], 0..255
    s = oracle.OracleScene(SCENES + "cornell-box", texture_dir=SCENES, width=256, height=256)
    img = s.render(25, seed=2025)
    mine = _blocks(oracle.quantize(img), 4)
    assert mine.shape == pub.shape
    assert abs(mine.mean() - pub.mean()) < 0.012 * pub.mean(), (mine.mean(), pub.mean())
    assert np.corrcoef(mine.ravel(), pub.ravel())[0, 1] > 0.975
    for name, (rs, cs) in {"ceiling": (slice(1, 6), slice(8, 22)), "back wall": (slice(12, 30), slice(12, 52)),
                           "floor": (slice(58, 63), slice(12, 52)), "left wall": (slice(10, 50), slice(1, 6)),
                           "right wall": (slice(10, 50), slice(58, 63)), "furniture": (slice(36, 54), slice(14, 50))}.items():
        a, b = mine[rs, cs].mean(), pub[rs, cs].mean()
        assert abs(a - b) < 0.05 * b + 0.5, (name, a, b)


def test_veach_matches_published_render_statistically(oracle):
    """veach-mis exercises five lights, the frozen light-area distribution (Q1) and the Phong lobe; the published files are
    truncated by one row (no fclose in imshow) but 899 rows decode."""
    g = np.load(os.path.join(GOLD, "published_renders.npz"))
    pub = g["veach_spp10"]                                                             # [56,75,3]
    s = oracle.OracleScene(SCENES + "veach-mis", texture_dir=SCENES, width=300, height=225)
    img = s.render(10, seed=7)
    mine = _blocks(oracle.quantize(img)[:224], 4)
    assert mine.shape == pub.shape
    assert abs(mine.mean() - pub.mean()) < 0.02 * pub.mean(), (mine.mean(), pub.mean())
    assert np.corrcoef(mine.ravel(), pub.ravel())[0, 1] > 0.98
