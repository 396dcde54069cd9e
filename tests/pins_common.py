"""What the reference's published renders (result/*.png, scene/cornell-box-SPP2.png) pin, shared by the CPU tests of the
oracle (test_oracle_pins.py) and the GPU tests of the product (test_gpu_parity.py).  Fixtures: tests/golden/published_masks.npz
(full-resolution masks of the pixels equal to (255,255,255)) and published_renders.npz (16x16-block means), both written by
tests/golden/make_golden.py from the reference's files.

  * Deterministic pixels.  A primary hit on an emitter returns the light's radiance un-weighted for every sample
    (MTPC/pathTracing.cpp:141-144); every shipped radiance is > 1 and imshow clamps (MTPC/MTPC.cpp:26-28), so such a pixel is
    (255,255,255) whatever the RNG did.  Emitter pixels of our primary hits must therefore all be saturated in every published
    render of the scene -- a pixel-exact pin of the camera model (Q11: pixel corners, running sum), the loader, the Morton
    order/BVH and the primary closest hit; in cornell-box-SPP25.png only 4 other pixels are saturated (fireflies far from the
    light), so there the silhouette is pinned from both sides: a camera shifted by 0.05 pixel already flips ~200 pixels.
  * Monte-Carlo pixels.  Everything else is time-seeded noise around an expectation that the restatement must share.  Per
    16x16 block and channel: z = (published block mean - our block mean) / sigma, where our mean comes from two independent
    renders at the published SPP and native resolution (same pixels, same quantisation) and sigma^2 = 1.5 x the block-mean
    variance estimated from the per-pixel differences of those two renders.  If the restatement is right the z are N(0,1).
    One thing is NOT shared with the reference by design (D1): its four RNG engines are read and advanced without a lock by the
    8 OpenMP threads that render one pixel's samples (MTPC/pathTracing.cpp:5,32,68,169,303), so samples of a pixel repeat draws and
    are positively correlated; ours are independent.  The expectation of a sample is the same, the spread of a pixel's mean is not,
    and imshow's clamp at 255 turns a wider spread into a darker expectation.  It shows exactly where it must: veach-mis blocks
    with no saturated pixel agree to 0.01-0.1 % (z mean 0.00-0.06), blocks in which 10-50 % of the pixels saturate are 0.5 %
    darker in the published SPP-10 render and 0.1 % darker at SPP 100 (a 10x narrower spread).  The z test therefore runs on
    blocks the clamp leaves alone (level < 140 of 255: < 1 % saturated pixels); the others are held to a one-sided band.
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
BLOCK = 16

# published render -> (scene, samples per pixel).  scene/cornell-box-SPP2.png and result/cornell-box-SPP100.png are four times
# darker than the others and SPP16/SPP50 are 2 % brighter with a 15 % brighter ceiling: other revisions of the integrator.
# Their emitter pixels are still deterministic, so they take part in the silhouette pin, not in the z test.
RENDERS = {
    "cornell_spp2": ("cornell-box", 2), "cornell_spp2_result": ("cornell-box", 2), "cornell_spp16": ("cornell-box", 16),
    "cornell_spp25": ("cornell-box", 25), "cornell_spp50": ("cornell-box", 50), "cornell_spp100": ("cornell-box", 100),
    "veach_spp10": ("veach-mis", 10), "veach_spp100": ("veach-mis", 100),
}
SAME_REVISION = ("cornell_spp2_result", "cornell_spp25", "veach_spp10", "veach_spp100")


def saturated_mask(name):
    """[rows present, W] bool: pixel == (255,255,255) in the published render"""
    g = np.load(os.path.join(GOLD, "published_masks.npz"))
    w, h, n = (int(v) for v in g[name + "_size"])
    return np.unpackbits(g[name])[: n * w].reshape(n, w).astype(bool)


def published_blocks(name):
    """[rows/16, W/16, 3] float32 block means (0..255) and (W, H, rows present)"""
    g = np.load(os.path.join(GOLD, "published_renders.npz"))
    return g[name], tuple(int(v) for v in g[name + "_size"])


def check_emitter_pixels(emit, scene):
    """emit: [H, W] bool, our primary hit is an emitter.  Returns {render: (emitter pixels, other saturated pixels)}."""
    out = {}
    for name, (sc, _) in RENDERS.items():
        if sc != scene:
            continue
        sat = saturated_mask(name)
        e = emit[: sat.shape[0]]
        missing = int((e & ~sat).sum())
        assert missing == 0, "%s: %d emitter pixels are not saturated in the published render" % (name, missing)
        out[name] = (int(e.sum()), int((sat & ~e).sum()))
    return out


def block_stats(name, q_a, q_b, block_step=1):
    """q_a, q_b: [H, W, 3] uint8 quantised renders of two seeds at the published SPP (only the selected blocks need to be
    filled).  Returns (diff [n, 3] = published block mean - ours, sigma [n, 3] of that difference, ours [n, 3], sat [n] = share of
    saturated pixel channels in our two renders) over the blocks (br, bc) with br % step == bc % step == 0."""
    pub, (w, h, rows) = published_blocks(name)
    a = q_a.astype(np.float64)
    b = q_b.astype(np.float64)
    diff, sigma, ours, sat = [], [], [], []
    npx = BLOCK * BLOCK
    for br in range(0, rows // BLOCK, block_step):
        for bc in range(0, w // BLOCK, block_step):
            sl = (slice(br * BLOCK, (br + 1) * BLOCK), slice(bc * BLOCK, (bc + 1) * BLOCK))
            d = a[sl] - b[sl]
            var_blk = ((d * d / 2).sum(axis=(0, 1)) + npx / 6.0) / npx ** 2      # + quantisation noise of both sides
            e = (a[sl] + b[sl]).mean(axis=(0, 1)) / 2
            diff.append(pub[br, bc].astype(np.float64) - e)
            sigma.append(np.sqrt(1.5 * var_blk))                                  # published run + the mean of our two
            ours.append(e)
            sat.append(((a[sl] >= 255).mean() + (b[sl] >= 255).mean()) / 2)
    return np.array(diff), np.array(sigma), np.array(ours), np.array(sat)


def selected_blocks(name, block_step):
    _, (w, h, rows) = published_blocks(name)
    return [(br, bc) for br in range(0, rows // BLOCK, block_step) for bc in range(0, w // BLOCK, block_step)]


def assert_standard_normal(diff, sigma, what, mean_tol=0.2, rms=(0.8, 1.25), tail=0.01):
    """Block z-scores are N(0,1) when both sides sample the same expectation.  A few blocks are heavy-tailed (fireflies), hence
    clipped moments and a counted tail; `whole` is the z-score of the summed difference per channel (the picture's brightness),
    which no per-block weighting can bias."""
    z = diff / sigma
    zc = np.clip(z, -6, 6)
    m, r, t = float(zc.mean()), float(np.sqrt((zc ** 2).mean())), float((np.abs(z) > 4).mean())
    whole = diff.sum(axis=0) / np.sqrt((sigma ** 2).sum(axis=0))
    msg = "%s: z mean %.3f rms %.3f tail(|z|>4) %.4f over %d values; whole-picture z per channel %s" % (
        what, m, r, t, z.size, np.round(whole, 2).tolist())
    assert abs(m) < mean_tol, msg
    assert rms[0] < r < rms[1], msg
    assert t < tail, msg
    assert np.abs(whole).max() < 4.0, msg
    return msg


def assert_matches_published(name, q_a, q_b, what, block_step=1, clamp_band=(-0.012, 0.004), **kw):
    """The z test on the blocks the clamp leaves alone, the one-sided band on the rest (module docstring): the published render may
    be darker there by up to 1.2 % of the blocks' brightness, not brighter.  "Left alone" = the block's level, averaged over the
    published and our renders (a selection symmetric in the two sides, so it cannot bias their difference), is below 140 of 255:
    such blocks have < 1 % saturated pixels."""
    diff, sigma, ours, sat = block_stats(name, q_a, q_b, block_step)
    free = (ours + diff / 2).max(axis=1) < 140.0
    msg = assert_standard_normal(diff[free], sigma[free], what + " (blocks below level 140: %d of %d)" % (free.sum(), free.size), **kw)
    rest = ~free & (np.abs(diff).max(axis=1) > 0)
    if rest.sum() >= 20:
        rel = float(diff[rest].sum() / ours[rest].sum())
        msg += "; %d blocks with saturating pixels: published %+.2f %%" % (rest.sum(), 100 * rel)
        assert clamp_band[0] < rel < clamp_band[1], msg
    return msg
