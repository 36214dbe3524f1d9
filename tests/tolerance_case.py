"""A constructed input on which update_mmap meets |m_old - m_new| == 1e-5f EXACTLY with an unchanged parent
(spec delta 4): there the keep rule's comparison type decides.  liblqr compares
(double) fabsf(d) < 1e-5 -- 1e-5f = 0x3727C5AC is below the double 1e-5, so the stale value is KEPT; a float
comparison fabsf(d) < 1e-5f would update it, and the second seam would run elsewhere.

Construction (found by the random search in the docstring of find_parameters): a 2000 x 3 image, NULL energy
(LQR_EF_NULL: en = bias / w_start), preservation mask with bias_factor 1 on row 0 only:
  column 0      RGBA (255, 187, 0, 21)   colour sum 442, alpha  21  ->  en = eb = 1.1895425e-05
  columns 1..   RGBA ( 85,   0, 0, 201)  colour sum  85, alpha 201  ->  en = ea = 2.1895425e-05,  ea - eb == 1e-5f exactly
Seam 1 is column 0 on every row (m = eb all the way).  Afterwards the row-1 pixel that was at column 1 loses its parent
and takes ea (its back pointer changes: updated), and on row 2 the pixel that was at column 2 keeps the SAME parent,
whose m went eb -> ea: |m_old - m_new| = ea - eb = 1e-5f.  Kept (liblqr's rule): row 2 reads [ea, eb, ea, ...] and seam
2 starts at that pixel -> original columns (1, 1, 2) on rows (0, 1, 2).  Updated (float rule): all ea, leftmost wins ->
original column 1 on every row.
"""
import numpy as np

W, H = 2000, 3
FACTOR = 1


def build():
    img = np.zeros((H, W, 4), np.uint8)
    img[..., 3] = 255
    mask = np.zeros((H, W, 4), np.uint8)
    mask[0, 0] = (255, 187, 0, 21)
    mask[0, 1:] = (85, 0, 0, 201)
    return img, mask


def energies():
    """the two row-0 energies as the engine's arithmetic produces them (k_mask_add, energy_at)"""
    def q(s, a):
        b = (float(FACTOR) * s) / float(2 * 255 * 3)
        b = b * (a / 255.0)
        return np.float32(np.float32(b) / np.float32(W))
    return q(85, 201), q(442, 21)


EXPECTED_SEAM2_COLUMNS = (1, 1, 2)        # original column of the second seam on rows 0, 1, 2 under liblqr's (double) rule


def find_parameters(seed=1, n=4_000_000, rounds=60):
    """the search that produced the numbers above: random (w, colour sums, alphas), bias_factor = the integer nearest to
    what would make ea - eb = 1e-5, kept if the float arithmetic lands on 0x3727C5AC exactly (a few hits per 10^8 draws)"""
    rng = np.random.default_rng(seed)
    target = np.float32(1e-5)
    t64 = float(target)
    for _ in range(rounds):
        w = rng.integers(8, 4000, n).astype(np.float64)
        sa = rng.integers(1, 766, n).astype(np.float64); aa = rng.integers(1, 256, n).astype(np.float64)
        sb = rng.integers(0, 766, n).astype(np.float64); ab = rng.integers(1, 256, n).astype(np.float64)
        dn = sa * aa - sb * ab
        ok = dn > 0
        freal = np.where(ok, t64 * 1530.0 * 255.0 * w / np.where(ok, dn, 1), 0)
        f = np.rint(freal)
        idx = np.nonzero(ok & (np.abs(freal - f) < 3e-4) & (f >= 1) & (f < 2 ** 31))[0]
        if not len(idx):
            continue
        fw = w[idx].astype(np.float32)

        def q(s, a):
            return ((((f[idx] * s) / 1530.0) * (a / 255.0)).astype(np.float32) / fw).astype(np.float32)
        qa, qb = q(sa[idx], aa[idx]), q(sb[idx], ab[idx])
        hit = ((qa - qb).astype(np.float32).view(np.uint32) == target.view(np.uint32)) & (qa < 6e-5)
        for j in np.nonzero(hit)[0]:
            i = idx[j]
            yield int(w[i]), int(sa[i]), int(aa[i]), int(sb[i]), int(ab[i]), int(f[i])
