"""CPU prototype of the WAVE-LOCAL 64x64 eigen-solver (csrc/evd_wave.h): the data flow of one wave, emulated with numpy arrays
indexed [register][lane] — lane c holds column c of G (and of the accumulated eigenvector matrix Q), register r holds row r.

Odd-even transposition ordering with swap (Luk-Park), as in the LDS solver of round 1/2: phase A pairs positions (2k, 2k+1),
phase B pairs (2k+1, 2k+2) (positions 0 and 63 idle); after every rotation the two columns / rows exchange places; 64 phases
= every pair once.  What this prototype pins down before the HIP kernel is written:
  * which lane / register holds the pivot of the NEXT phase (collected with one select per register: `b[lane] = g[lane+1][lane]`);
  * the closed-form diagonal (d + t b, a - t b) kept in a per-lane vector — the diagonal entries inside the register image are
    never read, the annihilated element is NOT zeroed (it stays at rounding level);
  * the per-lane column coefficients of both phases, incl. the idle lanes 0 and 63 of phase B;
  * everything in fp32, same jacobi_rot arithmetic (unit-norm correction) as the kernel.
Run: python tools/proto_evd_wave.py  -> prints off-diagonal decay per inner sweep and the eigenvalue / orthogonality errors."""
import numpy as np

f32 = np.float32


def jacobi_rot(a, d, b):
    """vectorised over lanes; returns c, s, t (fp32), identity where |cos| <= 1e-8 or the diagonal is not positive"""
    with np.errstate(all="ignore"):
        cosv = b / np.sqrt(a) / np.sqrt(d)
        zeta = (d - a) / (f32(2) * b)
        tt = np.copysign(f32(1), zeta) / (np.abs(zeta) + np.sqrt(zeta * zeta + f32(1)))
        cc = f32(1) / np.sqrt(tt * tt + f32(1))
        ss = tt * cc
        hd = f32(0.5) * (ss * ss + (cc * cc - f32(1)))
        cc = cc - cc * hd
        ss = ss - ss * hd
        rot = np.abs(cosv) > f32(1e-8)
    c = np.where(rot, cc, f32(1)).astype(f32)
    s = np.where(rot, ss, f32(0)).astype(f32)
    t = np.where(rot, tt, f32(0)).astype(f32)
    return c, s, t


def phase(g, q, diag, bpiv, par):
    """one phase on the [register][lane] images.  bpiv[L] must hold G[L+1][L] for the LOWER lane L of every pair of this phase."""
    n = 64
    lane = np.arange(n)
    if par == 0:
        lower = (lane & 1) == 0
        partner = lane ^ 1
        idle = np.zeros(n, bool)
    else:
        lower = (lane & 1) == 1
        partner = np.where(lower, lane + 1, lane - 1)
        idle = (lane == 0) | (lane == n - 1)
        partner = np.clip(partner, 0, n - 1)
    dpart = diag[partner]
    b = np.where(lower, bpiv, bpiv[partner])
    a_ = np.where(lower, diag, dpart)
    d_ = np.where(lower, dpart, diag)
    c, s, t = jacobi_rot(a_, d_, b)
    c = np.where(idle, f32(1), c)
    s = np.where(idle, f32(0), s)
    t = np.where(idle, f32(0), t)
    new_diag = np.where(lower, d_ + t * b, a_ - t * b).astype(f32)
    new_diag = np.where(idle, diag, new_diag)
    own = np.where(idle, f32(1), np.where(lower, s, -s)).astype(f32)
    parc = np.where(idle, f32(0), c).astype(f32)
    # rows (registers), coefficients of row pair = those of the lanes at the same positions
    y = g.copy()
    first = 0 if par == 0 else 1
    for p in range(first, n - 1, 2):
        qq = p + 1
        ck, sk = c[p], s[p]  # v_readlane from the lower lane of the pair
        x0, x1 = g[p], g[qq]
        y[p] = sk * x0 + ck * x1
        y[qq] = ck * x0 - sk * x1
    # columns (lanes)
    gn = (own[None, :] * y + parc[None, :] * y[:, partner]).astype(f32)
    qn = (own[None, :] * q + parc[None, :] * q[:, partner]).astype(f32)
    # pivot of the next phase: lower lanes there have the OTHER parity; they need register lane+1
    bnext = np.zeros(n, f32)
    for r in range(1, n):
        bnext[r - 1] = gn[r, r - 1]
    return gn, qn, new_diag, bnext


def solve(G, sweeps=1):
    n = 64
    g = G.astype(f32).copy()  # g[r][c]
    q = np.eye(n, dtype=f32)
    diag = np.diag(g).copy()
    bpiv = np.array([g[l + 1, l] if l + 1 < n else 0 for l in range(n)], f32)
    hist = []
    for sw in range(sweeps):
        for ph in range(32):
            g, q, diag, bpiv = phase(g, q, diag, bpiv, 0)
            g, q, diag, bpiv = phase(g, q, diag, bpiv, 1)
        off = g - np.diag(np.diag(g))
        hist.append(float(np.abs(off).max() / np.abs(diag).max()))
    return g, q, diag, hist


def main():
    rng = np.random.default_rng(0)
    for trial, cond in enumerate([1e1, 1e3, 1e5]):
        X = rng.standard_normal((256, 64)) * np.logspace(0, -np.log10(cond) / 2, 64)[None, :]
        X = X @ np.linalg.qr(rng.standard_normal((64, 64)))[0] * 0.3 + X  # couple the columns
        G = (X.T @ X).astype(f32)
        g, q, diag, hist = solve(G, sweeps=6)
        w = np.linalg.eigvalsh(G.astype(np.float64))[::-1]
        got = np.sort(diag.astype(np.float64))[::-1]
        qd = q.astype(np.float64)
        print(f"cond {cond:.0e}: off/diag per inner sweep {['%.1e' % h for h in hist]}")
        print(f"   eigenvalue rel err (vs max) {np.abs(got - w).max() / w[0]:.2e}   |Q^T Q - I| {np.abs(qd.T @ qd - np.eye(64)).max():.2e}"
              f"   |Q^T G Q - diag| / max {np.abs(qd.T @ G.astype(np.float64) @ qd - np.diag(diag)).max() / w[0]:.2e}")
        # register image consistency: tracked diagonal vs the image's own diagonal, and symmetry
        print(f"   tracked diag vs image diag {np.abs(np.diag(g) - diag).max() / w[0]:.2e}   asymmetry {np.abs(g - g.T).max() / w[0]:.2e}")
    # one inner sweep only (what the kernel runs per visit): the off-diagonal mass must drop
    g, q, diag, hist = solve(G, sweeps=1)
    print("one sweep:", hist)


if __name__ == "__main__":
    main()
