"""CPU simulation (numpy, no GPU) of the two-pass speculative Macenko transform:

  prior   -- a cheap presample of the tile (1024 sectors of 16 consecutive pixels at hashed positions) gives an approximate
             plane frame [V^, n^], brackets for the two angle percentiles as DIRECTIONS in that frame, and for each
             concentration slot two end directions psi1/psi2 with a lower threshold each;
  pass A  -- ONE pass over all pixels: exact moments + the pixels that cannot be ruled out (half-plane tests in the prior
             frame) are kept as candidates;
  resolve -- exact plane from the moments, exact keys of the candidates, exact order statistics, and the checks that make
             the speculation rigorous (the answer lies strictly inside the mapped brackets / above the combined threshold).

Prints, per tile: candidate fractions per slot and whether every check passed (a failed check = the slow exact path).
"""
import sys
import numpy as np
import torch

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
from stainx_amd import synth  # noqa: E402
from oracle import stain_oracle as so  # noqa: E402


def nearest_rank(n, q):
    return int(np.rint(0.01 * q * (n - 1)))


def hashed_units(P, unit, n_units, seed=0):
    """n_units sectors of `unit` consecutive pixels: sector j lies in cell j at a hashed offset."""
    sectors = P // unit
    n_units = min(n_units, sectors)
    cell = sectors // n_units
    j = np.arange(n_units, dtype=np.uint64)
    off = ((j * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF)) % np.uint64(cell)
    sec = (j * np.uint64(cell) + off).astype(np.int64)
    return (sec[:, None] * unit + np.arange(unit)[None, :]).reshape(-1)


def run(n_tiles=64, H=512, W=512, unit=16, n_units=1024, K=5.0, neff_per_unit=1.0, verbose=True):
    tiles = synth.as_dtype(synth.he_batch(n_tiles, H, W, seed0=1000), torch.float32).numpy()
    P = H * W
    tot = np.zeros(4)
    fails = 0
    for ti in range(n_tiles):
        x = tiles[ti]
        od = so.optical_density(so.to_unit_float(x[None]))[0].reshape(3, -1).T.astype(np.float64)  # (P,3)
        keep = od.min(1) >= 0.15
        # ---------------- prior ----------------
        idx = hashed_units(P, unit, n_units)
        s_od = od[idx]
        s_keep = keep[idx]
        sk = s_od[s_keep]
        cov = np.cov(sk.T)
        w_, v_ = np.linalg.eigh(cov)
        Vh = v_[:, [1, 2]] * np.sign(v_[:, [1, 2]].sum(0))
        nh = np.cross(Vh[:, 0], Vh[:, 1])
        th_s = np.arctan2(sk @ Vh[:, 1], sk @ Vh[:, 0])
        th_sorted = np.sort(th_s)
        mv = th_sorted.size
        n_eff = max(mv / unit * neff_per_unit, 8.0)   # effective sample size: ~one per sector
        def br(q):
            f = q / 100.0
            sd = np.sqrt(f * (1 - f) / n_eff)          # std of the sample quantile level
            lo = max(f - K * sd, 0.0); hi = min(f + K * sd, 1.0)
            return th_sorted[int(np.floor(lo * (mv - 1)))], th_sorted[int(np.ceil(hi * (mv - 1)))], lo <= 0.0, hi >= 1.0
        a1, a2, a_open_lo, _ = br(1.0)
        b1, b2, _, b_open_hi = br(99.0)
        # concentration slots: the rank order of C_H depends only on the direction of E (level sets parallel to E), and
        # vice versa.  Which percentile is H: the one whose vector has the larger first component (torch_backend.py:439).
        def vec(phi):
            return Vh[:, 0] * np.cos(phi) + Vh[:, 1] * np.sin(phi)
        min_first = vec(0.5 * (a1 + a2))[0] > vec(0.5 * (b1 + b2))[0]
        # HE = [vmin, vmax] if min_first else [vmax, vmin];  C = pinv(HE) od.  In the plane: HE columns at angles (pH, pE)
        # row 0 of pinv ~ direction perpendicular to E, row 1 perpendicular to H (signs so that C > 0 for pixels between)
        st = s_od @ Vh                                  # sample, all pixels (concentrations use every pixel)
        n_eff_all = max(st.shape[0] / unit * neff_per_unit, 8.0)
        f = 0.99
        sd = np.sqrt(f * (1 - f) / n_eff_all)
        q_lo = max(f - K * sd, 0.0)
        def perp_dir(phi_other, phi_self):
            # unit vector u with u . e(phi_other) = 0 and u . e(phi_self) > 0
            u = np.array([-np.sin(phi_other), np.cos(phi_other)])
            if u @ np.array([np.cos(phi_self), np.sin(phi_self)]) < 0:
                u = -u
            return u
        # slot C0: H concentration if min_first (H = vmin at angle in [a1,a2], E = vmax in [b1,b2])
        if min_first:
            ivH, ivE = (a1, a2), (b1, b2)
        else:
            ivH, ivE = (b1, b2), (a1, a2)
        slots = []
        for (iv_other, iv_self) in ((ivE, ivH), (ivH, ivE)):
            dirs = [perp_dir(iv_other[0], 0.5 * (iv_self[0] + iv_self[1])), perp_dir(iv_other[1], 0.5 * (iv_self[0] + iv_self[1]))]
            thr = []
            for u in dirs:
                ks = np.sort(st @ u)
                thr.append(ks[int(np.floor(q_lo * (ks.size - 1)))])
            slots.append((dirs, thr, iv_other))
        # ---------------- pass A ----------------
        t_hat = od @ Vh
        th_all = np.arctan2(t_hat[:, 1], t_hat[:, 0])
        cand_phi1 = keep & (th_all >= a1 if not a_open_lo else True) & (th_all <= a2)
        below1 = int((keep & (th_all < a1)).sum()) if not a_open_lo else 0
        cand_phi2 = keep & (th_all >= b1) & ((th_all <= b2) if not b_open_hi else True)
        below2 = int((keep & (th_all < b1)).sum())
        cand_c = []
        for dirs, thr, _ in slots:
            cand_c.append(((t_hat @ dirs[0]) >= thr[0]) | ((t_hat @ dirs[1]) >= thr[1]))
        # ---------------- resolve (exact) ----------------
        odk = od[keep]
        n = odk.shape[0]
        covx = np.cov(odk.T)
        wx, vx = np.linalg.eigh(covx)
        V = vx[:, [1, 2]] * np.sign(vx[:, [1, 2]].sum(0))
        tk = odk @ V
        th_exact = np.arctan2(tk[:, 1], tk[:, 0])
        k1, k2 = nearest_rank(n, 1.0), nearest_rank(n, 99.0)
        ths = np.sort(th_exact)
        p1, p2 = ths[k1], ths[k2]
        # the same from the candidates
        ok = True
        th_keep_exact = np.full(P, np.nan); th_keep_exact[keep] = th_exact
        for (cand, below, k, ref) in ((cand_phi1, below1, k1, p1), (cand_phi2, below2, k2, p2)):
            c = np.sort(th_keep_exact[cand])
            r = k - below
            if not (0 <= r < c.size) or c[r] != ref:
                ok = False
        tilt = np.abs(nh @ V).max()
        # exact concentrations
        def vex(phi):
            return V[:, 0] * np.cos(phi) + V[:, 1] * np.sin(phi)
        vmin, vmax = vex(p1), vex(p2)
        HE = np.stack([vmin, vmax], 1) if vmin[0] > vmax[0] else np.stack([vmax, vmin], 1)
        ok_order = (vmin[0] > vmax[0]) == min_first
        C = np.linalg.lstsq(HE, od.T, rcond=None)[0]   # (2,P)
        k99 = nearest_rank(P, 99.0)
        fr_c = []
        for s in range(2):
            cs = np.sort(C[s])
            ref = cs[k99]
            cc = np.sort(C[s][cand_c[s]])
            r = k99 - (P - cc.size)
            good = 0 <= r < cc.size and cc[r] == ref
            # rigorous check: every non-candidate is below the answer
            if good:
                good = C[s][~cand_c[s]].max() < cc[r]
            ok = ok and good and ok_order
            fr_c.append(cand_c[s].mean())
        fr = [cand_phi1.mean(), cand_phi2.mean(), fr_c[0], fr_c[1]]
        tot += fr
        fails += 0 if ok else 1
        if verbose:
            print(f"tile {ti:2d}: n_eff {n_eff:6.0f}  cand% phi1 {fr[0]*100:5.2f} phi2 {fr[1]*100:5.2f} C0 {fr[2]*100:5.2f} C1 {fr[3]*100:5.2f}  any {100*np.mean(cand_phi1|cand_phi2|cand_c[0]|cand_c[1]):5.2f}  tilt {tilt:.2e}  {'ok' if ok else 'FAIL'}")
    print(f"unit {unit} n_units {n_units} K {K} neff/unit {neff_per_unit}: mean cand% {tot / n_tiles * 100}  sum {tot.sum() / n_tiles * 100:.2f}%  fails {fails}/{n_tiles}")


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    for (unit, n_units, K, npu) in ((16, 1024, 5.0, 1.0), (16, 1024, 5.0, 2.0), (16, 2048, 5.0, 1.0), (4, 4096, 5.0, 1.0)):
        run(n, unit=unit, n_units=n_units, K=K, neff_per_unit=npu, verbose=(unit == 16 and n_units == 1024 and npu == 1.0))
