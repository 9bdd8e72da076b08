"""VERDICT r3 item 4 -- would a rotation or per-row norms give non-uniform data an INT8-rate filter?  Host-side study (numpy): for the
three non-uniform vector laws of include/hvs_gen.h the candidates a band lets through, per query, for
  (0) no band (what the guessed threshold alone hands over),
  (1) the shipped INT8 bound: one scale, band = |sd qq| E_D + e_q N_D with row MAXIMA E_D, N_D,
  (2) per-row norms: band_r = |sd qq| E_r + e_q N_r (VERDICT's variant b: the Cauchy-Schwarz bound per pair),
  (3) a rotation first: random signs + Walsh-Hadamard transform of the vectors padded 100 -> 128 (exactly orthogonal up to
      rounding; equalises per-dimension ranges), then (1) and (2) on the rotated vectors,
  (4) the FP16 bound: band = |q| E_D + e_q NB_D with half-precision rounding errors.
A row is a candidate when its true squared distance T <= tau + 2 band (tau = the query's k-th smallest distance: the proven
threshold; the inflation over (0) is what the planner's probe measures).  n rows and k are scaled together (k/n = 10^-5 as at
n = 10^7, k = 100 would need 10^7 rows here; the sample keeps k = 100 at n = 10^6: the band's RELATIVE effect is what is compared).
Usage: python scripts/nonuniform_int8_study.py [n]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import hvs_testlib as T

def hadamard128():
    H = np.array([[1.0]])
    while H.shape[0] < 128:
        H = np.block([[H, H], [H, -H]])
    return H / np.sqrt(128.0)

def int8_stats(D, Q):
    """centre, one scale, quantised images; returns per-row (E_r, N_r) and per-query (|sd qq|, e_q)"""
    lo, hi = D.min(0), D.max(0)
    c = 0.5 * (lo + hi)
    sd = np.abs(np.stack([lo - c, hi - c])).max() / 127.0
    Dc, Qc = D - c, Q - c
    Dq = np.clip(np.rint(Dc / sd), -127, 127) * sd
    Qq = np.clip(np.rint(Qc / sd), -127, 127) * sd
    return (np.linalg.norm(Dc - Dq, axis=1), np.linalg.norm(Dc, axis=1), np.linalg.norm(Qq, axis=1), np.linalg.norm(Qc - Qq, axis=1), sd)

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    nq, k = 48, (int(sys.argv[2]) if len(sys.argv) > 2 else 100)
    rng = np.random.default_rng(4)
    signs = np.sign(rng.standard_normal(128))
    H = hadamard128() * signs[None, :]
    print(f"n = {n}, {nq} type-0 queries, k = {k}; candidates per query (mean) with T <= tau + 2 band")
    print(f"{'law':12s} {'no band':>8s} {'int8 max':>9s} {'int8/row':>9s} {'rot max':>9s} {'rot/row':>9s} {'fp16':>8s}   sd, sd(rot), mean band int8 / rot / fp16, tau")
    for name, prof in (("gen-v1", T.GEN_V1), ("clustered", T.GEN_CLUSTER), ("pca-like", T.GEN_PCA), ("heavy-tail", T.GEN_HEAVY)):
        D = T.gen_data_numpy(n, T.SEED_DATA, prof, 100)[:, 2:].astype(np.float64)
        Q = T.gen_queries_numpy(nq, T.SEED_QUERY, prof, 100, 0)[:, 4:].astype(np.float64)
        Q = Q[np.all((Q >= D.min(0)) & (Q <= D.max(0)), axis=1)]      # in-box queries (the out-of-box 1 % pay a clip term on top)
        Dn = (D * D).sum(1)
        Er, Nr, nq8, eq8, sd = int8_stats(D, Q)
        Dr = np.pad(D, ((0, 0), (0, 28))) @ H.T
        Qr = np.pad(Q, ((0, 0), (0, 28))) @ H.T
        Err, Nrr, nq8r, eq8r, sdr = int8_stats(Dr, Qr)
        Dh = D.astype(np.float16).astype(np.float64)
        Qh = Q.astype(np.float16).astype(np.float64)
        Eh, NBh = np.linalg.norm(D - Dh, axis=1).max(), np.linalg.norm(Dh, axis=1).max()
        cnt = np.zeros(6)
        bands = np.zeros(3)
        taus = []
        for i in range(len(Q)):
            Tq = Dn - 2.0 * (D @ Q[i]) + (Q[i] * Q[i]).sum()
            tau = np.partition(Tq, k - 1)[k - 1]
            taus.append(tau)
            b_max = nq8[i] * Er.max() + eq8[i] * Nr.max()
            b_row = nq8[i] * Er + eq8[i] * Nr
            br_max = nq8r[i] * Err.max() + eq8r[i] * Nrr.max()
            br_row = nq8r[i] * Err + eq8r[i] * Nrr
            b_h = np.linalg.norm(Q[i]) * Eh + np.linalg.norm(Q[i] - Qh[i]) * NBh
            cnt += [(Tq <= tau).sum(), (Tq <= tau + 2 * b_max).sum(), (Tq <= tau + 2 * b_row).sum(), (Tq <= tau + 2 * br_max).sum(),
                    (Tq <= tau + 2 * br_row).sum(), (Tq <= tau + 2 * b_h).sum()]
            bands += [b_max, br_max, b_h]
        cnt /= len(Q)
        bands /= len(Q)
        print(f"{name:12s} {cnt[0]:8.0f} {cnt[1]:9.0f} {cnt[2]:9.0f} {cnt[3]:9.0f} {cnt[4]:9.0f} {cnt[5]:8.0f}   "
              f"{sd:.4f}, {sdr:.4f}, {bands[0]:.2f} / {bands[1]:.2f} / {bands[2]:.3f}, {np.mean(taus):.1f}")

if __name__ == "__main__":
    main()
