"""Statistical check of the dropout-mask mixers (emloco_amd/csrc/drop_hash.h): keep rate, correlations between the two halves of a hash,
adjacent keys / queries / heads, chi-square of the 16-bit halves, variance of the kept count per row and column -- the murmur finaliser of
rounds 3-4 (`old`) beside the full-rate candidates; `mixA` is the one shipped.   python tools/exp/hash_quality.py"""
import numpy as np
M32 = np.uint64(0xffffffff)
def u(x): return x & M32
def fmix32(x):
    x = u(x); x ^= x >> np.uint64(16); x = u(x * np.uint64(0x85EBCA6B)); x ^= x >> np.uint64(13); x = u(x * np.uint64(0xC2B2AE35)); x ^= x >> np.uint64(16); return x
def mul24(a, c): return u((a & np.uint64(0xffffff)) * np.uint64(c & 0xffffff))
def mixA5(x, c1=0x9E3779, c2=0x85EBCB):        # round 5's mixer: the key's top byte reaches the first multiply only through bits 8..15
    x = u(x); x ^= x >> np.uint64(16); x = mul24(x, c1); x ^= x >> np.uint64(13); x = mul24(x, c2); x ^= x >> np.uint64(16); return x
def mixA(x, c1=0x9E3779, c2=0x85EBCB):         # shipped (round 6): the key's upper 24 bits also enter behind the first multiply
    x = u(x); hi = x >> np.uint64(8); x ^= x >> np.uint64(16); x = mul24(x, c1); x ^= hi; x ^= x >> np.uint64(13); x = mul24(x, c2); x ^= x >> np.uint64(16); return x
def mixB(x, c1=0xB5297B, c2=0x68E31D):   # fold 15, mul, fold 12, mul, fold 15
    x = u(x); x ^= x >> np.uint64(15); x = mul24(x, c1); x ^= x >> np.uint64(12); x = mul24(x, c2); x ^= x >> np.uint64(15); return x
def mixC(x):   # one mul24 only
    x = u(x); x ^= x >> np.uint64(16); x = mul24(x, 0x9E3779); x ^= x >> np.uint64(15); return x
def old(qk, kp): return fmix32(qk ^ u(kp * np.uint64(0xC2B2AE35)))
def newh(mix):
    return lambda qk, kp: mix(qk ^ mul24(kp, 0xC2B2AF))
def stats(name, h):
    seed = 12345
    nq, nk = 453, 227   # queries, key pairs
    res = []
    for bh in range(16):
        hk = fmix32(np.uint64(seed) ^ u(np.uint64(bh) * np.uint64(0x9E3779B1)))
        q = np.arange(nq, dtype=np.uint64)[:, None]
        kp = np.arange(nk, dtype=np.uint64)[None, :]
        qk = u(hk + q * np.uint64(0x85EBCA6B))
        x = h(qk, kp)
        res.append(x)
    x = np.stack(res)            # [bh][q][kp]
    lo, hi = (x & np.uint64(0xffff)).astype(np.float64), (x >> np.uint64(16)).astype(np.float64)
    thr = 6553
    klo, khi = lo >= thr, hi >= thr
    keep = np.stack([klo, khi], -1).reshape(16, nq, 2 * nk).astype(np.float64)
    m = keep.mean()
    def corr(a, b): a = a - a.mean(); b = b - b.mean(); return (a * b).mean() / np.sqrt((a * a).mean() * (b * b).mean())
    c_pair = corr(klo.astype(float), khi.astype(float))
    c_key = corr(keep[:, :, :-1], keep[:, :, 1:])
    c_key2 = corr(keep[:, :, :-2], keep[:, :, 2:])
    c_q = corr(keep[:, :-1], keep[:, 1:])
    c_bh = corr(keep[:-1], keep[1:])
    # uniformity of the 16-bit values: chi-square over 256 buckets
    hist = np.bincount((lo.ravel() / 256).astype(int), minlength=256); e = lo.size / 256
    chi_lo = ((hist - e) ** 2 / e).sum()
    hist = np.bincount((hi.ravel() / 256).astype(int), minlength=256)
    chi_hi = ((hist - e) ** 2 / e).sum()
    # row sums (per query kept count) variance vs binomial
    rs = keep.sum(-1); var_ratio = rs.var() / (2 * nk * m * (1 - m))
    cs = keep.sum(1); var_ratio_c = cs.var() / (nq * m * (1 - m))
    print(f"{name:8s} keep {m:.5f}  corr pair {c_pair:+.4f} adj-key {c_key:+.4f} key+2 {c_key2:+.4f} adj-query {c_q:+.4f} adj-head {c_bh:+.4f}  chi2/255 lo {chi_lo/255:.2f} hi {chi_hi/255:.2f}  row-var ratio {var_ratio:.3f} col-var ratio {var_ratio_c:.3f}")
stats("old", old)
stats("mixA", newh(mixA))
stats("mixA r05", newh(mixA5))
stats("mixB", newh(mixB))
stats("mixC", newh(mixC))


def drop_keep_stats():
    """drop_keep (the GEMM epilogues' mask over a flat element index): an M x N = 4096 x 1024 output under three seeds"""
    M, N, p = 4096, 1024, 0.1
    thr = int(np.float32(p) * np.float32(65536.0))
    idx = np.arange(M * N, dtype=np.uint64)
    pair = idx >> np.uint64(1)
    for seed in (1, 0x9E3779B1, 123456789):
        rk = u(np.uint64(seed) + mul24(pair >> np.uint64(24), 0x7FEB35))
        lo = pair & np.uint64(0xffffff); lo = lo ^ (lo >> np.uint64(9)) ^ u(lo << np.uint64(11))
        x = mixA(rk ^ mul24(lo, 0xC2B2AF))
        half = np.where((idx & np.uint64(1)) == 1, x >> np.uint64(16), x & np.uint64(0xffff))
        keep = (half >= thr).reshape(M, N).astype(np.float64)
        def corr(a, b): a = a - a.mean(); b = b - b.mean(); return (a * b).mean() / np.sqrt((a * a).mean() * (b * b).mean())
        print(f"drop_keep seed {seed:#x}: keep {keep.mean():.5f}  corr adj-col {corr(keep[:, :-1], keep[:, 1:]):+.4f} col+2 {corr(keep[:, :-2], keep[:, 2:]):+.4f} "
              f"adj-row {corr(keep[:-1], keep[1:]):+.4f}  row-var ratio {keep.sum(1).var() / (N * 0.9 * 0.1):.3f} col-var ratio {keep.sum(0).var() / (M * 0.9 * 0.1):.3f}")
    # two launches with different seeds must be independent
    lo = pair & np.uint64(0xffffff); lo = lo ^ (lo >> np.uint64(9)) ^ u(lo << np.uint64(11))
    a = (np.where((idx & np.uint64(1)) == 1, mixA(u(np.uint64(11) + mul24(pair >> np.uint64(24), 0x7FEB35)) ^ mul24(lo, 0xC2B2AF)) >> np.uint64(16), 0) >= thr)
    b = (np.where((idx & np.uint64(1)) == 1, mixA(u(np.uint64(12) + mul24(pair >> np.uint64(24), 0x7FEB35)) ^ mul24(lo, 0xC2B2AF)) >> np.uint64(16), 0) >= thr)
    a, b = a[1::2].astype(float), b[1::2].astype(float)
    print(f"seeds 11 vs 12 (adjacent seeds, same elements): corr {((a - a.mean()) * (b - b.mean())).mean() / np.sqrt(a.var() * b.var()):+.4f}")


drop_keep_stats()


def row_key_collisions():
    """Two row keys must not share their whole mask.  Round 5's mixer did for keys that differ by (d << 24) | (d << 8): identical hashes
    for EVERY column.  Counted here: over 512 columns, pairs (key, key ^ pattern) whose hashes agree in every column, for the structured
    patterns of the finding and for random key pairs."""
    rng = np.random.default_rng(0)
    cols = np.arange(512, dtype=np.uint64)[None, :]
    keys = rng.integers(0, 1 << 32, size=4096, dtype=np.uint64)[:, None]
    for name, mix in (("mixA (shipped)", mixA), ("mixA r05", mixA5)):
        h = newh(mix)
        same = 0
        for d in (1, 0x5a, 0xff, 0x80, 0x33):
            pat = np.uint64((d << 24) | (d << 8))
            same += int((h(keys, cols) == h(keys ^ pat, cols)).all(axis=1).sum())
        rnd = int((h(keys, cols) == h(np.roll(keys, 1, axis=0), cols)).all(axis=1).sum())
        part = float((h(keys, cols) == h(keys ^ np.uint64((0x5a << 24) | (0x5a << 8)), cols)).mean())
        print(f"{name:16s} structured pairs with identical masks over 512 columns: {same} of {5 * 4096}; random pairs: {rnd}; share of equal hashes, d = 0x5a: {part:.5f}")


row_key_collisions()


def byte_stats(name, h, thr8=26):
    """Round 6: the fused attention's keep decisions are the four BYTES of one hash per four adjacent keys (at_keep_bit), against p 2^8."""
    seed, nq, nk4 = 12345, 453, 114
    res = []
    for bh in range(16):
        hk = fmix32(np.uint64(seed) ^ u(np.uint64(bh) * np.uint64(0x9E3779B1)))
        q = np.arange(nq, dtype=np.uint64)[:, None]
        k4 = np.arange(nk4, dtype=np.uint64)[None, :]
        res.append(h(u(hk + q * np.uint64(0x85EBCA6B)), k4))
    x = np.stack(res)                                             # [bh][q][key quad]
    by = np.stack([(x >> np.uint64(8 * e)) & np.uint64(0xff) for e in range(4)], -1).reshape(16, nq, 4 * nk4)[:, :, :453]
    keep = (by >= thr8).astype(np.float64)
    m = keep.mean()
    def corr(a, b): a = a - a.mean(); b = b - b.mean(); return (a * b).mean() / np.sqrt((a * a).mean() * (b * b).mean())
    chi = [((np.bincount(((x >> np.uint64(8 * e)) & np.uint64(0xff)).ravel().astype(int), minlength=256) - x.size / 256) ** 2 / (x.size / 256)).sum() / 255 for e in range(4)]
    rs = keep.sum(-1); cs = keep.sum(1)
    print(f"{name:8s} bytes, thr {thr8}/256: keep {m:.5f} (nominal {1 - thr8 / 256:.5f}; p = 0.1 asks 0.90000)  corr key+1 {corr(keep[:, :, :-1], keep[:, :, 1:]):+.4f} key+2 {corr(keep[:, :, :-2], keep[:, :, 2:]):+.4f} "
          f"key+4 {corr(keep[:, :, :-4], keep[:, :, 4:]):+.4f} adj-query {corr(keep[:, :-1], keep[:, 1:]):+.4f} adj-head {corr(keep[:-1], keep[1:]):+.4f}  chi2/255 per byte {' '.join(f'{c:.2f}' for c in chi)}  "
          f"var(kept per query) / binomial {rs.var() / (453 * m * (1 - m)):.3f}  per key {cs.var() / (nq * m * (1 - m)):.3f}")


byte_stats("mixA", newh(mixA))
byte_stats("old", old)
