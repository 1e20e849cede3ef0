"""Statistics of the attention-dropout hash of the BiMAU kernels (easydgl_amd/csrc/edgl_common.h: drop_hash_quad), restated in numpy:
one 32-bit multiply / xor-shift hash of the index of the first of four neighbouring (query, key) pairs, widened to 64 bits by one
32 x 32 -> 64 multiply; element r is kept iff its 16-bit field [16r, 16r + 16) is >= rate * 2^16.  Printed per candidate: drop rate
per field, largest correlation between the drop indicators of two fields / of elements `lag` apart (lags 1..32 and the row strides
of T = 101) / of the same elements under two unrelated keys, chi-square of every field over 256 bins.  The noise level of 2^18
samples is 0.002; the candidates built from multiplies alone (no xor-shift round) fail the lag and step tests.
    python tools/dropout_hash_stats.py"""
import numpy as np

MASK, S32 = np.uint64(0xffffffff), np.uint64(32)


def u64(x):
    return np.uint64(x)


def murmur(idx, k0, k1):
    h = ((idx.astype(np.uint64) ^ u64(k0)) & MASK) * u64(0x9E3779B1) + u64(k1)
    h &= MASK
    h ^= h >> u64(15)
    h = (h * u64(0x85ebca6b)) & MASK
    h ^= h >> u64(13)
    return h


def quad(idx, k0, k1, m2=0xFFF1AFD7):          # drop_hash_quad
    return murmur(idx, k0, k1) * u64(m2)


def pair(idx, k0, k1):                          # the paired form (drop_hash_pair): two hashes per four elements
    return murmur(idx, k0, k1) | (murmur(idx + u64(2), k0, k1) << S32)


def two_mads(idx, k0, k1):                      # multiplies only: cheaper, and not good enough
    a = ((idx.astype(np.uint64) ^ u64(k0)) & MASK) * u64(0x9E3779B1) + u64((k1 << 32) | k0)
    y = (a & MASK) ^ (a >> S32)
    return y * u64(0xFF51AFD7) + a


def fields(b):
    return np.stack([(b >> u64(16 * i)) & u64(0xffff) for i in range(4)], 1).astype(np.int64)


def evaluate(name, fn, rate=0.1, n=1 << 18):
    t16 = int(rate * 65536 + 0.5)
    rng = np.random.default_rng(2)
    worst_lag = worst_field = worst_step = 0.0
    chis, rates = [], []
    for _ in range(4):
        k0, k1, base = (int(rng.integers(0, 2 ** 32)) for _ in range(3))
        idx = np.arange(base % (1 << 20), base % (1 << 20) + n * 4, 4, dtype=np.uint64)
        f = fields(fn(idx, k0, k1))
        d = (f < t16).astype(float)
        rates.append(d.mean(0))
        c = np.corrcoef(d.T)
        worst_field = max(worst_field, np.abs(c - np.eye(4)).max())
        flat = d.reshape(-1)
        for lag in list(range(1, 33)) + [100, 101, 102, 104, 112, 202, 404, 808, 11312]:
            worst_lag = max(worst_lag, abs(np.corrcoef(flat[:-lag], flat[lag:])[0, 1]))
        chis.append([((np.bincount(f[:, i] >> 8, minlength=256) - n / 256) ** 2 / (n / 256)).sum() for i in range(4)])
        d2 = (fields(fn(idx, int(rng.integers(0, 2 ** 32)), int(rng.integers(0, 2 ** 32)))) < t16).astype(float)
        worst_step = max(worst_step, max(abs(np.corrcoef(d[:, i], d2[:, i])[0, 1]) for i in range(4)))
    print(f"{name:28s} rates {np.mean(rates, 0).round(5)}  field corr {worst_field:.4f}  lag corr {worst_lag:.4f}  "
          f"key corr {worst_step:.4f}  chi2(255) {np.max(chis, 0).round(0)}")


if __name__ == "__main__":
    evaluate("pair (two hashes)", pair)
    evaluate("quad (hash + one multiply)", quad)
    evaluate("two multiplies, no xor-shift", two_mads)
