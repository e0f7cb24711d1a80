"""Counter-based RNG spec shared by the oracle and the HIP kernels.

The reference draws all MCMC noise from NumPy's global Mersenne-Twister on the
host (``models/causalbgm/base.py:842,862,870``) and from ``tf.random.normal``
(``:704-706,753-755``); neither stream can be reproduced on a GPU.  The build
therefore fixes its *own* stream -- Philox4x32-10 (Salmon et al., SC'11,
"Parallel random numbers: as easy as 1, 2, 3") keyed by the user seed and
counted by (row, iteration, call, purpose) -- and this file is its CPU
restatement, so that a HIP chain and an oracle chain fed the same seed see the
same proposals / uniforms up to transcendental-function rounding.

Counter layout  ctr = (row, iteration, call, purpose),  key = (seed_lo, seed_hi)

purpose tags
  0  initial state      z0      ~ N(0,1)            (base.py:842)
  1  MH proposal noise  eps     ~ N(0,1)            (base.py:862)
  2  MH accept uniform  u       ~ U(0,1)            (base.py:870); u(it) = word (it & 3)
                                                    of the call with iteration field it >> 2
  3  outcome noise      eps_y   ~ N(0,1)            (base.py:704-706,753-755)
  4  HMC momentum, 5 HMC accept uniform, 6 posterior-predictive noise (BGM)

Normal for "feature" f of a row:  g = f & 3, s = f >> 2,
  call = g + 4*(s >> 2), element e = s & 3 of the 4 Box-Muller outputs of that
  call (outputs 0,1 from words 0,1; outputs 2,3 from words 2,3).
This is the layout in which one 16x16x4 MFMA lane (row = lane & 15,
g = lane >> 4) consumes all four outputs of one Philox call.
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)

TAG_INIT, TAG_PROP, TAG_ACC, TAG_YNOISE, TAG_MOM, TAG_HACC, TAG_XNOISE = range(7)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10.  Inputs broadcastable uint32 arrays / ints.
    Returns 4 uint32 arrays."""
    c0, c1, c2, c3 = np.broadcast_arrays(
        *[np.asarray(c, dtype=np.uint64) & MASK for c in (c0, c1, c2, c3)])
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0)), lo1, (hi0 ^ c3 ^ np.uint64(k1)), lo0
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def u01_open(x):
    """uint32 -> float32 in (0,1):  ((x >> 8) + 0.5) * 2^-24  (exact in fp32)."""
    return ((x >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24)


def u01_half(x):
    """uint32 -> float32 in [0,1):  (x >> 8) * 2^-24."""
    return (x >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)


def box_muller4(x0, x1, x2, x3):
    """4 uint32 words -> 4 standard normals (float32), matching the device code:
    r = sqrt(-2 ln u1), (r cos 2pi u2, r sin 2pi u2)."""
    outs = []
    for a, b in ((x0, x1), (x2, x3)):
        u1 = u01_open(a).astype(np.float64)
        u2 = u01_half(b).astype(np.float64)
        r = np.sqrt(-2.0 * np.log(u1))
        outs.append((r * np.cos(2 * np.pi * u2)).astype(np.float32))
        outs.append((r * np.sin(2 * np.pi * u2)).astype(np.float32))
    return outs


def normals(rows, iteration, n_feat, tag, seed):
    """[len(rows) x n_feat] float32 standard normals of the spec above."""
    rows = np.asarray(rows, dtype=np.uint32)
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    out = np.empty((rows.shape[0], n_feat), dtype=np.float32)
    n_s = (n_feat + 3) // 4
    for g in range(min(4, n_feat)):
        for sblk in range((n_s + 3) // 4):
            call = g + 4 * sblk
            bm = box_muller4(*philox4x32_10(rows, iteration, call, tag, k0, k1))
            for e in range(4):
                f = 4 * (4 * sblk + e) + g
                if f < n_feat:
                    out[:, f] = bm[e]
    return out


def uniforms(rows, iteration, tag, seed, call=0):
    """[len(rows)] float32 uniforms in (0,1) for `iteration`: word (iteration & 3) of
    Philox(row, iteration >> 2, call, tag) -- one Philox call serves four iterations."""
    rows = np.asarray(rows, dtype=np.uint32)
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    words = philox4x32_10(rows, int(iteration) >> 2, call, tag, k0, k1)
    return u01_open(words[int(iteration) & 3])


def normals_seq(rows, iteration, n_feat, tag, seed):
    """[len(rows) x n_feat] normals in *sequential* layout: feature k is Box-Muller
    output (k & 3) of call (k >> 2).  Used for the outcome noise (tag 3) where
    every lane of a row evaluates the same call."""
    rows = np.asarray(rows, dtype=np.uint32)
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    out = np.empty((rows.shape[0], n_feat), dtype=np.float32)
    for call in range((n_feat + 3) // 4):
        bm = box_muller4(*philox4x32_10(rows, iteration, call, tag, k0, k1))
        for e in range(4):
            k = 4 * call + e
            if k < n_feat:
                out[:, k] = bm[e]
    return out
