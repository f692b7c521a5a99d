"""TEST INFRASTRUCTURE — numpy restatement of the counter-based RNG the HIP path uses.

The reference draws from numpy's global MT19937 stream (voltage_control_env.py:49,337,384-398,498-508),
which a GPU cannot reproduce bit-for-bit across thousands of envs; the product instead keys
Philox4x32-10 (Salmon et al., SC'11; public algorithm) by (seed, global env id, per-env draw
counter, stream, block) so results do not depend on how envs are sharded over GPUs.  This file
restates exactly the same mapping so oracle and GPU see identical noise / start times / reset
actions.

  key     = (seed & 0xffffffff, seed >> 32)
  counter = (env_id, draw, stream, block)
  streams : 0 pv noise, 1 load-p noise, 2 load-q noise, 3 reset action, 4 start time
  normals : block b -> (x0,x1)->u1 in (0,1), (x2,x3)->u2 in [0,1); z0 = r cos(2 pi u2), z1 = r sin(2 pi u2)
            element j uses block j>>1, z[j&1]
  uniform : element j uses block j>>1, words (x0,x1) if j even else (x2,x3); u = 53-bit / 2^53 in [0,1)
  randint : (x * n) >> 32
"""
from __future__ import annotations

import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)

STREAM_PV, STREAM_LOAD_P, STREAM_LOAD_Q, STREAM_ACTION, STREAM_START = range(5)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over numpy arrays of uint32 counters; returns 4 uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & MASK for c in np.broadcast_arrays(c0, c1, c2, c3))
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0)) & MASK, lo1, (hi0 ^ c3 ^ np.uint64(k1)) & MASK, lo0
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def _u53(hi, lo):
    return ((hi.astype(np.uint64) >> np.uint64(5)) * np.uint64(67108864)
            + (lo.astype(np.uint64) >> np.uint64(6))).astype(np.float64)


def _key(seed):
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return seed & 0xFFFFFFFF, seed >> 32


def normals(seed, env_id, draw, stream, n):
    """n standard normals for (env, draw, stream)."""
    nblk = (n + 1) // 2
    k0, k1 = _key(seed)
    x0, x1, x2, x3 = philox4x32_10(env_id, draw, stream, np.arange(nblk, dtype=np.uint64), k0, k1)
    u1 = (_u53(x0, x1) + 0.5) * (1.0 / 9007199254740992.0)
    u2 = _u53(x2, x3) * (1.0 / 9007199254740992.0)
    r = np.sqrt(-2.0 * np.log(u1))
    ang = 2.0 * np.pi * u2
    z = np.empty(2 * nblk)
    z[0::2] = r * np.cos(ang)
    z[1::2] = r * np.sin(ang)
    return z[:n]


def uniforms(seed, env_id, draw, stream, n):
    nblk = (n + 1) // 2
    k0, k1 = _key(seed)
    x0, x1, x2, x3 = philox4x32_10(env_id, draw, stream, np.arange(nblk, dtype=np.uint64), k0, k1)
    u = np.empty(2 * nblk)
    u[0::2] = _u53(x0, x1) * (1.0 / 9007199254740992.0)
    u[1::2] = _u53(x2, x3) * (1.0 / 9007199254740992.0)
    return u[:n]


def start_time(seed, env_id, draw, n_days, n_intervals):
    """(hour, day, interval) as voltage_control_env.py:111-113 draws them, from one Philox block."""
    k0, k1 = _key(seed)
    x0, x1, x2, _ = philox4x32_10(env_id, draw, STREAM_START, 0, k0, k1)
    hour = (int(x0) * 24) >> 32
    day = (int(x1) * int(n_days)) >> 32
    interval = (int(x2) * int(n_intervals)) >> 32
    return hour, day, interval
