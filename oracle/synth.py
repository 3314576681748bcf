"""Stateless synthetic inputs (oracle side).  TEST INFRASTRUCTURE ONLY.

Host twin of csrc/synth.hip: element (seed, row, col) is a pure function of its coordinates, built only
from integer arithmetic, exact int->float conversions and single correctly-rounded float32 operations,
so numpy here and the HIP kernel there produce the same bits (tests/test_synth.py checks that on the GPU).

  value(seed,row,col) = (a+b+c+d - 131070) * SCALE     a..d = the four 16-bit fields of a 64-bit hash
                        (Irwin-Hall(4): mean 0, variance 1, |v| < 3.47, close to N(0,1))
  audio(seed,clip,n)  = 0.1*value(seed,clip,n) + 0.3*tri_P(n)   P = 20 + hash(seed,clip,2^40) % 181
                        tri_P(n) = (2*|2*(n mod P) - P| - P) / P   (a per-clip triangle wave, 88..800 Hz at 16 kHz)
"""
import numpy as np

SCALE = np.float32(1.0 / 37837.2271)     # 1/sqrt(4*(65536^2-1)/12)
NOISE_SCALE = np.float32(0.1) * SCALE      # folded on the host in float32, same constant on the device
TRI_AMP = np.float32(0.3)
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _hash(seed, row, col):
    """64-bit hash of (seed,row,col); row/col are uint64 arrays (broadcast)."""
    with np.errstate(over="ignore"):
        r = _splitmix64(row.astype(np.uint64) ^ _splitmix64(np.uint64(seed)))   # seed is hashed first: streams do not alias
        return _splitmix64(r ^ (col.astype(np.uint64) * np.uint64(0x100000001B3)))


def _ih4(h):
    s = (h & np.uint64(0xFFFF)) + ((h >> np.uint64(16)) & np.uint64(0xFFFF)) \
        + ((h >> np.uint64(32)) & np.uint64(0xFFFF)) + (h >> np.uint64(48))
    return s.astype(np.int64) - 131070


def rows(row0, n_rows, dim, seed):
    """float32 [n_rows, dim] block of the synthetic reference store starting at row0."""
    r = np.arange(row0, row0 + n_rows, dtype=np.uint64)[:, None]
    c = np.arange(dim, dtype=np.uint64)[None, :]
    return (_ih4(_hash(seed, r, c)).astype(np.float32) * SCALE).astype(np.float32)


def audio(clip0, n_clips, samples, seed):
    """float32 [n_clips, samples] synthetic 16 kHz clips."""
    c = np.arange(clip0, clip0 + n_clips, dtype=np.uint64)[:, None]
    n = np.arange(samples, dtype=np.uint64)[None, :]
    noise = _ih4(_hash(seed, c, n)).astype(np.float32) * NOISE_SCALE
    period = (20 + (_hash(seed, c, np.uint64(1 << 40) + np.zeros_like(c)) % np.uint64(181))).astype(np.int64)
    ph = n.astype(np.int64) % period
    num = (2 * np.abs(2 * ph - period) - period).astype(np.float32)
    tri = num / period.astype(np.float32)
    return (noise + TRI_AMP * tri).astype(np.float32)


def param(shape, seed, scale):
    """Deterministic stand-in for a trained parameter: hash-normal values * float32(scale)."""
    shape = tuple(int(s) for s in shape)
    r, c = (shape[0], int(np.prod(shape[1:]))) if len(shape) > 1 else (1, shape[0])
    return (rows(0, r, c, seed) * np.float32(scale)).astype(np.float32).reshape(shape)


def fill_state_dict(shapes, seed):
    """shapes: ordered {name: shape}.  Linear/LayerNorm weights ~ N(0, 1/fan_in) (LayerNorm gain around 1),
    biases ~ N(0, 0.05^2).  Used to give the reference's modules and this build's the SAME weights without
    storing them in the golden files."""
    out = {}
    for i, (name, shape) in enumerate(shapes.items()):
        shape = tuple(shape)
        if len(shape) >= 2:
            out[name] = param(shape, seed + i, 1.0 / np.sqrt(shape[-1]))
        elif name.endswith("weight"):          # LayerNorm / BatchNorm gain
            out[name] = (np.float32(1.0) + param(shape, seed + i, 0.1)).astype(np.float32)
        elif name.endswith("running_var"):
            out[name] = (np.float32(1.0) + np.abs(param(shape, seed + i, 0.1))).astype(np.float32)
        elif name.endswith("num_batches_tracked"):
            out[name] = np.zeros(shape, np.int64)
        else:
            out[name] = param(shape, seed + i, 0.05)
    return out
