"""ORACLE (test infrastructure): numpy twin of the synthetic-map channel kernel (terrain_diffusion_amd/csrc/compose_kernels.hip:perlin_map_kernel)
and the reference's `finalize` arithmetic (terrain_diffusion/inference/synthetic_map.py:232-252) / quantile transfer (perlin_transform.py:41-45).
The noise VALUES are this package's own (pyfastnoiselite absent: parity unpinned); the twin pins the kernel to its specification."""
import numpy as np

PX, PY = 501125321, 1136930381


def _i32(x):
    return ((np.asarray(x, dtype=np.int64) + 2 ** 31) % 2 ** 32 - 2 ** 31).astype(np.int64)


def _grad(seed, xp, yp, xd, yd):
    h = _i32(_i32(np.int64(seed) ^ xp ^ yp) * 0x27d4eb2d)
    h = h ^ (h >> 15)
    a = (h & 127).astype(np.float32) * np.float32(6.283185307179586 / 128) + np.float32(3.141592653589793 / 128)
    return xd * np.cos(a) + yd * np.sin(a)


def single(seed, x, y):
    x, y = x.astype(np.float32), y.astype(np.float32)
    fx, fy = np.floor(x), np.floor(y)
    xd0, yd0 = x - fx, y - fy
    xd1, yd1 = xd0 - 1, yd0 - 1
    xs = xd0 * xd0 * xd0 * (xd0 * (xd0 * 6 - 15) + 10)
    ys = yd0 * yd0 * yd0 * (yd0 * (yd0 * 6 - 15) + 10)
    x0, y0 = _i32(fx.astype(np.int64) * PX), _i32(fy.astype(np.int64) * PY)
    x1, y1 = _i32(x0 + PX), _i32(y0 + PY)
    a, b = _grad(seed, x0, y0, xd0, yd0), _grad(seed, x1, y0, xd1, yd0)
    c, d = _grad(seed, x0, y1, xd0, yd1), _grad(seed, x1, y1, xd1, yd1)
    xf0, xf1 = a + xs * (b - a), c + xs * (d - c)
    return ((xf0 + ys * (xf1 - xf0)) * np.float32(1.4247691104677813)).astype(np.float32)


def fbm_map(rows, cols, i1, j1, seed, frequency, octaves, lacunarity, gain, src_q, dst_q):
    r, c = np.meshgrid(np.arange(i1, i1 + rows), np.arange(j1, j1 + cols), indexing="ij")
    x, y = (r * np.float32(frequency)).astype(np.float32), (c * np.float32(frequency)).astype(np.float32)
    bound, a = 1.0, abs(gain)
    for _ in range(1, octaves):
        bound += a
        a *= abs(gain)
    amp, total = np.float32(1.0 / bound), np.zeros((rows, cols), np.float32)
    for o in range(octaves):
        total += single(seed + o, x, y) * amp
        x, y, amp = x * np.float32(lacunarity), y * np.float32(lacunarity), amp * np.float32(gain)
    return np.interp(total, src_q, dst_q, left=dst_q[0], right=dst_q[-1]).astype(np.float32)


def finalize(raw, a_temp_std, b_temp_std, temp_std_p1, temp_std_p99):
    """synthetic_map.py:232-252, verbatim arithmetic in numpy."""
    elev, temp, tstd, precip, pcv = (np.asarray(raw[k], np.float32) for k in range(5))
    lapse = (-6.5 + 0.0015 * precip).clip(-9.8, -4.0) / 1000
    temp = np.clip(temp + lapse * np.maximum(0, elev), -10, 40)
    temp = np.where(temp > 20, temp, (temp - 20) * 1.25 + 20)
    t = (tstd - temp_std_p1) / (temp_std_p99 - temp_std_p1)
    baseline = np.maximum(temp_std_p1, -(a_temp_std * temp + b_temp_std))
    tstd = t * (temp_std_p99 - baseline) + baseline
    tstd = np.maximum(tstd + (a_temp_std * temp + b_temp_std), 20)
    pcv = pcv * np.maximum(0, (185 - 0.04111 * precip) / 185)
    return np.stack([elev, temp, tstd, precip, pcv], axis=0)
