#!/usr/bin/env python3
"""How often is a direction chain c_{k+1} = unit_vector(c_k) (float32; Hitable::hit re-normalises at every nesting level,
transform.h:50-54) NOT at a fixed point by c3?  Camera-like directions against Lambertian scatter directions (normal +
point in the unit sphere) and uniform ones.  python tools/chain_modes.py   (CPU, numpy)"""
import numpy as np

rng = np.random.RandomState(0)


def unit(v):
    v = v.astype(np.float32)
    s2 = (v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1]).astype(np.float32) + v[:, 2] * v[:, 2]
    return (v / np.sqrt(s2.astype(np.float32)).astype(np.float32)[:, None]).astype(np.float32)


def eq(a, b):
    return (a.view(np.uint32) == b.view(np.uint32)).all(1)


n = 4000000
nn = unit(rng.normal(size=(n, 3)))
g = rng.normal(size=(n, 3))
cases = {
    "camera-like (|d0| ~ 10)": np.stack([rng.uniform(-6.5, 6.5, n), rng.uniform(-3.6, 3.6, n), np.full(n, -9.95)], 1),
    "Lambertian scatter (normal + point in the unit sphere)": nn + g / np.linalg.norm(g, axis=1)[:, None] * rng.uniform(size=(n, 1)) ** (1 / 3),
    "uniform on the sphere": unit(rng.normal(size=(n, 3))),
}
for name, d in cases.items():
    c1 = unit(d.astype(np.float32)); c2 = unit(c1); c3 = unit(c2); c4 = unit(c3); c5 = unit(c4)
    hot = eq(c4, c3)
    m0 = ~hot & eq(c5, c4)
    m1 = ~hot & ~eq(c5, c4) & eq(c5, c3)
    m2 = ~hot & ~m0 & ~m1
    p = 1.0 - hot.mean()
    print("%-56s c4 != c3: %.5f (a wavefront of 64 holds one: %.3f) | fixed at c4: %.5f | 2-cycle c3,c4: %.5f | neither by c5: %.6f (wavefront: %.4f)"
          % (name, p, 1 - (1 - p) ** 64, m0.mean(), m1.mean(), m2.mean(), 1 - (1 - m2.mean()) ** 64))
