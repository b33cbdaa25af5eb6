#!/usr/bin/env python3
"""shade on a bench mesh (C3 | C3B | C5), 4 materials, depth 8: every walk of the generic kernel against the skip links (variant 9),
whole frame, pixels and RNG states.   python tools/shade_variants_diff.py [spp] [mesh]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cudaraytracer_amd import api, scenes, _lib
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 2
mesh = sys.argv[2] if len(sys.argv) > 2 else "C3"
path, nx, ny, _ = scenes.baseline_obj(mesh, tempfile.mkdtemp())
scene = api.Scene.from_obj(path, scale=1.0)
cam = api.init_camera(nx, ny)
mats = [(api.LAMBERTIAN, (0.65, 0.05, 0.05), 0.0), (api.METAL, (0.8, 0.85, 0.88), 0.1), (api.DIELECTRIC, (1, 1, 1), 1.5), (api.DIFFUSE_LIGHT, (4.0, 4.0, 4.0), 0.0)]
st = api.SetCurandState(nx, ny, 0)
api.create_BVHfromList(scene, st)
scene.set_materials(mats, (np.arange(scene.n_triangles) % 4).astype(np.int32))
out = {}
for v in (9, 0, 11, 12):
    st = api.SetCurandState(nx, ny, 0)
    d = torch.zeros(ny * nx * 3, dtype=torch.float32, device="cuda")
    prev = _lib.lib().crt_set_kernel_variant(v)
    api.render_rows_ex(scene, cam, st, nx, ny, spp, 0, ny, d, world=api.WORLD_BVH, shader=api.SHADER_SHADE, max_depth=8, stream=torch.cuda.current_stream())
    torch.cuda.synchronize()
    _lib.lib().crt_set_kernel_variant(prev)
    out[v] = (d.cpu().numpy().view(np.uint32).reshape(ny, nx, 3), st.export())
for v in (0, 11, 12):
    diff = (out[v][0] != out[9][0]).any(axis=2)
    sd = (out[v][1] != out[9][1]).any(axis=1)
    ys, xs = np.nonzero(diff)
    print(mesh, "%dx%dx%d" % (nx, ny, spp), "variant %d against 9: %d pixels differ, %d states differ; first: %s" % (v, diff.sum(), sd.sum(), list(zip(ys[:8].tolist(), xs[:8].tolist()))))
