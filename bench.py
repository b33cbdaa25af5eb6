#!/usr/bin/env python3
"""bench.py — headline benchmark of the list-render path on MI355X.

Metric (BASELINE.json): Mrays/s at 1920x1080x64 spp on a 100 k-triangle synthetic OBJ mesh
(configs[2]); rays = primary rays = nx*ny*spp (LambertShade never traces secondary rays).
One "step" = one frame: the render kernel over the whole image (row bands over N GPUs, one
process per GPU) + the gather of the float3 framebuffer to rank 0.  Scene, camera and RNG
states are resident in HBM before the timed region; the RNG streams simply continue from
frame to frame like the reference's frame loop (core/render.h:174-186).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `roofline` is the HBM roofline BASELINE.json asks for
(algorithmic bytes of the triangle stream per launch / measured kernel time / 8 TB/s);
`cpu_baseline` is the CPU oracle (oracle/, a port — the reference cannot be built here)
timed on a bounded crop of the same frame on this host.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FRAME = "C3"                                # cudaraytracer_amd.scenes.BASELINE_FRAMES: the definition the tests share
NX, NY, SPP = 1920, 1080, 64
HBM_PEAK_GBS = 8000.0                       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
REC_BYTES = 64                              # derived triangle record (csrc/crt_device.hpp)
CLS_BYTES = 80                              # classification record: 5 float4 planes
BASE_BYTES = 48                             # per-triangle input of tri_prepass: 12 floats (csrc/crt_device.hpp kBaseFloats)


def build_scene_obj(tmpdir):
    """The timed frame = scenes.baseline_obj("C3"): the frame tests/test_gpu_configs.py checks
    against the brute-force walk and the oracle at the full 64 spp."""
    from cudaraytracer_amd import scenes
    path, nx, ny, spp = scenes.baseline_obj(FRAME, tmpdir)
    assert (nx, ny, spp) == (NX, NY, SPP)
    return path


def algorithmic_bytes_brute(nx, rows, spp, n_tris):
    """Brute-force variant: every 64-lane wavefront (8x8 pixel tile) reads each 64-B triangle
    record once per sample, plus the float3 store and the XORWOW state load/store (24 B each
    way) per pixel.  DESIGN.md §5."""
    tiles = ((nx + 7) // 8) * ((rows + 7) // 8)
    px = nx * rows
    return tiles * spp * n_tris * REC_BYTES + px * 12 + px * 48


def algorithmic_bytes_tiled(nx, rows, spp, n_tris, base_bytes, sum2):
    """Tiled variant.  binning (five small kernels, lane = triangle): tri_prepass
    reads the base record and writes the 64-B exact record, cls_prepass reads it and writes the
    80-B classification record, the count and the fill pass of bin_tri_kernel read that once
    each; every list entry is written once (4 B), every tile has a count and a descriptor (12 B).
    shade kernel: every tile reads, once, the list entry (4 B), the 64-B classification forms and the
    64-B exact record of each of its survivors (staged in LDS for all samples), plus the float3
    store (12 B) and the XORWOW state load/store (48 B) per pixel."""
    px = nx * rows
    tiles = ((nx + 7) // 8) * ((rows + 7) // 8)
    bin_b = n_tris * (base_bytes + REC_BYTES + REC_BYTES + CLS_BYTES + 2 * CLS_BYTES) + sum2 * 4 + tiles * 12
    shade_b = sum2 * (4 + 64 + REC_BYTES) + tiles * 8 + px * 12 + px * 48
    return bin_b, shade_b


def host_cpu():
    """Model string and usable core count of this host (state the core model and count)."""
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return model, n


CPU_SHARE = 16      # a one-GPU box of this pool gives a job 16 of the host's cores, whatever the affinity mask shows


def cpu_baseline(obj_path, reps=3):
    """Oracle (serial CPU port of the reference loop) on a crop of the same frame, SURVEY §8d's protocol: the
    single-thread figure (= the reference's serial loop; median of `reps` repetitions) and, labelled as such, an
    all-cores figure (the crop's rows dealt to one thread per usable core; ctypes releases the GIL)."""
    import threading
    from oracle import pyoracle as po
    pts, idx = po.parse_obj(obj_path)
    w = po.World(pts, idx, None, 1)
    cam = po.make_camera((0, 100, 1000), (0, 150, 0), (0, 1, 0), 40.0,
                         np.float32(NX) / np.float32(NY), 0.0, 10.0)
    cw, ch, cs = 32, 12, 4                  # 32x12 px x 4 spp = 1536 rays x 100 k triangles
    r0, c0 = 500, NX // 2 - 250             # on the mesh, near its left silhouette (the mesh covers rows ~20..620)
    win = (r0, r0 + ch, c0, c0 + cw)
    po.render(w, cam, NX, NY, 1, window=(r0, r0 + 1, c0, c0 + 2))      # warm caches / tables
    model, n_visible = host_cpu()
    n_cores = min(n_visible, CPU_SHARE)
    times, ref = [], None
    for _ in range(reps):
        states = po.fresh_states(NX, NY, window=win)                   # outside the timed part, like the GPU's SetCurandState
        t0 = time.perf_counter()
        ref, _ = po.render(w, cam, NX, NY, cs, window=win, states=states)
        times.append(time.perf_counter() - t0)
    dt = float(np.median(times))
    rays = cw * ch * cs
    # all cores: 4 rows of the same 32-pixel-wide window per thread, every thread with its own state array (get_ray
    # draws its lens sample from stream 0 for every pixel, core/render.h:118 — private copies keep the threads apart;
    # untouched pages of the zero-filled arrays are never committed)
    rows_per = 4
    wins = [(r0 + rows_per * k, r0 + rows_per * (k + 1), c0, c0 + cw) for k in range(n_cores)]
    all_times = []
    for _ in range(reps):
        sts = [po.fresh_states(NX, NY, window=wk) for wk in wins]
        ths = [threading.Thread(target=po.render, args=(w, cam, NX, NY, cs), kwargs={"window": wk, "states": sk}) for wk, sk in zip(wins, sts)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        all_times.append(time.perf_counter() - t0)
    dt_all = float(np.median(all_times))
    rays_all = cw * rows_per * cs * n_cores
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "port",
            "cpu_model": model, "host_cores_visible": n_visible, "host_cores": n_cores, "repetitions": reps, "seconds": [round(x, 3) for x in times],
            "sample": "oracle/crt_oracle.c (gcc -O2, 1 thread), %dx%d px crop (rows %d.., columns %d.., on the mesh) x %d spp "
                      "of the 1920x1080 frame against all %d triangles: %d rays, median of %d repetitions %.1f s"
                      % (cw, ch, r0, c0, cs, w.n_tris, rays, reps, dt),
            "all_cores": {"value": rays_all / dt_all / 1e6, "unit": "Mrays/s", "cores": n_cores,
                          "seconds": [round(x, 3) for x in all_times],
                          "sample": "the same oracle, %d threads (the job's CPU share: min(visible cores, 16)) x %d rows of the same %d-pixel-wide window "
                                    "x %d spp: %d rays, median of %d repetitions %.1f s (row-parallel; NOT the reference's serial loop)"
                                    % (n_cores, rows_per, cw, cs, rays_all, reps, dt_all)}}, win, cs, ref[r0:r0 + ch, c0:c0 + cw]


def hip_crop(scene, camera, win, spp):
    """The same crop through the product: fresh RNG states of the crop's rows (seed 0, stream =
    global pixel index), crt_render_rows at `spp`, columns cut on the host."""
    import torch
    from cudaraytracer_amd import api
    r0, r1, c0, c1 = win
    st = api.SetCurandState(NX, NY, 0, r0, r1 - r0)
    d = torch.zeros((r1 - r0) * NX * 3, dtype=torch.float32, device="cuda")
    api.render_rows(scene, camera, st, NX, NY, spp, r0, r1 - r0, d, stream=torch.cuda.current_stream())
    torch.cuda.synchronize()
    return d.cpu().numpy().reshape(r1 - r0, NX, 3)[:, c0:c1]


def timed_frames(frame, n, warm_s=0.2):
    """ms per call of `frame` (which only enqueues): untimed calls for >= warm_s first — building a scene on the host in between
    lets the shader clock drop, and ten frames are over before it is back — then n calls between two synchronisations."""
    import torch
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < warm_s:
        for _ in range(5):
            frame()
        torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(n):
        frame()
    torch.cuda.synchronize()
    return (time.perf_counter() - t1) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--spp", type=int, default=SPP, help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu-baseline", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--variant", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--keep-timing-events", action="store_true", help=argparse.SUPPRESS)   # A/B: the per-frame HIP events stay on in the timed region
    ap.add_argument("--no-extras", action="store_true", help=argparse.SUPPRESS)   # profiling: only launches of the timed frame
    ap.add_argument("--gather", choices=("c", "torch"), default="c", help=argparse.SUPPRESS)   # N > 1: the C ABI's exchange (crt_gather_strips over an ncclComm_t
                                                                                             # made with crt_comm_*) or torch.distributed.gather (fallback)
    ap.add_argument("--no-overlap", action="store_true", help=argparse.SUPPRESS)             # N > 1, C path: the exchange on the render's own stream
                                                                                             # (default: on a stream of its own, frame k's exchange under frame k + 1's kernels)
    ap.add_argument("--clock-warm-s", type=float, default=0.3, help=argparse.SUPPRESS)        # untimed frames before the timed passes (clock ramp)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from cudaraytracer_amd import _lib, api
    from cudaraytracer_amd.dist import BandRenderer

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)"
                         % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no HIP device — the product has no CPU fallback")
    # Rehearsal hook (tests only): CRT_BENCH_SHARE_GPU=1 lets several ranks share the one GPU of a
    # test box over gloo; the driver's runs use one GPU per rank over RCCL ("nccl").
    share = os.environ.get("CRT_BENCH_SHARE_GPU") == "1"
    dev = local_rank % torch.cuda.device_count() if share else local_rank
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))

    if not os.path.exists(_lib.library_path()):
        if local_rank == 0:
            _lib.build_library()
        if world > 1:
            dist.barrier()
    _lib.lib().crt_set_kernel_variant(args.variant)

    tmpdir = tempfile.mkdtemp(prefix="crt_bench_%d_" % rank)
    obj_path = build_scene_obj(tmpdir)
    scene = api.Scene.from_obj(obj_path, scale=1.0)
    camera = api.init_camera(NX, NY)                       # the reference's live camera
    n_tris = scene.n_triangles
    # N > 1 over RCCL: the exchange step is the C ABI's (crt_gather_strips: one grouped ncclSend/ncclRecv + the reassembly
    # kernel) on a communicator made through the C ABI as well (crt_comm_unique_id on rank 0, broadcast over the process
    # group that torchrun set up, crt_comm_init_rank everywhere); gloo rehearsals and --gather torch use dist.gather.
    gather = "c" if (world > 1 and not share and args.gather == "c") else "torch"
    gather_note = None
    stream = torch.cuda.current_stream()
    br = None
    if gather == "c":
        # The C path is tried first — communicator, then one whole step — and every rank learns whether it worked everywhere;
        # if not (an RCCL build the library's dlopen does not find, an IPC restriction of the node), ALL ranks go on with
        # torch.distributed.gather and the line says so, rather than the N-GPU run dying without a number.
        ok, why = 1, ""
        try:
            br = BandRenderer(scene, camera, NX, NY, rank, world, seed=0, gather="c", overlap=not args.no_overlap)
            br.render(args.spp, stream=stream)
            br.gather()
            torch.cuda.synchronize()
        except Exception as e:   # noqa: BLE001 — whatever went wrong, the fallback decision is collective
            ok, why = 0, "%s: %s" % (type(e).__name__, e)
        flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if br is not None:
                try:
                    br.close()
                except Exception:   # noqa: BLE001
                    pass
            br, gather = None, "torch"
            gather_note = "C-path exchange failed on some rank (%s): torch.distributed.gather instead" % (why or "another rank")
    if br is None:
        br = BandRenderer(scene, camera, NX, NY, rank, world, seed=0, gather=gather)

    def step():
        br.render(args.spp, stream=stream)
        full = br.gather()
        return full

    def fence():
        br.wait(stream)                                    # overlapped exchange: the launch stream behind the last one, so the barrier's
        if world > 1:                                      # collective is enqueued after every exchange of this rank, as without overlap
            dist.barrier()
        torch.cuda.synchronize()

    if not args.keep_timing_events:
        _lib.lib().crt_set_kernel_timing(0)                # no one reads kernel times of the timed frames: no event markers between their kernels
    # Clock ramp: a fresh box needs a few hundred milliseconds of work before the shader clock settles, and the timed
    # region is 15-40 ms.  Untimed frames for >= clock_warm_s first (reported as clock_warm_ms; `warmup` stays what the
    # contract says: W untimed steps right before the timed ones).
    t_w = time.perf_counter()
    warm_frames = 0
    while time.perf_counter() - t_w < args.clock_warm_s:
        for _ in range(20):
            step()
        fence()
        warm_frames += 20
    clock_warm_ms = (time.perf_counter() - t_w) * 1e3
    # The K steps are timed THREE times (each pass: W warm-up steps, fence, exactly K steps, fence); ms_per_step is the
    # median pass, all three are reported.
    passes = []
    for _ in range(3):
        for _ in range(args.warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()                                         # enqueue only: no host synchronisation inside the timed region
        fence()
        passes.append(time.perf_counter() - t0)
    _lib.lib().crt_set_kernel_timing(1)

    # per-kernel durations (HIP events on the launch stream; reading them synchronises, so these
    # frames are outside the timed region)
    kernel_ms = []
    for _ in range(min(args.steps, 5)):
        br.render(args.spp, stream=stream)
        kernel_ms.append(api.last_kernel_times())

    t = torch.tensor(passes, dtype=torch.float64, device="cpu" if (world > 1 and share) else "cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)          # per pass: the slowest rank
    passes = [float(x) for x in t.tolist()]
    elapsed = float(np.median(passes))

    # survivor statistics of this rank's rows (one extra, untimed frame of the same shape, so
    # that every launch a profiler sees is alike): they define the algorithmic bytes of the
    # tiled kernels
    api.tiled_stats(True)
    br.render(args.spp, stream=stream)
    torch.cuda.synchronize()
    st = api.tiled_stats(False)
    tiled = st[4] > 0

    # The headline depends on what the frame shows (the tiled kernels skip what no ray can hit;
    # the brute-force walk did not): the same mesh seen from 300 units fills the whole screen.
    filling = None
    if rank == 0 and world == 1 and tiled and not args.no_extras:
        cam_near = api.create_camera(NX, NY, (0.0, 0.0, 300.0), (0.0, 0.0, 0.0), 10.0, 0.0, 40)
        st_near = api.SetCurandState(NX, NY, 0)
        buf = torch.empty(NX * NY * 3, dtype=torch.float32, device="cuda")
        api.tiled_stats(True)
        api.render_rows(scene, cam_near, st_near, NX, NY, args.spp, 0, NY, buf, stream=stream)
        torch.cuda.synchronize()
        st_n = api.tiled_stats(False)
        ms_near = timed_frames(lambda: api.render_rows(scene, cam_near, st_near, NX, NY, args.spp, 0, NY, buf, stream=stream), 10)
        filling = {"camera": "lookfrom (0,0,300) lookat (0,0,0): the mesh covers every pixel",
                   "ms_per_frame": ms_near, "Mrays/s": NX * NY * args.spp / ms_near / 1e3,
                   "tiles_with_geometry_frac": 1.0 - st_n[8] / max(st_n[4], 1),
                   "tile_survivors_sum": st_n[5], "tile_survivors_max": st_n[6]}

    # SURVEY §8d's OTHER C3 mesh (small_bunny midpoint-subdivided twice on a ground grid, 100 000 triangles, no pole fan):
    # the same configuration, the same camera, the same parity bar (tests/test_gpu_configs.py: C3B); reported beside
    # the headline, never as it.
    second = None
    if rank == 0 and world == 1 and tiled and not args.no_extras:
        from cudaraytracer_amd import scenes
        path_b, nxb, nyb, _ = scenes.baseline_obj("C3B", tmpdir)
        scene_b = api.Scene.from_obj(path_b, scale=1.0)
        st_bb = api.SetCurandState(nxb, nyb, 0)
        buf = torch.empty(nxb * nyb * 3, dtype=torch.float32, device="cuda")
        api.tiled_stats(True)
        api.render_rows(scene_b, camera, st_bb, nxb, nyb, args.spp, 0, nyb, buf, stream=stream)
        torch.cuda.synchronize()
        st_s = api.tiled_stats(False)
        ms_2 = timed_frames(lambda: api.render_rows(scene_b, camera, st_bb, nxb, nyb, args.spp, 0, nyb, buf, stream=stream), 20)
        second = {"workload": "BASELINE configs[2] on the other synthetic mesh SURVEY names: small_bunny.obj midpoint-subdivided twice (79 488 "
                              "triangles) on a 641 x 16 ground grid (20 512) = %d triangles, no pole; triangle list, reference camera"
                              % scene_b.n_triangles,
                  "ms_per_frame": ms_2, "Mrays/s": nxb * nyb * args.spp / ms_2 / 1e3,
                  "tiles_with_geometry_frac": 1.0 - st_s[8] / max(st_s[4], 1), "tile_survivors_sum": st_s[5], "tile_survivors_max": st_s[6]}
        scene_b.close()

    # The same frame through the BVH world (the reference's shipped configuration, kernel.cu:97): not the
    # headline metric (north_star names the list path), reported beside it.
    bvh_world = None
    if rank == 0 and world == 1 and tiled and not args.no_extras:
        st_b = api.SetCurandState(NX, NY, 0)
        t1 = time.perf_counter()
        api.create_BVHfromList(scene, st_b)
        build_s = time.perf_counter() - t1
        buf = torch.empty(NX * NY * 3, dtype=torch.float32, device="cuda")
        ms_b = timed_frames(lambda: api.render_rows(scene, camera, st_b, NX, NY, args.spp, 0, NY, buf, stream=stream, bvh=True), 20)
        bvh_world = {"what": "same frame, world = BVHNode tree (crt_render_rows_bvh): tiled kernels + chain of boxes per winner",
                     "parity": "tree shape pinned against rocThrust's thrust::sort run inside one device thread; AABB::hit, box arithmetic and "
                               "the triangle test pinned against the reference's own headers on this GPU (tests/test_ref_thrust.py, "
                               "tests/test_ref_hit.py); BVHNode's merge rule and the render loop: in-repo oracle only (UNPINNED)",
                     "ms_per_frame": ms_b, "Mrays/s": NX * NY * args.spp / ms_b / 1e3, "host_build_s": build_s}

    # the recursive `shade` (core/render.h:48-67) with Lambertian / Metal / Dielectric / DiffuseLight dealt over
    # the mesh in bands of 2 500 triangles, depth 8, BVH world, 4 samples per pixel (the workload of tools/shade_time.py and
    # profiles/r3_f): the generic kernel.  Reported beside the headline, never as it.
    f2_shade = None
    if rank == 0 and world == 1 and tiled and not args.no_extras:
        scene_f = api.Scene.from_obj(obj_path, scale=1.0)
        st_f = api.SetCurandState(NX, NY, 0)
        api.create_BVHfromList(scene_f, st_f)
        mats = [(api.LAMBERTIAN, (0.65, 0.05, 0.05), 0.0), (api.METAL, (0.8, 0.85, 0.88), 0.1), (api.DIELECTRIC, (1, 1, 1), 1.5),
                (api.DIFFUSE_LIGHT, (4.0, 4.0, 4.0), 0.0)]
        scene_f.set_materials(mats, (np.arange(scene_f.n_triangles) // 2500 % 4).astype(np.int32))
        buf = torch.empty(NX * NY * 3, dtype=torch.float32, device="cuda")
        f_spp, f_depth = 4, 8
        ms_f = timed_frames(lambda: api.render_rows_ex(scene_f, camera, st_f, NX, NY, f_spp, 0, NY, buf, world=api.WORLD_BVH, shader=api.SHADER_SHADE,
                                                       max_depth=f_depth, stream=stream), 10)
        f2_shade = {"what": "recursive shade, depth %d, 4 material classes in bands, BVH world, %d spp, same mesh and camera: render_generic_kernel "
                            "(persistent wavefronts, pair walk shared by the wavefront)" % (f_depth, f_spp),
                    "parity": "PARITY UNPINNED (in-repo oracle; pow(x, 5.0f) and the tree's shape by stated rules)",
                    "ms_per_frame": ms_f, "M primary rays/s": NX * NY * f_spp / ms_f / 1e3}
        scene_f.close()

    # The reference's OWN default workload (kernel.cu:44-51,97; render.h:191-237): 1024x512, 4 spp, max_depth 8 (dead under the
    # live LambertShade, render.h:120), world = the BVHNode tree, frames 0..30, every frame: pose update, Update_BVH (refit),
    # render + synchronise + copy back — through the same calls crt_driver makes (api.renderBVH mirrors kernel.cu:24-39), with
    # the rows of the reference's timing table (frame, rendering, update, build; Loader/CSVWriter.h).  The reference's mesh
    # (low_walking.fbx) needs the FBX SDK: the bench mesh stands in, its pose breathing by +-2 % over the 31 frames.
    reference_default = None
    if rank == 0 and world == 1 and not args.no_extras:
        rnx, rny, rspp, rdepth, f0, f1 = 1024, 512, 4, 8, 0, 30
        pts, idx = api.parse_obj(obj_path)
        scene_r = api.Scene.from_mesh(pts, idx, None, 1)
        cam_r = api.init_camera(rnx, rny)
        st_r = api.SetCurandState(rnx, rny, 0)
        poses = {f: (pts * np.float32(1.0 + 0.02 * np.sin(2.0 * np.pi * f / 31.0))).astype(np.float32) for f in range(f0, f1 + 1)}
        data = [["frame", "rendering", "update", "build"]]
        api.renderBVH(rnx, rny, rspp, rdepth, f0, f1, scene_r, cam_r, st_r, data, write_png=False, poses=poses)
        rows = data[2:]
        ren = np.array([float(r[1]) for r in rows])
        upd = np.array([float(r[2]) for r in rows])
        csv_path = os.path.join(tmpdir, "output.csv")
        api.writeCSV(csv_path, data)
        reference_default = {
            "what": "the reference's default run (kernel.cu:44-51,97): %dx%d, %d spp, max_depth %d, BVH world, LambertShade, frames %d..%d, "
                    "pose update + refit + render + sync + D2H per frame (renderBVHAnimation, render.h:191-237), %d triangles"
                    % (rnx, rny, rspp, rdepth, f0, f1, scene_r.n_triangles),
            "parity": "tree shape and hit path pinned against rocThrust's thrust::sort and the reference's own headers "
                      "(tests/test_ref_thrust.py, tests/test_ref_hit.py); render loop / RNG: in-repo oracle (PARITY UNPINNED)",
            "frames": len(rows), "build_s": float(data[1][3]),
            "rendering_ms_mean": float(ren.mean() * 1e3), "rendering_ms_median": float(np.median(ren) * 1e3),
            "rendering_ms_first": float(ren[0] * 1e3),
            "update_ms_mean": float(upd.mean() * 1e3), "update_ms_median": float(np.median(upd) * 1e3),
            "update_ms_first": float(upd[0] * 1e3), "update_ms_max": float(upd.max() * 1e3),
            "Mrays/s_per_frame_median": rnx * rny * rspp / float(np.median(ren)) / 1e6,
            "csv_bytes": os.path.getsize(csv_path),
            "note": "host wall clock per frame as the reference's StopWatch brackets it: 'rendering' includes the synchronise and the "
                    "6.3 MB device-to-host copy of the frame, 'update' the refit and its synchronise"}
        scene_r.close()

    # The one-call surface crt_render(width, height, samples, list, camera) -> HOST RGB buffer: states restored from
    # the seed, the frame, 24.9 MB over PCIe into a pageable buffer.  Reported beside the resident figure, never as it.
    pcie = None
    if rank == 0 and world == 1 and not args.no_extras:
        api.render(NX, NY, args.spp, scene, camera, seed=0)
        t1 = time.perf_counter()
        for _ in range(5):
            api.render(NX, NY, args.spp, scene, camera, seed=0)
        ms_p = (time.perf_counter() - t1) / 5 * 1e3
        pinned = torch.empty(NX * NY * 3, dtype=torch.float32, pin_memory=True)
        api.render(NX, NY, args.spp, scene, camera, seed=0, out=pinned)
        t1 = time.perf_counter()
        for _ in range(5):
            api.render(NX, NY, args.spp, scene, camera, seed=0, out=pinned)
        ms_pin = (time.perf_counter() - t1) / 5 * 1e3
        pcie = {"what": "crt_render: RNG states restored for the seed + the frame + D2H copy of the 24.9 MB float3 frame into a pageable "
                        "host buffer (a fresh numpy array per call), per call (host wall clock, 5 calls)", "ms_per_call": ms_p, "Mrays/s": NX * NY * args.spp / ms_p / 1e3,
                "page_locked_buffer": {"what": "the same call into the caller's page-locked buffer, reused", "ms_per_call": ms_pin,
                                       "Mrays/s": NX * NY * args.spp / ms_pin / 1e3}}
        api.stream_release(None)

    if rank == 0:
        rays = NX * NY * args.spp * args.steps
        value = rays / elapsed / 1e6
        k_tot, k_bin, k_shade = [float(x) for x in np.mean(np.array(kernel_ms), axis=0)]
        if tiled:
            bin_b, shade_b = algorithmic_bytes_tiled(NX, br.row_count, args.spp, n_tris, BASE_BYTES, st[5])
            kernels = {
                "binning": {"ms": k_bin, "algorithmic_bytes": bin_b, "GB/s": bin_b / (k_bin * 1e-3) / 1e9,
                            "note": "per-frame prepass + bin_tri_kernel<count> + bin_alloc_kernel + "
                                    "bin_tri_kernel<fill>: lane = triangle, one coalesced pass each over the "
                                    "per-triangle records; launch-latency bound at this size"},
                "shade_tiles_kernel": {"ms": k_shade, "algorithmic_bytes": shade_b,
                                       "GB/s": shade_b / (k_shade * 1e-3) / 1e9,
                                       "note": "lane = pixel, exact reference arithmetic on tile survivors; "
                                               "VALU bound (DESIGN.md §5)"},
            }
            dom = "shade_tiles_kernel" if k_shade >= k_bin else "binning"
            alg, k_ms = kernels[dom]["algorithmic_bytes"], kernels[dom]["ms"]
            survivors = {"bin_wavefronts": st[0], "regions_entered": st[1], "tiles": st[4],
                         "tile_survivors_sum": st[5], "tile_survivors_max": st[6], "pool_overflows": st[7],
                         "tiles_with_geometry_frac": 1.0 - st[8] / max(st[4], 1)}
        else:
            alg, k_ms, dom = algorithmic_bytes_brute(NX, br.row_count, args.spp, n_tris), k_tot, "render_list_kernel"
            kernels, survivors = {"render_list_kernel": {"ms": k_tot, "algorithmic_bytes": alg}}, None
        achieved = alg / (k_ms * 1e-3) / 1e9
        # HBM traffic and VALU instruction count of the dominant kernel: NOT from this run — from the
        # committed rocprofv3 PMC passes of this very command (tools/profile_bench.sh; the summary
        # names the commit it was taken at).  The kernel time they are divided by IS this run's.
        traffic, traffic_src, valu = None, None, None
        for prof in ("r4c_tiled", "r4b_tiled", "r4_tiled", "r3_tiled", "r2_tiled", "r1_tiled"):
            pmc = os.path.join(ROOT, "profiles", prof, "pmc_summary_%s.json" % ("shade" if dom.startswith("shade") else "bin_walk"))
            if os.path.exists(pmc):
                break
        if tiled and world == 1 and args.spp == SPP and os.path.exists(pmc):
            try:
                summary = json.load(open(pmc))
                traffic = summary["derived"]["hbm_traffic_bytes_corrected"]
                traffic_src = "prior rocprofv3 --pmc pass of this command: %s" % os.path.relpath(pmc, ROOT)
                # What actually bounds the kernel: the VALU.  Every figure below comes from ONE source, the committed SQ pass:
                # its instruction count, ITS kernel duration (counter passes run the kernel 20-45 % slower than the trace
                # pass) and the clock of the same pass — nothing of this run is mixed in (round 3 divided the pass's
                # instructions by this run's time).  issue: wave64 VALU instructions per clock per SIMD against the 2.25-cycle
                # rate of VGPR-only v_mul/v_add measured on this chip (profiles/r1_ubench_valu2.txt); pipe_busy: the share of
                # the pass's cycles the VALU pipe was occupied (SQ_ACTIVE_INST_VALU counts quad-cycles) — the saturation
                # measure, since the kernel's mix is dominated by 4-cycle forms; lanes_active: SQ_THREAD_CYCLES_VALU /
                # SQ_ACTIVE_INST_VALU where the pass collected it.
                pl, dv = summary["per_launch_counters"], summary["derived"]
                insts, sq_s = pl.get("SQ_INSTS_VALU"), summary.get("pmc_pass_avg_seconds", {}).get("pmc_sq")
                ghz = dv.get("effective_clock_GHz")
                if insts and sq_s and ghz:
                    cycles = ghz * 1e9 * sq_s
                    rate = insts / 1024 / cycles
                    valu = {"unit": "wave64 VALU instructions per clock per SIMD", "achieved": rate, "peak": 1 / 2.25, "frac": rate * 2.25,
                            "pipe_busy_frac": (pl["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cycles) if pl.get("SQ_ACTIVE_INST_VALU") else None,
                            "lanes_active_of_64": dv.get("valu_lanes_active_of_64"),
                            "mix_bound_frac": dv.get("valu_mix_bound_frac"),
                            "insts_per_launch": insts, "clock_GHz": ghz, "clock_source": dv.get("effective_clock_source"),
                            "pass_kernel_ms": sq_s * 1e3, "this_run_kernel_ms": k_ms,
                            "source": "all from the committed SQ counter pass (its instructions, its duration, its clock): %s" % os.path.relpath(pmc, ROOT)}
            except Exception:
                traffic = None
        out = {
            "metric": "Mrays/s at 1920x1080x64spp; achieved HBM GB/s vs peak",
            "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_runs": [x / args.steps * 1e3 for x in passes],
            "clock_warm_ms": clock_warm_ms, "clock_warm_frames": warm_frames,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: %dx%d, %d spp, %d-triangle synthetic OBJ mesh "
                                   "(displaced sphere; bunny.obj is absent from the reference), "
                                   "triangle list (HitableList path), reference camera" % (NX, NY, args.spp, n_tris),
                       "nx": NX, "ny": NY, "spp": args.spp, "triangles": n_tris,
                       "parallelism": ("1 GPU, one launch" if world == 1 else
                                       "32-row strips dealt round-robin x%d + %s + reassembly kernel"
                                       % (world, "crt_gather_strips (C ABI: one grouped RCCL send/recv to rank 0 on a crt_comm_* communicator%s)"
                                          % (", on a stream of its own: frame k's exchange runs under frame k+1's kernels, two pieces in turn"
                                             if br.overlap else "")
                                          if gather == "c" else "torch.distributed.gather to rank 0")),
                       "exchange_overlapped": bool(br.overlap) if world > 1 else None,
                       "gather_path": gather if world > 1 else None, "gather_note": gather_note,
                       "kernel_variant": args.variant},
            # rays x triangles: what the reference's list walk would test, NOT what the tiled kernels
            # execute (they test `tile_survivors_sum` list entries per sample and tile)
            "nominal_list_pairs_per_s": value * 1e6 * n_tris,
            "kernel_ms": k_tot,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": dom, "algorithmic_bytes_per_launch": alg,
                         "bound_note": "the contract's labels are hbm | mfma; this kernel is neither: the HBM figure is reported "
                                       "because the metric asks for it, the measured limiter is below",
                         "limiter": "valu issue, not HBM: see valu_issue (DESIGN.md §5)"},
            "valu_issue": valu,
            "kernels": kernels,
            "survivors": survivors,
            "screen_filling_variant": filling,
            "second_workload": second,
            "bvh_world": bvh_world,
            "f2_shade": f2_shade,
            "reference_default": reference_default,
            "pcie_inclusive_one_call": pcie,
        }
        if world == 1 and not args.no_cpu_baseline:
            base, win, cs, ref = cpu_baseline(obj_path)
            out["cpu_baseline"] = base
            # the same crop through the product, outside the timed region: the frame that is timed is
            # the frame that is checked (bit-exact against the oracle)
            got = hip_crop(scene, camera, win, cs)
            out["parity_crop"] = bool(np.array_equal(np.ascontiguousarray(got).view(np.uint32), np.ascontiguousarray(ref).view(np.uint32)))
            out["parity_crop_hit_pixels"] = int(((ref[..., 0] - ref[..., 2]) > 0.05).sum())
        print(json.dumps(out), flush=True)

    br.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
