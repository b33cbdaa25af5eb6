"""GPU tests of the multi-GPU half of the C ABI on ONE device (the GPU box has one):
launch contexts per (device, stream), the root's reassembly kernel, the grouped RCCL send/recv
of crt_gather_strips (world 1: the root sends to itself through RCCL) and the single-process
driver crt_multi_* (ncclCommInitAll over one device).  The frame of every path must equal the
single-launch frame bit for bit."""
import ctypes as C

import numpy as np
import pytest

from cudaraytracer_amd import _lib, api, scenes

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def make(sc, nx, ny):
    scene = api.Scene.from_mesh(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
    c = sc["camera"]
    return scene, api.create_camera(nx, ny, c["lookfrom"], c["lookat"], c["focus"], c["aperture"], c["vfov"])


def full_frame(gpu, scene, cam, nx, ny, spp):
    st = api.SetCurandState(nx, ny, 0)
    d = gpu.zeros(ny * nx * 3, dtype=gpu.float32, device="cuda")
    api.render_rows(scene, cam, st, nx, ny, spp, 0, ny, d, stream=gpu.cuda.current_stream())
    gpu.cuda.synchronize()
    return d.cpu().numpy().reshape(ny, nx, 3)


def test_two_streams_one_thread_one_scene(gpu):
    """Two bands of one scene enqueued back to back on two streams from one host thread (and a
    second camera on a third): every launch has its own context (derived records, binning
    scratch), so the bands equal the single-launch frame."""
    sc = scenes.displaced_sphere(60, 80)
    nx, ny, spp = 640, 384, 8
    scene, cam = make(sc, nx, ny)
    want = full_frame(gpu, scene, cam, nx, ny, spp)
    cam2 = api.create_camera(nx, ny, (300.0, 400.0, 900.0), (0.0, 150.0, 0.0), 10.0, 0.0, 35.0)
    want2 = full_frame(gpu, scene, cam2, nx, ny, spp)
    s1, s2, s3 = gpu.cuda.Stream(), gpu.cuda.Stream(), gpu.cuda.Stream()
    half = ny // 2
    for _ in range(3):                                   # repeated: a race would not show every time
        st_a = api.SetCurandState(nx, ny, 0, 0, half)
        st_b = api.SetCurandState(nx, ny, 0, half, ny - half)
        st_c = api.SetCurandState(nx, ny, 0)
        a = gpu.zeros(half * nx * 3, dtype=gpu.float32, device="cuda")
        b = gpu.zeros((ny - half) * nx * 3, dtype=gpu.float32, device="cuda")
        c = gpu.zeros(ny * nx * 3, dtype=gpu.float32, device="cuda")
        gpu.cuda.synchronize()
        api.render_rows(scene, cam, st_a, nx, ny, spp, 0, half, a, stream=s1)
        api.render_rows(scene, cam2, st_c, nx, ny, spp, 0, ny, c, stream=s3)
        api.render_rows(scene, cam, st_b, nx, ny, spp, half, ny - half, b, stream=s2)
        gpu.cuda.synchronize()
        got = np.concatenate([a.cpu().numpy().reshape(half, nx, 3), b.cpu().numpy().reshape(ny - half, nx, 3)], axis=0)
        assert np.array_equal(bits(got), bits(want))
        assert np.array_equal(bits(c.cpu().numpy().reshape(ny, nx, 3)), bits(want2))


@pytest.mark.parametrize("nx,ny,world", [(640, 360, 8), (90, 150, 3), (33, 70, 2), (1920, 1080, 8)])
def test_assemble_strips_kernel(gpu, crt_built, nx, ny, world):
    """Every rank's strip set rendered on this GPU into the root's receive layout
    (rank r at r * part_stride), then crt_assemble_strips == the single-launch frame."""
    sc = scenes.displaced_sphere(24, 30)
    spp = 2
    scene, cam = make(sc, nx, ny)
    want = full_frame(gpu, scene, cam, nx, ny, spp)
    stride = api.strip_part_stride(nx, ny, 32, world)
    parts = gpu.full((world * stride,), float("nan"), dtype=gpu.float32, device="cuda")
    for r in range(world):
        if r * 32 >= ny:
            continue
        st = api.CurandState.for_strips(nx, ny, r, 32, world, 0)
        api.render_strips(scene, cam, st, nx, ny, spp, r, 32, world, parts.data_ptr() + 4 * r * stride,
                          stream=gpu.cuda.current_stream())
    full = gpu.zeros(ny * nx * 3, dtype=gpu.float32, device="cuda")
    api.assemble_strips(parts, stride, nx, ny, 32, world, full, stream=gpu.cuda.current_stream())
    gpu.cuda.synchronize()
    assert np.array_equal(bits(full.cpu().numpy().reshape(ny, nx, 3)), bits(want))


def test_dist_assemble_parts_device_branch(gpu):
    """cudaraytracer_amd.dist.assemble_parts on device tensors (what rank 0 runs after the RCCL gather in
    bench.py --gpus N): the [W, max] receive buffer of padded pieces -> the frame, via crt_assemble_strips;
    equal to the CPU branch on the same data."""
    from cudaraytracer_amd import dist
    sc = scenes.displaced_sphere(24, 30)
    for nx, ny, world in [(642, 360, 8), (90, 150, 3)]:          # 642 * 3 is not a multiple of 4: the scalar kernel
        spp = 2
        scene, cam = make(sc, nx, ny)
        want = full_frame(gpu, scene, cam, nx, ny, spp)
        rows_of = [sum(c for _, c in dist.partition_strips(ny, world, r)) for r in range(world)]
        parts = gpu.zeros((world, max(rows_of) * nx * 3), dtype=gpu.float32, device="cuda")
        for r in range(world):
            if rows_of[r] == 0:
                continue
            st = api.CurandState.for_strips(nx, ny, r, 32, world, 0)
            api.render_strips(scene, cam, st, nx, ny, spp, r, 32, world, parts[r], stream=gpu.cuda.current_stream())
        got = dist.assemble_parts(parts, nx, ny, world)
        gpu.cuda.synchronize()
        assert np.array_equal(bits(got.cpu().numpy()), bits(want))
        assert np.array_equal(bits(dist.assemble_parts(parts.cpu(), nx, ny, world).numpy()), bits(want))


def _rccl():
    """The RCCL copy this process holds (torch's), through ctypes: ncclCommInitAll over device 0."""
    lib = C.CDLL("librccl.so.1")
    lib.ncclCommInitAll.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int)]
    lib.ncclCommDestroy.argtypes = [C.c_void_p]
    return lib


def test_gather_strips_through_rccl_world_one(gpu):
    """crt_gather_strips with a real ncclComm_t: at world 1 the root's send to itself and its
    receive go through RCCL's grouped send/recv, then the reassembly kernel runs."""
    sc = scenes.displaced_sphere(24, 30)
    nx, ny, spp = 200, 120, 2
    scene, cam = make(sc, nx, ny)
    want = full_frame(gpu, scene, cam, nx, ny, spp)
    rccl = _rccl()
    comm = C.c_void_p()
    devs = (C.c_int * 1)(gpu.cuda.current_device())
    assert rccl.ncclCommInitAll(C.byref(comm), 1, devs) == 0
    try:
        st = api.CurandState.for_strips(nx, ny, 0, 32, 1, 0)
        piece = gpu.zeros(ny * nx * 3, dtype=gpu.float32, device="cuda")
        stride = api.strip_part_stride(nx, ny, 32, 1)
        parts = gpu.zeros(stride, dtype=gpu.float32, device="cuda")
        full = gpu.zeros(ny * nx * 3, dtype=gpu.float32, device="cuda")
        stream = gpu.cuda.Stream()
        gpu.cuda.synchronize()
        api.render_strips(scene, cam, st, nx, ny, spp, 0, 32, 1, piece, stream=stream)
        api.gather_strips(piece, nx, ny, 32, 0, 1, parts, stride, full, comm.value, stream=stream)
        stream.synchronize()
        assert np.array_equal(bits(full.cpu().numpy().reshape(ny, nx, 3)), bits(want))
    finally:
        rccl.ncclCommDestroy(comm)


def test_single_process_driver_on_one_device(gpu, oracle):
    """crt_multi_* with n_gpus = 1 (ncclCommInitAll, stream + host worker per device): two frames
    continue the RNG streams like the reference's frame loop; a pose update reaches the replica."""
    sc = scenes.displaced_sphere(16, 20)
    nx, ny, spp = 96, 80, 3
    m = api.MultiRenderer(sc["points"], sc["idx"], sc["normals"], sc["index_base"], nx, ny, 1)
    c = sc["camera"]
    cam = api.create_camera(nx, ny, c["lookfrom"], c["lookat"], c["focus"], c["aperture"], c["vfov"])
    f0 = m.render(cam, spp)
    moved = (sc["points"] * np.float32(1.1)).astype(np.float32)
    m.update_pose(moved)
    f1 = m.render(cam, spp, frame_index=1)
    m.close()
    w = oracle.World(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
    ocam = oracle.make_camera(c["lookfrom"], c["lookat"], (0, 1, 0), c["vfov"], np.float32(nx) / np.float32(ny),
                              c["aperture"], c["focus"])
    r0, states = oracle.render(w, ocam, nx, ny, spp)
    assert np.array_equal(bits(f0), bits(r0))
    w.update_pose(moved)
    r1, _ = oracle.render(w, ocam, nx, ny, spp, states=states, frame=1)
    assert np.array_equal(bits(f1).reshape(-1, 3)[1:], bits(r1).reshape(-1, 3)[1:])   # pixel 0 of frame 1: DESIGN.md §1
    with pytest.raises(_lib.CrtError):
        api.MultiRenderer(sc["points"], sc["idx"], sc["normals"], sc["index_base"], nx, ny, 99)


def test_cpp_driver_program(gpu, oracle, tmp_path):
    """cudaraytracer_amd/crt_driver (csrc/crt_driver.cpp): the C++ host program in place of the
    reference's main() — OBJ in, PNG per frame + output.csv out, C ABI only.  Its frame equals the
    oracle's (BVH world, LambertShade, the reference's live configuration)."""
    import os
    import subprocess
    from test_abi import decode_png
    exe = os.path.join(os.path.dirname(_lib.library_path()), "crt_driver")
    assert os.path.exists(exe), "run make -C cudaraytracer_amd/csrc"
    sc = scenes.displaced_sphere(16, 20, index_base=1)
    obj = scenes.write_obj(str(tmp_path / "m.obj"), sc)
    nx, ny, spp = 96, 64, 2
    out, csv = str(tmp_path / "frames"), str(tmp_path / "output.csv")
    subprocess.check_call([exe, "--obj", obj, "--scale", "1", "--width", str(nx), "--height", str(ny), "--spp", str(spp),
                           "--frames", "2", "--world", "bvh", "--out", out, "--csv", csv], cwd=str(tmp_path))
    pts, idx = oracle.parse_obj(obj)
    w = oracle.World(pts, idx, None, 1)
    cam = oracle.make_camera((0, 100, 1000), (0, 150, 0), (0, 1, 0), 40.0, np.float32(nx) / np.float32(ny), 0.0, 10.0)
    states = oracle.fresh_states(nx, ny)
    bvh = oracle.Bvh(w, states)
    ref, states = oracle.render_bvh(bvh, cam, nx, ny, spp, states)
    assert np.array_equal(decode_png(os.path.join(out, "picture_0.png")), oracle.to_rgba8(ref))
    rows = open(csv).read().splitlines()
    assert rows[0] == "frame,rendering,update,build" and rows[1].startswith(",,,") and rows[2].startswith("0,") and rows[3].startswith("1,")


def test_bench_under_torchrun_two_ranks_sharing_the_gpu(gpu):
    """The driver's N>1 launch line (python -m torch.distributed.run ... bench.py --gpus N) with two ranks on the one GPU
    of the test box (CRT_BENCH_SHARE_GPU=1: gloo instead of RCCL, same strips, same gather, same reassembly kernel):
    rank 0 prints ONE JSON line for the whole job."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, CRT_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--spp", "4",
           "--no-cpu-baseline", "--no-extras"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["value"] > 0 and out["scaling"] == "strong"
    assert "strips" in out["config"]["parallelism"]


def test_bench_line_carries_the_contract_fields(gpu):
    """python bench.py (the driver's N=1 command, here with few steps and samples): ONE JSON line with the contract's
    fields, the roofline and cpu_baseline objects, and the oracle crop of the timed frame bit-identical."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--spp", "2", "--no-extras"],
                       cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["warmup"] == 1 and out["unit"] == "Mrays/s" and out["dtype"] == "f32"
    assert out["vs_baseline"] is None and out["higher_is_better"] is True and out["data"] == "synthetic" and "workload" in out["config"]
    rf = out["roofline"]
    # `bound` carries one of the contract's two labels; what actually limits the kernel is in `limiter` (VALU issue), said next to it
    assert rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert rf["bound"] in ("hbm", "mfma") and "valu" in rf["limiter"] and rf["bound_note"]
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["sample"]
    assert cb["repetitions"] >= 3 and len(cb["seconds"]) == cb["repetitions"] and cb["cpu_model"] and cb["host_cores"] >= 1
    assert cb["all_cores"]["cores"] == cb["host_cores"] and cb["all_cores"]["value"] > 0
    assert out["parity_crop"] is True and out["parity_crop_hit_pixels"] > 0


def test_failed_send_leaves_a_clean_state(gpu, crt_built):
    """An RCCL error inside the grouped exchange (injected: the library's next ncclSend reports ncclInternalError without
    reaching RCCL) must not leave the NCCL group open, work in flight or the wrong device current: the call fails, the
    NEXT call on the same communicator succeeds and delivers the frame.  Both entry points: crt_gather_strips (caller's
    communicator) and crt_multi_render (the single-process driver)."""
    sc = scenes.displaced_sphere(24, 30)
    nx, ny, spp = 200, 120, 2
    scene, cam = make(sc, nx, ny)
    want = full_frame(gpu, scene, cam, nx, ny, spp)
    dev_before = gpu.cuda.current_device()
    rccl = _rccl()
    comm = C.c_void_p()
    devs = (C.c_int * 1)(dev_before)
    assert rccl.ncclCommInitAll(C.byref(comm), 1, devs) == 0
    try:
        piece = gpu.zeros(ny * nx * 3, dtype=gpu.float32, device="cuda")
        stride = api.strip_part_stride(nx, ny, 32, 1)
        parts = gpu.zeros(stride, dtype=gpu.float32, device="cuda")
        full = gpu.zeros(ny * nx * 3, dtype=gpu.float32, device="cuda")
        stream = gpu.cuda.Stream()
        gpu.cuda.synchronize()
        for attempt in range(2):
            st = api.CurandState.for_strips(nx, ny, 0, 32, 1, 0)
            api.render_strips(scene, cam, st, nx, ny, spp, 0, 32, 1, piece, stream=stream)
            if attempt == 0:
                assert crt_built.crt_debug_inject_send_failure(1) == 0
                with pytest.raises(_lib.CrtError) as ei:
                    api.gather_strips(piece, nx, ny, 32, 0, 1, parts, stride, full, comm.value, stream=stream)
                assert "ncclSend" in str(ei.value)
                assert crt_built.crt_debug_inject_send_failure(0) == 0          # the injected failure was consumed
                assert gpu.cuda.current_device() == dev_before
            else:
                api.gather_strips(piece, nx, ny, 32, 0, 1, parts, stride, full, comm.value, stream=stream)
                stream.synchronize()
                assert np.array_equal(bits(full.cpu().numpy().reshape(ny, nx, 3)), bits(want))
    finally:
        crt_built.crt_debug_inject_send_failure(0)
        rccl.ncclCommDestroy(comm)
    # the single-process driver
    m = api.MultiRenderer(sc["points"], sc["idx"], sc["normals"], sc["index_base"], nx, ny, 1)
    try:
        crt_built.crt_debug_inject_send_failure(1)
        with pytest.raises(_lib.CrtError) as ei:
            m.render(cam, spp)
        assert "ncclSend" in str(ei.value)
        assert gpu.cuda.current_device() == dev_before
        # the failed frame DID advance the RNG states (its strips were rendered before the exchange failed), as a frame
        # of the reference's loop would: a fresh renderer gives frame 0 again
    finally:
        crt_built.crt_debug_inject_send_failure(0)
        m.close()
    m = api.MultiRenderer(sc["points"], sc["idx"], sc["normals"], sc["index_base"], nx, ny, 1)
    try:
        assert np.array_equal(bits(m.render(cam, spp)), bits(want))
    finally:
        m.close()


def test_streams_come_and_go_without_growing_device_memory(gpu, crt_built):
    """A host that makes a fresh stream per task: crt_stream_release before the stream is destroyed frees its launch
    context (records, binning scratch, events).  100 streams, each used for one tiled frame: free device memory ends where
    it was after the first; and the renderer of crt_multi_* releases the contexts of its own streams when it is closed."""
    sc = scenes.displaced_sphere(40, 50)
    nx, ny, spp = 320, 192, 2
    scene, cam = make(sc, nx, ny)
    want = full_frame(gpu, scene, cam, nx, ny, spp)
    d = gpu.zeros(ny * nx * 3, dtype=gpu.float32, device="cuda")
    st = api.SetCurandState(nx, ny, 0)

    hip = C.CDLL("libamdhip64.so")        # the HIP runtime this process already holds (torch's): real create / destroy cycles
    hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]

    class RawStream:                       # what api.render_rows needs of a stream object
        def __init__(self, handle):
            self.cuda_stream = handle

    def one_stream(release=True):
        h = C.c_void_p()
        assert hip.hipStreamCreate(C.byref(h)) == 0
        st.reset(0)
        api.render_rows(scene, cam, st, nx, ny, spp, 0, ny, d, stream=RawStream(h.value))
        assert hip.hipStreamSynchronize(h) == 0
        if release:
            assert crt_built.crt_stream_release(h) == 0
        assert hip.hipStreamDestroy(h) == 0

    for _ in range(8):                     # the runtime's own per-stream pools settle
        one_stream()
    gpu.cuda.synchronize()
    free0, _ = gpu.cuda.mem_get_info()
    for _ in range(100):
        one_stream()
    assert np.array_equal(bits(d.cpu().numpy().reshape(ny, nx, 3)), bits(want))
    gpu.cuda.synchronize()
    free1, _ = gpu.cuda.mem_get_info()
    assert free0 - free1 < (8 << 20), "device memory shrank by %.1f MB over 100 released streams" % ((free0 - free1) / 2**20)
    assert crt_built.crt_stream_release(C.c_void_p(12345)) == 0                # nothing to release: still CRT_OK
    # A host that destroys streams WITHOUT crt_stream_release leaves their contexts behind, and the runtime hands the
    # handles out again.  Where the process's HIP runtime has hipStreamGetId (>= 7.1) the library notices by itself; torch's
    # bundled 7.0 runtime does not have it, so the branch is driven through crt_debug_force_stream_recycle here: the
    # context is reset (device synchronised, events and cached occupancy dropped), and the frame is still the frame.
    for _ in range(4):
        one_stream(release=False)
    h = C.c_void_p()
    assert hip.hipStreamCreate(C.byref(h)) == 0
    try:
        for forced in (0, 1, 1):
            assert crt_built.crt_debug_force_stream_recycle(forced) == 0
            st.reset(0)
            api.render_rows(scene, cam, st, nx, ny, spp, 0, ny, d, stream=RawStream(h.value))   # 2nd, 3rd: look-up of the kept context
            assert hip.hipStreamSynchronize(h) == 0
            assert crt_built.crt_debug_force_stream_recycle(0) == 0            # the look-up consumed the forced recycle
            assert np.array_equal(bits(d.cpu().numpy().reshape(ny, nx, 3)), bits(want))
            assert api.last_kernel_times()[0] > 0.0                            # the reset context's fresh events work
    finally:
        crt_built.crt_debug_force_stream_recycle(0)
        assert crt_built.crt_stream_release(h) == 0
        assert hip.hipStreamDestroy(h) == 0
    def one_renderer():
        m = api.MultiRenderer(sc["points"], sc["idx"], sc["normals"], sc["index_base"], nx, ny, 1)
        m.render(cam, spp)
        m.close()

    one_renderer()                         # RCCL's own first-use allocations
    gpu.cuda.synchronize()
    free1, _ = gpu.cuda.mem_get_info()
    for _ in range(5):
        one_renderer()
    gpu.cuda.synchronize()
    free2, _ = gpu.cuda.mem_get_info()
    assert free1 - free2 < (16 << 20), "device memory shrank by %.1f MB over 5 renderers" % ((free1 - free2) / 2**20)


def test_one_call_render_is_repeatable_and_keeps_its_buffers(gpu, crt_built, oracle):
    """crt_render = SetCurandState(seed) + one frame into a host buffer: the same seed gives the same frame on every
    call (the states are restored from the pristine copy), another seed another frame, another size reallocates; the
    oracle agrees; crt_stream_release(NULL) frees what the calls kept."""
    sc = scenes.displaced_sphere(16, 20)
    scene = api.Scene.from_mesh(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
    nx, ny, spp = 96, 80, 3
    cam = api.init_camera(nx, ny)
    a = api.render(nx, ny, spp, scene, cam, seed=0)
    b = api.render(nx, ny, spp, scene, cam, seed=0)
    c = api.render(nx, ny, spp, scene, cam, seed=7)
    d = api.render(nx, ny, spp, scene, cam, seed=0)
    assert np.array_equal(bits(a), bits(b)) and np.array_equal(bits(a), bits(d)) and not np.array_equal(bits(a), bits(c))
    w = oracle.World(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
    ocam = oracle.make_camera((0, 100, 1000), (0, 150, 0), (0, 1, 0), 40.0, np.float32(nx) / np.float32(ny), 0.0, 10.0)
    ref, _ = oracle.render(w, ocam, nx, ny, spp)
    assert np.array_equal(bits(a), bits(ref))
    cam2 = api.init_camera(64, 48)
    e = api.render(64, 48, 2, scene, cam2, seed=0)
    ocam2 = oracle.make_camera((0, 100, 1000), (0, 150, 0), (0, 1, 0), 40.0, np.float32(64) / np.float32(48), 0.0, 10.0)
    ref2, _ = oracle.render(w, ocam2, 64, 48, 2)
    assert np.array_equal(bits(e), bits(ref2))
    assert crt_built.crt_stream_release(None) == 0
    assert np.array_equal(bits(api.render(nx, ny, spp, scene, cam, seed=0)), bits(a))
    # into the caller's buffers: a page-locked tensor, a numpy array; wrong shapes are refused before the call
    pinned = gpu.empty(ny * nx * 3, dtype=gpu.float32, pin_memory=True)
    got = api.render(nx, ny, spp, scene, cam, seed=0, out=pinned)
    assert np.array_equal(bits(got), bits(a)) and got.ctypes.data == pinned.data_ptr()
    mine = np.full((ny, nx, 3), np.nan, dtype=np.float32)
    assert np.array_equal(bits(api.render(nx, ny, spp, scene, cam, seed=0, out=mine)), bits(a)) and np.array_equal(bits(mine), bits(a))
    with pytest.raises(ValueError):
        api.render(nx, ny, spp, scene, cam, seed=0, out=np.zeros(5, dtype=np.float32))
    with pytest.raises(ValueError):
        api.render(nx, ny, spp, scene, cam, seed=0, out=gpu.empty(ny * nx * 3, dtype=gpu.float64))


@pytest.mark.parametrize("overlap", [False, True])
def test_band_renderer_c_path_gather_world_one(gpu, overlap):
    """What bench.py --gpus N runs per rank over RCCL, at world 1: the communicator comes from the C ABI
    (crt_comm_unique_id + crt_comm_init_rank = ncclGetUniqueId + ncclCommInitRank), the exchange step is crt_gather_strips
    (the root's send to itself goes through RCCL), the frame equals the single-launch frame; the frames continue the
    streams.  overlap: the exchange of frame k on a stream of its own while frame k + 1 renders into the other piece —
    five frames enqueued back to back without a host synchronisation in between, each copied out on the caller's stream
    behind wait()."""
    from cudaraytracer_amd.dist import BandRenderer
    sc = scenes.displaced_sphere(24, 30)
    nx, ny, spp = 200, 150, 3
    scene, cam = make(sc, nx, ny)
    st = api.SetCurandState(nx, ny, 0)
    d = gpu.zeros(ny * nx * 3, dtype=gpu.float32, device="cuda")
    wants = []
    n_frames = 5 if overlap else 2
    for _ in range(n_frames):
        api.render_rows(scene, cam, st, nx, ny, spp, 0, ny, d, stream=gpu.cuda.current_stream())
        gpu.cuda.synchronize()
        wants.append(d.cpu().numpy().reshape(ny, nx, 3).copy())
    br = BandRenderer(scene, cam, nx, ny, 0, 1, seed=0, mode="strips", gather="c", overlap=overlap)
    try:
        assert br.gather_path == "c" and br.comm and br.overlap == overlap
        if overlap:
            copies = []
            for k in range(n_frames):
                br.render(spp)
                full = br.gather()
                br.wait()                       # the caller's stream behind the exchange of frame k
                copies.append(full.clone())     # ... and the next exchange behind this copy (it follows the next render's event)
            gpu.cuda.synchronize()
            for k in range(n_frames):
                assert np.array_equal(bits(copies[k].cpu().numpy()), bits(wants[k])), k
        else:
            for k in range(n_frames):
                br.render(spp)
                full = br.gather()
                gpu.cuda.synchronize()
                assert np.array_equal(bits(full.cpu().numpy()), bits(wants[k])), k
    finally:
        br.close()
    assert br.comm is None


def test_comm_helpers_reject_bad_arguments(gpu, crt_built):
    ident = api.comm_unique_id()
    assert len(ident) == api.COMM_ID_BYTES and any(ident)
    with pytest.raises(_lib.CrtError):
        api.comm_init_rank(2, ident, 2)                 # rank out of range
    with pytest.raises(_lib.CrtError):
        api.comm_init_rank(1, ident[:5], 0)
    assert crt_built.crt_comm_destroy(None) == 0


@pytest.mark.parametrize("world_kind,shader", [(api.WORLD_BVH, api.SHADER_LAMBERT), (api.WORLD_BVH, api.SHADER_SHADE),
                                               (api.WORLD_LIST, api.SHADER_NORMAL)])
def test_strips_ex_reassembled_equal_rows_ex(gpu, world_kind, shader):
    """crt_render_strips_ex: the strip sets of a 3-way split with an explicit world and shader, reassembled, equal the
    whole frame through crt_render_rows_ex (pixels; the RNG stream of a pixel is its global index)."""
    sc = scenes.displaced_sphere(20, 26)
    nx, ny, spp, world = 160, 136, 3, 3
    scene, cam = make(sc, nx, ny)
    scene.set_materials([(api.LAMBERTIAN, (0.65, 0.05, 0.05), 0.0), (api.METAL, (0.8, 0.85, 0.88), 0.1), (api.DIELECTRIC, (1, 1, 1), 1.5),
                         (api.DIFFUSE_LIGHT, (4.0, 4.0, 4.0), 0.0)], (np.arange(scene.n_triangles) // 60 % 4).astype(np.int32))
    scene.build_bvh(None, 0)
    st = api.SetCurandState(nx, ny, 0)
    d = gpu.zeros(ny * nx * 3, dtype=gpu.float32, device="cuda")
    api.render_rows_ex(scene, cam, st, nx, ny, spp, 0, ny, d, world=world_kind, shader=shader, max_depth=6, stream=gpu.cuda.current_stream())
    gpu.cuda.synchronize()
    want = d.cpu().numpy().reshape(ny, nx, 3)
    stride = api.strip_part_stride(nx, ny, 32, world)
    parts = gpu.full((world * stride,), float("nan"), dtype=gpu.float32, device="cuda")
    for r in range(world):
        stt = api.CurandState.for_strips(nx, ny, r, 32, world, 0)
        api.render_strips_ex(scene, cam, stt, nx, ny, spp, r, 32, world, parts.data_ptr() + 4 * r * stride, world=world_kind,
                             shader=shader, max_depth=6, stream=gpu.cuda.current_stream())
    full = gpu.zeros(ny * nx * 3, dtype=gpu.float32, device="cuda")
    api.assemble_strips(parts, stride, nx, ny, 32, world, full, stream=gpu.cuda.current_stream())
    gpu.cuda.synchronize()
    assert np.array_equal(bits(full.cpu().numpy().reshape(ny, nx, 3)), bits(want))


def test_single_process_driver_runs_the_shipped_configuration(gpu, oracle):
    """crt_multi_configure(CRT_WORLD_BVH): the reference's shipped world (kernel.cu:97 renderBVH) behind the multi-device
    driver — tree built per replica, frames equal the oracle's BVH frames, a pose update refits (Update_BVH, render.h:215);
    then the recursive shader with materials through the same driver against crt_render_rows_ex."""
    sc = scenes.displaced_sphere(16, 20)
    nx, ny, spp = 96, 80, 3
    c = sc["camera"]
    cam = api.create_camera(nx, ny, c["lookfrom"], c["lookat"], c["focus"], c["aperture"], c["vfov"])
    m = api.MultiRenderer(sc["points"], sc["idx"], sc["normals"], sc["index_base"], nx, ny, 1)
    try:
        m.configure(api.WORLD_BVH, api.SHADER_LAMBERT)
        f0 = m.render(cam, spp)
        moved = (sc["points"] * np.float32(1.1)).astype(np.float32)
        m.update_pose(moved)
        f1 = m.render(cam, spp, frame_index=1)
        w = oracle.World(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
        ocam = oracle.make_camera(c["lookfrom"], c["lookat"], (0, 1, 0), c["vfov"], np.float32(nx) / np.float32(ny), c["aperture"], c["focus"])
        # the driver builds its trees from a private stream-0 sequence: the frame's own stream 0 is untouched by the build
        bvh = oracle.Bvh(w, oracle.fresh_states(nx, ny))
        states = oracle.fresh_states(nx, ny)
        r0, states = oracle.render_bvh(bvh, ocam, nx, ny, spp, states)
        assert np.array_equal(bits(f0), bits(r0))
        w.update_pose(moved)
        bvh.update()
        r1, _ = oracle.render_bvh(bvh, ocam, nx, ny, spp, states, frame=1)
        assert np.array_equal(bits(f1).reshape(-1, 3)[1:], bits(r1).reshape(-1, 3)[1:])   # pixel 0 of frame 1: DESIGN.md §1
        # shade with four material classes: the driver against the single-device entry point on an identical scene
        table = [(api.LAMBERTIAN, (0.65, 0.05, 0.05), 0.0), (api.METAL, (0.8, 0.85, 0.88), 0.1), (api.DIELECTRIC, (1, 1, 1), 1.5),
                 (api.DIFFUSE_LIGHT, (4.0, 4.0, 4.0), 0.0)]
        n_tris = len(sc["idx"])
        of = (np.arange(n_tris) // 40 % 4).astype(np.int32)
        m2 = api.MultiRenderer(sc["points"], sc["idx"], sc["normals"], sc["index_base"], nx, ny, 1)
        try:
            m2.configure(api.WORLD_BVH, api.SHADER_SHADE, table, of)
            g = m2.render(cam, spp, max_depth=5)
        finally:
            m2.close()
        scene = api.Scene.from_mesh(sc["points"], sc["idx"], sc["normals"], sc["index_base"])
        scene.set_materials(table, of)
        scene.build_bvh(None, 0)
        st = api.SetCurandState(nx, ny, 0)
        d = gpu.zeros(ny * nx * 3, dtype=gpu.float32, device="cuda")
        api.render_rows_ex(scene, cam, st, nx, ny, spp, 0, ny, d, world=api.WORLD_BVH, shader=api.SHADER_SHADE, max_depth=5,
                           stream=gpu.cuda.current_stream())
        gpu.cuda.synchronize()
        assert np.array_equal(bits(g), bits(d.cpu().numpy().reshape(ny, nx, 3)))
        with pytest.raises(_lib.CrtError):
            m.configure(7, api.SHADER_LAMBERT)
    finally:
        m.close()
