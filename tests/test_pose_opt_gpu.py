"""GPU side of Optimizer::PoseOptimization: the HIP kernel, through the C ABI, against
the CPU restatement (both its Ceres-shaped QR form and the normal-equation form the kernel computes),
the committed vectors, and the reference-shaped host class."""
import ctypes as C

import numpy as np
import pytest

from dsdtm_amd import capi, synth
from dsdtm_amd.optimizer import Optimizer, pose_optimization
from tests import helpers as H

pytestmark = pytest.mark.gpu

TOL = 1e-10     # rad / m / normalised-image units: the two sides differ only in summation order


def run_gpu(ctx, P, max_iterations=100, **over):
    T = np.ascontiguousarray(over.get("T", P.T_seed), np.float64).reshape(12).copy()
    rn, sm = pose_optimization(ctx, over.get("bearing", P.bearing), over.get("p_world", P.p_world), over.get("level", P.level),
                               over.get("use", P.use), T, max_iterations)
    return T.reshape(3, 4), rn, sm


def assert_same(gpu, cpu, what=""):
    (Tg, rg, sg), (Tc, rc, sc) = gpu, cpu
    for k in ("iterations", "successful_steps", "termination", "n_residual_blocks"):
        assert sg[k] == sc[k], (what, k, sg, sc)
    ang, dt = synth.pose_error(Tg, Tc)
    assert ang <= TOL and dt <= TOL, (what, ang, dt)
    assert np.allclose(rg, rc, rtol=0, atol=TOL), what
    assert np.allclose([sg["initial_cost"], sg["final_cost"]], [sc["initial_cost"], sc["final_cost"]], rtol=1e-10, atol=1e-300), what
    assert np.allclose(sg["x"], sc["x"], rtol=0, atol=TOL), what


CASES = [dict(seed=1, n=200, max_level=3), dict(seed=2, n=200, max_level=0), dict(seed=3, n=500, max_level=4, outlier_frac=0.2),
         dict(seed=4, n=30, max_level=2), dict(seed=5, n=150, max_level=3, seed_t=0.12, seed_w=0.1), dict(seed=6, n=7, max_level=1, unused_frac=0.0),
         dict(seed=7, n=2000, max_level=3), dict(seed=8, n=64, max_level=2), dict(seed=9, n=65, max_level=2), dict(seed=10, n=700, max_level=4, noise_px=1.5),
         # the few-frames kernel keeps one / two features per lane in registers up to 256 / 512 features
         dict(seed=11, n=256, max_level=3, unused_frac=0.3), dict(seed=12, n=257, max_level=3, unused_frac=0.3),
         dict(seed=13, n=512, max_level=2, outlier_frac=0.1), dict(seed=14, n=513, max_level=2, outlier_frac=0.1)]


@pytest.mark.parametrize("kw", CASES)
def test_kernel_equals_the_restatement(gpu_ctx, oracle, kw):
    P = synth.make_pose_problem(**kw)
    g = run_gpu(gpu_ctx, P)
    assert_same(g, oracle.pose_optimization(P.bearing, P.p_world, P.level, P.use, P.T_seed, linear_solver=1), "normal equations")
    assert_same(g, oracle.pose_optimization(P.bearing, P.p_world, P.level, P.use, P.T_seed, linear_solver=0), "Householder QR")


def test_committed_vectors(gpu_ctx):
    g = np.load(H.golden_path("pose_opt.npz"))
    for k in range(int(g["n_cases"])):
        T = np.ascontiguousarray(g[f"T_seed{k}"], np.float64).reshape(12).copy()
        rn, sm = pose_optimization(gpu_ctx, g[f"bearing{k}"], g[f"p_world{k}"], g[f"level{k}"], g[f"use{k}"], T)
        assert [sm["iterations"], sm["successful_steps"], sm["termination"], sm["n_residual_blocks"]] == list(g[f"summary{k}"])
        ang, dt = synth.pose_error(T.reshape(3, 4), g[f"T_out{k}"])
        assert ang <= TOL and dt <= TOL
        assert np.allclose(rn, g[f"residual_norm{k}"], rtol=0, atol=TOL)
        assert np.allclose([sm["initial_cost"], sm["final_cost"]], g[f"cost{k}"], rtol=1e-10)


def test_edge_cases(gpu_ctx, oracle):
    P = synth.make_pose_problem(31, n=60, max_level=2, unused_frac=0.3)
    z = np.zeros_like(P.use)
    # no residual block / no feature at all: pose through log/exp once, nothing else
    for kw in (dict(use=z), dict(bearing=np.zeros((0, 3)), p_world=np.zeros((0, 3)), level=np.zeros(0, np.int32), use=np.zeros(0, np.uint8))):
        T, rn, sm = run_gpu(gpu_ctx, P, **kw)
        assert sm["termination"] == capi.PO_NO_RESIDUALS and len(rn) == 0 and sm["iterations"] == 0
        assert max(synth.pose_error(T, P.T_seed)) < 1e-15
    # iteration cap, including 0 (norms at the seed pose, in residual-block order)
    for cap in (0, 1, 3):
        assert_same(run_gpu(gpu_ctx, P, cap), oracle.pose_optimization(P.bearing, P.p_world, P.level, P.use, P.T_seed, cap, 1), f"cap {cap}")
    # a map point exactly on the camera plane: the initial evaluation fails, parameters stay
    pw = P.p_world.copy()
    i = int(np.nonzero(P.use)[0][0])
    Tid = np.eye(4)[:3]
    pw[i] = [0.3, -0.2, 0.0]
    T, rn, sm = run_gpu(gpu_ctx, P, T=Tid, p_world=pw)
    Tc, rnc, smc = oracle.pose_optimization(P.bearing, pw, P.level, P.use, Tid, linear_solver=1)
    assert sm["termination"] == smc["termination"] == capi.PO_EVALUATION_FAILED
    assert max(synth.pose_error(T, Tid)) < 1e-15 and sm["iterations"] == 0
    # under-determined problems (1 and 2 features) end in a valid state, whatever path they take
    for n in (1, 2):
        Pn = synth.make_pose_problem(40 + n, n=n, unused_frac=0.0)
        T, rn, sm = run_gpu(gpu_ctx, Pn)
        assert 0 <= sm["termination"] <= 7 and np.isfinite(T).all() and len(rn) == n and sm["final_cost"] <= sm["initial_cost"]
    # bad arguments
    T12 = np.ascontiguousarray(P.T_seed).reshape(12).copy()
    lv = P.level.copy(); lv[i] = 40
    with pytest.raises(capi.DsdtmError):
        pose_optimization(gpu_ctx, P.bearing, P.p_world, lv, P.use, T12)


@pytest.mark.parametrize("F", [96, 8])
def test_batch_on_the_device(gpu_ctx, oracle, F):
    """n_frames independent problems in one launch (ragged feature counts), device pointers, own stream.
    96 frames: one wavefront per frame; 8 frames: four wavefronts per frame (the few-frames shape)."""
    import torch
    dev = torch.device("cuda:0")
    maxf = 512
    rng = np.random.default_rng(3)
    probs = [synth.make_pose_problem(100 + k, n=(maxf if k == 0 else 3 if k == 1 else int(rng.integers(1, maxf + 1))),
                                     max_level=int(rng.integers(0, 5)), unused_frac=(0.0 if k == 1 else 0.1))
             for k in range(F)]
    bearing = np.zeros((F, maxf, 3)); pw = np.zeros((F, maxf, 3)); level = np.zeros((F, maxf), np.int32)
    use = np.ones((F, maxf), np.uint8)        # garbage beyond n_features must not be read
    nf = np.zeros(F, np.int32); T = np.zeros((F, 12))
    for k, P in enumerate(probs):
        n = len(P.use); nf[k] = n
        bearing[k, :n] = P.bearing; pw[k, :n] = P.p_world; level[k, :n] = P.level; use[k, :n] = P.use; T[k] = P.T_seed.reshape(12)
    t = lambda a: torch.from_numpy(a).to(dev)
    d_b, d_p, d_l, d_u, d_n, d_T = t(bearing), t(pw), t(level), t(use), t(nf), t(T)
    d_rn = torch.full((F, maxf), -1.0, dtype=torch.float64, device=dev)
    d_sm = torch.zeros((F, C.sizeof(capi.PoseOptSummary)), dtype=torch.uint8, device=dev)
    prm = capi.PoseOptParams(100, 0)
    f = gpu_ctx.lib.dsdtm_pose_optimization_batch_device
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.POINTER(capi.PoseOptParams), C.c_void_p, C.c_void_p, C.c_void_p]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        gpu_ctx.check(f(gpu_ctx.handle, F, maxf, d_n.data_ptr(), d_b.data_ptr(), d_p.data_ptr(), d_l.data_ptr(), d_u.data_ptr(),
                        d_T.data_ptr(), C.byref(prm), d_rn.data_ptr(), d_sm.data_ptr(), s.cuda_stream))
    s.synchronize()
    Tg, rng_, smb = d_T.cpu().numpy(), d_rn.cpu().numpy(), d_sm.cpu().numpy()
    for k, P in enumerate(probs):
        sm = capi.PoseOptSummary.from_buffer_copy(smb[k].tobytes()).as_dict()
        cpu = oracle.pose_optimization(P.bearing, P.p_world, P.level, P.use, P.T_seed, linear_solver=1)
        nb = sm["n_residual_blocks"]
        assert_same((Tg[k].reshape(3, 4), rng_[k, :nb], sm), cpu, f"frame {k}")
        assert np.all(rng_[k, nb:] == -1.0)                   # nothing written past the block count


class _MP:
    def __init__(self, p, found=1, bad=False):
        self.p, self.mnFound, self.mbBad = np.asarray(p, np.float64), found, bad

    def Get_Pose(self): return self.p
    def IsBad(self): return self.mbBad

    def EraseFound(self, n=1):
        self.mnFound -= n
        if self.mnFound <= 0:
            self.mbBad = True


def test_host_class_pose_and_erase_walk(gpu_ctx, oracle):
    """Optimizer::PoseOptimization as Tracking calls it (src/Tracking.cpp:236): the pose is written back and
    EraseFound hits the map points the reference's walk hits — residual i (block order) against the map
    point of FEATURE i (src/Optimizer.cpp:80-92)."""
    from dsdtm_amd.frame import Frame
    cam = synth.Camera.tum()
    P = synth.make_pose_problem(77, n=120, cam=cam, outlier_frac=0.25, unused_frac=0.0, max_level=2)
    fr = Frame(cam, [np.zeros((8, 8), np.uint8)], P.T_seed)
    initial = np.ones(120, np.uint8); initial[5::11] = 0
    fr.set_features(np.zeros((120, 2), np.float32), P.bearing, P.p_world, initial, P.level)
    mpts = [_MP(P.p_world[i], found=1 + i % 3) for i in range(120)]
    for i in (3, 40, 41):
        mpts[i] = None
    mpts[17].mbBad = True
    fr.mvMapPoints = mpts
    found0 = [m.mnFound if m else None for m in mpts]
    sm = Optimizer.PoseOptimization(fr, 10, ctx=gpu_ctx)          # tIterations is ignored
    use = np.array([m is not None and not (i == 17) and bool(initial[i]) for i, m in enumerate(mpts)], np.uint8)
    Tc, rnc, smc = oracle.pose_optimization(P.bearing, P.p_world, P.level, use, P.T_seed, linear_solver=1)
    assert sm["iterations"] == smc["iterations"] and sm["n_residual_blocks"] == int(use.sum())
    assert max(synth.pose_error(fr.Get_Pose(), Tc)) <= TOL
    thresh = 2.0 / float(np.float32(cam.f))
    expect = list(found0)
    bad = [m.mbBad if m else None for m in [(_MP(0, f, i == 17) if f is not None else None) for i, f in enumerate(found0)]]
    for i in range(len(rnc)):                                     # sequential restatement of :80-92
        if rnc[i] > thresh and use[i] and not bad[i]:
            expect[i] -= 1
            if expect[i] <= 0:
                bad[i] = True
    assert [m.mnFound if m else None for m in mpts] == expect
    assert [m.mbBad if m else None for m in mpts] == bad
    assert sum(e != f for e, f in zip(expect, found0) if e is not None) > 3       # the walk did something
