"""GPU: the xGMI mailbox (mtd_comm_*, metadynamics.xgmi) — SURVEY.md §8e.  The one-GPU box has no second card, so the
ranks of these tests are separate PROCESSES that all use cuda:0: IPC handles, peer mappings, the wire protocol, the
slot double-buffering and the fused kernels' send / receive paths are the real ones, only the link is missing."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import util

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_world(world, n_global, steps=6, timeout=300, extra_env=None, worker="_comm_worker.py"):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", MTD_COMM_TIMEOUT_MS="3000")
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, worker), str(n_global), str(steps)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s\n%s" % (r, so[-2000:], se[-4000:])
    line = [l for l in outs[0][0].splitlines() if l.startswith("{")][-1]
    return json.loads(line)


@pytest.mark.parametrize("n_cv,dtype", [(2, np.float32), (1, np.float64), (3, np.float32), (3, np.float64)])
def test_mailbox_single_rank_matches_plain_fused_step(abi, n_cv, dtype):
    """world = 1: the send / receive code of the fused kernels with the rank talking to itself"""
    import ctypes as C
    from metadynamics import xgmi
    from metadynamics.sharded import HipLamellarBackend
    lib = abi.load()
    h = C.c_void_p()
    abi.check(lib.mtd_comm_create(C.byref(h), 0, 1, 8))
    box = xgmi.Mailbox(h, 0, 1)
    v = torch.tensor([1.5, -2.0, 3.25], dtype=torch.float64, device="cuda")
    for _ in range(4):
        box.all_reduce(v)
    assert v.cpu().tolist() == [1.5, -2.0, 3.25]
    N, L = 50000, 30.0
    pos, types = util.snapshot_random(N, L, seed=9, modulated=True, dtype=np.float32)
    grid = dict(sigma=[0.02, 0.02, 0.03][:n_cv], cv_min=[-1.0] * n_cv, cv_max=[1.0] * n_cv, num_points=[64, 48, 20][:n_cv])
    kw = dict(W=1.0, T_shift=7.0, T=1.0, stride=2, mode="well_tempered")
    cvs = [(util.CV1_VECTORS, util.MODE_AB), (util.CV2_VECTORS, util.MODE_AB), ([(0, 0, 3), (1, 2, 0)], [0.5, -1.5])][:n_cv]
    d = torch.from_numpy(util.pack_postype(pos.astype(dtype), types, dtype)).cuda()
    a = HipLamellarBackend(cvs, d, N, L, grid, fast_trig=False, **kw)
    b = HipLamellarBackend(cvs, d, N, L, grid, fast_trig=False, **kw)
    b.attach_mailbox(box)
    for t in range(7):
        a.step_single(t)
        b.step_single(t)
        sa, sb = a.state(), b.state()
        assert np.allclose(sb["cv"], sa["cv"], rtol=1e-13, atol=0)
        assert np.allclose(sb["bias"], sa["bias"], rtol=1e-9, atol=1e-12 * max(abs(x) for x in sa["bias"]))
        assert sb["V"] == pytest.approx(sa["V"], rel=1e-10) and sb["w"] == pytest.approx(sa["w"], rel=1e-10)
        assert sb["num_gaussians"] == sa["num_gaussians"]
    for c in range(n_cv):
        fa, fb = a.forces[c].cpu().numpy(), b.forces[c].cpu().numpy()
        assert np.abs(fa - fb).max() <= 1e-6 * np.abs(fa).max()
    assert box.timeouts() == 0
    b.attach_mailbox(None)
    a.close(); b.close(); box.close()


def test_nan_positions_are_not_a_communication_failure(abi):
    """a diverged simulation (a NaN position) gives NaN CV sums; with a mailbox attached that used to be read as an expired wait
    (poisoned step, nothing deposited, the cause misreported).  An expired wait hands on a NaN with a payload of its own
    (comm_device.hpp); an arithmetic NaN goes through the step exactly as without a mailbox — the reference deposits it too"""
    import ctypes as C
    from metadynamics import xgmi
    from metadynamics.sharded import HipLamellarBackend
    lib = abi.load()
    h = C.c_void_p()
    abi.check(lib.mtd_comm_create(C.byref(h), 0, 1, 8))
    box = xgmi.Mailbox(h, 0, 1)
    N, L = 20000, 30.0
    pos, types = util.snapshot_random(N, L, seed=9, modulated=True, dtype=np.float32)
    pos[1234, 1] = np.nan
    grid = dict(sigma=[0.02, 0.02], cv_min=[-1.0] * 2, cv_max=[1.0] * 2, num_points=[64, 48])
    kw = dict(W=1.0, T_shift=7.0, T=1.0, stride=1, mode="well_tempered")
    cvs = [(util.CV1_VECTORS, util.MODE_AB), (util.CV2_VECTORS, util.MODE_AB)]
    d = torch.from_numpy(util.pack_postype(pos, types, np.float32)).cuda()
    a = HipLamellarBackend(cvs, d, N, L, grid, fast_trig=False, **kw)
    b = HipLamellarBackend(cvs, d, N, L, grid, fast_trig=False, **kw)
    b.attach_mailbox(box)
    for t in range(3):
        a.step_single(t)
        b.step_single(t)                                              # (raises on MTD_ERR_COMM_TIMEOUT)
    sa, sb = a.state(), b.state()
    assert all(np.isnan(x) for x in sa["cv"]) and all(np.isnan(x) for x in sb["cv"])
    assert sb["num_gaussians"] == sa["num_gaussians"] == 3            # deposited like the plain path (and the reference)
    assert box.timeouts() == 0
    G = lib.mtd_metad_num_elements(a.h)
    ga, gb = np.zeros(G), np.zeros(G)
    abi.check(lib.mtd_metad_get_array(a.h, 0, ga.ctypes.data, None))
    abi.check(lib.mtd_metad_get_array(b.h, 0, gb.ctypes.data, None))
    assert np.array_equal(np.isnan(ga), np.isnan(gb)) and np.isnan(ga).any()
    b.attach_mailbox(None)
    a.close(); b.close(); box.close()


def test_mailbox_rejects_bad_arguments(abi):
    import ctypes as C
    lib = abi.load()
    h = C.c_void_p()
    assert lib.mtd_comm_create(C.byref(h), 0, 9, 8) == -1          # more ranks than one node has GPUs
    assert lib.mtd_comm_create(C.byref(h), 2, 2, 8) == -1
    assert lib.mtd_comm_create(C.byref(h), 0, 2, 0) == -1
    abi.check(lib.mtd_comm_create(C.byref(h), 0, 2, 8))
    v = torch.zeros(4, dtype=torch.float64, device="cuda")
    assert lib.mtd_comm_allreduce_small(h, v.data_ptr(), 4, None) == -1      # not connected yet
    abi.check(lib.mtd_comm_destroy(h))


@pytest.mark.parametrize("world", [2, 3, 4])
def test_mailbox_between_processes(world):
    r = _run_world(world, 60000)
    assert r["connected"], "the mailbox could not be set up between processes on this box"
    assert r["timeouts"] == 0
    assert r["allreduce_max_err"] < 1e-13
    assert r["bitwise_same_on_all_ranks"] and r["replicated_state_bitwise"]
    assert r["num_gaussians"] == 6
    # fp32 per-thread partial sums: the sharding shows at 1e-8 in the CV (cf. test_gpu_sharded)
    assert r["errs"]["cv"] < 1e-6 and r["errs"]["V"] < 1e-5 and r["errs"]["w"] < 1e-5
    assert r["errs"]["bias"] < 1e-5
    assert r["force_rel_err"] < 1e-5
    # mesh CV with the mesh decomposed into slabs over the ranks, against the whole mesh on one rank
    for key, v in r["slab"].items():
        assert v["timeouts"] == 0, key
        assert v["cv_rel"] < 1e-9, (key, v)
        assert v["force_rel"] < 1e-7 and v["force_max"] > 1e-12, (key, v)
        # ... and against the oracle on the whole snapshot (stated tolerances: CV 1e-6, forces 1e-5 of max|F|)
        assert v["cv_rel_oracle"] < 1e-8, (key, v)
        assert v["force_rel_oracle"] < 1e-5, (key, v)
    assert "16x24x24" in r["slab"]
    # host classes, domain decomposed: one hills file (header + prepRun deposit + 4 steps), written by the root rank alone
    hd = r["host_dd"]
    assert hd["fused"] and hd["hills_lines"] == 6 and hd["hills_ok"], hd
    assert "hills.log" in hd["files"] and len(hd["dump_lines"]) >= 1 and len(hd["files"]) == 1 + len(hd["dump_lines"]), hd
    assert all(n == 4 + 32 * 32 for n in hd["dump_lines"]), hd           # three header lines, the column names, one line per cell
    # generic CV set through the stand-alone mailbox all-reduce
    assert r["timeouts_set"] == 0
    assert r["set_errs"]["cv"] < 1e-6 and r["set_errs"]["bias"] < 1e-5
    assert r["set_energy_cv"][0] == pytest.approx(r["set_energy_cv"][1], rel=1e-12)
    assert r["set_force_err"] < 1e-5


def test_host_classes_take_the_mailbox_as_communicator(abi):
    """metadynamics.cv / integrate over the C++ host classes with a (one-rank) mailbox in the execution configuration:
    same step as without; a CV set that cannot take the fused step goes CV by CV (round 4: every variable reduces its own sums over
    the ranks — tests/test_gpu_host_dd.py runs that between processes)"""
    import ctypes as C
    from metadynamics import context, cv, integrate, xgmi
    lib = abi.load()
    N, L = 30000, 30.0
    pos, types = util.snapshot_random(N, L, seed=21, modulated=True, dtype=np.float32)

    def run(box, umbrella=False):
        context.initialize(pos, types, ["A", "B"], L, dtype=np.float32, n_global=N)
        if box is not None:
            context.exec_conf.setMailbox(box.handle.value)
            assert context.exec_conf.getNRanks() == 1 and context.exec_conf.getRank() == 0
        meta = integrate.mode_metadynamics(dt=0.005, stride=1, mode="well_tempered", W=1.0, deltaT=7.0, T=1.0)
        cvs = []
        for i, vecs in enumerate((util.CV1_VECTORS, util.CV2_VECTORS)):
            c = cv.lamellar(sigma=0.02, mode=dict(A=1.0, B=-1.0), lattice_vectors=vecs, name="c%d" % i)
            c.set_grid(-1.0, 1.0, 48)
            cvs.append(c)
        if umbrella:
            cvs[0].set_params(umbrella="harmonic", kappa=1.0, cv0=0.0)
        context.run(5)
        t = context.current.system.getCurrentTimeStep()
        integ = meta.cpp_integrator
        out = dict(cv=[c.cpp_force.getCurrentValue(t) for c in cvs], V=integ.getLogValue("bias", t), w=integ.getLogValue("weight", t),
                   n=integ.getNumGaussians(), fused=integ.usedFusedPath(), f=cvs[0].cpp_force.getForces().copy())
        context.current = None
        return out

    plain = run(None)
    h = C.c_void_p()
    abi.check(lib.mtd_comm_create(C.byref(h), 0, 1, 8))
    box = xgmi.Mailbox(h, 0, 1)
    try:
        dd = run(box)
        assert dd["fused"] and plain["fused"] and dd["n"] == plain["n"]
        assert np.allclose(dd["cv"], plain["cv"], rtol=1e-13, atol=0)
        assert dd["V"] == pytest.approx(plain["V"], rel=1e-10) and dd["w"] == pytest.approx(plain["w"], rel=1e-10)
        assert np.abs(dd["f"] - plain["f"]).max() <= 1e-6 * np.abs(plain["f"]).max()
        assert box.timeouts() == 0
        plain_u, dd_u = run(None, umbrella=True), run(box, umbrella=True)
        assert not dd_u["fused"] and not plain_u["fused"] and dd_u["n"] == plain_u["n"]
        assert np.allclose(dd_u["cv"], plain_u["cv"], rtol=1e-12, atol=0)
        assert dd_u["V"] == pytest.approx(plain_u["V"], rel=1e-9)
        assert np.abs(dd_u["f"] - plain_u["f"]).max() <= 1e-6 * np.abs(plain_u["f"]).max()
        assert box.timeouts() == 0
    finally:
        context.current = None
        box.close()


def test_mailbox_setup_failure_is_agreed_by_all_ranks():
    """one rank cannot map its peers: every rank gets None back from xgmi.connect (and keeps the collective path)"""
    r = _run_world(2, 1000, extra_env={"MTD_XGMI_TEST_FAIL_RANK": "1"})
    assert r["connected"] is False


def test_slab_close_then_whole_mesh_sequence(abi, ref):
    """Regression for the round-1 memory access fault (gpurun_out/dbg_slab.log): whole mesh -> one-rank slab mesh over exported
    (uncached) buffers -> slab close -> comm destroy -> next whole mesh, several sizes in a row so that later meshes land on
    addresses earlier buffers used.  Every CV and the forces are checked against the ORACLE (ref_mesh_*,
    OrderParameterMesh.cc:517-968), not against the other HIP path.  Cause and fix: comm.hip (uncached pool), DESIGN.md §6."""
    import ctypes as C
    from test_gpu_mesh import GpuMesh
    lib = abi.load()
    rng = np.random.default_rng(5)
    N, Ls, mode = 4000, (9.0, 11.0, 7.5), [1.0, -0.6]
    box, rbox = abi.Box.make(Ls), ref.Box.make(Ls)
    for it in range(8):
        dims = [(48, 48, 32), (16, 24, 24), (32, 32, 32), (20, 12, 8)][it % 4]
        pos = ((rng.random((N, 3)) - 0.5) * np.array(Ls)).astype(np.float32)
        types = rng.integers(0, 2, N).astype(np.int32)
        d_pos = torch.from_numpy(util.pack_postype(pos, types, np.float32)).cuda()
        opt = util.oracle_postype(pos, types)
        r = ref.Mesh(*dims, mode)
        s_ref = r.cv(opt, rbox)
        F_ref = r.forces(opt, rbox, 0.8)
        fm = np.abs(F_ref[:, :3]).max()
        whole = GpuMesh(abi, dims, mode, N)
        s_whole = whole.cv(d_pos, abi.MTD_F32, box, N)
        F_whole = whole.forces(d_pos, abi.MTD_F32, box, N, 0.8)
        whole.close()
        assert s_whole == pytest.approx(s_ref, rel=1e-9), (it, dims)
        assert np.abs(F_whole[:, :3] - F_ref[:, :3]).max() <= 1e-5 * fm, (it, dims)
        h = C.c_void_p()
        abi.check(lib.mtd_comm_create(C.byref(h), 0, 1, 8))
        slab = GpuMesh(abi, dims, mode, N)
        sizes = (C.c_size_t * 4)()
        abi.check(lib.mtd_mesh_slab_bytes(slab.h, 1, sizes))
        peers = []
        for k in range(4):
            local, slot, hd = C.c_void_p(), C.c_uint(), (C.c_ubyte * 64)()
            abi.check(lib.mtd_comm_share(h, sizes[k], C.byref(local), C.byref(slot), hd))
            pp = (C.c_void_p * 1)()
            abi.check(lib.mtd_comm_open(h, slot.value, None, pp))
            peers.append(pp)
        abi.check(lib.mtd_mesh_slab_attach(slab.h, h, peers[0], peers[1], peers[2], peers[3]))
        for rep in range(2):                                             # twice: the exported buffers are re-used between steps
            cv_sum = C.c_void_p()
            abi.check(lib.mtd_mesh_slab_compute_cv(slab.h, N, abi.ptr(d_pos), abi.MTD_F32, C.byref(box), N, C.byref(cv_sum), None))
            out = torch.zeros(1, dtype=torch.float64, device="cuda")
            abi.check(lib.mtd_reduce_partials(cv_sum.value, 1, 1, 1, 0.5, 0.0, out.data_ptr(), None))
            torch.cuda.synchronize()
            assert out.item() == pytest.approx(s_ref, rel=1e-9), (it, dims, rep)
        F_slab = slab.forces(d_pos, abi.MTD_F32, box, N, 0.8)
        assert np.abs(F_slab[:, :3] - F_ref[:, :3]).max() <= 1e-5 * fm, (it, dims)
        n_to = C.c_uint()
        abi.check(lib.mtd_comm_status(h, C.byref(n_to), None))
        assert n_to.value == 0
        slab.close()
        abi.check(lib.mtd_comm_destroy(h))


def test_mailbox_timeout_is_fatal():
    """A peer that never sends: the bounded wait expires (MTD_COMM_TIMEOUT_MS = 200) and the step is POISONED, not continued on
    a stale sum — NaN CV values, bias factors and forces, nothing deposited (the grid keeps the hills it had), the deferred pass
    skipped; the status is sticky: mtd_metad_get_state and every later call that would use the mailbox return
    MTD_ERR_COMM_TIMEOUT (-4).  Reference semantics: a failed MPI_Allreduce aborts (LamellarOrderParameterGPU.cc:69-77)."""
    r = _run_world(2, 0, extra_env={"MTD_COMM_TIMEOUT_MS": "200"}, worker="_comm_timeout_worker.py")
    assert r["connected"]
    assert r["rc_launch"] == [0, 0]                     # the failing launches themselves are asynchronous
    assert r["timeouts"] > 0
    assert r["cv_nan"] and r["bias_nan"] and r["V_nan"] and r["forces_all_nan"]
    assert r["rc_state"] == -4 and r["rc_next"] == -4 and r["rc_allreduce"] == -4
    assert r["rc_grid"] == 0 and r["grid_finite"] and r["grid_max"] > 0       # the good step's hill, nothing else
    assert r["num_gaussians"] == [1, 1]
    assert "mailbox" in r["status_string"]


def test_rccl_allreduce_large_and_walker_update(abi, ref):
    """mtd_rccl_* / mtd_comm_allreduce_large / mtd_metad_update_bias_walkers (rccl.hip): RCCL bound at run time, communicator from
    a caller-supplied unique id.  One rank here (RCCL refuses two ranks on one device): the call sequence, both element types,
    and the walker update (IntegratorMetaDynamics.cc:363-451 with m_multiple_walkers) against the oracle's grid engine — with
    one walker the sums over walkers are the walker's own increments."""
    import ctypes as C
    lib = abi.load()
    uid = (C.c_ubyte * 128)()
    abi.check(lib.mtd_rccl_unique_id(uid))
    r = C.c_void_p()
    abi.check(lib.mtd_rccl_create(C.byref(r), uid, 0, 1))
    assert lib.mtd_rccl_world(r) == 1 and lib.mtd_rccl_rank(r) == 0
    v = torch.arange(100000, dtype=torch.float64, device="cuda") * 0.5
    u = torch.arange(7777, dtype=torch.int32, device="cuda")
    abi.check(lib.mtd_comm_allreduce_large(r, v.data_ptr(), v.numel(), 1, None))
    abi.check(lib.mtd_comm_allreduce_large(r, u.data_ptr(), u.numel(), 2, None))
    torch.cuda.synchronize()
    assert torch.equal(v.cpu(), torch.arange(100000, dtype=torch.float64) * 0.5) and torch.equal(u.cpu(), torch.arange(7777, dtype=torch.int32))
    assert lib.mtd_comm_allreduce_large(r, v.data_ptr(), 10, 7, None) == -1
    from test_gpu_metad import GpuMetad, compare
    kw = dict(sigma=[0.05, 0.1], cv_min=[-1.0, 0.0], cv_max=[1.0, 2.0], num_points=[33, 20], W=0.7, T_shift=5.0, T=1.3, stride=2, mode="well_tempered")
    g = GpuMetad(abi, **kw)
    o = ref.Metad(**kw)
    try:
        rng = np.random.default_rng(3)
        for t in range(9):
            vals = [rng.uniform(-0.8, 0.8), rng.uniform(0.2, 1.8)]
            for c, x in enumerate(vals):
                abi.check(lib.mtd_metad_set_cv_value(g.h, c, float(x)))
            abi.check(lib.mtd_metad_update_bias_walkers(g.h, r, t, None))
            b = o.update_bias(t, vals)
            compare(g, o, b, label="walker step %d" % t)
    finally:
        g.close()
    abi.check(lib.mtd_rccl_destroy(r))


def test_host_multiple_walkers(abi):
    """integrate.set_params(multiple_walkers=True) through the C++ host classes: without a walker communicator the step
    is the only walker (the reference in a serial run, test/test_2d.py:29; a notice says so — it used to be a silent no-op that
    kept the fused path); with one (a one-walker RCCL communicator) the increments go through the all-reduce; both equal the plain run"""
    import ctypes as C
    from metadynamics import context, cv, integrate
    lib = abi.load()
    N, L = 20000, 30.0
    pos, types = util.snapshot_random(N, L, seed=21, modulated=True, dtype=np.float32)

    def run(walkers, comm):
        context.initialize(pos, types, ["A", "B"], L, dtype=np.float32)
        if comm is not None:
            context.exec_conf.setWalkerCommunicator(comm.value)
        meta = integrate.mode_metadynamics(dt=0.005, stride=1, mode="well_tempered", W=1.0, deltaT=7.0, T=1.0)
        cvs = []
        for i, vecs in enumerate((util.CV1_VECTORS, util.CV2_VECTORS)):
            c = cv.lamellar(sigma=0.02, mode=dict(A=1.0, B=-1.0), lattice_vectors=vecs, name="c%d" % i)
            c.set_grid(-1.0, 1.0, 40)
            cvs.append(c)
        if walkers:
            meta.set_params(multiple_walkers=True)
        try:
            context.run(4)
            t = context.current.system.getCurrentTimeStep()
            integ = meta.cpp_integrator
            return dict(V=integ.getLogValue("bias", t), w=integ.getLogValue("weight", t), n=integ.getNumGaussians(),
                        fused=integ.usedFusedPath(), f=cvs[0].cpp_force.getForces().copy())
        finally:
            context.current = None

    plain = run(False, None)
    alone = run(True, None)                    # no communicator: a single walker (with a notice), the walker code path
    assert not alone["fused"] and alone["n"] == plain["n"] and alone["V"] == pytest.approx(plain["V"], rel=1e-6)
    uid = (C.c_ubyte * 128)()
    abi.check(lib.mtd_rccl_unique_id(uid))
    r = C.c_void_p()
    abi.check(lib.mtd_rccl_create(C.byref(r), uid, 0, 1))
    try:
        w = run(True, r)
        assert plain["fused"] and not w["fused"] and w["n"] == plain["n"]
        assert w["V"] == pytest.approx(plain["V"], rel=1e-6) and w["w"] == pytest.approx(plain["w"], rel=1e-6)
        assert np.abs(w["f"] - plain["f"]).max() <= 1e-5 * np.abs(plain["f"]).max()
    finally:
        abi.check(lib.mtd_rccl_destroy(r))


@pytest.mark.parametrize("n_cv", [1, 2, 3])
def test_one_launch_step_with_mailbox_single_rank(abi, ref, n_cv):
    """mtd_fused_step as the persistent kernel with a (one-rank) mailbox attached: block 0 collects the blocks' sums and sends
    them, every block's chain polls the local mailbox — same step as without the mailbox, and against the oracle"""
    import ctypes as C
    from metadynamics import xgmi
    from test_gpu_fused import Fused, make_traj
    from test_gpu_metad import GpuMetad, compare
    lib = abi.load()
    N, L, steps = 100_003, 30.0, 5
    cvs = [(util.CV1_VECTORS, util.MODE_AB), (util.CV2_VECTORS, util.MODE_AB), ([(0, 0, 3), (1, 2, 0)], [0.5, -1.5])][:n_cv]
    traj, types = make_traj(N, L, steps, np.float32)
    box, rbox = abi.Box.make(L), ref.Box.make(L)
    kw = dict(sigma=[0.02, 0.01, 0.03][:n_cv], cv_min=[-0.6, -0.3, -0.5][:n_cv], cv_max=[0.4, 0.3, 0.5][:n_cv], num_points=[37, 21, 9][:n_cv],
              W=1.0, T_shift=7.0, T=1.0, stride=1, mode="well_tempered")
    h = C.c_void_p()
    abi.check(lib.mtd_comm_create(C.byref(h), 0, 1, 8))
    mbox = xgmi.Mailbox(h, 0, 1)
    g = GpuMetad(abi, **kw)
    r = ref.Metad(**kw)
    try:
        abi.check(lib.mtd_metad_set_comm(g.h, mbox.handle))
        f = Fused(abi, g, N, np.float32, True, cvs)
        for t in range(steps):
            d_pos = torch.from_numpy(util.pack_postype(traj[t], types, np.float32)).cuda()
            f.step(t, d_pos, box)
            torch.cuda.synchronize()
            F = [x.cpu().numpy().astype(np.float64) for x in f.forces]
            st = g.state()
            opt = util.oracle_postype(traj[t], types)
            s_ref = [ref.lamellar_cv(v, opt, m, rbox) for v, m in cvs]
            for c in range(n_cv):
                assert abs(st["cv"][c] - s_ref[c]) <= max(1e-6 * abs(s_ref[c]), 1e-6 * 8 / np.sqrt(N)), (t, c)
            b = r.update_bias(t, st["cv"])
            compare(g, r, b, label="one launch + mailbox step %d" % t)
            for c, (v, m) in enumerate(cvs):
                F_ref = ref.lamellar_forces(v, opt, m, rbox, b[c])
                assert np.abs(F[c][:, :3] - F_ref[:, :3]).max() <= 1e-5 * np.abs(F_ref[:, :3]).max(), (t, c)
        assert mbox.timeouts() == 0
        abi.check(lib.mtd_metad_set_comm(g.h, None))
    finally:
        g.close()
        mbox.close()
