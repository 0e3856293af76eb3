"""GPU: the reference's Python API (metadynamics.cv / metadynamics.integrate) over the C++ host classes,
checked against the oracle — including the reference's own test/test_2d.py scenario end to end."""
import os

import numpy as np
import pytest

import util

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu


@pytest.fixture()
def api():
    from metadynamics import context, cv, integrate
    yield context, cv, integrate
    context.current = None


def _loadgrid(path):
    return np.loadtxt(path, skiprows=4)


def test_reference_test_2d_scenario(api, ref, tmp_path):
    """test/test_2d.py: N=1, V=10; density (sigma .25, [0,1]x20) + aspect_ratio (sigma .1, [0,2]x30); well-tempered,
    stride 1, deltaT=1, W=1; dump every step; run(1); box volume x0.125; run(1); then a fresh context restarted from
    bias.dat_1.  "bias.restart_0.dat and bias.dat_2 should be identical up to rounding errors" (test_2d.py:1-2)."""
    context, cv, integrate = api
    L1 = 10 ** (1.0 / 3.0)
    s = 0.125 ** (1.0 / 3.0)
    os.chdir(tmp_path)

    def block(restart):
        context.initialize(np.zeros((1, 3)), [0], ["A"], L1, dtype=np.float64)
        meta = integrate.mode_metadynamics(dt=0.005, mode="well_tempered", stride=1, deltaT=1, W=1)
        density = cv.density(sigma=0.25)
        density.set_grid(cv_min=0, cv_max=1, num_points=20)
        aspect = cv.aspect_ratio(sigma=0.1, dir1=0, dir2=1)
        aspect.set_grid(cv_min=0, cv_max=2, num_points=30)
        pdata = context.current.system_definition.getParticleData()
        if not restart:
            meta.dump_grid("bias.dat", period=1)
            meta.set_params(multiple_walkers=True)
            context.run(1)
            pdata.setGlobalBox(pdata.getGlobalBox().scale(s))
            context.run(1)
        else:
            meta.restart_from_grid("bias.dat_1")
            meta.dump_grid("bias_restart.dat", period=1)
            meta.set_params(multiple_walkers=True)
            pdata.setGlobalBox(pdata.getGlobalBox().scale(s))
            context.run(1)
        assert not meta.cpp_integrator.usedFusedPath()     # box CVs go through the generic path
        return meta

    block(False)
    block(True)

    # the same call sequence on the oracle
    kw = dict(sigma=[0.25, 0.1], cv_min=[0.0, 0.0], cv_max=[1.0, 2.0], num_points=[20, 30], W=1.0, T_shift=1.0, T=1.0,
              stride=1, mode="well_tempered")
    names = ["cv_density_all", "cv_aspect_ratio"]
    b1, b2 = ref.Box.make(L1), ref.Box.make(L1 * s)
    v1 = [ref.density(b1, 1), ref.aspect_ratio(b1, 0, 1)]
    v2 = [ref.density(b2, 1), ref.aspect_ratio(b2, 0, 1)]
    odir = tmp_path / "oracle"
    odir.mkdir()
    a = ref.Metad(**kw)
    for t, v in ((0, v1), (1, v1), (1, v2), (2, v2)):    # prepRun(0), update(0); prepRun(1), update(1)  (Q17)
        a.update_bias(t, v)
        a.write_grid(str(odir / "bias.dat"), t, names)
    b = ref.Metad(**kw)
    b.read_grid(str(odir / "bias.dat_1"))
    for t in (0, 1):
        b.update_bias(t, v2)
        b.write_grid(str(odir / "bias_restart.dat"), t, names)

    for f in ("bias.dat_0", "bias.dat_1", "bias.dat_2", "bias_restart.dat_0", "bias_restart.dat_1"):
        got, want = _loadgrid(tmp_path / f), _loadgrid(odir / f)
        assert got.shape == (600, 8)
        assert np.allclose(got, want, rtol=1e-9, atol=1e-12), f
        assert open(tmp_path / f).read().splitlines()[:4] == open(odir / f).read().splitlines()[:4], f
    # the reference's own acceptance criterion
    assert np.allclose(_loadgrid(tmp_path / "bias_restart.dat_0"), _loadgrid(tmp_path / "bias.dat_2"), rtol=1e-8, atol=1e-12)
    assert open(tmp_path / "bias.dat_2").read().splitlines()[2] == "#num_gaussians: 4"


@pytest.mark.parametrize("fused", [True, False])
def test_config0b_through_api(api, ref, tmp_path, fused):
    """BASELINE.json configs[0]: 4096 particles, 1 lamellar CV [(0,0,4)], grid [-1,1]x128, sigma .05, W=1, dT=7, T=1,
    stride 1, 10 steps — CV, bias, weight and forces vs the oracle; fused and generic paths"""
    context, cv, integrate = api
    pos, types, L = util.snapshot_config0b()
    pos = pos.astype(np.float32)
    context.initialize(pos, types, ["A", "B"], L, dtype=np.float32)
    hills = str(tmp_path / "hills.log")
    meta = integrate.mode_metadynamics(dt=0.005, stride=1, mode="well_tempered", W=1.0, deltaT=7.0, T=1.0, filename=hills)
    lam = cv.lamellar(sigma=0.05, mode=dict(A=1.0, B=-1.0), lattice_vectors=[(0, 0, 4)])
    lam.set_grid(cv_min=-1.0, cv_max=1.0, num_points=128)
    meta.cpp_integrator.setFusedPath(fused)
    context.run(10)
    assert meta.cpp_integrator.usedFusedPath() == fused

    rbox = ref.Box.make(L)
    opt = util.oracle_postype(pos, types)
    s_ref = ref.lamellar_cv([(0, 0, 4)], opt, util.MODE_AB, rbox)
    r = ref.Metad([0.05], [-1.0], [1.0], [128], W=1.0, T_shift=7.0, T=1.0, stride=1, mode="well_tempered")
    for t in range(11):              # prepRun(0) + 10 updates (timesteps 1..10)
        b = r.update_bias(t, [s_ref])
    t_now = context.current.system.getCurrentTimeStep()
    assert t_now == 10
    # the forces of the last step were written with the last bias factor
    F = lam.cpp_force.getForceArray().astype(np.float64)
    F_ref = ref.lamellar_forces([(0, 0, 4)], opt, util.MODE_AB, rbox, b[0])
    assert np.abs(F[:, :3] - F_ref[:, :3]).max() <= 1e-5 * np.abs(F_ref[:, :3]).max()
    # log quantities: bias potential and reweighting factor (IntegratorMetaDynamics.h:161-189)
    assert meta.cpp_integrator.getLogValue("bias", t_now) == pytest.approx(r.curr_bias, rel=2e-6)
    assert meta.cpp_integrator.getLogValue("weight", t_now) == pytest.approx(r.curr_weight, rel=2e-6)
    assert meta.cpp_integrator.getLogValue("det_sigma", t_now) == pytest.approx(1 / 0.05)
    assert lam.cpp_force.getCurrentValue(t_now) == pytest.approx(s_ref, rel=1e-6)
    assert lam.cpp_force.getLogValue("cv_lamellar", t_now) == pytest.approx(s_ref, rel=1e-6)
    # hills file: header + one line per deposit (:98-119, :523-550)
    lines = open(hills).read().splitlines()
    assert lines[0].split("\t")[:3] == ["timestep", "W", "cv_lamellar"]
    assert len(lines) == 1 + 11
    assert int(lines[1].split("\t")[0]) == 0 and int(lines[-1].split("\t")[0]) == 10


def test_two_lamellar_cvs_and_umbrella_fallback(api, ref):
    """two fused lamellar CVs; adding an umbrella to one of them makes the integrator fall back to the generic
    path (the umbrella needs the host CV value, CollectiveVariable.cc:22-60) with the same grid evolution"""
    context, cv, integrate = api
    N, L = 20011, 30.0
    pos, types = util.snapshot_random(N, L, seed=8, modulated=True, dtype=np.float32)
    grids = {}
    for variant in ("fused", "umbrella"):
        context.initialize(pos, types, ["A", "B"], L, dtype=np.float32)
        meta = integrate.mode_metadynamics(dt=0.005, stride=2, mode="well_tempered", W=0.5, deltaT=5.0, T=1.0)
        c1 = cv.lamellar(sigma=0.02, mode=dict(A=1.0, B=-1.0), lattice_vectors=util.CV1_VECTORS, name="one")
        c1.set_grid(-0.6, 0.4, 40)
        c2 = cv.lamellar(sigma=0.01, mode=dict(A=1.0, B=-1.0), lattice_vectors=util.CV2_VECTORS, name="two")
        c2.set_grid(-0.3, 0.3, 30)
        if variant == "umbrella":
            c2.set_params(umbrella="harmonic", kappa=0.0, cv0=0.0)   # zero stiffness: no change in the bias factor
        context.run(5)
        assert meta.cpp_integrator.usedFusedPath() == (variant == "fused")
        meta.dump_grid("/tmp/_mtd_api_%s" % variant)
        grids[variant] = _loadgrid("/tmp/_mtd_api_%s_0" % variant)
        assert c1.cpp_force.getName() == "cv_lamellar_one"
    assert np.allclose(grids["fused"], grids["umbrella"], rtol=1e-9, atol=1e-14)
    assert grids["fused"][:, 2].max() > 0


def test_potential_energy_cv(api, ref):
    """cv.potential_energy (WellTemperedEnsemble): PE = sum net_force.w + external energy; net force, torque and
    virial scaled by 1 + dV/dE, torque.w too like the CPU path (Q18); net_force_first ordering (.cc:259-301)"""
    context, cv, integrate = api
    N = 5003
    rng = np.random.default_rng(4)
    pos = rng.random((N, 3)) * 10 - 5
    context.initialize(pos, np.zeros(N, dtype=int), ["A"], 10.0, dtype=np.float64)
    pdata = context.current.system_definition.getParticleData()
    nf = rng.normal(size=(N, 4)); nf[:, 3] = rng.normal(-2.0, 0.5, N)
    nt = rng.normal(size=(N, 4))
    nv = rng.normal(size=(6, N))
    pdata.setNetForce(nf); pdata.setNetTorque(nt); pdata.setNetVirial(nv)
    pdata.setExternalEnergy(12.5)
    for i in range(6):
        pdata.setExternalVirial(i, float(i + 1))
    pe_ref = ref.wte_potential_energy(nf, 12.5)
    meta = integrate.mode_metadynamics(dt=0.005, stride=1, mode="well_tempered", W=2.0, deltaT=50.0, T=1.0)
    pe = cv.potential_energy(sigma=40.0)
    pe.set_grid(cv_min=pe_ref - 300.0, cv_max=pe_ref + 300.0, num_points=64)
    assert pe.cpp_force.requiresNetForce()
    assert pe.cpp_force.getCurrentValue(0) == pytest.approx(pe_ref, rel=1e-13)
    context.run(1)
    # oracle: prepRun deposits at t=0 (bias evaluated, forces of the CV computed at t=0 are not applied to the net
    # force in prepRun: the WTE compute is not in the force list), update(0): updateBias(1) then cv.compute(0) scales.
    r = ref.Metad([40.0], [pe_ref - 300.0], [pe_ref + 300.0], [64], W=2.0, T_shift=50.0, T=1.0, stride=1, mode="well_tempered")
    r.update_bias(0, [pe_ref])
    b = r.update_bias(1, [pe_ref])
    f2, t2, v2, e2 = ref.wte_scale(nf, nt, nv.reshape(-1), N, [1, 2, 3, 4, 5, 6], b[0])
    assert np.allclose(pdata.getNetForce(), f2, rtol=1e-12, atol=1e-14)
    assert np.allclose(pdata.getNetTorque(), t2, rtol=1e-12, atol=1e-14)
    assert np.allclose(pdata.getNetVirial().reshape(-1), v2, rtol=1e-12, atol=1e-14)
    assert np.allclose([pdata.getExternalVirial(i) for i in range(6)], e2, rtol=1e-12)


def test_api_errors(api):
    context, cv, integrate = api
    context.initialize(np.zeros((4, 3)), [0, 1, 0, 1], ["A", "B"], 5.0)
    with pytest.raises(RuntimeError):
        cv.lamellar(mode=dict(A=1.0), lattice_vectors=[(0, 0, 1)])            # missing mode amplitude (cv.py:245-247)
    with pytest.raises(RuntimeError):
        cv.lamellar(mode=dict(A=1.0, B=-1.0), lattice_vectors=[])             # empty list (cv.py:232-234)
    with pytest.raises(RuntimeError):
        cv.lamellar(mode=[1.0, -1.0], lattice_vectors=[(0, 0, 1)])            # not a dict (cv.py:236-238)
    with pytest.raises(RuntimeError):
        integrate.mode_metadynamics(dt=0.005, stride=1, mode="flux_tempered")  # integrate.py:214-216
    context.current.forces.clear()
    meta = integrate.mode_metadynamics(dt=0.005, stride=1)
    c = cv.lamellar(mode=dict(A=1.0, B=-1.0), lattice_vectors=[(0, 0, 1)])
    c.set_grid(cv_min=1.0, cv_max=1.0, num_points=10)
    with pytest.raises(RuntimeError):
        context.run(1)                                                         # cv_min >= cv_max (.cc:800-805)
    with pytest.raises(RuntimeError):
        c.set_params(umbrella="bogus")


def test_mesh_cv_with_umbrella_through_api(api, ref):
    """test/test_mesh.py's set-up in miniature: cv.mesh under a harmonic umbrella (no metadynamics grid): the force is
    -(dU/ds) grad s with dU/ds = kappa (s - cv0) (CollectiveVariable.cc:43-50) and umbrella_energy_mesh = kappa/2 (s-cv0)^2;
    plus cv.mesh registered with the bias grid next to a lamellar CV (config 3's CV set, generic path)"""
    context, cv, integrate = api
    N, L = 8000, 20.0
    pos, types = util.snapshot_random(N, L, seed=21, modulated=True, dtype=np.float64)
    rbox = ref.Box.make(L)
    opt = util.oracle_postype(pos, types)
    r = ref.Mesh(16, 16, 16, [1.0, -1.0])
    s_ref = r.cv(opt, rbox)

    context.initialize(pos, types, ["A", "B"], L, dtype=np.float64)
    meta = integrate.mode_metadynamics(dt=0.005, stride=1, mode="well_tempered", W=1.0, deltaT=7.0, T=1.0)
    mesh = cv.mesh(nx=16, mode={"A": 1.0, "B": -1.0})
    cv0 = 0.5 * s_ref
    kappa = 10.0 / cv0 ** 2
    mesh.set_params(umbrella="harmonic", cv0=cv0, kappa=kappa)
    context.run(2)
    t = context.current.system.getCurrentTimeStep()
    assert mesh.cpp_force.getCurrentValue(t) == pytest.approx(s_ref, rel=1e-9)
    assert mesh.cpp_force.getLogValue("cv_mesh", t) == pytest.approx(s_ref, rel=1e-9)
    dU = ref.umbrella_bias("harmonic", s_ref, 0.0, cv0, kappa, 0.0, 1.0)
    assert mesh.cpp_force.getLogValue("umbrella_energy_mesh", t) == pytest.approx(ref.umbrella_energy("harmonic", s_ref, cv0, kappa, 0.0, 1.0), rel=1e-8)
    F = mesh.cpp_force.getForceArray()
    F_ref = r.forces(opt, rbox, dU)
    assert np.abs(F[:, :3] - F_ref[:, :3]).max() <= 1e-7 * np.abs(F_ref[:, :3]).max()

    # config 3's CV set: one lamellar CV + the mesh CV on one 2-d grid (generic path, both CVs device resident)
    context.initialize(pos, types, ["A", "B"], L, dtype=np.float64)
    meta = integrate.mode_metadynamics(dt=0.005, stride=1, mode="well_tempered", W=1.0, deltaT=7.0, T=1.0)
    lam = cv.lamellar(sigma=0.02, mode=dict(A=1.0, B=-1.0), lattice_vectors=util.CV1_VECTORS)
    lam.set_grid(-0.6, 0.4, 32)
    mesh = cv.mesh(nx=16, mode={"A": 1.0, "B": -1.0}, sigma=0.05 * abs(s_ref))
    # the CV must not sit symmetrically between two nodes (there dV/ds is pure rounding noise)
    mesh.set_grid(0.25 * s_ref, 1.6 * s_ref, 40)
    context.run(3)
    assert not meta.cpp_integrator.usedFusedPath()
    s_lam = ref.lamellar_cv(util.CV1_VECTORS, opt, util.MODE_AB, rbox)
    cv_gpu = meta.cpp_integrator.getCurrentValues()
    assert cv_gpu[0] == pytest.approx(s_lam, rel=1e-6) and cv_gpu[1] == pytest.approx(s_ref, rel=1e-9)
    g = ref.Metad([0.02, 0.05 * abs(s_ref)], [-0.6, 0.25 * s_ref], [0.4, 1.6 * s_ref], [32, 40], W=1.0, T_shift=7.0, T=1.0,
                  stride=1, mode="well_tempered")
    for tt in range(4):
        b = g.update_bias(tt, cv_gpu)          # oracle grid driven with the device's CV values
    t = context.current.system.getCurrentTimeStep()
    assert meta.cpp_integrator.getLogValue("bias", t) == pytest.approx(g.curr_bias, rel=1e-9)
    assert np.allclose(meta.cpp_integrator.getBiasFactors(), b, rtol=1e-7)
    F = mesh.cpp_force.getForceArray()
    F_ref = r.forces(opt, rbox, b[1])
    assert np.abs(F[:, :3] - F_ref[:, :3]).max() <= 1e-5 * np.abs(F_ref[:, :3]).max()
    F = lam.cpp_force.getForceArray()
    F_ref = ref.lamellar_forces(util.CV1_VECTORS, opt, util.MODE_AB, rbox, b[0])
    assert np.abs(F[:, :3] - F_ref[:, :3]).max() <= 1e-5 * np.abs(F_ref[:, :3]).max()
    with pytest.raises(RuntimeError):
        cv.mesh(nx=300, mode={"A": 1.0, "B": -1.0})           # neither <= 256 nor a power of two


def test_steinhardt_through_api(api, ref):
    """cv.steinhardt on a noisy fcc crystal registered with the bias grid: CV, log quantities (incl. the reference's
    off-by-one steinhardt_Q{l} = Ql[l-1], Q19), bias and forces vs the oracle"""
    context, cv, integrate = api
    pos, L = util.fcc_lattice(5)
    rng = np.random.default_rng(12)
    pos = pos + rng.normal(0, 0.05, pos.shape)
    N = len(pos)
    types = np.zeros(N, dtype=np.int32)
    context.initialize(pos, types, ["A"], L, dtype=np.float64)
    meta = integrate.mode_metadynamics(dt=0.005, stride=1, mode="well_tempered", W=1.0, deltaT=7.0, T=1.0)
    nl = cv.nlist_cell(r_cut=1.5)
    lists = nl.update()
    ref_lists = util.build_nlist(pos, L, 1.5)
    assert all(np.array_equal(a, b) for a, b in zip(lists, ref_lists))
    Ql_ref = [0, 0, 0, 0, 1, 0, 1]
    rbox = ref.Box.make(L)
    pt = util.oracle_postype(pos, types)
    val, Qlm, Ql = ref.ql_compute_cv(pt, rbox, *lists, 1.4, 1.2, 6, 0, Ql_ref)
    st = cv.steinhardt(r_cut=1.4, r_on=1.2, lmax=6, Ql_ref=Ql_ref, nlist=nl, type="A", sigma=0.02 * val)
    st.set_grid(0.55 * val, 1.3 * val, 64)
    context.run(2)
    t = context.current.system.getCurrentTimeStep()
    assert st.cpp_force.getCurrentValue(t) == pytest.approx(val, rel=1e-10)
    assert st.cpp_force.getLogValue("cv_steinhardt", t) == pytest.approx(val, rel=1e-10)
    assert st.cpp_force.getLogValue("steinhardt_Q6", t) == pytest.approx(Ql[5], abs=1e-12)     # off by one (Q19)
    assert st.cpp_force.getLogValue("steinhardt_Q5", t) == pytest.approx(Ql[4], rel=1e-10)
    g = ref.Metad([0.02 * val], [0.55 * val], [1.3 * val], [64], W=1.0, T_shift=7.0, T=1.0, stride=1, mode="well_tempered")
    cvg = meta.cpp_integrator.getCurrentValues()
    for tt in range(3):
        b = g.update_bias(tt, cvg)
    assert np.allclose(meta.cpp_integrator.getBiasFactors(), b, rtol=1e-7)
    F = st.cpp_force.getForceArray()
    F_ref = ref.ql_compute_forces(pt, rbox, *lists, 1.4, 1.2, 6, 0, Ql_ref, Qlm, b[0])
    assert np.abs(F[:, :3] - F_ref[:, :3]).max() <= 1e-7 * np.abs(F_ref[:, :3]).max()
    with pytest.raises(RuntimeError):
        cv.steinhardt(r_cut=1.4, r_on=1.2, lmax=6, Ql_ref=[1, 2, 3], nlist=nl, type="A")      # SteinhardtQl.cc:25-29
    with pytest.raises(RuntimeError):
        cv.steinhardt(r_cut=1.4, r_on=1.2, lmax=6, Ql_ref=Ql_ref, nlist=nl, type="Z")         # cv.py:591-593


def test_steinhardt_half_storage_through_host_classes(api, ref):
    """a NeighborList in HALF storage mode (HOOMD's CPU default; cv.steinhardt asks for full storage like the reference does on a
    GPU, so the mode is set on the C++ object as a C++ caller would): the host class turns the half list into the symmetric full
    list it stands for once per list update (mtd_ql_symmetrize_half_list) — CV and forces are the reference's HALF-list results
    (SteinhardtQl.cc:80, 173-179, 328-333), the force pass uses no atomics"""
    context, cv, integrate = api
    from metadynamics import _metadynamics
    pos, L = util.fcc_lattice(5)
    pos = pos + np.random.default_rng(21).normal(0, 0.05, pos.shape)
    N = len(pos)
    types = np.zeros(N, dtype=np.int32)
    context.initialize(pos, types, ["A"], L, dtype=np.float64)
    meta = integrate.mode_metadynamics(dt=0.005, stride=1, mode="well_tempered", W=1.0, deltaT=7.0, T=1.0)
    nl = cv.nlist_cell(r_cut=1.5)
    half = util.build_nlist(pos, L, 1.5, half=True)
    Ql_ref = [0, 0, 0, 0, 1, 0, 1]
    rbox = ref.Box.make(L)
    pt = util.oracle_postype(pos, types)
    val, Qlm, Ql = ref.ql_compute_cv(pt, rbox, *half, 1.4, 1.2, 6, 0, Ql_ref, half=True)
    st = cv.steinhardt(r_cut=1.4, r_on=1.2, lmax=6, Ql_ref=Ql_ref, nlist=nl, type="A", sigma=0.02 * val)
    nl.cpp_nlist.setStorageMode(_metadynamics.NeighborList.storageMode.half)
    nl.set_lists(*half)
    st.set_grid(0.55 * val, 1.3 * val, 64)
    context.run(2)
    t = context.current.system.getCurrentTimeStep()
    assert st.cpp_force.getCurrentValue(t) == pytest.approx(val, rel=1e-10)
    b = meta.cpp_integrator.getBiasFactors()
    F = st.cpp_force.getForceArray()
    F_ref = ref.ql_compute_forces(pt, rbox, *half, 1.4, 1.2, 6, 0, Ql_ref, Qlm, b[0], half=True)
    assert np.abs(F_ref[:, :3]).max() > 0
    assert np.abs(F[:, :3] - F_ref[:, :3]).max() <= 1e-7 * np.abs(F_ref[:, :3]).max()
    # a new list (the pairs stored at their other end) is picked up: same result, bit for bit
    i_of = np.repeat(np.arange(N), half[1])
    a, bb = half[2].astype(np.int64), i_of
    order = np.lexsort((bb, a))
    a, bb = a[order], bb[order]
    nn2 = np.bincount(a, minlength=N).astype(np.uint32)
    head2 = np.zeros(N, dtype=np.uint32)
    head2[1:] = np.cumsum(nn2)[:-1]
    nl.set_lists(head2, nn2, bb.astype(np.uint32))
    context.current.system.run(0)
    st.cpp_force.compute(context.current.system.getCurrentTimeStep()) if hasattr(st.cpp_force, "compute") else None
    context.run(1)
    assert st.cpp_force.getCurrentValue(context.current.system.getCurrentTimeStep()) == pytest.approx(val, rel=1e-12)


def test_wrap_through_api(api, ref):
    """cv.wrap around a prescribed force: energy CV incl. the external energy, grid evolution, and the wrapped
    compute's own arrays scaled by the bias factor (CollectiveWrapper.cc:136-179)"""
    context, cv, integrate = api
    from metadynamics import force
    N, L = 5000, 20.0
    rng = np.random.default_rng(3)
    pos = rng.random((N, 3)) * L - L / 2
    context.initialize(pos, np.zeros(N, dtype=np.int32), ["A"], L, dtype=np.float64)
    f = rng.normal(size=(N, 4))
    f[:, 3] = rng.normal(-0.5, 0.1, N)
    tq = rng.normal(size=(N, 4))
    vir = rng.normal(size=(6, N))
    meta = integrate.mode_metadynamics(dt=0.005, stride=1, mode="well_tempered", W=2.0, deltaT=50.0, T=1.0)
    frc = force.prescribed(f, tq, vir, external_energy=12.5, name="lj")
    e_ref = ref.wrapper_energy(f, 12.5)
    w = cv.wrap(frc, sigma=40.0)
    assert w.name == "cv_lj"
    w.set_grid(e_ref - 250.0, e_ref + 330.0, 64)                       # CV off the node mid-point: dV/ds != 0
    context.run(3)
    t = context.current.system.getCurrentTimeStep()
    assert w.cpp_force.getCurrentValue(t) == pytest.approx(e_ref, rel=1e-12)
    assert w.cpp_force.getLogValue("cv_lj", t) == pytest.approx(e_ref, rel=1e-12)
    g = ref.Metad([40.0], [e_ref - 250.0], [e_ref + 330.0], [64], W=2.0, T_shift=50.0, T=1.0, stride=1, mode="well_tempered")
    for tt in range(4):
        b = g.update_bias(tt, [e_ref])
    assert abs(b[0]) > 1e-5
    assert np.allclose(meta.cpp_integrator.getBiasFactors(), b, rtol=1e-8)
    f2, t2, v2 = ref.wrapper_scale(f, tq, vir.reshape(-1), N, b[0])
    assert np.allclose(frc.cpp_force.getForces(), f2, rtol=1e-8, atol=0)
    assert np.allclose(frc.cpp_force.getTorques(), t2, rtol=1e-8, atol=0)
    assert np.allclose(frc.cpp_force.getVirial().reshape(-1), v2, rtol=1e-8, atol=0)
    with pytest.raises(RuntimeError):
        cv.wrap("not a force")


def test_adaptive_gaussians_through_api(api, ref, tmp_path):
    """set_params(adaptive=True, sigma_g): every deposit step the derivative arrays (bias factor 1) give the width
    matrix (IntegratorMetaDynamics.cc:333-341, 1205-1294); a box CV keeps its registered sigma on the diagonal"""
    context, cv, integrate = api
    pos, types, L = util.snapshot_config0b()
    context.initialize(pos, types, ["A", "B"], L, dtype=np.float64)
    hills = str(tmp_path / "hills_adaptive.log")
    meta = integrate.mode_metadynamics(dt=0.005, stride=2, mode="well_tempered", W=1.0, deltaT=7.0, T=1.0, filename=hills)
    lv1, lv2 = [(0, 0, 4)], [(0, 0, 4), (0, 4, 0)]
    lam1 = cv.lamellar(sigma=0.05, mode=dict(A=1.0, B=-1.0), lattice_vectors=lv1, name="a")
    lam2 = cv.lamellar(sigma=0.05, mode=dict(A=1.0, B=-1.0), lattice_vectors=lv2, name="b")
    dens = cv.density(sigma=0.01)
    lam1.set_grid(-1.0, 1.0, 48)
    lam2.set_grid(-1.0, 1.0, 40)
    dens.set_grid(0.5, 1.5, 16)
    meta.set_params(adaptive=True, sigma_g=0.02)
    context.run(5)
    assert not meta.cpp_integrator.usedFusedPath()

    rbox = ref.Box.make(L)
    pt = util.oracle_postype(pos, types)
    d1 = ref.lamellar_forces(lv1, pt, util.MODE_AB, rbox, 1.0)
    d2 = ref.lamellar_forces(lv2, pt, util.MODE_AB, rbox, 1.0)
    sq, inv = ref.compute_sigma([d1, d2, np.zeros_like(d1)], [1, 1, 0], [0.05, 0.05, 0.01], 0.02)
    assert np.isfinite(inv).all()
    got = np.array(meta.cpp_integrator.getSigmaInv()).reshape(3, 3)
    # the derivative arrays come from the lamellar force kernel (1e-5 force tolerance, fp32 trig): widths agree to ~1e-7
    assert np.allclose(got, inv, rtol=1e-6, atol=1e-6 * np.abs(inv).max())
    assert got[2, 2] == pytest.approx(100.0) and got[0, 2] == 0
    # the hills file records the width matrix of every deposit: the rows of h_sigma_inv as computeSigma left them
    # (IntegratorMetaDynamics.cc:536-541), each row written without delimiter (Q16) — not diag(1 / sigma)
    last = open(hills).read().splitlines()[-1].split("\t")
    for i in range(3):
        assert last[3 + 2 * i] == "".join("%.10g" % got[i, j] for j in range(3)), (i, last)
    assert abs(got[0, 1]) > 0

    s = [ref.lamellar_cv(lv1, pt, util.MODE_AB, rbox), ref.lamellar_cv(lv2, pt, util.MODE_AB, rbox), len(pos) / L ** 3]
    g = ref.Metad([0.05, 0.05, 0.01], [-1.0, -1.0, 0.5], [1.0, 1.0, 1.5], [48, 40, 16], W=1.0, T_shift=7.0, T=1.0, stride=2,
                  mode="well_tempered")
    for tt in range(6):
        if tt % 2 == 0:
            g.set_sigma_inv(inv)
        b = g.update_bias(tt, s)
    # lamellar CVs to the stated 1e-6 (the trig mode is a process-wide switch other tests may have left on "fast")
    cv_now = meta.cpp_integrator.getCurrentValues()
    assert np.allclose(cv_now, s, rtol=1e-6)
    b = None
    g = ref.Metad([0.05, 0.05, 0.01], [-1.0, -1.0, 0.5], [1.0, 1.0, 1.5], [48, 40, 16], W=1.0, T_shift=7.0, T=1.0, stride=2,
                  mode="well_tempered")
    for tt in range(6):                                              # the oracle grid driven with the device's CV values
        if tt % 2 == 0:
            g.set_sigma_inv(inv)
        b = g.update_bias(tt, cv_now)
    assert np.allclose(meta.cpp_integrator.getBiasFactors(), b, rtol=1e-6, atol=1e-9 * np.abs(b).max())
    t_now = context.current.system.getCurrentTimeStep()
    assert meta.cpp_integrator.getLogValue("det_sigma", t_now) == pytest.approx(g.sigma_determinant, rel=1e-6)


def test_mesh_log_quantities_and_virial_through_api(api, ref):
    """cv.mesh log quantities qx_max..sq_max (OrderParameterMesh.cc:1077-1179), set_kernel / use_table (cv.py:423-466)
    and the external virial under the pressure flag (:1062-1072)"""
    context, cv, integrate = api
    N, L = 8000, 20.0
    pos, types = util.snapshot_random(N, L, seed=21, modulated=True, dtype=np.float64)
    context.initialize(pos, types, ["A", "B"], L, dtype=np.float64)
    integrate.mode_metadynamics(dt=0.005, stride=1)
    mesh = cv.mesh(nx=16, mode={"A": 1.0, "B": -1.0})
    assert set(["cv_mesh", "qx_max", "qy_max", "qz_max", "sq_max"]) <= set(mesh.cpp_force.getProvidedLogQuantities())
    rbox = ref.Box.make(L)
    opt = util.oracle_postype(pos, types)
    r = ref.Mesh(16, 16, 16, [1.0, -1.0])
    s_ref = r.cv(opt, rbox)
    q_ref = r.qmax(N)
    got_q = np.array([mesh.cpp_force.getLogValue(n, 1) for n in ("qx_max", "qy_max", "qz_max")])
    assert np.abs(q_ref[:3]).max() > 0                               # defined up to the sign (|f(k)| = |f(-k)|)
    assert np.allclose(got_q, q_ref[:3], rtol=1e-10, atol=1e-13) or np.allclose(got_q, -q_ref[:3], rtol=1e-10, atol=1e-13)
    assert mesh.cpp_force.getLogValue("sq_max", 1) == pytest.approx(q_ref[3], rel=1e-10)
    assert mesh.cpp_force.getLogValue("cv_mesh", 1) == pytest.approx(s_ref, rel=1e-9)
    with pytest.raises(RuntimeError):
        mesh.set_params(use_table=True)                               # no kernel set yet

    def kernel(k, kmin, kmax, a):
        return np.exp(-a * k * k), -2 * a * k * np.exp(-a * k * k)

    mesh.set_kernel(kernel, 0.3, 6.0, 40, coeff=dict(a=0.1))
    mesh.set_params(use_table=True, umbrella="harmonic", kappa=3.0, cv0=0.5 * s_ref)
    pdata = context.current.system_definition.getParticleData()
    context.run(1)
    assert [mesh.cpp_force.getExternalVirial(i) for i in range(6)] == [0.0] * 6      # pressure flag not set (:1069-1072)
    pdata.setPressureFlag(True)
    context.run(1)
    kt = np.linspace(0.3, 6.0, 40)
    K, dK = kernel(kt, 0.3, 6.0, 0.1)
    r.set_table(K, dK, 0.3, 6.0)
    r.set_use_table(True)
    r.cv(opt, rbox)
    bias = 3.0 * (s_ref - 0.5 * s_ref)                                # harmonic umbrella dU/ds (CollectiveVariable.cc:43-50)
    v_ref = r.virial(N, bias)
    got = np.array([mesh.cpp_force.getExternalVirial(i) for i in range(6)])
    assert np.abs(v_ref).max() > 0
    assert np.allclose(got, v_ref, rtol=1e-8, atol=1e-9 * np.abs(v_ref).max())


def test_umbrella_force_descends_the_umbrella_energy(api):
    """end to end, no oracle: moving the particles a small step along the bias force the API reports must lower the umbrella
    energy U(s) = kappa/2 (s - cv0)^2 and pull the CV towards cv0 — the sign and direction of F = -dU/ds grad s through the
    whole stack (CollectiveVariable.cc:22-66 on top of the CV's computeBiasForces)"""
    context, cv, integrate = api
    pos, types, L = util.snapshot_config0b()
    context.initialize(pos, types, ["A", "B"], L, dtype=np.float64)
    integrate.mode_metadynamics(dt=0.005, stride=1)
    lam = cv.lamellar(sigma=0.05, mode=dict(A=1.0, B=-1.0), lattice_vectors=[(0, 0, 4)])
    mesh = cv.mesh(nx=16, mode={"A": 1.0, "B": -1.0})
    pdata = context.current.system_definition.getParticleData()
    for c in (lam, mesh):
        name = [q for q in c.cpp_force.getProvidedLogQuantities() if q.startswith("umbrella_energy_")][0]
        context.run(1)
        t = context.current.system.getCurrentTimeStep()
        s0 = c.cpp_force.getCurrentValue(t)
        cv0 = 0.8 * s0
        c.set_params(umbrella="harmonic", cv0=cv0, kappa=50.0 / s0 ** 2)
        energies, values = [], []
        for it in range(6):
            context.run(1)
            t = context.current.system.getCurrentTimeStep()
            energies.append(c.cpp_force.getLogValue(name, t))
            values.append(c.cpp_force.getCurrentValue(t))
            F = c.cpp_force.getForceArray().astype(np.float64)
            P = pdata.getPositions().astype(np.float64)
            step = 0.02 / np.abs(F[:, :3]).max()                    # largest displacement 0.02: a small step downhill
            P[:, :3] += step * F[:, :3]
            pdata.setPositions(P)
        # the first step goes downhill; later ones may overshoot the minimum with this fixed step length (the mesh CV is
        # quartic in the density), but never back up to where the descent started
        assert energies[1] < energies[0], (name, energies)
        assert max(energies[1:]) < energies[0] and min(energies) < 0.2 * energies[0], (name, energies)
        assert abs(values[-1] - cv0) < abs(values[0] - cv0)
        c.set_params(umbrella="no_umbrella")


@pytest.mark.parametrize("kind,stride", [("mesh", 1), ("mesh", 3), ("steinhardt", 1), ("steinhardt", 2)])
def test_graph_replay_matches_plain_launches(api, kind, stride, monkeypatch):
    """System::run replays long runs from a HIP graph where the integrator allows it (IntegratorMetaDynamics::graphPeriod): one
    period of lcm(2, stride) steps is captured after a settling period and replayed.  The replay must leave EXACTLY what the plain
    launches leave — hill count, CV values, V, w, dV/ds, every cell of the bias grid, the forces: a captured sequence that is not
    periodic (an extra deferred pass in the capture, the mesh's alternating cursor sets out of phase) would show here."""
    import ctypes as C
    from metadynamics import _abi
    context, cv, integrate = api
    monkeypatch.setenv("MTD_GRAPH_VERIFY", "1")
    lib = _abi.load()

    s_probe = [None]

    def build():
        if kind == "mesh":
            N, L = 20000, 20.0
            pos, types = util.snapshot_random(N, L, seed=21, modulated=True, dtype=np.float32)
            pos = (np.mod(pos.astype(np.float64) + L / 2, L) - L / 2).astype(np.float32)
            if s_probe[0] is None:                                   # one evaluation supplies the value the grid is laid around
                context.initialize(pos, types, ["A", "B"], L, dtype=np.float32)
                integrate.mode_metadynamics(dt=0.005, stride=1, mode="well_tempered", W=1.0, deltaT=7.0, T=1.0)
                probe = cv.mesh(nx=32, mode={"A": 1.0, "B": -1.0})
                probe.set_grid(0.0, 1.0, 8)
                context.run(1)
                s_probe[0] = probe.cpp_force.getCurrentValue(1)
                context.current = None
            sm = s_probe[0]
            context.initialize(pos, types, ["A", "B"], L, dtype=np.float32)
            meta = integrate.mode_metadynamics(dt=0.005, stride=stride, mode="well_tempered", W=1.0, deltaT=7.0, T=1.0)
            lam = cv.lamellar(sigma=0.02, mode=dict(A=1.0, B=-1.0), lattice_vectors=util.CV1_VECTORS)
            lam.set_grid(-0.6, 0.4, 48)
            mesh = cv.mesh(nx=32, mode={"A": 1.0, "B": -1.0}, sigma=0.05 * abs(sm))
            mesh.set_grid(0.25 * sm, 1.6 * sm, 40)
            return meta, [lam, mesh]
        pos, L = util.fcc_lattice(6)
        pos = pos + np.random.default_rng(12).normal(0, 0.05, pos.shape)
        N = len(pos)
        context.initialize(pos, np.zeros(N, dtype=np.int32), ["A"], L, dtype=np.float64)
        meta = integrate.mode_metadynamics(dt=0.005, stride=stride, mode="well_tempered", W=1.0, deltaT=7.0, T=1.0)
        nl = cv.nlist_cell(r_cut=1.5)
        nl.update()
        st = cv.steinhardt(r_cut=1.4, r_on=1.2, lmax=6, Ql_ref=[0, 0, 0, 0, 1, 0, 1], nlist=nl, type="A", sigma=0.5)
        st.set_grid(20.0, 60.0, 64)
        return meta, [st]

    def run(graph):
        meta, cvs = build()
        context.current.system.setGraphMode(1 if graph else 0)
        context.run(3)                      # short: plain launches in both arms
        assert context.current.system.lastRunGraphSteps() == 0
        context.run(61)
        integ = meta.cpp_integrator
        t = context.current.system.getCurrentTimeStep()
        h = C.c_void_p(integ.getEngineHandle())
        G = lib.mtd_metad_num_elements(h)
        arrays = {}
        for which, name in enumerate(_abi.ARRAY_NAMES):
            out = np.zeros(G, dtype=np.float64 if which < 6 else np.uint32)
            _abi.check(lib.mtd_metad_get_array(h, which, out.ctypes.data, None))
            arrays[name] = out
        res = dict(cv=list(integ.getCurrentValues()), bias=list(integ.getBiasFactors()), V=integ.getLogValue("bias", t), w=integ.getLogValue("weight", t),
                   n=integ.getNumGaussians(), arrays=arrays, F=[c.cpp_force.getForceArray().copy() for c in cvs],
                   graph_steps=context.current.system.lastRunGraphSteps(), period=integ.graphPeriod(), t=t)
        context.current = None
        return res

    plain, graph = run(False), run(True)
    period = 2 * stride if stride % 2 else stride
    assert plain["graph_steps"] == 0 and graph["period"] == period
    assert graph["graph_steps"] >= 2 * period and graph["graph_steps"] % period == 0      # (a graph holds several periods)
    assert graph["t"] == plain["t"] == 64 and graph["n"] == plain["n"] and plain["n"] >= 60 // stride
    assert graph["cv"] == plain["cv"] and graph["bias"] == plain["bias"] and graph["V"] == plain["V"] and graph["w"] == plain["w"]
    for name in _abi.ARRAY_NAMES:
        assert np.array_equal(graph["arrays"][name], plain["arrays"][name], equal_nan=True), name
    assert np.isfinite(plain["arrays"]["reweighted"]).all() and plain["arrays"]["grid"].max() > 0 and plain["V"] > 0     # (on the grid: a real bias)
    for a, b in zip(graph["F"], plain["F"]):
        assert np.array_equal(a, b) and np.abs(a).max() > 0


def test_graph_replay_is_refused_where_steps_are_not_replayable(api, tmp_path):
    """a hills file (host values every deposit), an umbrella, a pure lamellar set (two-launch step: slower from a graph): plain launches"""
    context, cv, integrate = api
    pos, types = util.snapshot_random(4000, 12.0, seed=3, modulated=True, dtype=np.float32)
    context.initialize(pos, types, ["A", "B"], 12.0, dtype=np.float32)
    meta = integrate.mode_metadynamics(dt=0.005, stride=1, mode="well_tempered", W=1.0, deltaT=7.0, T=1.0)
    lam = cv.lamellar(sigma=0.02, mode=dict(A=1.0, B=-1.0), lattice_vectors=util.CV1_VECTORS)
    lam.set_grid(-1.0, 1.0, 32)
    context.current.system.setGraphMode(1)
    context.run(40)
    assert meta.cpp_integrator.graphPeriod() == 0 and context.current.system.lastRunGraphSteps() == 0 and meta.cpp_integrator.usedFusedPath()
    lam.set_params(umbrella="harmonic", kappa=1.0, cv0=0.0)
    context.run(40)
    assert meta.cpp_integrator.graphPeriod() == 0 and context.current.system.lastRunGraphSteps() == 0
