"""Worker of tests/test_gpu_host_dd.py: one rank of a DOMAIN-DECOMPOSED run through the reference-shaped API (metadynamics.cv /
integrate over the C++ host classes) — the ranks are separate processes that all use cuda:0 (a one-GPU box), control plane gloo,
data plane the xGMI mailbox: the small per-step sums through mtd_comm_allreduce_small, the replicated mesh through
mtd_comm_allreduce_pull (RCCL refuses two ranks on one device), the slab mesh through its exported buffers.  Every CV the
reference runs under MPI (OrderParameterMesh.cc:630, 911; SteinhardtQl.cc:183-191; WellTemperedEnsemble.cc:57-63;
CollectiveWrapper.cc:64-70; IntegratorMetaDynamics.cc:1259-1268) against the ORACLE on the whole snapshot.
Prints one JSON line on rank 0.   RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT from the environment.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "metadynamics-plugin_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch
import torch.distributed as dist

import mtd_ref as ref
import util
from metadynamics import context, cv, integrate, xgmi

KW = dict(dt=0.005, stride=1, mode="well_tempered", W=1.0, deltaT=7.0, T=1.0)


def same_on_all_ranks(values):
    """the same BITS on every rank (float64 array)"""
    mine = torch.from_numpy(np.ascontiguousarray(values, dtype=np.float64).view(np.int64).copy())
    ref0 = mine.clone()
    dist.broadcast(ref0, 0)
    same = torch.tensor([int(torch.equal(mine, ref0))])
    dist.all_reduce(same, op=dist.ReduceOp.MIN)
    return bool(int(same.item()))


def replicated_state(meta):
    integ = meta.cpp_integrator
    t = context.current.system.getCurrentTimeStep()
    import ctypes as C
    from metadynamics import _abi
    lib = _abi.load()
    h = C.c_void_p(integ.getEngineHandle())
    G = lib.mtd_metad_num_elements(h)
    grid = np.zeros(G)
    _abi.check(lib.mtd_metad_get_array(h, 0, grid.ctypes.data, None))
    return np.concatenate([integ.getCurrentValues(), integ.getBiasFactors(), [integ.getLogValue("bias", t), integ.getLogValue("weight", t)], grid])


def shards(n, world):
    cut = [r * n // world + (17 if 0 < r < world else 0) for r in range(world + 1)]      # unequal shards
    return [slice(cut[r], cut[r + 1]) for r in range(world)]


def mesh_set(rank, world, box, decomposition, out):
    """config 3's CV set: cv.lamellar + cv.mesh on one 2-d grid"""
    N, L, dims = 6007, 20.0, (16, 24, 24)
    pos, types = util.snapshot_random(N, L, seed=21, modulated=True, dtype=np.float32)
    sl = shards(N, world)[rank]
    rbox, opt = ref.Box.make(L), util.oracle_postype(pos, types)
    mode = [1.0, -0.7]
    rm = ref.Mesh(dims[0], dims[1], dims[2], mode)
    s_mesh, s_lam = rm.cv(opt, rbox), ref.lamellar_cv(util.CV1_VECTORS, opt, util.MODE_AB, rbox)
    context.initialize(pos[sl].copy(), types[sl].copy(), ["A", "B"], L, dtype=np.float32, n_global=N)
    xgmi.attach(dist, context.exec_conf, box)
    meta = integrate.mode_metadynamics(**KW)
    lam = cv.lamellar(sigma=0.02, mode=dict(A=1.0, B=-1.0), lattice_vectors=util.CV1_VECTORS)
    lam.set_grid(-0.6, 0.4, 32)
    mesh = cv.mesh(nx=dims[0], ny=dims[1], nz=dims[2], mode={"A": mode[0], "B": mode[1]}, sigma=0.05 * abs(s_mesh))
    mesh.set_grid(0.25 * s_mesh, 1.6 * s_mesh, 40)
    mesh.set_decomposition(decomposition)
    context.run(3)
    integ = meta.cpp_integrator
    t = context.current.system.getCurrentTimeStep()
    cvg = list(integ.getCurrentValues())
    g = ref.Metad([0.02, 0.05 * abs(s_mesh)], [-0.6, 0.25 * s_mesh], [0.4, 1.6 * s_mesh], [32, 40], W=1.0, T_shift=7.0, T=1.0, stride=1,
                  mode="well_tempered")
    for tt in range(4):                                            # prepRun(0) + 3 updates; the oracle's grid driven with the device's values
        b = g.update_bias(tt, cvg)
    rec = dict(fused=bool(integ.usedFusedPath()), exchange=context.exec_conf.largeExchangeName(),
               cv_rel=[abs(cvg[0] - s_lam) / abs(s_lam), abs(cvg[1] - s_mesh) / abs(s_mesh)],
               bias_rel=float(np.abs(np.array(integ.getBiasFactors()) - b).max() / np.abs(b).max()),
               V_rel=abs(integ.getLogValue("bias", t) - g.curr_bias) / abs(g.curr_bias),
               hills=integ.getNumGaussians(), replicated_bitwise=same_on_all_ranks(replicated_state(meta)))
    # forces per unit bias factor (the particles do not move: dV/ds at the hill's centre is ~0) against the oracle's slice
    mesh.cpp_force.computeDerivatives(t)
    lam.cpp_force.computeDerivatives(t)
    Fm, Fl = mesh.cpp_force.getForceArray().astype(np.float64), lam.cpp_force.getForceArray().astype(np.float64)
    Fm_ref, Fl_ref = rm.forces(opt, rbox, 1.0), ref.lamellar_forces(util.CV1_VECTORS, opt, util.MODE_AB, rbox, 1.0)
    err = torch.tensor([np.abs(Fm[:, :3] - Fm_ref[sl, :3]).max() / np.abs(Fm_ref[:, :3]).max(),
                        np.abs(Fl[:, :3] - Fl_ref[sl, :3]).max() / np.abs(Fl_ref[:, :3]).max()], dtype=torch.float64)
    dist.all_reduce(err, op=dist.ReduceOp.MAX)
    rec["force_rel"] = err.tolist()
    # the synchronising read-back of a CV is collective too (every rank asks)
    rec["getCurrentValue_rel"] = [abs(lam.cpp_force.getCurrentValue(t) - s_lam) / abs(s_lam), abs(mesh.cpp_force.getCurrentValue(t) - s_mesh) / abs(s_mesh)]
    rec["timeouts"] = box.timeouts()
    out["mesh_" + decomposition] = rec
    context.current = None
    dist.barrier()


def steinhardt_set(rank, world, box, out):
    """cv.steinhardt over z slabs with ghost particles"""
    pos, L = util.fcc_lattice(6)
    pos = pos + np.random.default_rng(12).normal(0, 0.05, pos.shape)
    pos = np.mod(pos + L / 2, L) - L / 2
    N = len(pos)
    types = np.zeros(N, dtype=np.int32)
    r_list = 1.5
    owner = np.minimum((np.mod(pos[:, 2] + L / 2, L) / L * world).astype(int), world - 1)
    mine = np.where(owner == rank)[0]
    lo, hi = -L / 2 + rank * L / world, -L / 2 + (rank + 1) * L / world
    z = pos[:, 2]

    def zdist(a, b):
        d = np.abs(a - b)
        return np.minimum(d, L - d)

    near = (owner != rank) & ((zdist(z, lo) <= r_list) | (zdist(z, hi) <= r_list))
    ghosts = np.where(near)[0]
    context.initialize(pos[mine], types[mine], ["A"], L, dtype=np.float64, n_global=N, ghost_positions=pos[ghosts], ghost_types=types[ghosts])
    xgmi.attach(dist, context.exec_conf, box)
    meta = integrate.mode_metadynamics(**KW)
    nl = cv.nlist_cell(r_cut=r_list)
    nl.update()
    Ql_ref = [0, 0, 0, 0, 1, 0, 1]
    rbox, pt = ref.Box.make(L), util.oracle_postype(pos, types)
    lists = util.build_nlist(pos, L, r_list)
    val, Qlm, Ql = ref.ql_compute_cv(pt, rbox, *lists, 1.4, 1.2, 6, 0, Ql_ref)
    st = cv.steinhardt(r_cut=1.4, r_on=1.2, lmax=6, Ql_ref=Ql_ref, nlist=nl, type="A", sigma=0.02 * val)
    st.set_grid(0.55 * val, 1.3 * val, 64)
    context.run(2)
    integ = meta.cpp_integrator
    t = context.current.system.getCurrentTimeStep()
    cvg = list(integ.getCurrentValues())
    g = ref.Metad([0.02 * val], [0.55 * val], [1.3 * val], [64], W=1.0, T_shift=7.0, T=1.0, stride=1, mode="well_tempered")
    for tt in range(3):
        b = g.update_bias(tt, cvg)
    st.cpp_force.computeDerivatives(t)
    F = st.cpp_force.getForceArray().astype(np.float64)
    F_ref = ref.ql_compute_forces(pt, rbox, *lists, 1.4, 1.2, 6, 0, Ql_ref, Qlm, 1.0)
    err = torch.tensor([np.abs(F[:, :3] - F_ref[mine, :3]).max() / np.abs(F_ref[:, :3]).max()], dtype=torch.float64)
    dist.all_reduce(err, op=dist.ReduceOp.MAX)
    n_loc = torch.tensor([len(mine), len(ghosts)])
    dist.all_reduce(n_loc)
    out["steinhardt"] = dict(cv_rel=abs(cvg[0] - val) / abs(val), bias_rel=float(np.abs(np.array(integ.getBiasFactors()) - b).max() / max(np.abs(b).max(), 1e-300)),
                             force_rel=float(err.item()), getCurrentValue_rel=abs(st.cpp_force.getCurrentValue(t) - val) / abs(val),
                             Q6_abs=abs(st.cpp_force.getLogValue("steinhardt_Q6", t) - Ql[5]),
                             replicated_bitwise=same_on_all_ranks(replicated_state(meta)), locals_total=int(n_loc[0]), ghosts_total=int(n_loc[1]),
                             n_global=N, timeouts=box.timeouts())
    context.current = None
    dist.barrier()


def energy_sets(rank, world, box, out):
    """cv.potential_energy (net force arrays scaled by 1 + bias) and cv.wrap (a compute's own arrays scaled by the bias)"""
    from metadynamics import force as force_mod
    N = 5003
    rng = np.random.default_rng(4)
    pos = rng.random((N, 3)) * 10 - 5
    nf = rng.normal(size=(N, 4)); nf[:, 3] = rng.normal(-2.0, 0.5, N)
    nt = rng.normal(size=(N, 4))
    nv = rng.normal(size=(6, N))
    sl = shards(N, world)[rank]
    n_loc = sl.stop - sl.start
    ext = [12.5 * (r + 1) for r in range(world)]
    pe_ref = ref.wte_potential_energy(nf, sum(ext))
    context.initialize(pos[sl], np.zeros(n_loc, dtype=int), ["A"], 10.0, dtype=np.float64, n_global=N)
    xgmi.attach(dist, context.exec_conf, box)
    pdata = context.current.system_definition.getParticleData()
    pdata.setNetForce(nf[sl].copy()); pdata.setNetTorque(nt[sl].copy()); pdata.setNetVirial(np.ascontiguousarray(nv[:, sl]))
    pdata.setExternalEnergy(ext[rank])
    meta = integrate.mode_metadynamics(dt=0.005, stride=1, mode="well_tempered", W=2.0, deltaT=50.0, T=1.0)
    pe = cv.potential_energy(sigma=40.0)
    pe.set_grid(cv_min=pe_ref - 300.0, cv_max=pe_ref + 300.0, num_points=64)
    v0 = pe.cpp_force.getCurrentValue(0)
    context.run(1)
    r = ref.Metad([40.0], [pe_ref - 300.0], [pe_ref + 300.0], [64], W=2.0, T_shift=50.0, T=1.0, stride=1, mode="well_tempered")
    r.update_bias(0, [pe_ref])
    b = r.update_bias(1, [pe_ref])
    f2, t2, v2, _ = ref.wte_scale(nf, nt, nv.reshape(-1), N, [0, 0, 0, 0, 0, 0], b[0])
    v2 = v2.reshape(6, N)
    err = torch.tensor([np.abs(pdata.getNetForce() - f2[sl]).max(), np.abs(pdata.getNetTorque() - t2[sl]).max(),
                        np.abs(pdata.getNetVirial() - v2[:, sl]).max()], dtype=torch.float64)
    dist.all_reduce(err, op=dist.ReduceOp.MAX)
    out["potential_energy"] = dict(cv_rel=abs(v0 - pe_ref) / abs(pe_ref), scaled_abs_err=err.tolist(), bias=b[0],
                                   replicated_bitwise=same_on_all_ranks(replicated_state(meta)), timeouts=box.timeouts())
    context.current = None
    dist.barrier()

    # cv.wrap next to a lamellar CV under an umbrella (the generic path, CV by CV)
    N, L = 6000, 20.0
    pos, types = util.snapshot_random(N, L, seed=5, modulated=True, dtype=np.float32)
    frc = rng.normal(size=(N, 4)).astype(np.float32)
    frc[:, 3] = rng.normal(-0.5, 0.1, N)
    sl = shards(N, world)[rank]
    n_loc = sl.stop - sl.start
    e_tot = float(frc[:, 3].astype(np.float64).sum()) + sum(1.0 + r for r in range(world))

    def build(sel, ext_energy, dd):
        n = len(pos[sel])
        context.initialize(pos[sel].copy(), types[sel].copy(), ["A", "B"], L, dtype=np.float32, n_global=N)
        if dd:
            xgmi.attach(dist, context.exec_conf, box)
        meta = integrate.mode_metadynamics(**KW)
        lam = cv.lamellar(sigma=0.02, mode=dict(A=1.0, B=-1.0), lattice_vectors=util.CV1_VECTORS)
        lam.set_grid(-0.6, 0.4, 40)
        lam.set_params(umbrella="harmonic", kappa=3.0, cv0=-0.1)
        fc = force_mod.prescribed(frc[sel].copy(), np.zeros((n, 4), dtype=np.float32), np.zeros((6, n), dtype=np.float32), external_energy=ext_energy)
        w = cv.wrap(fc, sigma=40.0)
        w.set_grid(e_tot - 250.0, e_tot + 330.0, 64)                  # (the CV off the node mid-point: dV/ds != 0)
        context.run(3)
        integ = meta.cpp_integrator
        t = context.current.system.getCurrentTimeStep()
        return dict(cv=list(integ.getCurrentValues()), bias=list(integ.getBiasFactors()), V=integ.getLogValue("bias", t),
                    f_lam=lam.cpp_force.getForceArray().astype(np.float64), f_wrap=fc.cpp_force.getForces().astype(np.float64),
                    umbrella=lam.cpp_force.getLogValue("umbrella_energy_" + lam.cpp_force.getName(), t)), meta

    got, meta = build(sl, 1.0 + rank, True)
    bitwise = same_on_all_ranks(replicated_state(meta))
    to = box.timeouts()
    context.current = None
    dist.barrier()
    rec = dict(replicated_bitwise=bitwise, timeouts=to, energy_cv=[got["cv"][1], e_tot])
    if rank == 0:
        one, _ = build(slice(0, N), sum(1.0 + r for r in range(world)), False)      # one rank holding everything, no mailbox
        context.current = None
        rec.update(bias_one=one["bias"], cv_rel=[abs(a - b) / abs(b) for a, b in zip(got["cv"], one["cv"])],
                   bias_rel=float(np.abs(np.array(got["bias"]) - np.array(one["bias"])).max() / np.abs(one["bias"]).max()),
                   V_rel=abs(got["V"] - one["V"]) / abs(one["V"]), umbrella_rel=abs(got["umbrella"] - one["umbrella"]) / abs(one["umbrella"]),
                   f_lam_rel=float(np.abs(got["f_lam"][:, :3] - one["f_lam"][sl, :3]).max() / np.abs(one["f_lam"][:, :3]).max()),
                   f_wrap_rel=float(np.abs(got["f_wrap"][:, :3] - one["f_wrap"][sl, :3]).max() / np.abs(one["f_wrap"][:, :3]).max()))
    out["wrap_umbrella"] = rec
    dist.barrier()


def adaptive_set(rank, world, box, out):
    """adaptive Gaussians: the derivative products of computeSigma summed over the ranks (IntegratorMetaDynamics.cc:1259-1268)"""
    pos, types, L = util.snapshot_config0b()
    N = len(pos)
    sl = shards(N, world)[rank]
    lv1, lv2 = [(0, 0, 4)], [(0, 0, 4), (0, 4, 0)]
    context.initialize(pos[sl], types[sl], ["A", "B"], L, dtype=np.float64, n_global=N)
    xgmi.attach(dist, context.exec_conf, box)
    meta = integrate.mode_metadynamics(dt=0.005, stride=2, mode="well_tempered", W=1.0, deltaT=7.0, T=1.0)
    lam1 = cv.lamellar(sigma=0.05, mode=dict(A=1.0, B=-1.0), lattice_vectors=lv1, name="a")
    lam2 = cv.lamellar(sigma=0.05, mode=dict(A=1.0, B=-1.0), lattice_vectors=lv2, name="b")
    dens = cv.density(sigma=0.01)
    lam1.set_grid(-1.0, 1.0, 48)
    lam2.set_grid(-1.0, 1.0, 40)
    dens.set_grid(0.5, 1.5, 16)
    meta.set_params(adaptive=True, sigma_g=0.02)
    context.run(5)
    rbox, pt = ref.Box.make(L), util.oracle_postype(pos, types)
    d1 = ref.lamellar_forces(lv1, pt, util.MODE_AB, rbox, 1.0)
    d2 = ref.lamellar_forces(lv2, pt, util.MODE_AB, rbox, 1.0)
    sq, inv = ref.compute_sigma([d1, d2, np.zeros_like(d1)], [1, 1, 0], [0.05, 0.05, 0.01], 0.02)
    got = np.array(meta.cpp_integrator.getSigmaInv()).reshape(3, 3)
    s = [ref.lamellar_cv(lv1, pt, util.MODE_AB, rbox), ref.lamellar_cv(lv2, pt, util.MODE_AB, rbox), N / L ** 3]
    cvg = list(meta.cpp_integrator.getCurrentValues())
    out["adaptive"] = dict(sigma_inv_rel=float(np.abs(got - inv).max() / np.abs(inv).max()), box_cv_diag=float(got[2, 2]),
                           cv_rel=float(max(abs(a - b) / abs(b) for a, b in zip(cvg, s))), hills=meta.cpp_integrator.getNumGaussians(),
                           replicated_bitwise=same_on_all_ranks(np.concatenate([replicated_state(meta), got.ravel()])), timeouts=box.timeouts())
    context.current = None
    dist.barrier()


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = {"world": world}
    box = xgmi.connect(dist, max_doubles=256)
    out["connected"] = box is not None
    if box is not None:
        mesh_set(rank, world, box, "replicated", out)
        mesh_set(rank, world, box, "slab", out)
        steinhardt_set(rank, world, box, out)
        energy_sets(rank, world, box, out)
        adaptive_set(rank, world, box, out)
        torch.cuda.synchronize()
        dist.barrier()
        box.close()
    if rank == 0:
        print(json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
