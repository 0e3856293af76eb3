"""Worker of tests/test_gpu_comm.py: one rank of a world of processes that all use cuda:0 (a one-GPU box), control
plane gloo, data plane the xGMI mailbox (mtd_comm_*).  Prints one JSON line on rank 0.

    RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT from the environment; argv[1] = particles in the global snapshot
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

import util
from metadynamics import _abi, xgmi
from metadynamics.sharded import HipLamellarBackend, ShardedBiasStep

GRID = dict(sigma=[0.02, 0.02], cv_min=[-1.0, -1.0], cv_max=[1.0, 1.0], num_points=[64, 48])
KW = dict(W=1.0, T_shift=7.0, T=1.0, stride=1, mode="well_tempered")
CVS = [(util.CV1_VECTORS, util.MODE_AB), (util.CV2_VECTORS, util.MODE_AB)]


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    n_global = int(sys.argv[1])
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = {"world": world}
    box = xgmi.connect(dist, max_doubles=64)
    out["connected"] = box is not None
    if box is None:
        if rank == 0:
            print(json.dumps(out), flush=True)
        dist.destroy_process_group()
        return

    # ---- stand-alone all-reduce of n doubles, many rounds (both slot parities, message sizes up to max_doubles)
    worst = 0.0
    for n in (1, 2, 3, 8, 33, 64):
        for it in range(5):
            mine = np.random.default_rng(1000 * n + 10 * it + rank).normal(size=n)
            want = sum(np.random.default_rng(1000 * n + 10 * it + r).normal(size=n) for r in range(world))
            v = torch.from_numpy(mine).cuda()
            box.all_reduce(v)
            got = v.cpu().numpy()
            worst = max(worst, float(np.abs(got - want).max()))
    out["allreduce_max_err"] = worst
    # every rank must hold the same bits: compare against rank 0's result through the control plane
    v = torch.from_numpy(np.random.default_rng(77 + rank).normal(size=16)).cuda()
    box.all_reduce(v)
    mine = v.cpu()
    ref0 = mine.clone()
    dist.broadcast(ref0, 0)
    same = torch.tensor([int(torch.equal(mine, ref0))])
    dist.all_reduce(same, op=dist.ReduceOp.MIN)
    out["bitwise_same_on_all_ranks"] = bool(int(same.item()))

    # ---- the fused lamellar bias step, particles sharded, exchange through the mailbox
    L = 30.0
    pos, types = util.snapshot_random(n_global, L, seed=5, modulated=True, dtype=np.float32)
    cut = [r * n_global // world + (17 if 0 < r < world else 0) for r in range(world + 1)]     # unequal shards
    sl = slice(cut[rank], cut[rank + 1])
    d_pos = torch.from_numpy(util.pack_postype(pos[sl].copy(), types[sl].copy(), np.float32)).cuda()
    be = HipLamellarBackend(CVS, d_pos, n_global, L, GRID, fast_trig=False, fused=True, **KW)
    be.attach_mailbox(box)
    step = ShardedBiasStep(be, dist)
    states = []
    for t in range(steps):
        step.step(t)
        states.append(be.state())
    torch.cuda.synchronize()
    out["timeouts"] = box.timeouts()
    forces = [f.cpu().numpy() for f in be.forces]

    # all ranks hold the same replicated state, bit for bit
    mine = torch.tensor([x for st in states for x in st["cv"] + st["bias"] + [st["V"], st["w"]]], dtype=torch.float64)
    ref0 = mine.clone()
    dist.broadcast(ref0, 0)
    same = torch.tensor([int(torch.equal(mine, ref0))])
    dist.all_reduce(same, op=dist.ReduceOp.MIN)
    out["replicated_state_bitwise"] = bool(int(same.item()))

    # rank 0: the same steps on one GPU holding all particles, no mailbox
    if rank == 0:
        d_all = torch.from_numpy(util.pack_postype(pos, types, np.float32)).cuda()
        one = HipLamellarBackend(CVS, d_all, n_global, L, GRID, fast_trig=False, fused=True, **KW)
        errs = dict(cv=0.0, bias=0.0, V=0.0, w=0.0)
        for t in range(steps):
            one.step_single(t)
            st1, st = one.state(), states[t]
            errs["cv"] = max(errs["cv"], max(abs(a - b) / max(abs(b), 1e-300) for a, b in zip(st["cv"], st1["cv"])))
            bmax = max(abs(b) for b in st1["bias"]) or 1.0
            errs["bias"] = max(errs["bias"], max(abs(a - b) / bmax for a, b in zip(st["bias"], st1["bias"])))
            errs["V"] = max(errs["V"], abs(st["V"] - st1["V"]) / max(abs(st1["V"]), 1e-300))
            errs["w"] = max(errs["w"], abs(st["w"] - st1["w"]) / max(abs(st1["w"]), 1e-300))
        torch.cuda.synchronize()
        ferr = 0.0
        for c in range(2):
            full = one.forces[c].cpu().numpy()
            ferr = max(ferr, float(np.abs(forces[c][:, :3] - full[sl, :3]).max() / np.abs(full[:, :3]).max()))
        out.update(errs=errs, force_rel_err=ferr, num_gaussians=states[-1]["num_gaussians"], cv=states[-1]["cv"])
        one.close()
    # ---- a generic CV set (lamellar + wrapped energy): its small exchange buffers go through the mailbox's stand-alone
    #      all-reduce (ShardedBiasStep(mailbox=...)), the grid engine runs replicated
    from metadynamics import sharded
    rng = np.random.default_rng(4)
    frc = rng.normal(size=(n_global, 4)).astype(np.float32)
    frc[:, 3] = rng.normal(-0.5, 0.1, n_global)
    e_tot = float(frc[:, 3].astype(np.float64).sum()) + 1.0 * world
    grid2 = dict(sigma=[0.02, 5.0], cv_min=[-1.0, e_tot - 300.0], cv_max=[1.0, e_tot + 300.0], num_points=[40, 30])

    def make_set(sl_, ext):
        dp = torch.from_numpy(util.pack_postype(pos[sl_].copy(), types[sl_].copy(), np.float32)).cuda()
        f = torch.from_numpy(frc[sl_].copy()).cuda()
        parts = [sharded.LamellarPart(util.CV1_VECTORS, util.MODE_AB, dp, n_global, L),
                 sharded.EnergyPart(f, torch.zeros_like(f), torch.zeros((6, f.shape[0]), dtype=f.dtype, device="cuda"), f.shape[0], ext, wrapper=True)]
        return sharded.HipCvSetBackend(parts, grid2, **KW), f

    cs, f_loc = make_set(sl, 1.0)
    set_step = ShardedBiasStep(cs, dist, mailbox=box)
    set_states = []
    for t in range(3):
        set_step.step(t)
        set_states.append(cs.state())
    torch.cuda.synchronize()
    out["timeouts_set"] = box.timeouts()
    if rank == 0:
        one_set, f_all = make_set(slice(0, n_global), 1.0 * world)
        e = dict(cv=0.0, bias=0.0)
        for t in range(3):
            one_set.force_pass(one_set.cv_pass(), t)
            s1, s2 = one_set.state(), set_states[t]
            e["cv"] = max(e["cv"], max(abs(a - b) / max(abs(b), 1e-300) for a, b in zip(s2["cv"], s1["cv"])))
            bmax = max(abs(b) for b in s1["bias"]) or 1.0
            e["bias"] = max(e["bias"], max(abs(a - b) / bmax for a, b in zip(s2["bias"], s1["bias"])))
        torch.cuda.synchronize()
        out["set_errs"] = e
        out["set_energy_cv"] = [set_states[-1]["cv"][1], e_tot]
        out["set_force_err"] = float((f_loc.cpu() - f_all.cpu()[sl]).abs().max() / f_all.cpu().abs().max())
        one_set.close()
    # ---- the C++ host classes with the mailbox as communicator: every rank runs the same System loop; the hills file and the
    #      grid dump belong to the root rank only (IntegratorMetaDynamics.cc:124-146, 835-839)
    import tempfile
    from metadynamics import context, cv as cvmod, integrate
    tmpdir = [tempfile.mkdtemp(prefix="mtd_dd_") if rank == 0 else None]
    dist.broadcast_object_list(tmpdir, 0)
    hills, dump = os.path.join(tmpdir[0], "hills.log"), os.path.join(tmpdir[0], "grid.dat")
    context.initialize(pos[sl].copy(), types[sl].copy(), ["A", "B"], L, dtype=np.float32, n_global=n_global)
    context.exec_conf.setMailbox(box.handle.value)
    meta = integrate.mode_metadynamics(dt=0.005, stride=1, mode="well_tempered", W=1.0, deltaT=7.0, T=1.0, filename=hills)
    hcvs = []
    for i, vecs in enumerate((util.CV1_VECTORS, util.CV2_VECTORS)):
        c = cvmod.lamellar(sigma=0.02, mode=dict(A=1.0, B=-1.0), lattice_vectors=vecs, name="c%d" % i)
        c.set_grid(-1.0, 1.0, 32)
        hcvs.append(c)
    meta.dump_grid(dump, period=2)
    context.run(4)
    torch.cuda.synchronize()
    dist.barrier()
    if rank == 0:
        lines = open(hills).read().splitlines()
        out["host_dd"] = dict(fused=bool(meta.cpp_integrator.usedFusedPath()), hills_lines=len(lines),
                              hills_ok=all(len(l.split("\t")) == 6 for l in lines[1:]), files=sorted(os.listdir(tmpdir[0])))
        dumps = [f for f in out["host_dd"]["files"] if f.startswith("grid.dat_")]
        out["host_dd"]["dump_lines"] = [len(open(os.path.join(tmpdir[0], f)).read().splitlines()) for f in dumps]
    context.current = None
    dist.barrier()
    # ---- the mesh CV with the mesh decomposed into slabs over the ranks (mtd_mesh_slab_*), against one rank holding all
    #      particles and the whole mesh; nz = ny = 24 divides by 2, 3 and 4 (direct transforms), 32 by 2 and 4 (radix-4)
    slab = {}
    for dims in [(16, 24, 24)] + ([(32, 32, 32)] if world in (2, 4) else []):
        mode = [1.0, -0.7]
        dpos = torch.from_numpy(util.pack_postype(pos[sl].copy(), types[sl].copy(), np.float32)).cuda()
        part = sharded.MeshSlabPart(dims[0], dims[1], dims[2], mode, dpos, n_global, L, box, dist)
        lam = sharded.LamellarPart(util.CV1_VECTORS, util.MODE_AB, dpos, n_global, L)
        one_part = None
        if rank == 0:
            d_all = torch.from_numpy(util.pack_postype(pos, types, np.float32)).cuda()
            one_part = sharded.MeshPart(dims[0], dims[1], dims[2], mode, d_all, n_global, L)
            one_part.local_pass()
            probe = sharded.HipCvSetBackend([one_part], dict(sigma=[1.0], cv_min=[-1.0], cv_max=[1.0], num_points=[8]), **KW)
            probe.force_pass(None, 0)
            s_one = probe.state()["cv"][0]
        else:
            s_one = 0.0
        t = torch.tensor([s_one], dtype=torch.float64)
        dist.broadcast(t, 0)
        s_one = float(t.item())
        grid3 = dict(sigma=[0.02, 0.05 * abs(s_one)], cv_min=[-1.0, 0.25 * s_one], cv_max=[1.0, 1.75 * s_one], num_points=[40, 30])
        bs = sharded.HipCvSetBackend([lam, part], grid3, **KW)
        st_step = ShardedBiasStep(bs, dist, mailbox=box)
        for t_ in range(3):
            st_step.step(t_)
        torch.cuda.synchronize()
        st_slab = bs.state()
        # the particles do not move, so every hill lands on the same point and dV/ds there is ~0: the force comparison
        # takes a bias factor of one instead
        unit_bias = torch.ones(1, dtype=torch.float64, device="cuda")
        part.forces(unit_bias.data_ptr())
        torch.cuda.synchronize()
        key = "x".join(str(d) for d in dims)
        slab[key] = dict(timeouts=box.timeouts())
        if rank == 0:
            lam1 = sharded.LamellarPart(util.CV1_VECTORS, util.MODE_AB, d_all, n_global, L)
            one2 = sharded.MeshPart(dims[0], dims[1], dims[2], mode, d_all, n_global, L)
            b1 = sharded.HipCvSetBackend([lam1, one2], grid3, **KW)
            for t_ in range(3):
                b1.force_pass(b1.cv_pass(), t_)
            one2.forces(unit_bias.data_ptr())
            torch.cuda.synchronize()
            st1 = b1.state()
            f_slab, f_one = part.force.cpu().numpy(), one2.force.cpu().numpy()
            # the ORACLE on the whole snapshot (ref_mesh_*: OrderParameterMesh.cc:517-968), bias factor one
            import mtd_ref
            rmesh = mtd_ref.Mesh(dims[0], dims[1], dims[2], mode)
            opt = util.oracle_postype(pos, types)
            s_orc = rmesh.cv(opt, mtd_ref.Box.make(L))
            f_orc = rmesh.forces(opt, mtd_ref.Box.make(L), 1.0)
            slab[key].update(cv=[st_slab["cv"][1], st1["cv"][1], s_orc],
                             cv_rel=abs(st_slab["cv"][1] - st1["cv"][1]) / abs(st1["cv"][1]),
                             cv_rel_oracle=abs(st_slab["cv"][1] - s_orc) / abs(s_orc),
                             force_rel=float(np.abs(f_slab[:, :3] - f_one[sl, :3]).max() / np.abs(f_one[:, :3]).max()),
                             force_rel_oracle=float(np.abs(f_slab[:, :3] - f_orc[sl, :3]).max() / np.abs(f_orc[:, :3]).max()),
                             force_max=float(np.abs(f_one[:, :3]).max()))
            b1.close()
            probe.close()
        dist.barrier()
        bs.close()
    out["slab"] = slab
    dist.barrier()
    cs.close()
    be.close()
    box.close()
    if rank == 0:
        print(json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
