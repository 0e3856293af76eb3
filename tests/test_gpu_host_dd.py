"""GPU: domain decomposition INSIDE the C++ host classes, for every collective variable the reference runs under MPI
(OrderParameterMesh.cc:630, 911 / OrderParameterMeshGPU.cc:235, 494; SteinhardtQl.cc:183-191; WellTemperedEnsemble.cc:57-63;
CollectiveWrapper.cc:64-70; IntegratorMetaDynamics.cc:1259-1268) — the same `metadynamics.cv` / `integrate` script on every
rank, `System::run` in C++, the execution configuration's mailbox in the role of HOOMD's MPI communicator.  The ranks are
separate processes sharing cuda:0 (tests/_host_dd_worker.py); every result is compared with the ORACLE on the whole snapshot
and must be bit-identical across the ranks (the bias grid is replicated, never broadcast)."""
import pytest

from test_gpu_comm import _run_world

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 3])
def test_host_classes_domain_decomposed(world):
    r = _run_world(world, 0, worker="_host_dd_worker.py", timeout=400)
    assert r["connected"], "the mailbox could not be set up between processes on this box"
    # config 3's set (cv.lamellar + cv.mesh), replicated mesh summed by remote loads and slab-decomposed mesh
    for key in ("mesh_replicated", "mesh_slab"):
        v = r[key]
        assert v["timeouts"] == 0 and not v["fused"] and v["hills"] == 4, (key, v)
        assert v["exchange"] == "xgmi-pull", v                        # (no RCCL between two processes on one device)
        assert v["cv_rel"][0] < 1e-6 and v["cv_rel"][1] < 1e-8, (key, v)       # stated tolerance: 1e-6 on CV values
        assert v["getCurrentValue_rel"][0] < 1e-6 and v["getCurrentValue_rel"][1] < 1e-8, (key, v)
        assert v["bias_rel"] < 1e-7 and v["V_rel"] < 1e-9, (key, v)           # the oracle's grid on the device's CV values
        assert v["force_rel"][0] < 1e-5 and v["force_rel"][1] < 1e-5, (key, v)  # stated tolerance: 1e-5 of max|F|
        assert v["replicated_bitwise"], (key, v)
    # cv.steinhardt over z slabs with ghost particles
    s = r["steinhardt"]
    assert s["timeouts"] == 0 and s["locals_total"] == s["n_global"] and s["ghosts_total"] > 0, s
    assert s["cv_rel"] < 1e-10 and s["getCurrentValue_rel"] < 1e-10 and s["Q6_abs"] < 1e-11, s
    assert s["bias_rel"] < 1e-7 and s["force_rel"] < 1e-7 and s["replicated_bitwise"], s
    # cv.potential_energy: every rank's external energy counts (WellTemperedEnsemble.cc:56-63)
    e = r["potential_energy"]
    assert e["timeouts"] == 0 and e["cv_rel"] < 1e-13 and max(e["scaled_abs_err"]) < 1e-12 and e["replicated_bitwise"], e
    # cv.wrap + a lamellar CV under an umbrella: the generic path against one rank holding everything
    w = r["wrap_umbrella"]
    assert w["timeouts"] == 0 and w["replicated_bitwise"], w
    assert w["energy_cv"][0] == pytest.approx(w["energy_cv"][1], rel=1e-12)
    assert max(w["cv_rel"]) < 1e-6 and w["bias_rel"] < 1e-5 and w["V_rel"] < 1e-5 and w["umbrella_rel"] < 1e-5, w
    assert abs(w["bias_one"][1]) > 1e-6, w                                   # (the wrapped compute's arrays are scaled by a bias that is not zero)
    assert w["f_lam_rel"] < 1e-5 and w["f_wrap_rel"] < 1e-5, w
    # adaptive Gaussians: the width matrix from derivative products summed over the ranks, against the oracle
    a = r["adaptive"]
    assert a["timeouts"] == 0 and a["hills"] == 3 and a["replicated_bitwise"], a
    assert a["sigma_inv_rel"] < 1e-6 and a["box_cv_diag"] == pytest.approx(100.0) and a["cv_rel"] < 1e-6, a
