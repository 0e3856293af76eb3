// metadynamics_host.cc — see metadynamics_host.h.  Reference citations are file:line under
// /root/reference/metadynamics/.
#include "metadynamics_host.h"
#include "grid_file.h"
#include "prof.h"

#include <algorithm>
#include <sys/stat.h>

#include <cstdio>
#include <iomanip>
#include <iostream>
#include <sstream>

namespace mtdhost
{

// ------------------------------------------------------------------------------------------------
// CollectiveVariable
// ------------------------------------------------------------------------------------------------

CollectiveVariable::CollectiveVariable(std::shared_ptr<SystemDefinition> sysdef, const std::string &name)
    : ForceCompute(sysdef), m_bias(0.0), m_bias_device(nullptr), m_cv_name(name), m_umbrella(no_umbrella), m_cv0(0.0),
      m_kappa(1.0), m_width_flat(0.0), m_scale(1.0)
    {
    }

void CollectiveVariable::enqueueCurrentValue(unsigned int timestep, mtd_metad *engine, unsigned int slot)
    {
    mtd_check(mtd_metad_set_cv_value(engine, slot, getCurrentValue(timestep)), "mtd_metad_set_cv_value");
    }

// CollectiveVariable.cc:22-66
void CollectiveVariable::computeForces(unsigned int timestep)
    {
    if (m_umbrella != no_umbrella)
        {
        // the umbrella adds to a HOST bias factor: fetch the device-resident one first (synchronises)
        if (m_bias_device)
            {
            double b = 0.0;
            hip_check(hipMemcpy(&b, m_bias_device, sizeof(double), hipMemcpyDeviceToHost), "bias read-back");
            setBiasFactor(b);
            }
        double val = getCurrentValue(timestep);
        if ((val < m_cv0 + m_width_flat / 2.0) && (val > m_cv0 - m_width_flat / 2.0))
            {
            // leave bias as it is
            }
        else
            {
            double delta = 0.0;
            if (val > m_cv0)
                delta = val - m_cv0 - m_width_flat / 2.0;
            else
                delta = val - m_cv0 + m_width_flat / 2.0;

            if (m_umbrella == linear)
                setBiasFactor(m_bias + m_scale * 1.0);
            else if (m_umbrella == harmonic)
                setBiasFactor(m_bias + m_kappa * delta);
            else if (m_umbrella == wall)
                setBiasFactor(m_bias + m_scale * 12.0 * std::pow(delta / m_kappa, 11.0) / m_kappa);
            else if (m_umbrella == gaussian)
                setBiasFactor(m_bias - m_scale * (val - m_cv0) * std::exp(-(val - m_cv0) * (val - m_cv0) / m_kappa / m_kappa / 2.0));
            }
        }

    computeBiasForces(timestep);

    // reset bias factor
    setBiasFactor(0.0);
    }

// CollectiveVariable.cc:68-106
double CollectiveVariable::getUmbrellaPotential(unsigned int timestep)
    {
    if (m_umbrella != no_umbrella)
        {
        double val = getCurrentValue(timestep);
        if ((val < m_cv0 + m_width_flat / 2.0) && (val > m_cv0 - m_width_flat / 2.0)) return 0.0;
        double delta = 0.0;
        if (val > m_cv0)
            delta = val - m_cv0 - m_width_flat / 2.0;
        else if (val < m_cv0)
            delta = val - m_cv0 + m_width_flat / 2.0;
        if (m_umbrella == linear) return m_scale * delta;
        if (m_umbrella == harmonic) return (1.0 / 2.0) * delta * delta * m_kappa;
        if (m_umbrella == wall) return m_scale * std::pow(delta / m_kappa, 12.0);
        if (m_umbrella == gaussian) return m_scale * std::exp(-(val - m_cv0) * (val - m_cv0) / m_kappa / m_kappa / 2.0) - m_scale;
        }
    return 0.0;
    }

// ------------------------------------------------------------------------------------------------
// LamellarOrderParameterGPU
// ------------------------------------------------------------------------------------------------

LamellarOrderParameterGPU::LamellarOrderParameterGPU(std::shared_ptr<SystemDefinition> sysdef, const std::vector<double> &mode,
                                                     const std::vector<int3> &lattice_vectors, const std::string &suffix)
    : CollectiveVariable(sysdef, "cv_lamellar"), m_mode(mode), m_lattice_vectors(lattice_vectors), m_n_partials(0), m_cv(0.0),
      m_cv_last_updated(0)
    {
    if (mode.size() != m_pdata->getNTypes())                          // LamellarOrderParameter.cc:14-18
        throw std::runtime_error("cv.lamellar: Number of mode parameters has to equal the number of particle types!");
    if (lattice_vectors.empty() || lattice_vectors.size() > MTD_MAX_MODES)
        throw std::runtime_error("cv.lamellar: between 1 and " + std::to_string(MTD_MAX_MODES) + " lattice vectors are supported");
    if (mode.size() > MTD_MAX_TYPES) throw std::runtime_error("cv.lamellar: too many particle types");
    m_cv_name += suffix;
    m_log_name = m_cv_name;

    std::memset(&m_set, 0, sizeof(m_set));
    m_set.n_cv = 1;
    m_set.n_types = (unsigned int)mode.size();
    m_set.n_modes = (unsigned int)lattice_vectors.size();
    m_set.first[0] = 0;
    m_set.first[1] = m_set.n_modes;
    for (unsigned int k = 0; k < m_set.n_modes; ++k)
        {
        m_set.hkl[k][0] = lattice_vectors[k].x;
        m_set.hkl[k][1] = lattice_vectors[k].y;
        m_set.hkl[k][2] = lattice_vectors[k].z;
        }
    for (unsigned int t = 0; t < m_set.n_types; ++t) m_set.coeff[0][t] = mode[t];

    m_partials.resize(sizeof(double) * mtd_lamellar_scratch_doubles(m_pdata->getN()));
    m_cv_dev.resize(sizeof(double));
    }

void LamellarOrderParameterGPU::enqueuePartials()
    {
    const mtd_box box = m_pdata->getGlobalBox().toMtd();
    mtd_check(mtd_lamellar_cv_partials(&m_set, m_pdata->getN(), m_pdata->positionsPtr(), m_pdata->getDtype(), &box,
                                       (double *)m_partials.data(), &m_n_partials, m_exec_conf->getStream()),
              "mtd_lamellar_cv_partials");
    }

void LamellarOrderParameterGPU::enqueueCurrentValue(unsigned int timestep, mtd_metad *engine, unsigned int slot)
    {
    enqueuePartials();
    if (distributed())
        {
        // reduce Fourier modes on all processors (LamellarOrderParameterGPU.cc:69-77): this rank's sum, the mailbox, then 1 / N_global
        hipStream_t s = m_exec_conf->getStream();
        mtd_check(mtd_reduce_partials((const double *)m_partials.data(), m_n_partials, 1, 1, 1.0, 0.0, (double *)m_cv_dev.data(), s),
                  "mtd_reduce_partials");
        m_exec_conf->allreduceSmall((double *)m_cv_dev.data(), 1, s);
        mtd_check(mtd_metad_set_cv_source(engine, slot, (const double *)m_cv_dev.data(), 1, 1, 0, 1.0 / (double)m_pdata->getNGlobal(), 0.0),
                  "mtd_metad_set_cv_source");
        }
    else
        mtd_check(mtd_metad_set_cv_source(engine, slot, (const double *)m_partials.data(), m_n_partials, 1, 0,
                                          1.0 / (double)m_pdata->getNGlobal(), 0.0),
                  "mtd_metad_set_cv_source");
    m_cv_last_updated = timestep;
    }

double LamellarOrderParameterGPU::getCurrentValue(unsigned int timestep)
    {
    ProfRange prof_range("Lamellar");
    enqueuePartials();
    hipStream_t s = m_exec_conf->getStream();
    const double inv_n = 1.0 / (double)m_pdata->getNGlobal();
    mtd_check(mtd_reduce_partials((const double *)m_partials.data(), m_n_partials, 1, 1, distributed() ? 1.0 : inv_n, 0.0,
                                  (double *)m_cv_dev.data(), s),
              "mtd_reduce_partials");
    if (distributed()) m_exec_conf->allreduceSmall((double *)m_cv_dev.data(), 1, s);   // LamellarOrderParameterGPU.cc:69-77
    m_exec_conf->sync();
    m_cv_dev.download(&m_cv, sizeof(double));
    if (distributed()) m_cv = 0.0 + inv_n * m_cv;                      // (shift + scale * sum: the grid engine's own expression)
    m_cv_last_updated = timestep;
    return m_cv;
    }

void LamellarOrderParameterGPU::computeBiasForces(unsigned int timestep)
    {
    ProfRange prof_range("Lamellar");
    const mtd_box box = m_pdata->getGlobalBox().toMtd();
    void *f[1] = {m_force.data()};
    if (m_bias_device)
        mtd_check(mtd_lamellar_forces(&m_set, m_pdata->getN(), m_pdata->positionsPtr(), f, m_pdata->getDtype(),
                                      m_pdata->getNGlobal(), m_bias_device, &box, m_exec_conf->getStream()),
                  "mtd_lamellar_forces");
    else
        {
        std::vector<int> lat(3 * m_lattice_vectors.size());
        for (size_t k = 0; k < m_lattice_vectors.size(); ++k)
            {
            lat[3 * k] = m_lattice_vectors[k].x;
            lat[3 * k + 1] = m_lattice_vectors[k].y;
            lat[3 * k + 2] = m_lattice_vectors[k].z;
            }
        mtd_check(mtd_compute_sq_forces(m_pdata->getN(), m_pdata->positionsPtr(), m_force.data(), m_pdata->getDtype(),
                                        (unsigned int)m_lattice_vectors.size(), lat.data(), m_mode.data(),
                                        (unsigned int)m_mode.size(), m_pdata->getNGlobal(), m_bias, &box, m_exec_conf->getStream()),
                  "mtd_compute_sq_forces");
        }
    }

// ------------------------------------------------------------------------------------------------
// WellTemperedEnsemble
// ------------------------------------------------------------------------------------------------

WellTemperedEnsemble::WellTemperedEnsemble(std::shared_ptr<SystemDefinition> sysdef, const std::string &name)
    : CollectiveVariable(sysdef, name), m_pe(0.0), m_log_name("cv_potential_energy"), m_n_partials(0)
    {
    m_partials.resize(sizeof(double) * mtd_wte_scratch_doubles(m_pdata->getN()));
    m_sum.resize(sizeof(double));
    }

void WellTemperedEnsemble::enqueuePartials()
    {
    mtd_check(mtd_wte_energy_partials(m_pdata->getN(), m_pdata->getNetForce().data(), m_pdata->getDtype(),
                                      (double *)m_partials.data(), &m_n_partials, m_exec_conf->getStream()),
              "mtd_wte_energy_partials");
    }

void WellTemperedEnsemble::enqueueCurrentValue(unsigned int, mtd_metad *engine, unsigned int slot)
    {
    enqueuePartials();
    // PE = sum_j net_force_j.w + external energy (WellTemperedEnsemble.cc:45-56)
    if (distributed())
        {
        // ... of this rank, then the sum over the ranks (:57-63: every rank adds ITS external energy before the MPI_Allreduce)
        hipStream_t s = m_exec_conf->getStream();
        mtd_check(mtd_reduce_partials((const double *)m_partials.data(), m_n_partials, 1, 1, 1.0, m_pdata->getExternalEnergy(),
                                      (double *)m_sum.data(), s),
                  "mtd_reduce_partials");
        m_exec_conf->allreduceSmall((double *)m_sum.data(), 1, s);
        mtd_check(mtd_metad_set_cv_source(engine, slot, (const double *)m_sum.data(), 1, 1, 0, 1.0, 0.0), "mtd_metad_set_cv_source");
        return;
        }
    mtd_check(mtd_metad_set_cv_source(engine, slot, (const double *)m_partials.data(), m_n_partials, 1, 0, 1.0,
                                      m_pdata->getExternalEnergy()),
              "mtd_metad_set_cv_source");
    }

double WellTemperedEnsemble::getCurrentValue(unsigned int)
    {
    ProfRange prof_range("Well-Tempered Ensemble");
    enqueuePartials();
    mtd_check(mtd_reduce_partials((const double *)m_partials.data(), m_n_partials, 1, 1, 1.0, m_pdata->getExternalEnergy(),
                                  (double *)m_sum.data(), m_exec_conf->getStream()),
              "mtd_reduce_partials");
    if (distributed()) m_exec_conf->allreduceSmall((double *)m_sum.data(), 1, m_exec_conf->getStream());   // :57-63
    m_exec_conf->sync();
    m_sum.download(&m_pe, sizeof(double));
    return m_pe;
    }

// WellTemperedEnsemble.cc:135-188; the CPU path (the parity target) also scales net_torque.w (Q18)
void WellTemperedEnsemble::computeBiasForces(unsigned int)
    {
    ProfRange prof_range("Well-Tempered Ensemble");
    mtd_check(mtd_wte_scale_netforce(m_pdata->getN(), m_pdata->getNetForce().data(), m_pdata->getNetTorqueArray().data(),
                                     m_pdata->getNetVirial().data(), m_pdata->getNetVirialPitch(), m_pdata->getDtype(),
                                     m_bias_device, m_bias, 1, m_exec_conf->getStream()),
              "mtd_wte_scale_netforce");
    bool any_virial = false;
    for (unsigned int i = 0; i < 6; ++i) any_virial = any_virial || (m_pdata->getExternalVirial(i) != 0.0);
    if (any_virial)
        {
        double bias = m_bias;
        if (m_bias_device) hip_check(hipMemcpy(&bias, m_bias_device, sizeof(double), hipMemcpyDeviceToHost), "bias read-back");
        const double fac = 1.0 + bias;
        for (unsigned int i = 0; i < 6; ++i) m_pdata->setExternalVirial(i, fac * m_pdata->getExternalVirial(i));   // :180-184
        }
    }

// ------------------------------------------------------------------------------------------------
// OrderParameterMeshGPU
// ------------------------------------------------------------------------------------------------

OrderParameterMeshGPU::OrderParameterMeshGPU(std::shared_ptr<SystemDefinition> sysdef, unsigned int nx, unsigned int ny,
                                             unsigned int nz, std::vector<double> mode, std::vector<int3> zero_modes)
    : CollectiveVariable(sysdef, "mesh"), m_keep_fourier(false), m_q_max_last_computed(0), m_q_max{0.0, 0.0, 0.0}, m_sq_max(0.0),
      m_mesh(nullptr), m_mode(mode), m_zero_modes(zero_modes), m_k_min(0.0), m_k_max(0.0), m_delta_k(0.0), m_use_table(false),
      m_is_first_step(true), m_partials(nullptr), m_n_partials(0), m_cv_last_updated(0), m_cv(0.0)
    {
    if (mode.size() != m_pdata->getNTypes()) throw std::runtime_error("Error setting up cv.mesh");   // OrderParameterMesh.cc:44-49
    int rc = mtd_mesh_create(&m_mesh, nx, ny, nz, mode.data(), (unsigned int)mode.size(), m_pdata->getN());
    if (rc == MTD_ERR_UNSUPPORTED)
        throw std::runtime_error("cv.mesh: mesh points per direction: 4 ... 256, or a power of two up to 1024");
    mtd_check(rc, "mtd_mesh_create");
    // the normalised Fourier mesh is written only once a log quantity (q_max) or the virial has asked for it
    mtd_check(mtd_mesh_set_keep_fourier(m_mesh, 0), "mtd_mesh_set_keep_fourier");
    m_keep_fourier = false;
    m_cv_dev.resize(sizeof(double));
    }

// q_max / virial read the Fourier mesh: from the first request on every spectral step keeps it; the step at hand is redone
// from the real mesh, which is still in place (same CV partial sums)
void OrderParameterMeshGPU::needFourierMesh()
    {
    if (m_keep_fourier) return;
    m_keep_fourier = true;
    mtd_check(mtd_mesh_set_keep_fourier(m_mesh, 1), "mtd_mesh_set_keep_fourier");
    if (!m_is_first_step)
        {
        const mtd_box box = m_pdata->getGlobalBox().toMtd();
        mtd_check(mtd_mesh_spectral(m_mesh, &box, m_pdata->getNGlobal(), &m_partials, &m_n_partials, m_exec_conf->getStream()), "mtd_mesh_spectral");
        }
    }

OrderParameterMeshGPU::~OrderParameterMeshGPU()
    {
    if (m_mesh) (void)mtd_mesh_destroy(m_mesh);
    }

void OrderParameterMeshGPU::setBugCompatible(bool on)
    {
    mtd_check(mtd_mesh_set_bug_compat(m_mesh, on ? 1 : 0), "mtd_mesh_set_bug_compat");
    m_is_first_step = true;
    }

// OrderParameterMesh.cc:148-189
void OrderParameterMeshGPU::setTable(const std::vector<double> &K, const std::vector<double> &d_K, double kmin, double kmax)
    {
    if (kmin < 0 || kmax < 0 || kmax <= kmin) throw std::runtime_error("Error setting up OrderParameterMesh");
    if (K.size() != d_K.size()) throw std::runtime_error("Error setting up OrderParameterMesh");
    m_k_min = kmin;
    m_k_max = kmax;
    m_delta_k = (kmax - kmin) / (double)(K.size() - 1);
    m_table = K;
    m_table_d = d_K;
    mtd_check(mtd_mesh_set_table(m_mesh, K.data(), d_K.data(), (unsigned int)K.size(), kmin, kmax), "mtd_mesh_set_table");
    }

void OrderParameterMeshGPU::setUseTable(bool use_table)
    {
    if (use_table && m_table.empty()) throw std::runtime_error("cv.mesh: set_kernel() before use_table=True");
    m_use_table = use_table;
    mtd_check(mtd_mesh_set_use_table(m_mesh, use_table ? 1 : 0), "mtd_mesh_set_use_table");
    }

// :1108-1179
void OrderParameterMeshGPU::computeQmax(unsigned int timestep)
    {
    enqueueCV(timestep);                                               // compute Fourier grid (:1111)
    needFourierMesh();
    if (timestep && m_q_max_last_computed == timestep) return;         // :1113
    m_q_max_last_computed = timestep;
    const mtd_box box = m_pdata->getGlobalBox().toMtd();
    double out[4];
    mtd_check(mtd_mesh_qmax(m_mesh, &box, m_pdata->getNGlobal(), out, m_exec_conf->getStream()), "mtd_mesh_qmax");
    m_q_max[0] = out[0]; m_q_max[1] = out[1]; m_q_max[2] = out[2];
    m_sq_max = out[3];
    }

// :1077-1106
double OrderParameterMeshGPU::getLogValue(const std::string &quantity, unsigned int timestep)
    {
    if (quantity == "cv_mesh") return getCurrentValue(timestep);
    if (quantity == "qx_max" || quantity == "qy_max" || quantity == "qz_max" || quantity == "sq_max")
        {
        computeQmax(timestep);
        if (quantity == "qx_max") return m_q_max[0];
        if (quantity == "qy_max") return m_q_max[1];
        if (quantity == "qz_max") return m_q_max[2];
        return m_sq_max;
        }
    return CollectiveVariable::getLogValue(quantity, timestep);
    }

// :970-1050 — needs the host value of the bias factor (the reference multiplies on the host too)
void OrderParameterMeshGPU::computeVirial()
    {
    double bias = m_bias;
    if (m_bias_device)
        {
        m_exec_conf->sync();
        hip_check(hipMemcpy(&bias, m_bias_device, sizeof(double), hipMemcpyDeviceToHost), "bias read-back");
        }
    needFourierMesh();
    const mtd_box box = m_pdata->getGlobalBox().toMtd();
    double v[6];
    mtd_check(mtd_mesh_virial(m_mesh, &box, m_pdata->getNGlobal(), bias, v, m_exec_conf->getStream()), "mtd_mesh_virial");
    for (unsigned int i = 0; i < 6; ++i) m_external_virial[i] = v[i];
    }

void OrderParameterMeshGPU::setSlabDecomposition(bool on)
    {
    if (m_slab_attached && !on) throw std::runtime_error("cv.mesh: the slab decomposition cannot be switched off once it is attached");
    m_slab = on;
    m_is_first_step = true;
    }

// the four exported buffers of the slab decomposition (local assignment, transformed slab, pencils of G, slab of Re(inv)):
// allocated on every rank, their handles gathered through the control plane, mapped, handed to the mesh.  Collective.
void OrderParameterMeshGPU::attachSlab()
    {
    if (m_slab_attached) return;
    const unsigned int world = m_exec_conf->getNRanks();
    size_t sizes[4];
    int rc = mtd_mesh_slab_bytes(m_mesh, world, sizes);
    if (rc == MTD_ERR_INVALID_ARGUMENT || rc == MTD_ERR_UNSUPPORTED)
        throw std::runtime_error("cv.mesh: the slab decomposition needs ny and nz to be multiples of the number of ranks");
    mtd_check(rc, "mtd_mesh_slab_bytes");
    std::vector<void *> peers[4];
    for (int k = 0; k < 4; ++k) m_exec_conf->shareBuffer(sizes[k], peers[k]);
    mtd_check(mtd_mesh_slab_attach(m_mesh, m_exec_conf->getMailbox(), peers[0].data(), peers[1].data(), peers[2].data(), peers[3].data()),
              "mtd_mesh_slab_attach");
    m_slab_attached = true;
    }

bool OrderParameterMeshGPU::armLamellarRider(unsigned int timestep, mtd_metad *engine, const mtd_lamellar_set *set, double *d_partials,
                                             unsigned int *n_partials)
    {
    if (distributed()) return false;                                   // the riders' sums are this rank's only
    if (m_cv_last_updated == timestep && !m_is_first_step) return false;    // enqueueCV would not launch anything this step
    const mtd_box box = m_pdata->getGlobalBox().toMtd();
    const int rc = mtd_mesh_set_lamellar_rider(m_mesh, engine, set, &box, m_pdata->getN(), d_partials, n_partials, m_exec_conf->getStream());
    if (rc == MTD_ERR_UNSUPPORTED) return false;
    mtd_check(rc, "mtd_mesh_set_lamellar_rider");
    return true;
    }

bool OrderParameterMeshGPU::clearRider()
    {
    int was = 0;
    mtd_check(mtd_mesh_clear_rider(m_mesh, &was), "mtd_mesh_clear_rider");
    return was != 0;
    }

void OrderParameterMeshGPU::enqueueCV(unsigned int timestep)
    {
    ProfRange prof_range("Mesh");
    if (m_cv_last_updated == timestep && !m_is_first_step) return;   // :927-928
    const mtd_box box = m_pdata->getGlobalBox().toMtd();
    hipStream_t s = m_exec_conf->getStream();
    if (distributed() && m_slab)
        {
        // the mesh decomposed over the ranks: z slabs for the x / y passes, y rows for the z pass, the transposes as remote loads
        // (OrderParameterMesh.cc:263-316, 659-746: ghost-cell exchange + dfft); every exchange is inside the call
        attachSlab();
        mtd_check(mtd_mesh_slab_compute_cv(m_mesh, m_pdata->getN(), m_pdata->positionsPtr(), m_pdata->getDtype(), &box,
                                           m_pdata->getNGlobal(), &m_slab_sum, s),
                  "mtd_mesh_slab_compute_cv");
        m_partials = m_slab_sum;
        m_n_partials = 1;
        }
    else if (distributed())
        {
        // replicated mesh: this rank's particles are spread, the ranks sum their meshes and sum(mode^2) — M + 1 doubles, what the
        // reference moves as ghost cells + the MPI_Allreduce of :630 (OrderParameterMeshGPU.cc:235) — and every rank transforms
        // the whole mesh; the CV is then the same on every rank (the reference's sum over local cells, :911 / GPU.cc:494)
        mtd_check(mtd_mesh_assign(m_mesh, m_pdata->getN(), m_pdata->positionsPtr(), m_pdata->getDtype(), &box, s), "mtd_mesh_assign");
        double *buf = nullptr;
        size_t count = 0;
        mtd_check(mtd_mesh_exchange_buffer(m_mesh, &buf, &count), "mtd_mesh_exchange_buffer");
        m_exec_conf->allreduceLarge(buf, count, s);
        mtd_check(mtd_mesh_spectral(m_mesh, &box, m_pdata->getNGlobal(), &m_partials, &m_n_partials, s), "mtd_mesh_spectral");
        }
    else
        mtd_check(mtd_mesh_compute_cv(m_mesh, m_pdata->getN(), m_pdata->positionsPtr(), m_pdata->getDtype(), &box,
                                      m_pdata->getNGlobal(), &m_partials, &m_n_partials, s),
                  "mtd_mesh_compute_cv");
    m_is_first_step = false;
    m_cv_last_updated = timestep;
    }

void OrderParameterMeshGPU::enqueueCurrentValue(unsigned int timestep, mtd_metad *engine, unsigned int slot)
    {
    enqueueCV(timestep);
    mtd_check(mtd_metad_set_cv_source(engine, slot, m_partials, m_n_partials, 1, 0, 0.5, 0.0), "mtd_metad_set_cv_source");   // :907
    }

double OrderParameterMeshGPU::getCurrentValue(unsigned int timestep)
    {
    enqueueCV(timestep);
    mtd_check(mtd_reduce_partials(m_partials, m_n_partials, 1, 1, 0.5, 0.0, (double *)m_cv_dev.data(), m_exec_conf->getStream()),
              "mtd_reduce_partials");
    m_exec_conf->sync();
    m_cv_dev.download(&m_cv, sizeof(double));
    return m_cv;
    }

// OrderParameterMesh.cc:1052-1075
bool OrderParameterMeshGPU::forcesWithBiasUpdate(unsigned int timestep, mtd_metad *engine, unsigned int mesh_slot, const mtd_lamellar_set *set,
                                                 const unsigned int *slots, void *const *lamellar_forces)
    {
    // (the pressure tensor is formed from the Fourier mesh on the host side of computeBiasForces; a distributed mesh reduces there too)
    if (distributed() || m_pdata->getPressureFlag()) return false;
    if (m_is_first_step || m_cv_last_updated != timestep) return false;  // (the force pass walks the lists of THIS step's assignment)
    ProfRange prof_range("forces");
    const mtd_box box = m_pdata->getGlobalBox().toMtd();
    const int rc = mtd_mesh_forces_update_bias(m_mesh, engine, mesh_slot, set, slots, m_pdata->getN(), m_pdata->positionsPtr(), m_force.data(),
                                               lamellar_forces, m_pdata->getDtype(), m_pdata->getNGlobal(), &box, timestep,
                                               m_exec_conf->getStream());
    if (rc == MTD_ERR_UNSUPPORTED) return false;
    mtd_check(rc, "mtd_mesh_forces_update_bias");
    for (unsigned int i = 0; i < 6; ++i) m_external_virial[i] = 0.0;
    markComputed(timestep);
    return true;
    }

void OrderParameterMeshGPU::computeBiasForces(unsigned int timestep)
    {
    ProfRange prof_range("forces");
    if (m_is_first_step || m_cv_last_updated != timestep) enqueueCV(timestep);   // :1055-1056
    const mtd_box box = m_pdata->getGlobalBox().toMtd();
    mtd_check(mtd_mesh_forces(m_mesh, m_pdata->getN(), m_pdata->positionsPtr(), m_force.data(), m_pdata->getDtype(), &box,
                              m_pdata->getNGlobal(), m_bias_device, m_bias, m_exec_conf->getStream()),
              "mtd_mesh_forces");
    if (m_pdata->getPressureFlag())                                    // :1062-1067 (pdata_flag::pressure_tensor / isotropic_virial)
        computeVirial();
    else
        for (unsigned int i = 0; i < 6; ++i) m_external_virial[i] = 0.0;
    }

// ------------------------------------------------------------------------------------------------
// CollectiveWrapper
// ------------------------------------------------------------------------------------------------

CollectiveWrapper::CollectiveWrapper(std::shared_ptr<SystemDefinition> sysdef, std::shared_ptr<ForceCompute> fc,
                                     const std::string &name)
    : CollectiveVariable(sysdef, name), m_fc(fc), m_energy(0.0), m_n_partials(0)
    {
    if (!fc) throw std::runtime_error("cv.wrap: a force is required");
    m_partials.resize(sizeof(double) * mtd_wte_scratch_doubles(m_pdata->getN()));
    m_sum.resize(sizeof(double));
    }

void CollectiveWrapper::enqueuePartials(unsigned int timestep)
    {
    m_fc->compute(timestep);                                           // CollectiveWrapper.cc:35 (once per time step)
    mtd_check(mtd_wte_energy_partials(m_pdata->getN(), m_fc->getForceArray().data(), m_pdata->getDtype(),
                                      (double *)m_partials.data(), &m_n_partials, m_exec_conf->getStream()),
              "mtd_wte_energy_partials");
    }

void CollectiveWrapper::enqueueCurrentValue(unsigned int timestep, mtd_metad *engine, unsigned int slot)
    {
    enqueuePartials(timestep);
    // energy = sum_j force_j.w + external energy of the wrapped compute (:50-61)
    if (distributed())
        {
        hipStream_t s = m_exec_conf->getStream();                      // :64-70: summed over the ranks
        mtd_check(mtd_reduce_partials((const double *)m_partials.data(), m_n_partials, 1, 1, 1.0, m_fc->getExternalEnergy(),
                                      (double *)m_sum.data(), s),
                  "mtd_reduce_partials");
        m_exec_conf->allreduceSmall((double *)m_sum.data(), 1, s);
        mtd_check(mtd_metad_set_cv_source(engine, slot, (const double *)m_sum.data(), 1, 1, 0, 1.0, 0.0), "mtd_metad_set_cv_source");
        return;
        }
    mtd_check(mtd_metad_set_cv_source(engine, slot, (const double *)m_partials.data(), m_n_partials, 1, 0, 1.0,
                                      m_fc->getExternalEnergy()),
              "mtd_metad_set_cv_source");
    }

double CollectiveWrapper::getCurrentValue(unsigned int timestep)
    {
    enqueuePartials(timestep);
    mtd_check(mtd_reduce_partials((const double *)m_partials.data(), m_n_partials, 1, 1, 1.0, m_fc->getExternalEnergy(),
                                  (double *)m_sum.data(), m_exec_conf->getStream()),
              "mtd_reduce_partials");
    if (distributed()) m_exec_conf->allreduceSmall((double *)m_sum.data(), 1, m_exec_conf->getStream());   // :64-70
    m_exec_conf->sync();
    m_sum.download(&m_energy, sizeof(double));
    return m_energy;
    }

// CollectiveWrapper.cc:136-179 — the wrapped compute's own arrays are scaled by the bias factor; the CPU path (the parity
// target) also scales torque.w
void CollectiveWrapper::computeBiasForces(unsigned int timestep)
    {
    ProfRange prof_range("Collective wrap");
    m_fc->compute(timestep);                                           // :139
    mtd_check(mtd_wrapper_scale_forces(m_pdata->getN(), m_fc->getForceArray().data(), m_fc->getTorqueArray().data(),
                                       m_fc->getVirialArray().data(), m_fc->getVirialPitch(), m_pdata->getDtype(), m_bias_device,
                                       m_bias, 1, m_exec_conf->getStream()),
              "mtd_wrapper_scale_forces");
    }

// ------------------------------------------------------------------------------------------------
// SteinhardtQl
// ------------------------------------------------------------------------------------------------

SteinhardtQl::SteinhardtQl(std::shared_ptr<SystemDefinition> sysdef, double rcut, double ron, unsigned int lmax,
                           std::shared_ptr<NeighborList> nlist, unsigned int type, const std::vector<double> &Ql_ref,
                           const std::string &log_suffix)
    : CollectiveVariable(sysdef, "steinhardt" + log_suffix), m_rcut(rcut), m_ron(ron), m_lmax(lmax), m_nlist(nlist), m_type(type),
      m_Ql_ref(Ql_ref), m_Ql(lmax + 1, 0.0), m_cv_last_updated(0), m_have_computed(false), m_value(0.0), m_d_value(nullptr),
      m_d_Ql(nullptr), m_d_Qlm(nullptr), m_sym_version(0), m_sym_ok(false)
    {
    if (Ql_ref.size() != lmax + 1) throw std::runtime_error("Error setting up Steinhardt CV");   // SteinhardtQl.cc:25-29
    if (lmax > 12) throw std::runtime_error("cv.steinhardt: lmax <= 12 in this build");
    if (!nlist) throw std::runtime_error("cv.steinhardt: a neighbour list is required");
    m_scratch.resize(sizeof(double) * mtd_ql_scratch_doubles(lmax));
    }

// the arrays and the list mode of this step's mtd_ql_* calls.  A half list (NeighborList::half: every pair once, reaction force by
// Newton's third law, SteinhardtQl.cc:80, 173-179, 328-333) is turned ONCE PER LIST UPDATE into the symmetric full list it
// stands for; the symmetric-full-list mode then gives the half-list result (each pair counted once in the CV pass, even degrees
// doubled, odd ones zero) with a force pass that gathers instead of scattering reaction forces with atomics.  A half list with
// ghost particles keeps the third-law pass.
SteinhardtQl::Lists SteinhardtQl::lists()
    {
    Lists l;
    l.head = (const unsigned int *)m_nlist->getHeadList().data();
    l.n_neigh = (const unsigned int *)m_nlist->getNNeighArray().data();
    l.nlist = (const unsigned int *)m_nlist->getNListArray().data();
    l.mode = m_nlist->getStorageMode() == NeighborList::half ? 1 : (m_nlist->isSymmetricFull() ? 2 : 0);
    static const bool keep_third_law = std::getenv("MTD_QL_HALF_THIRD_LAW") != nullptr;    // diagnostic: the atomic pass
    if (l.mode != 1 || keep_third_law) return l;
    if (m_sym_version != m_nlist->getVersion())
        {
        const unsigned int N = m_pdata->getN();
        const size_t cap = 2 * (m_nlist->getNListArray().bytes() / sizeof(unsigned int));
        m_sym_head.resize(sizeof(unsigned int) * (N ? N : 1));
        m_sym_nneigh.resize(sizeof(unsigned int) * (N ? N : 1));
        m_sym_nlist.resize(sizeof(unsigned int) * (cap ? cap : 1));
        m_sym_work.resize(sizeof(unsigned int) * mtd_ql_symmetrize_workspace_uints(N));      // (kept across the list updates of a run)
        size_t n_full = 0;
        const int rc = mtd_ql_symmetrize_half_list_ws(N, l.head, l.n_neigh, l.nlist, (unsigned int *)m_sym_head.data(),
                                                      (unsigned int *)m_sym_nneigh.data(), (unsigned int *)m_sym_nlist.data(), cap, &n_full,
                                                      (unsigned int *)m_sym_work.data(), m_exec_conf->getStream());
        if (rc != MTD_SUCCESS && rc != MTD_ERR_UNSUPPORTED) mtd_check(rc, "mtd_ql_symmetrize_half_list");
        m_sym_ok = rc == MTD_SUCCESS;
        m_sym_version = m_nlist->getVersion();
        }
    if (m_sym_ok)
        {
        l.head = (const unsigned int *)m_sym_head.data();
        l.n_neigh = (const unsigned int *)m_sym_nneigh.data();
        l.nlist = (const unsigned int *)m_sym_nlist.data();
        l.mode = 2;
        }
    return l;
    }

// the pair pass of computeCV and, in a domain-decomposed run, the sum of its result over the ranks: Q'_lm (m >= 0) in m_scratch
void SteinhardtQl::accumulateSums(unsigned int timestep)
    {
    m_nlist->compute(timestep);                                      // :68
    const mtd_box box = m_pdata->getBox().toMtd();
    const Lists l = lists();
    // this rank's central particles (their neighbours may be ghosts, stored behind the local particles), the sum of the
    // Q'_lm over the ranks (SteinhardtQl.cc:183-191)
    if (distributed() && l.mode == 1 && m_pdata->getNGhosts())
        throw std::runtime_error("cv.steinhardt: a half neighbour list cannot be combined with ghost particles (the reaction "
                                 "force on a ghost is dropped, SteinhardtQl.cc:328): use a full list in domain-decomposed runs");
    hipStream_t s = m_exec_conf->getStream();
    double *d_sums = nullptr;
    unsigned int n_sums = 0;
    mtd_check(mtd_ql_accumulate_local(m_pdata->getN(), m_pdata->positionsPtr(), m_pdata->getDtype(), &box, l.head, l.n_neigh, l.nlist,
                                      l.mode, m_rcut, m_ron, m_lmax, m_type, m_pdata->getNGlobal(), (double *)m_scratch.data(), &d_sums,
                                      &n_sums, s),
              "mtd_ql_accumulate_local");
    if (distributed()) m_exec_conf->allreduceSmall(d_sums, n_sums, s);
    }

void SteinhardtQl::computeCV(unsigned int timestep)
    {
    ProfRange prof_range("CV");
    if (m_cv_last_updated == timestep && m_have_computed) return;    // :64-65
    accumulateSums(timestep);
    // Q_lm, Q_l and the value on every rank
    mtd_check(mtd_ql_finalize(lists().mode, m_lmax, m_Ql_ref.data(), m_pdata->getNGlobal(), (double *)m_scratch.data(), &m_d_value, &m_d_Ql,
                              &m_d_Qlm, m_exec_conf->getStream()),
              "mtd_ql_finalize");
    m_have_computed = true;
    m_cv_last_updated = timestep;
    }

// the only variable of the grid: the finalize step is the head of the grid engine's launch (mtd_ql_finalize_update_bias)
bool SteinhardtQl::enqueueValueAndBias(unsigned int timestep, mtd_metad *engine)
    {
    static const bool off = [] { const char *e = std::getenv("MTD_QL_MERGED"); return e && e[0] == '0'; }();   // diagnostic: the separate launches
    if (off) return false;
    ProfRange prof_range("CV");
    const bool cached = m_cv_last_updated == timestep && m_have_computed;     // (its sums are still in m_scratch)
    if (!cached) accumulateSums(timestep);
    const int rc = mtd_ql_finalize_update_bias(engine, lists().mode, m_lmax, m_Ql_ref.data(), m_pdata->getNGlobal(), (double *)m_scratch.data(),
                                               timestep, &m_d_value, &m_d_Ql, &m_d_Qlm, m_exec_conf->getStream());
    if (rc == MTD_ERR_UNSUPPORTED)
        {
        if (!cached)
            mtd_check(mtd_ql_finalize(lists().mode, m_lmax, m_Ql_ref.data(), m_pdata->getNGlobal(), (double *)m_scratch.data(), &m_d_value,
                                      &m_d_Ql, &m_d_Qlm, m_exec_conf->getStream()),
                      "mtd_ql_finalize");
        m_have_computed = true;
        m_cv_last_updated = timestep;
        return false;
        }
    mtd_check(rc, "mtd_ql_finalize_update_bias");
    m_have_computed = true;
    m_cv_last_updated = timestep;
    return true;
    }

void SteinhardtQl::enqueueCurrentValue(unsigned int timestep, mtd_metad *engine, unsigned int slot)
    {
    computeCV(timestep);
    mtd_check(mtd_metad_set_cv_source(engine, slot, m_d_value, 1, 1, 0, 1.0, 0.0), "mtd_metad_set_cv_source");
    }

double SteinhardtQl::getCurrentValue(unsigned int timestep)
    {
    computeCV(timestep);
    m_exec_conf->sync();
    hip_check(hipMemcpy(&m_value, m_d_value, sizeof(double), hipMemcpyDeviceToHost), "cv read-back");
    hip_check(hipMemcpy(m_Ql.data(), m_d_Ql, sizeof(double) * (m_lmax + 1), hipMemcpyDeviceToHost), "Ql read-back");
    return m_value;
    }

void SteinhardtQl::computeBiasForces(unsigned int timestep)
    {
    ProfRange prof_range("Force");
    m_nlist->compute(timestep);                                      // :206
    // the reference relies on the Q_lm of the computeCV the integrator triggered earlier in the step (Q20); make sure one exists
    if (!m_have_computed) computeCV(timestep);
    const mtd_box box = m_pdata->getBox().toMtd();
    const Lists l = lists();
    mtd_check(mtd_ql_forces(m_pdata->getN(), m_pdata->positionsPtr(), m_force.data(), m_pdata->getDtype(), &box, l.head, l.n_neigh, l.nlist, l.mode,
                            m_rcut, m_ron, m_lmax, m_type, m_Ql_ref.data(), m_pdata->getNGlobal(), (const double *)m_scratch.data(),
                            m_bias_device, m_bias, m_exec_conf->getStream()),
              "mtd_ql_forces");
    }

std::vector<std::string> SteinhardtQl::getProvidedLogQuantities()
    {
    auto list = CollectiveVariable::getProvidedLogQuantities();
    for (unsigned int l = 1; l <= m_lmax; l++) list.push_back("steinhardt_Q" + std::to_string(l));   // SteinhardtQl.h:38-40
    list.push_back("cv_" + m_cv_name);
    return list;
    }

double SteinhardtQl::getLogValue(const std::string &quantity, unsigned int timestep)
    {
    if (quantity == "cv_" + m_cv_name) return getCurrentValue(timestep);
    for (unsigned int l = 1; l <= m_lmax; ++l)
        if (quantity == "steinhardt_Q" + std::to_string(l))
            {
            getCurrentValue(timestep);
            return m_Ql[l - 1];                                      // off by one like the reference (SteinhardtQl.h:61, Q19)
            }
    return CollectiveVariable::getLogValue(quantity, timestep);
    }

// ------------------------------------------------------------------------------------------------
// AspectRatio, Density
// ------------------------------------------------------------------------------------------------

AspectRatio::AspectRatio(std::shared_ptr<SystemDefinition> sysdef, unsigned int dir1, unsigned int dir2)
    : CollectiveVariable(sysdef, "cv_aspect_ratio"), m_dir1(dir1), m_dir2(dir2)
    {
    if (dir1 == dir2 || dir1 >= 3 || dir2 >= 3) throw std::runtime_error("Error setting up metadynamics.aspect_ratio");   // AspectRatio.cc:8-12
    }

// AspectRatio.cc:24-57, including `length1 = L.x` in the dir2 switch (:46)
double AspectRatio::getCurrentValue(unsigned int)
    {
    const auto L = m_pdata->getGlobalBox().getL();
    double length1 = 0.0, length2 = 0.0;
    switch (m_dir1)
        {
        case 0: length1 = L[0]; break;
        case 1: length1 = L[1]; break;
        case 2: length1 = L[2]; break;
        }
    switch (m_dir2)
        {
        case 0: length1 = L[0]; break;
        case 1: length2 = L[1]; break;
        case 2: length2 = L[2]; break;
        }
    return length1 / length2;
    }

static double host_bias(double bias, const double *d_bias)
    {
    if (d_bias) hip_check(hipMemcpy(&bias, d_bias, sizeof(double), hipMemcpyDeviceToHost), "bias read-back");
    return bias;
    }

// AspectRatio.cc:59-130
void AspectRatio::computeBiasForces(unsigned int)
    {
    const double bias = host_bias(m_bias, m_bias_device);
    const BoxDim &box = m_pdata->getGlobalBox();
    const auto L = box.getL();
    double d_l_x = 0.0, d_l_y = 0.0, d_l_z = 0.0;
    switch (m_dir1)
        {
        case 0:
            if (m_dir2 == 1) { d_l_x = 1.0 / L[1]; d_l_y = -L[0] / L[1] / L[1]; }
            if (m_dir2 == 2) { d_l_x = 1.0 / L[2]; d_l_z = -L[0] / L[2] / L[2]; }
            break;
        case 1:
            if (m_dir2 == 0) { d_l_x = -L[1] / L[0] / L[0]; d_l_y = 1.0 / L[0]; }
            if (m_dir2 == 2) { d_l_y = 1.0 / L[2]; d_l_z = -L[1] / L[2] / L[2]; }
            break;
        case 2:
            if (m_dir2 == 0) { d_l_x = -L[2] / L[0] / L[0]; d_l_z = 1.0 / L[0]; }
            if (m_dir2 == 1) { d_l_y = -L[2] / L[1] / L[1]; d_l_z = 1.0 / L[1]; }
            break;
        }
    const double xy = box.getTiltFactorXY(), xz = box.getTiltFactorXZ(), yz = box.getTiltFactorYZ();
    m_external_virial[0] = -bias * d_l_x * L[0];
    m_external_virial[1] = -bias * d_l_x * (L[1] * xy);
    m_external_virial[2] = -bias * d_l_x * (L[2] * xz);
    m_external_virial[3] = -bias * d_l_y * L[1];
    m_external_virial[4] = -bias * d_l_y * (L[2] * yz);
    m_external_virial[5] = -bias * d_l_z * L[2];
    }

Density::Density(std::shared_ptr<SystemDefinition> sysdef, const std::string &suffix)
    : CollectiveVariable(sysdef, "cv_density" + (suffix != "" ? "_" + suffix : ""))   // Density.cc:8
    {
    }

double Density::getCurrentValue(unsigned int)
    {
    const double V = m_pdata->getGlobalBox().getVolume();
    return (double)m_pdata->getNGlobal() / V;                          // Density.cc:22-26
    }

void Density::computeBiasForces(unsigned int)
    {
    const double bias = host_bias(m_bias, m_bias_device);
    const double V = m_pdata->getGlobalBox().getVolume();
    const double fac = -(double)m_pdata->getNGlobal() / (V * V);       // Density.cc:44
    const auto L = m_pdata->getGlobalBox().getL();
    const double v = -bias * fac * L[0] * L[1] * L[2];
    m_external_virial = {v, 0.0, 0.0, v, 0.0, v};                      // Density.cc:47-52
    }

// ------------------------------------------------------------------------------------------------
// IntegratorMetaDynamics
// ------------------------------------------------------------------------------------------------

IntegratorMetaDynamics::IntegratorMetaDynamics(std::shared_ptr<SystemDefinition> sysdef, double deltaT, double W, double T_shift,
                                               double T, unsigned int stride, bool add_bias, const std::string &filename,
                                               bool overwrite, const Enum mode)
    : m_sysdef(sysdef), m_pdata(sysdef->getParticleData()), m_exec_conf(sysdef->getExecConf()), m_deltaT(deltaT), m_W(W),
      m_T_shift(T_shift), m_stride(stride), m_is_initialized(false), m_filename(filename), m_overwrite(overwrite),
      m_is_appending(false), m_delimiter("\t"), m_use_grid(false), m_add_bias(add_bias), m_grid_period(0), m_cur_file(0),
      m_sigma_g(1.0), m_adaptive(false), m_temp(T), m_mode(mode), m_multiple_walkers(false), m_warned_single_walker(false), m_engine(nullptr),
      m_allow_fused(true), m_used_fused(false), m_fused_n_partials(0)
    {
    if (!(T_shift > 0.0) || !(W > 0.0)) throw std::runtime_error("IntegratorMetaDynamics: W and deltaT must be positive");   // asserts :58-59
    if (stride == 0) throw std::runtime_error("IntegratorMetaDynamics: stride must be positive");
    m_log_names = {"bias", "det_sigma", "weight"};                     // :61-63
    std::memset(&m_fused_set, 0, sizeof(m_fused_set));
    }

IntegratorMetaDynamics::~IntegratorMetaDynamics()
    {
    if (m_engine) (void)mtd_metad_destroy(m_engine);
    }

void IntegratorMetaDynamics::registerCollectiveVariable(std::shared_ptr<CollectiveVariable> cv, double sigma, double cv_min,
                                                        double cv_max, int num_points)
    {
    if (!cv) throw std::runtime_error("registerCollectiveVariable: null collective variable");
    if (!(sigma > 0.0)) throw std::runtime_error("registerCollectiveVariable: sigma must be positive");
    CollectiveVariableItem item;
    item.m_cv = cv;
    item.m_sigma = sigma;
    item.m_cv_min = cv_min;
    item.m_cv_max = cv_max;
    item.m_num_points = (unsigned int)num_points;
    m_variables.push_back(item);
    }

// :778-815
void IntegratorMetaDynamics::setGrid(bool use_grid)
    {
    if (m_is_initialized) throw std::runtime_error("Error setting up metadynamics parameters.");   // :785-789
    m_use_grid = use_grid;
    if (use_grid)
        for (const auto &it : m_variables)
            {
            if (it.m_cv_min >= it.m_cv_max) throw std::runtime_error("Error creating collective variable.");   // :800-805
            if (it.m_num_points < 2) throw std::runtime_error("Error creating collective variable.");         // :807-811
            }
    }

void IntegratorMetaDynamics::setMode(Enum mode)
    {
    m_mode = mode;
    if (m_engine) mtd_check(mtd_metad_set_mode(m_engine, mode == mode_well_tempered ? MTD_MODE_WELL_TEMPERED : MTD_MODE_STANDARD), "mtd_metad_set_mode");
    }

void IntegratorMetaDynamics::setStride(unsigned int stride)
    {
    if (stride == 0) throw std::runtime_error("integrate.mode_metadynamics: stride must be positive");
    m_stride = stride;
    if (m_engine) mtd_check(mtd_metad_set_stride(m_engine, stride), "mtd_metad_set_stride");
    }

void IntegratorMetaDynamics::setAddHills(bool add_bias)
    {
    m_add_bias = add_bias;
    if (m_engine) mtd_check(mtd_metad_set_add_hills(m_engine, add_bias ? 1 : 0), "mtd_metad_set_add_hills");
    }

// :1205-1294 — derivative products reduced on the device, n_cv x n_cv sqrt / inverse on the host
void IntegratorMetaDynamics::computeSigma()
    {
    ProfRange prof_range("Derivatives");
    const unsigned int ncv = (unsigned int)m_variables.size();
    if (m_sigma_scratch.bytes() == 0) m_sigma_scratch.resize(sizeof(double) * mtd_sigma_scratch_doubles());
    std::vector<const void *> force(ncv, nullptr);
    for (unsigned int i = 0; i < ncv; ++i)
        if (m_variables[i].m_cv->canComputeDerivatives()) force[i] = m_variables[i].m_cv->getForceArray().data();
    std::vector<double> sigmasq(ncv * ncv, 0.0);
    mtd_check(mtd_sigma_products(ncv, force.data(), m_pdata->getN(), m_pdata->getDtype(), m_sigma_g,
                                 (double *)m_sigma_scratch.data(), sigmasq.data(), m_exec_conf->getStream()),
              "mtd_sigma_products");
    const bool is_root = m_exec_conf->getRank() == 0;
    for (unsigned int i = 0; i < ncv; ++i)
        if (!force[i] && is_root) sigmasq[i * ncv + i] = m_variables[i].m_sigma * m_variables[i].m_sigma;   // :1249 (root only: summed below)
    if (m_exec_conf->getMailbox())
        {
        // the products of this rank's particles, summed over the ranks (:1259-1268)
        m_sigma_exchange.resize(sizeof(double) * ncv * ncv);
        m_exec_conf->sync();
        m_sigma_exchange.upload(sigmasq.data(), sizeof(double) * ncv * ncv);
        m_exec_conf->allreduceSmall((double *)m_sigma_exchange.data(), ncv * ncv, m_exec_conf->getStream());
        m_exec_conf->sync();
        m_sigma_exchange.download(sigmasq.data(), sizeof(double) * ncv * ncv);
        }
    m_sigma_inv.assign(ncv * ncv, 0.0);
    mtd_check(mtd_sigma_inverse(ncv, sigmasq.data(), m_sigma_inv.data()), "mtd_sigma_inverse");
    mtd_check(mtd_metad_set_sigma_inv(m_engine, m_sigma_inv.data()), "mtd_metad_set_sigma_inv");
    }

void IntegratorMetaDynamics::resetHistogram()
    {
    if (m_engine) mtd_check(mtd_metad_reset_histogram(m_engine, m_exec_conf->getStream()), "mtd_metad_reset_histogram");
    }

// :74-96
void IntegratorMetaDynamics::openOutputFile()
    {
    struct stat buffer;
    bool file_exists = stat(m_filename.c_str(), &buffer) == 0;
    if (file_exists && !m_overwrite)
        {
        m_file.open(m_filename.c_str(), std::ios_base::in | std::ios_base::out | std::ios_base::ate);
        m_is_appending = true;
        }
    else
        {
        m_file.open(m_filename.c_str(), std::ios_base::out);
        m_is_appending = false;
        }
    if (!m_file.good()) throw std::runtime_error("Error initializing IntegratorMetadynamics");
    }

// :98-119
void IntegratorMetaDynamics::writeFileHeader()
    {
    m_file << "timestep" << m_delimiter << "W" << m_delimiter;
    for (size_t i = 0; i < m_variables.size(); ++i)
        {
        m_file << m_variables[i].m_cv->getName();
        for (size_t j = 0; j < m_variables.size(); ++j)
            m_file << m_delimiter << "sigma_" << m_variables[i].m_cv->getName() << "_" << i << "_" << j;
        m_file << m_delimiter;
        }
    m_file << std::endl;
    }

// :590-661 on the device
void IntegratorMetaDynamics::setupGrid()
    {
    std::vector<double> sigma, lo, hi;
    std::vector<unsigned int> n;
    for (const auto &it : m_variables)
        {
        sigma.push_back(it.m_sigma);
        lo.push_back(it.m_cv_min);
        hi.push_back(it.m_cv_max);
        n.push_back(it.m_num_points);
        }
    if (m_engine)
        {
        mtd_check(mtd_metad_destroy(m_engine), "mtd_metad_destroy");
        m_engine = nullptr;
        }
    int rc = mtd_metad_create(&m_engine, (unsigned int)m_variables.size(), sigma.data(), lo.data(), hi.data(), n.data(), m_W,
                              m_T_shift, m_temp, m_stride, m_mode == mode_well_tempered ? MTD_MODE_WELL_TEMPERED : MTD_MODE_STANDARD,
                              m_add_bias ? 1 : 0);
    if (rc == MTD_ERR_INVALID_ARGUMENT) throw std::runtime_error("Error creating collective variable.");
    mtd_check(rc, "mtd_metad_create");
    // domain decomposition (the reference: MPI_Allreduce of the CV sums, root computes the bias, MPI_Bcast, :346-351, :571-575):
    // here every rank keeps the replicated grid and the CV sums of the fused step travel through the xGMI mailbox
    // (the mailbox is attached to the engine per step, where the step's form is known: updateBiasPotential)
    }

// :121-217
void IntegratorMetaDynamics::prepRun(unsigned int timestep)
    {
    // the hills file belongs to the root rank of a domain-decomposed run (is_root block, IntegratorMetaDynamics.cc:124-146)
    if (!m_is_initialized && m_filename != "" && m_exec_conf->getRank() == 0)
        {
        openOutputFile();
        if (!m_is_appending) writeFileHeader();
        }
    if (!m_is_initialized && !m_variables.empty())
        {
        if (!m_use_grid)
            throw std::runtime_error("integrate.mode_metadynamics: only grid mode is available (integrate.py:266-267 always enables it)");
        setupGrid();
        if (m_restart_filename != "")
            {
            readGrid(m_restart_filename);                              // :189-197
            m_restart_filename = "";
            }
        }
    m_is_initialized = true;

    // initial update of the potential (:214) — deposits a hill when timestep % stride == 0 (Q17)
    updateBiasPotential(timestep);
    // IntegratorTwoStep::prepRun computes the net force at `timestep`
    computeNetForce(timestep);
    }

void IntegratorMetaDynamics::computeNetForce(unsigned int timestep)
    {
    // HOOMD's Integrator::computeNetForce calls compute(timestep) on every ForceCompute and sums them into the net
    // force; the summation itself is HOOMD core and not part of the plugin
    for (auto &f : m_forces) f->compute(timestep);
    }

// :219-312 (no integration methods in the stand-alone system: integrateStepOne/Two are HOOMD's)
void IntegratorMetaDynamics::update(unsigned int timestep)
    {
    if (!m_is_initialized) throw std::runtime_error("IntegratorMetaDynamics::update called before prepRun");
    bool net_force_first = (m_variables.size() == 1 && m_variables[0].m_cv->requiresNetForce());   // :259
    if (!net_force_first)
        for (const auto &it : m_variables)
            if (it.m_cv->requiresNetForce())
                throw std::runtime_error("Only one collective variable requiring the potential energy may be defined.\n");   // :264-270

    if (net_force_first) computeNetForce(timestep + 1);                // :273-282
    updateBiasPotential(timestep + 1);                                 // :285
    if (!net_force_first)
        computeNetForce(timestep + 1);                                 // :287-296
    else
        m_variables[0].m_cv->compute(timestep);                        // :300 — bias forces *after* everything else
    }

bool IntegratorMetaDynamics::fusedLamellarPossible() const
    {
    if (m_adaptive) return false;                                      // the deposit width changes between the two launches
    if (m_multiple_walkers) return false;                              // the walkers' increments are summed between the grid passes
    if (!m_allow_fused || m_variables.empty() || m_variables.size() > MTD_METAD_MAX_CV) return false;
    unsigned int n_modes = 0;
    for (const auto &it : m_variables)
        {
        auto lam = std::dynamic_pointer_cast<LamellarOrderParameterGPU>(it.m_cv);
        if (!lam || lam->hasUmbrella()) return false;
        n_modes += (unsigned int)lam->getLatticeVectors().size();
        }
    return n_modes <= MTD_MAX_MODES;
    }

// the two-launch step (fused.hip): launch A = CV partial sums (+ deferred grid pass), launch B = grid update + forces
void IntegratorMetaDynamics::fusedLamellarStep(unsigned int timestep)
    {
    const unsigned int n_cv = (unsigned int)m_variables.size();
    std::memset(&m_fused_set, 0, sizeof(m_fused_set));
    m_fused_set.n_cv = n_cv;
    m_fused_set.n_types = m_pdata->getNTypes();
    unsigned int k = 0;
    m_fused_force_ptrs.assign(n_cv, nullptr);
    for (unsigned int c = 0; c < n_cv; ++c)
        {
        auto lam = std::static_pointer_cast<LamellarOrderParameterGPU>(m_variables[c].m_cv);
        m_fused_set.first[c] = k;
        for (const auto &v : lam->getLatticeVectors())
            {
            m_fused_set.hkl[k][0] = v.x;
            m_fused_set.hkl[k][1] = v.y;
            m_fused_set.hkl[k][2] = v.z;
            ++k;
            }
        for (unsigned int t = 0; t < m_fused_set.n_types; ++t) m_fused_set.coeff[c][t] = lam->getMode()[t];
        m_fused_force_ptrs[c] = lam->getForceArray().data();
        m_fused_set.trig_mode = std::max(m_fused_set.trig_mode, lam->getTrigMode());     // accurate (2) wins over hardware (1) over default
        }
    m_fused_set.first[n_cv] = k;
    m_fused_set.n_modes = k;
    if (m_fused_partials.bytes() == 0) m_fused_partials.resize(sizeof(double) * mtd_lamellar_scratch_doubles(m_pdata->getN()));

    const mtd_box box = m_pdata->getGlobalBox().toMtd();
    hipStream_t s = m_exec_conf->getStream();
    // the whole step through one entry point: the two-launch form (CV pass + deferred grid pass, then chain + grid pass +
    // forces) or, when selected (mtd_fused_step_set_mode / MTD_FUSED_STEP=1), the one-launch persistent kernel
    mtd_check(mtd_fused_step(m_engine, &m_fused_set, m_pdata->getN(), m_pdata->positionsPtr(), m_fused_force_ptrs.data(),
                             m_pdata->getDtype(), m_pdata->getNGlobal(), &box, (double *)m_fused_partials.data(), timestep, s),
              "mtd_fused_step");
    // the step wrote every CV's force array for `timestep`: computeNetForce's cv->compute(timestep) is a no-op
    for (auto &it : m_variables) it.m_cv->markComputed(timestep);
    m_used_fused = true;
    }

// A mixed set (e.g. lamellar + mesh): its lamellar CVs (those without an umbrella) are served by the two fused launches of
// the pure lamellar step — launch A sums all of them in one pass over the positions and carries the deferred grid pass,
// and the grid-engine launch of the step (the fused force kernel, which mtd_metad_update_bias runs without particles)
// also carries their force blocks: no k_apply launch, no CV and force kernel per lamellar CV.  `slots`: their indices
// among the grid's variables.
std::vector<unsigned int> IntegratorMetaDynamics::mixedLamellarSlots() const
    {
    std::vector<unsigned int> slots;
    if (!m_allow_fused || m_adaptive || m_multiple_walkers || m_variables.size() > 3) return slots;     // the one-wave chain handles <= 3 variables
    unsigned int n_modes = 0;
    for (unsigned int i = 0; i < m_variables.size(); ++i)
        {
        auto lam = std::dynamic_pointer_cast<LamellarOrderParameterGPU>(m_variables[i].m_cv);
        if (lam && !lam->hasUmbrella())
            {
            slots.push_back(i);
            n_modes += (unsigned int)lam->getLatticeVectors().size();
            }
        }
    if (n_modes > MTD_MAX_MODES) slots.clear();
    return slots;
    }

void IntegratorMetaDynamics::buildMixedLamellarSet(const std::vector<unsigned int> &slots)
    {
    std::memset(&m_fused_set, 0, sizeof(m_fused_set));
    m_fused_set.n_cv = (unsigned int)slots.size();
    m_fused_set.n_types = m_pdata->getNTypes();
    m_fused_force_ptrs.assign(slots.size(), nullptr);
    unsigned int k = 0;
    for (unsigned int c = 0; c < slots.size(); ++c)
        {
        auto lam = std::static_pointer_cast<LamellarOrderParameterGPU>(m_variables[slots[c]].m_cv);
        m_fused_set.first[c] = k;
        for (const auto &v : lam->getLatticeVectors())
            {
            m_fused_set.hkl[k][0] = v.x;
            m_fused_set.hkl[k][1] = v.y;
            m_fused_set.hkl[k][2] = v.z;
            ++k;
            }
        for (unsigned int t = 0; t < m_fused_set.n_types; ++t) m_fused_set.coeff[c][t] = lam->getMode()[t];
        m_fused_force_ptrs[c] = lam->getForceArray().data();
        m_fused_set.trig_mode = std::max(m_fused_set.trig_mode, lam->getTrigMode());
        }
    m_fused_set.first[slots.size()] = k;
    m_fused_set.n_modes = k;
    if (m_fused_partials.bytes() == 0) m_fused_partials.resize(sizeof(double) * mtd_lamellar_scratch_doubles(m_pdata->getN()));
    }

void IntegratorMetaDynamics::setMixedLamellarSources(const std::vector<unsigned int> &slots, unsigned int n_partials)
    {
    for (unsigned int c = 0; c < slots.size(); ++c)
        mtd_check(mtd_metad_set_cv_source(m_engine, slots[c], (const double *)m_fused_partials.data(), n_partials,
                                          (unsigned int)slots.size(), c, 1.0 / (double)m_pdata->getNGlobal(), 0.0),
                  "mtd_metad_set_cv_source");
    }

void IntegratorMetaDynamics::mixedLamellarCvPass(const std::vector<unsigned int> &slots, hipStream_t stream)
    {
    buildMixedLamellarSet(slots);
    const mtd_box box = m_pdata->getGlobalBox().toMtd();
    unsigned int n_partials = 0;
    mtd_check(mtd_fused_cv_pass(m_engine, &m_fused_set, m_pdata->getN(), m_pdata->positionsPtr(), m_pdata->getDtype(), &box,
                                (double *)m_fused_partials.data(), &n_partials, stream),
              "mtd_fused_cv_pass");
    setMixedLamellarSources(slots, n_partials);
    }

void IntegratorMetaDynamics::mixedLamellarForcePass(const std::vector<unsigned int> &slots, unsigned int timestep, hipStream_t stream,
                                                    const std::shared_ptr<OrderParameterMeshGPU> &mesh, unsigned int mesh_slot)
    {
    // one mesh variable beside the lamellar ones: the engine's launch inside the mesh's force pass (one launch instead of two)
    if (mesh && stream == m_exec_conf->getStream()
        && mesh->forcesWithBiasUpdate(timestep, m_engine, mesh_slot, &m_fused_set, slots.data(), m_fused_force_ptrs.data()))
        {
        for (unsigned int i : slots) m_variables[i].m_cv->markComputed(timestep);
        return;
        }
    const mtd_box box = m_pdata->getGlobalBox().toMtd();
    mtd_check(mtd_fused_force_pass_slots(m_engine, &m_fused_set, slots.data(), m_pdata->getN(), m_pdata->positionsPtr(),
                                         m_fused_force_ptrs.data(), m_pdata->getDtype(), m_pdata->getNGlobal(), &box, timestep,
                                         stream),
              "mtd_fused_force_pass_slots");
    for (unsigned int i : slots) m_variables[i].m_cv->markComputed(timestep);
    }

// :314-588, grid branch
void IntegratorMetaDynamics::updateBiasPotential(unsigned int timestep)
    {
    ProfRange prof_range("Metadynamics");
    if (m_variables.empty()) return;                                   // :317-318
    hipStream_t s = m_exec_conf->getStream();

    if (m_adaptive && (timestep % m_stride == 0))                      // :333-341
        {
        // compute derivatives of collective variables, then the instantaneous estimate of the standard deviation matrix
        for (auto &it : m_variables) it.m_cv->computeDerivatives(timestep);
        computeSigma();
        }

    // Domain decomposition (the reference: MPI_Allreduce inside every CV's computeCV, the root rank computes the bias and broadcasts
    // it, :346-351, :571-575): every rank keeps the replicated grid.  A pure lamellar set takes the fused step, whose two launches
    // carry the sums through the mailbox themselves; any other set goes CV by CV — each variable reduces its own sums over the
    // ranks inside enqueueCurrentValue (mailbox; the mesh: allreduceLarge or its slab decomposition) and the grid engine then
    // runs on identical values everywhere, no broadcast.
    mtd_comm *const mailbox = m_exec_conf->getMailbox();
    if (fusedLamellarPossible() && (!mailbox || m_variables.size() <= 3))          // (the mailbox form of the step: at most three sums)
        {
        if (mailbox) mtd_check(mtd_metad_set_comm(m_engine, mailbox), "mtd_metad_set_comm");
        fusedLamellarStep(timestep);
        }
    else
        {
        if (mailbox) mtd_check(mtd_metad_set_comm(m_engine, nullptr), "mtd_metad_set_comm");
        m_used_fused = false;
        // collect values of collective variables (:321-327) — they stay on the device
        // (the shared launches of a mixed set sum this rank's particles only: off in a domain-decomposed run)
        const std::vector<unsigned int> lam_slots = mailbox ? std::vector<unsigned int>() : mixedLamellarSlots();
        // The lamellar CVs' two launches depend on little of what the other CVs do: they go to a SIDE STREAM — launch A (one
        // pass over the positions + the deferred grid pass) runs beside the other CVs' kernels (the mesh assignment), the
        // grid-engine launch with the lamellar force blocks waits for the event that says "all CV values are there" and runs
        // beside what the main stream still has to do for the forces (the mesh's two inverse transforms); the main stream
        // takes up again after it.  Three events order the streams.  Opt-in (MTD_SIDE_STREAM=1): measured slower, mini_hoomd.h.
        hipStream_t side = lam_slots.empty() ? nullptr : m_exec_conf->getSideStream();
        hipStream_t ls = side ? side : s;
        std::shared_ptr<OrderParameterMeshGPU> hooked;
        if (side)
            {
            hip_check(hipEventRecord(m_exec_conf->getEvent(0), s), "hipEventRecord");              // positions / last step's work
            hip_check(hipStreamWaitEvent(side, m_exec_conf->getEvent(0), 0), "hipStreamWaitEvent");
            // exactly one other variable and it is a mesh: its CV sums are complete after the fused z pass, two passes early
            if (m_variables.size() == lam_slots.size() + 1)
                for (unsigned int i = 0; i < m_variables.size(); ++i)
                    if (std::find(lam_slots.begin(), lam_slots.end(), i) == lam_slots.end())
                        hooked = std::dynamic_pointer_cast<OrderParameterMeshGPU>(m_variables[i].m_cv);
            if (hooked) hooked->setCvEvent(m_exec_conf->getEvent(1));
            }
        // One mesh CV beside the lamellar ones (BASELINE.json's config 3): the lamellar sums and the deferred grid pass RIDE in the
        // mesh's assignment launches (mtd_mesh_set_lamellar_rider) instead of taking a launch and a second read of the position array
        // of their own (launch A of the fused step, 7.9 us at config 3).  In the bin pipeline — every assignment of a mesh but its
        // first — the sums are formed while the binning blocks wait for their atomics and the grid pass travels as extra blocks of
        // the scatter launch: 140 -> 133 us per step.  (The first form, the sums inside the counting kernel's per-particle chain,
        // measured SLOWER than launch A, 161.5 against 159.3 us, profiles/r3/mesh_rider_ab.log; it still carries the first step and
        // MTD_MESH_RIDER=count.)  MTD_MESH_RIDER=0 keeps launch A.
        std::shared_ptr<OrderParameterMeshGPU> carrier;
        static const bool use_rider = [] { const char *e = std::getenv("MTD_MESH_RIDER"); return !(e && (e[0] == '0' || std::strcmp(e, "off") == 0)); }();
        if (!lam_slots.empty() && !side && use_rider && lam_slots.size() <= 3)
            {
            unsigned int n_mesh = 0;
            for (unsigned int i = 0; i < m_variables.size(); ++i)
                if (std::find(lam_slots.begin(), lam_slots.end(), i) == lam_slots.end())
                    if (auto mesh = std::dynamic_pointer_cast<OrderParameterMeshGPU>(m_variables[i].m_cv))
                        {
                        carrier = mesh;
                        ++n_mesh;
                        }
            if (n_mesh != 1) carrier.reset();
            }
        bool ridden = false;
        if (carrier)
            {
            buildMixedLamellarSet(lam_slots);
            unsigned int n_partials = 0;
            ridden = carrier->armLamellarRider(timestep, m_engine, &m_fused_set, (double *)m_fused_partials.data(), &n_partials);
            if (ridden) setMixedLamellarSources(lam_slots, n_partials);
            }
        if (!lam_slots.empty() && !ridden) mixedLamellarCvPass(lam_slots, ls);
        // The grid-engine launch of such a step (chain + grid pass + lamellar forces, 11 us, HBM-bound) needs the CV sums, which
        // are complete after the mesh's z pass; the inverse x/y transform that follows (17 us, one LDS-filling block per CU,
        // latency-bound) needs nothing of the engine: the engine's launch can go to a second stream BESIDE the inverse transform.
        // Two events: "z pass done" (mtd_mesh_set_cv_event) lets the second stream start; "engine done" holds the main stream
        // before the first kernel that reads the bias factors.
        // MEASURED SLOWER, opt-in (MTD_MESH_OVERLAP=1): config 3 takes 141.1 / 142.9 / 142.8 us per step with it against 126.7 /
        // 126.3 / 126.5 on one stream (alternating processes on one box, profiles/r4/mesh_ab.log) — the two cross-queue
        // dependencies cost more than twice what the overlap of an 11 us launch can return, as round 3's wider form did (mini_hoomd.h).
        static const bool overlap_on = [] { const char *e = std::getenv("MTD_MESH_OVERLAP"); return e && e[0] == '1'; }();
        hipStream_t beside = nullptr;
        if (ridden && !side && overlap_on)
            {
            beside = m_exec_conf->getSideStream(true);
            carrier->setCvEvent(m_exec_conf->getEvent(1));
            }
        // a variable that is alone on the grid may run the engine's update fused with the tail of its own value (cv.steinhardt)
        const bool self_updated = m_variables.size() == 1 && !m_multiple_walkers && m_variables[0].m_cv->enqueueValueAndBias(timestep, m_engine);
        for (unsigned int i = 0; i < m_variables.size() && !self_updated; ++i)
            if (std::find(lam_slots.begin(), lam_slots.end(), i) == lam_slots.end())
                m_variables[i].m_cv->enqueueCurrentValue(timestep, m_engine, i);
        if (ridden && carrier->clearRider())
            {
            mixedLamellarCvPass(lam_slots, ls);                        // (nothing consumed the riders: the sums are formed by launch A after all)
            if (beside) hip_check(hipEventRecord(m_exec_conf->getEvent(1), s), "hipEventRecord");      // (no z pass ran either)
            }
        if (beside)
            {
            carrier->setCvEvent(nullptr);
            hip_check(hipStreamWaitEvent(beside, m_exec_conf->getEvent(1), 0), "hipStreamWaitEvent");
            ls = beside;
            }
        if (side)
            {
            if (hooked)
                hooked->setCvEvent(nullptr);
            else
                hip_check(hipEventRecord(m_exec_conf->getEvent(1), s), "hipEventRecord");          // every other CV's value is enqueued
            hip_check(hipStreamWaitEvent(side, m_exec_conf->getEvent(1), 0), "hipStreamWaitEvent");
            }
        if (!lam_slots.empty())
            {
            unsigned int carrier_slot = 0;
            for (unsigned int i = 0; i < m_variables.size(); ++i)
                if (m_variables[i].m_cv == carrier) carrier_slot = i;
            mixedLamellarForcePass(lam_slots, timestep, ls, carrier, carrier_slot);
            if (side || beside)
                {
                hip_check(hipEventRecord(m_exec_conf->getEvent(2), ls), "hipEventRecord");         // bias factors, grid arrays, lamellar forces
                hip_check(hipStreamWaitEvent(s, m_exec_conf->getEvent(2), 0), "hipStreamWaitEvent");
                }
            }
        else if (m_multiple_walkers)
            {
            // sum up the walkers' increments between the two grid passes (:393-409)
            // without a communicator this process is the only walker (a run with one partition: test/test_2d.py:29 sets the
            // flag in a serial run) — said once, so that nobody runs independent walkers believing the bias is shared
            if (!m_exec_conf->getWalkerCommunicator() && !m_warned_single_walker)
                {
                std::cerr << "integrate.mode_metadynamics: multiple_walkers is set but no walker communicator "
                             "(ExecutionConfiguration::setWalkerCommunicator): running as a single walker." << std::endl;
                m_warned_single_walker = true;
                }
            mtd_check(mtd_metad_update_bias_walkers(m_engine, m_exec_conf->getWalkerCommunicator(), timestep, s), "mtd_metad_update_bias_walkers");
            }
        else if (!self_updated)
            {
            // one mesh variable on the grid (alone or beside variables of other kinds): the engine's launch inside its force pass
            std::shared_ptr<OrderParameterMeshGPU> only_mesh;
            unsigned int mesh_slot = 0, n_mesh = 0;
            for (unsigned int i = 0; i < m_variables.size(); ++i)
                if (auto mesh = std::dynamic_pointer_cast<OrderParameterMeshGPU>(m_variables[i].m_cv))
                    {
                    only_mesh = mesh;
                    mesh_slot = i;
                    ++n_mesh;
                    }
            // (setFusedPath(false): every CV its own kernels and the engine's own launch, as the reference's classes do it)
            if (!(m_allow_fused && n_mesh == 1 && only_mesh->forcesWithBiasUpdate(timestep, m_engine, mesh_slot, nullptr, nullptr, nullptr)))
                mtd_check(mtd_metad_update_bias(m_engine, timestep, s), "mtd_metad_update_bias");
            }
        // update current bias potential derivative for every collective variable (:578-584)
        const double *d_bias = mtd_metad_bias_device(m_engine);
        for (unsigned int i = 0; i < m_variables.size(); ++i) m_variables[i].m_cv->setBiasFactorDevice(d_bias + i);
        }

    // write hills information (:523-550) — needs host values, only when a hills file was requested
    if (m_is_initialized && (timestep % m_stride == 0) && m_add_bias && m_file.is_open())
        {
        std::vector<double> cv(m_variables.size());
        double V = 0.0;
        mtd_check(mtd_metad_get_state(m_engine, cv.data(), nullptr, &V, nullptr, nullptr, nullptr, s), "mtd_metad_get_state");
        const double W = m_W * std::exp(-V / m_T_shift);               // :528 (written even in standard mode, Q16)
        // row i of h_sigma_inv (:536-541): diag(1 / sigma) (:177) until computeSigma overwrites it in adaptive mode (:1271-1290)
        std::vector<double> sinv(cv.size() * cv.size(), 0.0);
        for (size_t i = 0; i < cv.size(); ++i)
            for (size_t j = 0; j < cv.size(); ++j)
                sinv[i * cv.size() + j] = m_sigma_inv.size() == cv.size() * cv.size() ? m_sigma_inv[i * cv.size() + j]
                                                                                      : (i == j ? 1.0 / m_variables[i].m_sigma : 0.0);
        format_hills_line(m_file, timestep, W, cv, sinv, m_delimiter);  // grid_file.h
        }

    // dump grid information if required using alternating scheme (:555-565)
    if (m_grid_period && (timestep % m_grid_period == 0))
        {
        if (m_grid_fname2 != "")
            {
            writeGrid(m_cur_file ? m_grid_fname2 : m_grid_fname1, timestep);
            m_cur_file = m_cur_file ? 0 : 1;
            }
        else
            writeGrid(m_grid_fname1, timestep);
        }
    }

double IntegratorMetaDynamics::getLogValue(const std::string &quantity, unsigned int)
    {
    if (!m_engine) throw std::runtime_error("Error getting log value");
    double V = 0.0, w = 1.0;
    if (quantity == m_log_names[0] || quantity == m_log_names[2])
        mtd_check(mtd_metad_get_state(m_engine, nullptr, nullptr, &V, &w, nullptr, nullptr, m_exec_conf->getStream()), "mtd_metad_get_state");
    if (quantity == m_log_names[0]) return V;
    if (quantity == m_log_names[1]) return mtd_metad_sigma_determinant(m_engine);
    if (quantity == m_log_names[2]) return w;
    throw std::runtime_error("Error getting log value");               // .h:184-188
    }

std::vector<double> IntegratorMetaDynamics::getCurrentValues()
    {
    std::vector<double> cv(m_variables.size());
    if (m_engine) mtd_check(mtd_metad_get_state(m_engine, cv.data(), nullptr, nullptr, nullptr, nullptr, nullptr, m_exec_conf->getStream()), "mtd_metad_get_state");
    return cv;
    }

std::vector<double> IntegratorMetaDynamics::getBiasFactors()
    {
    std::vector<double> b(m_variables.size());
    if (m_engine) mtd_check(mtd_metad_get_state(m_engine, nullptr, b.data(), nullptr, nullptr, nullptr, nullptr, m_exec_conf->getStream()), "mtd_metad_get_state");
    return b;
    }

unsigned int IntegratorMetaDynamics::getNumGaussians()
    {
    unsigned int n = 0;
    if (m_engine) mtd_check(mtd_metad_get_state(m_engine, nullptr, nullptr, nullptr, nullptr, &n, nullptr, m_exec_conf->getStream()), "mtd_metad_get_state");
    return n;
    }

// :817-829
void IntegratorMetaDynamics::dumpGrid(const std::string &filename1, const std::string &filename2, unsigned int period)
    {
    if (period == 0)
        {
        writeGrid(filename1, 0);
        return;
        }
    m_grid_period = period;
    m_grid_fname1 = filename1;
    m_grid_fname2 = filename2;
    }

// :831-926 — the on-disk format users post-process
void IntegratorMetaDynamics::writeGrid(const std::string &filename, unsigned int timestep)
    {
    if (m_exec_conf->getRank() != 0) return;                          // only on the root processor (:835-839)
    if (!m_use_grid || !m_engine) throw std::runtime_error("Error dumping grid.");   // :841-845
    const unsigned int len = mtd_metad_num_elements(m_engine);
    hipStream_t s = m_exec_conf->getStream();
    std::vector<double> grid(len), sigma_grid(len), rew(len), weight(len);
    std::vector<unsigned int> hist(len), hist_gauss(len);
    mtd_check(mtd_metad_get_array(m_engine, 0, grid.data(), s), "get grid");
    mtd_check(mtd_metad_get_array(m_engine, 4, sigma_grid.data(), s), "get sigma_grid");
    mtd_check(mtd_metad_get_array(m_engine, 2, rew.data(), s), "get reweighted");
    mtd_check(mtd_metad_get_array(m_engine, 3, weight.data(), s), "get weight");
    mtd_check(mtd_metad_get_array(m_engine, 6, hist.data(), s), "get hist");
    mtd_check(mtd_metad_get_array(m_engine, 8, hist_gauss.data(), s), "get hist_gauss");
    unsigned int num_gaussians = 0;
    mtd_check(mtd_metad_get_state(m_engine, nullptr, nullptr, nullptr, nullptr, &num_gaussians, nullptr, s), "get state");

    std::ofstream file;
    file.open((filename + "_" + std::to_string(timestep)).c_str(), std::ios_base::out);
    GridFileData d;
    d.num_gaussians = num_gaussians;
    d.grid.swap(grid);
    d.sigma_grid.swap(sigma_grid);
    d.rew.swap(rew);
    d.weight.swap(weight);
    d.hist.swap(hist);
    d.hist_gauss.swap(hist_gauss);
    std::vector<std::string> names;
    std::vector<double> lo, hi;
    std::vector<unsigned int> pts;
    for (const auto &v : m_variables)
        {
        names.push_back(v.m_cv->getName());
        lo.push_back(v.m_cv_min);
        hi.push_back(v.m_cv_max);
        pts.push_back(v.m_num_points);
        }
    format_grid_file(file, names, lo, hi, pts, m_delimiter, d);        // grid_file.h: the text format, host only
    file.close();
    }

// :928-1000
void IntegratorMetaDynamics::readGrid(const std::string &filename)
    {
    if (!m_use_grid || !m_engine) throw std::runtime_error("Error reading grid.");
    std::ifstream file(filename.c_str());
    const unsigned int len = mtd_metad_num_elements(m_engine);
    GridFileData d;
    parse_grid_file(file, m_variables.size(), len, d);                 // grid_file.h: the text format, host only
    std::vector<double> &grid = d.grid, &sigma_grid = d.sigma_grid, &rew = d.rew, &weight = d.weight;
    std::vector<unsigned int> &hist = d.hist, &hist_gauss = d.hist_gauss;
    const unsigned int num_gaussians = d.num_gaussians;
    hipStream_t s = m_exec_conf->getStream();
    mtd_check(mtd_metad_set_array(m_engine, 0, grid.data(), s), "set grid");
    mtd_check(mtd_metad_set_array(m_engine, 4, sigma_grid.data(), s), "set sigma_grid");
    mtd_check(mtd_metad_set_array(m_engine, 2, rew.data(), s), "set reweighted");
    mtd_check(mtd_metad_set_array(m_engine, 3, weight.data(), s), "set weight");
    mtd_check(mtd_metad_set_array(m_engine, 6, hist.data(), s), "set hist");
    mtd_check(mtd_metad_set_array(m_engine, 8, hist_gauss.data(), s), "set hist_gauss");
    mtd_check(mtd_metad_set_num_gaussians(m_engine, num_gaussians, s), "set num_gaussians");
    }

// ------------------------------------------------------------------------------------------------
// System
// ------------------------------------------------------------------------------------------------

unsigned int IntegratorMetaDynamics::graphPeriod() const
    {
    if (!m_is_initialized || !m_use_grid || !m_engine || m_variables.empty()) return 0;
    if (m_adaptive || m_multiple_walkers || m_file.is_open() || m_grid_period) return 0;
    if (m_exec_conf->getMailbox() || m_exec_conf->getSideStream()) return 0;
    if (fusedLamellarPossible()) return 0;                             // two launches per step: measured slower from a graph
    if (m_pdata->getPressureFlag()) return 0;                          // (the mesh CV's virial is read back on the host)
    for (const auto &it : m_variables)
        {
        if (it.m_cv->hasUmbrella()) return 0;                           // the umbrella adds to a HOST bias factor every step
        const bool known = std::dynamic_pointer_cast<LamellarOrderParameterGPU>(it.m_cv) || std::dynamic_pointer_cast<OrderParameterMeshGPU>(it.m_cv)
                           || std::dynamic_pointer_cast<SteinhardtQl>(it.m_cv);
        if (!known) return 0;
        if (auto st = std::dynamic_pointer_cast<SteinhardtQl>(it.m_cv))
            if (!st->graphSafe()) return 0;
        }
    for (const auto &f : m_forces)                                     // every compute of the step is one of the variables
        {
        bool mine = false;
        for (const auto &it : m_variables) mine = mine || it.m_cv.get() == f.get();
        if (!mine) return 0;
        }
    unsigned int period = m_stride;
    if (period % 2) period *= 2;
    return period <= 64 ? period : 0;
    }

System::~System()
    {
    if (m_graph_stream) (void)hipStreamDestroy(m_graph_stream);
    }

// `nsteps` >= 3 periods of update() calls, most of them replayed from a capture of one period.  The captured sequence has to be
// PERIODIC: the same launches with the same arguments every `period` steps, and host-side hand-offs (the engine's pending deferred
// pass, keyed by stream; the mesh's alternating cursor sets) in the same state at its end as at its start.  Hence: one period of
// plain steps on the capture stream first (settles everything that is keyed by the stream), then the capture — which executes
// NOTHING, while the host state of the classes advances by a period — and at least one replay, which brings the device to where
// the host already is.  MTD_GRAPH_VERIFY=1 captures a second period and compares the two graphs node by node (kernel, grid,
// block, dynamic LDS): a sequence that is not periodic is then run with plain launches.  Returns the steps done.
unsigned int System::runGraph(unsigned int nsteps, unsigned int period)
    {
    auto exec = m_sysdef->getExecConf();
    if (!m_graph_stream) hip_check(hipStreamCreateWithFlags(&m_graph_stream, hipStreamNonBlocking), "hipStreamCreateWithFlags");
    const hipStream_t user_stream = exec->getStream();
    struct Restore
        {
        std::shared_ptr<ExecutionConfiguration> e;
        hipStream_t s;
        ~Restore() { e->setStream(reinterpret_cast<uintptr_t>(s)); }
        } restore{exec, user_stream};
    // nothing orders the caller's stream against ours: what it has enqueued is waited for here, and this function returns only
    // when its own stream has drained
    hip_check(hipStreamSynchronize(user_stream), "hipStreamSynchronize");
    exec->setStream(reinterpret_cast<uintptr_t>(m_graph_stream));
    unsigned int done = 0;
    for (unsigned int i = 0; i < period; ++i, ++done) m_integrator->update(m_cur_tstep++);       // settle
    // one graph = several periods, ~24 steps: every graph launch costs ~13 us in front of its first kernel (measured: two-step graphs
    // ran config 5 at 105.6 us per step against 99.0 with plain launches, twenty-step graphs at 94.5 — profiles/r4/graph_ab.log)
    const unsigned int base_period = period;
    period *= std::max(1u, 24u / period);
    if (nsteps < done + 2 * period) period = base_period;

    auto capture = [&](hipGraph_t &graph)
        {
        hip_check(hipStreamBeginCapture(m_graph_stream, hipStreamCaptureModeRelaxed), "hipStreamBeginCapture");
        try
            {
            for (unsigned int i = 0; i < period; ++i) m_integrator->update(m_cur_tstep + i);
            }
        catch (...)
            {
            hipGraph_t broken = nullptr;
            (void)hipStreamEndCapture(m_graph_stream, &broken);
            if (broken) (void)hipGraphDestroy(broken);
            throw;
            }
        hip_check(hipStreamEndCapture(m_graph_stream, &graph), "hipStreamEndCapture");
        m_cur_tstep += period;
        done += period;
        };
    auto signature = [](hipGraph_t graph)
        {
        std::vector<std::array<size_t, 8>> sig;
        size_t n = 0;
        hip_check(hipGraphGetNodes(graph, nullptr, &n), "hipGraphGetNodes");
        std::vector<hipGraphNode_t> nodes(n);
        if (n) hip_check(hipGraphGetNodes(graph, nodes.data(), &n), "hipGraphGetNodes");
        for (hipGraphNode_t node : nodes)
            {
            hipGraphNodeType type;
            hip_check(hipGraphNodeGetType(node, &type), "hipGraphNodeGetType");
            std::array<size_t, 8> e{};
            e[0] = (size_t)type;
            if (type == hipGraphNodeTypeKernel)
                {
                hipKernelNodeParams p;
                hip_check(hipGraphKernelNodeGetParams(node, &p), "hipGraphKernelNodeGetParams");
                e = {(size_t)type, (size_t)p.func, p.gridDim.x, p.gridDim.y, p.gridDim.z, p.blockDim.x, p.sharedMemBytes, 0};
                }
            sig.push_back(e);
            }
        std::sort(sig.begin(), sig.end());                             // (the node list of a graph has no defined order)
        return sig;
        };

    hipGraph_t graph = nullptr;
    hipGraphExec_t gexec = nullptr;
    capture(graph);
    bool periodic = true;
    static const bool verify = [] { const char *e = std::getenv("MTD_GRAPH_VERIFY"); return e && e[0] == '1'; }();
    if (verify && nsteps >= done + 2 * period)
        {
        hipGraphExec_t first = nullptr;
        hip_check(hipGraphInstantiate(&first, graph, nullptr, nullptr, 0), "hipGraphInstantiate");
        hip_check(hipGraphLaunch(first, m_graph_stream), "hipGraphLaunch");                  // the device catches up with the host state
        hipGraph_t second = nullptr;
        capture(second);
        periodic = signature(graph) == signature(second);
        (void)hipGraphExecDestroy(first);
        (void)hipGraphDestroy(graph);
        graph = second;
        if (!periodic) std::cerr << "System::run: the captured steps are not periodic — plain launches" << std::endl;
        }
    hip_check(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0), "hipGraphInstantiate");
    hip_check(hipGraphLaunch(gexec, m_graph_stream), "hipGraphLaunch");                      // (the captured period itself)
    unsigned int replayed = period;
    if (periodic)
        while (nsteps - done >= period)
            {
            hip_check(hipGraphLaunch(gexec, m_graph_stream), "hipGraphLaunch");
            m_cur_tstep += period;
            done += period;
            replayed += period;
            }
    m_last_graph_steps = periodic ? replayed : 0;
    m_last_graph_period = base_period;
    // the handles may go while the launches are still in flight? No: the executable graph must outlive them
    hip_check(hipStreamSynchronize(m_graph_stream), "hipStreamSynchronize");
    (void)hipGraphExecDestroy(gexec);
    (void)hipGraphDestroy(graph);
    return done;
    }

void System::run(unsigned int nsteps)
    {
    if (!m_integrator) throw std::runtime_error("System::run: no integrator set");
    m_integrator->prepRun(m_cur_tstep);      // HOOMD calls prepRun at the start of every run() (Q17)
    m_last_graph_steps = 0;
    unsigned int i = 0;
    // MEASURED SLOWER, opt-in (setGraphMode(1) / MTD_GRAPH=1): replaying the steps from a HIP graph costs config 5 102.7 us per
    // step against 99.3 with plain launches and config 3 134.3 against 132.7 (System::run, 24-step graphs, alternating processes on
    // one box); the headline's two-launch step 18.86 against 18.72 (profiles/r4/graph_ab.log).  The launches of a step are long
    // enough for the host to stay ahead; what a graph removes — the host's per-launch work — was never on the critical path.
    int mode = m_graph_mode;
    if (mode < 0)
        {
        static const int env = [] { const char *e = std::getenv("MTD_GRAPH"); return (e && e[0] == '1') ? 1 : 0; }();
        mode = env;
        }
    if (mode > 0)
        {
        const unsigned int period = m_integrator->graphPeriod();
        if (period && nsteps >= 4u * period) i = runGraph(nsteps, period);
        }
    for (; i < nsteps; ++i)
        {
        m_integrator->update(m_cur_tstep);
        m_cur_tstep++;
        }
    }

} // namespace mtdhost
