// MtdHoomd.cc — see MtdHoomd.h (never compiled in this project: no HOOMD tree on either box).
#ifdef MTD_WITH_HOOMD
#include "MtdHoomd.h"

#include <hoomd/extern/pybind/include/pybind11/pybind11.h>
#include <hoomd/extern/pybind/include/pybind11/stl_bind.h>

#include <stdexcept>

namespace py = pybind11;

namespace mtdhoomd
{

static void check(int rc, const char *what)
    {
    if (rc != MTD_SUCCESS) throw std::runtime_error(std::string("metadynamics (") + what + "): " + mtd_status_string(rc));
    }

// ---------------------------------------------------------------------------------------------- lamellar
LamellarOrderParameterGPU::LamellarOrderParameterGPU(std::shared_ptr<SystemDefinition> sysdef, const std::vector<Scalar> &mode,
                                                     const std::vector<int3> lattice_vectors, const std::string &suffix)
    : CollectiveVariable(sysdef, "cv_lamellar" + suffix), m_n_partials(0)
    {
    if (mode.size() != m_pdata->getNTypes())                                      // LamellarOrderParameter.cc:14-18
        throw std::runtime_error("Error setting up cv.lamellar");
    memset(&m_set, 0, sizeof(m_set));
    m_set.n_cv = 1;
    m_set.n_types = (unsigned int)mode.size();
    m_set.n_modes = (unsigned int)lattice_vectors.size();
    m_set.first[1] = m_set.n_modes;
    for (unsigned int k = 0; k < m_set.n_modes; ++k)
        {
        m_set.hkl[k][0] = lattice_vectors[k].x; m_set.hkl[k][1] = lattice_vectors[k].y; m_set.hkl[k][2] = lattice_vectors[k].z;
        }
    for (unsigned int t = 0; t < m_set.n_types; ++t) m_set.coeff[0][t] = mode[t];
    GPUArray<double> partials(mtd_lamellar_scratch_doubles(m_pdata->getN()), m_exec_conf);
    m_partials.swap(partials);
    }

void LamellarOrderParameterGPU::enqueuePartials()
    {
    ArrayHandle<Scalar4> d_postype(m_pdata->getPositions(), access_location::device, access_mode::read);
    ArrayHandle<double> d_partials(m_partials, access_location::device, access_mode::overwrite);
    const mtd_box box = to_mtd_box(m_pdata->getGlobalBox());
    check(mtd_lamellar_cv_partials(&m_set, m_pdata->getN(), d_postype.data, mtd_dtype(), &box, d_partials.data, &m_n_partials, 0),
          "mtd_lamellar_cv_partials");
    }

void LamellarOrderParameterGPU::enqueueCurrentValue(unsigned int, mtd_metad *engine, unsigned int slot)
    {
    enqueuePartials();
    ArrayHandle<double> d_partials(m_partials, access_location::device, access_mode::read);
    // in a domain-decomposed run the per-rank sums are added by mtd_comm_allreduce_small before the engine reads them
    check(mtd_metad_set_cv_source(engine, slot, d_partials.data, m_n_partials, 1, 0, 1.0 / (double)m_pdata->getNGlobal(), 0.0),
          "mtd_metad_set_cv_source");
    }

Scalar LamellarOrderParameterGPU::getCurrentValue(unsigned int)
    {
    enqueuePartials();
    GPUArray<double> sum(1, m_exec_conf);
        {
        ArrayHandle<double> d_partials(m_partials, access_location::device, access_mode::read);
        ArrayHandle<double> d_sum(sum, access_location::device, access_mode::overwrite);
        check(mtd_reduce_partials(d_partials.data, m_n_partials, 1, 1, 1.0 / (double)m_pdata->getNGlobal(), 0.0, d_sum.data, 0), "mtd_reduce_partials");
        }
    ArrayHandle<double> h_sum(sum, access_location::host, access_mode::read);          // the synchronising read-back
    return (Scalar)h_sum.data[0];
    }

void LamellarOrderParameterGPU::computeBiasForces(unsigned int)
    {
    ArrayHandle<Scalar4> d_postype(m_pdata->getPositions(), access_location::device, access_mode::read);
    ArrayHandle<Scalar4> d_force(m_force, access_location::device, access_mode::overwrite);
    const mtd_box box = to_mtd_box(m_pdata->getGlobalBox());
    void *f[1] = { d_force.data };
    if (m_bias_device)
        check(mtd_lamellar_forces(&m_set, m_pdata->getN(), d_postype.data, f, mtd_dtype(), m_pdata->getNGlobal(), m_bias_device, &box, 0), "mtd_lamellar_forces");
    else
        {
        int flat[3 * MTD_MAX_MODES];
        for (unsigned int k = 0; k < m_set.n_modes; ++k) for (int d = 0; d < 3; ++d) flat[3 * k + d] = m_set.hkl[k][d];
        check(mtd_compute_sq_forces(m_pdata->getN(), d_postype.data, d_force.data, mtd_dtype(), m_set.n_modes, flat, m_set.coeff[0], m_set.n_types,
                                    m_pdata->getNGlobal(), (double)m_bias, &box, 0), "mtd_compute_sq_forces");
        }
    }

// ---------------------------------------------------------------------------------------------- integrator
IntegratorMetaDynamics::IntegratorMetaDynamics(std::shared_ptr<SystemDefinition> sysdef, Scalar deltaT, Scalar W, Scalar T_shift, Scalar T,
                                               unsigned int stride, bool add_bias, const std::string &, bool, const Enum mode)
    : IntegratorTwoStep(sysdef, deltaT), m_W(W), m_T_shift(T_shift), m_temp(T), m_stride(stride), m_add_bias(add_bias), m_use_grid(false),
      m_is_initialized(false), m_multiple_walkers(false), m_mode(mode), m_engine(nullptr), m_walkers(nullptr)
    {
    if (!(m_W > 0.0) || !(m_T_shift > 0.0)) throw std::runtime_error("Error initializing IntegratorMetaDynamics");   // asserts :58-59
    }

IntegratorMetaDynamics::~IntegratorMetaDynamics() { if (m_engine) mtd_metad_destroy(m_engine); }

void IntegratorMetaDynamics::registerCollectiveVariable(std::shared_ptr<CollectiveVariable> cv, Scalar sigma, Scalar cv_min, Scalar cv_max, int num_points)
    {
    Item it = { cv, sigma, cv_min, cv_max, (unsigned int)num_points };
    m_variables.push_back(it);
    }

void IntegratorMetaDynamics::setAddHills(bool add_bias) { m_add_bias = add_bias; if (m_engine) mtd_metad_set_add_hills(m_engine, add_bias); }
void IntegratorMetaDynamics::setMode(Enum mode) { m_mode = mode; if (m_engine) mtd_metad_set_mode(m_engine, mode == mode_well_tempered); }
void IntegratorMetaDynamics::setStride(unsigned int stride) { m_stride = stride; if (m_engine) mtd_metad_set_stride(m_engine, stride); }

void IntegratorMetaDynamics::setupGrid()
    {
    std::vector<double> sigma, lo, hi;
    std::vector<unsigned int> n;
    for (const Item &it : m_variables) { sigma.push_back(it.m_sigma); lo.push_back(it.m_cv_min); hi.push_back(it.m_cv_max); n.push_back(it.m_num_points); }
    check(mtd_metad_create(&m_engine, (unsigned int)m_variables.size(), sigma.data(), lo.data(), hi.data(), n.data(), m_W, m_T_shift, m_temp,
                           m_stride, m_mode == mode_well_tempered ? MTD_MODE_WELL_TEMPERED : MTD_MODE_STANDARD, m_add_bias), "mtd_metad_create");
    GPUArray<double> scratch(mtd_lamellar_scratch_doubles(m_pdata->getN()), m_exec_conf);
    m_scratch.swap(scratch);
    }

bool IntegratorMetaDynamics::allLamellar() const
    {
    if (m_variables.empty() || m_variables.size() > 3 || m_multiple_walkers) return false;
    for (const Item &it : m_variables) if (!std::dynamic_pointer_cast<LamellarOrderParameterGPU>(it.m_cv)) return false;
    return true;
    }

void IntegratorMetaDynamics::prepRun(unsigned int timestep)
    {
    if (!m_is_initialized && !m_variables.empty())
        {
        if (!m_use_grid) throw std::runtime_error("integrate.mode_metadynamics: only grid mode is available");
        setupGrid();
        }
    m_is_initialized = true;
    updateBiasPotential(timestep);                                               // :214
    IntegratorTwoStep::prepRun(timestep);
    }

void IntegratorMetaDynamics::updateBiasPotential(unsigned int timestep)
    {
    if (m_variables.empty()) return;
    if (allLamellar())
        {
        // the fused step: every lamellar CV in one set (CV c of the set is CV c of the grid), forces written into the CVs' own arrays
        mtd_lamellar_set set;
        memset(&set, 0, sizeof(set));
        std::vector<std::unique_ptr<ArrayHandle<Scalar4>>> handles;
        void *force[MTD_MAX_CV] = { nullptr };
        unsigned int k = 0;
        for (unsigned int c = 0; c < m_variables.size(); ++c)
            {
            auto lam = std::static_pointer_cast<LamellarOrderParameterGPU>(m_variables[c].m_cv);
            const mtd_lamellar_set &one = lam->getSet();
            set.first[c] = k;
            for (unsigned int q = 0; q < one.n_modes; ++q, ++k) for (int d = 0; d < 3; ++d) set.hkl[k][d] = one.hkl[q][d];
            for (unsigned int t = 0; t < one.n_types; ++t) set.coeff[c][t] = one.coeff[0][t];
            set.n_types = one.n_types;
            handles.emplace_back(new ArrayHandle<Scalar4>(lam->getForceArray(), access_location::device, access_mode::overwrite));
            force[c] = handles.back()->data;
            }
        set.n_cv = (unsigned int)m_variables.size();
        set.first[set.n_cv] = k;
        set.n_modes = k;
        ArrayHandle<Scalar4> d_postype(m_pdata->getPositions(), access_location::device, access_mode::read);
        ArrayHandle<double> d_scratch(m_scratch, access_location::device, access_mode::overwrite);
        const mtd_box box = to_mtd_box(m_pdata->getGlobalBox());
        check(mtd_fused_step(m_engine, &set, m_pdata->getN(), d_postype.data, force, mtd_dtype(), m_pdata->getNGlobal(), &box, d_scratch.data, timestep, 0),
              "mtd_fused_step");
        return;
        }
    for (unsigned int i = 0; i < m_variables.size(); ++i) m_variables[i].m_cv->enqueueCurrentValue(timestep, m_engine, i);   // :321-327
    if (m_multiple_walkers)
        {
        if (!m_walkers) throw std::runtime_error("integrate.mode_metadynamics: multiple_walkers needs a communicator between the walkers");
        check(mtd_metad_update_bias_walkers(m_engine, m_walkers, timestep, 0), "mtd_metad_update_bias_walkers");       // :393-409
        }
    else
        check(mtd_metad_update_bias(m_engine, timestep, 0), "mtd_metad_update_bias");
    const double *d_bias = mtd_metad_bias_device(m_engine);
    for (unsigned int i = 0; i < m_variables.size(); ++i) m_variables[i].m_cv->setBiasFactorDevice(d_bias + i);          // :578-584
    }

// :219-312 — HOOMD's two-step integration around the bias update
void IntegratorMetaDynamics::update(unsigned int timestep)
    {
    if (!m_is_initialized) throw std::runtime_error("IntegratorMetaDynamics::update called before prepRun");
    for (auto &method : m_methods) method->integrateStepOne(timestep);            // :233-235
    updateBiasPotential(timestep + 1);                                            // :285
    computeNetForceGPU(timestep + 1);                                             // :287-296 (CVs are ForceComputes of the system)
    for (auto &method : m_methods) method->integrateStepTwo(timestep);            // :305-307
    }

void export_all(py::module &m)
    {
    py::bind_vector<std::vector<int3>>(m, "std_vector_int3");
    py::class_<CollectiveVariable, std::shared_ptr<CollectiveVariable>> cv(m, "CollectiveVariable", py::base<ForceCompute>());
    cv.def(py::init<std::shared_ptr<SystemDefinition>, const std::string &>())
        .def("getCurrentValue", &CollectiveVariable::getCurrentValue).def("setUmbrella", &CollectiveVariable::setUmbrella)
        .def("setKappa", &CollectiveVariable::setKappa).def("setWidthFlat", &CollectiveVariable::setWidthFlat)
        .def("setMinimum", &CollectiveVariable::setMinimum).def("setScale", &CollectiveVariable::setScale)
        .def("requiresNetForce", &CollectiveVariable::requiresNetForce);
    py::enum_<CollectiveVariable::umbrella_Enum>(cv, "umbrella").value("no_umbrella", CollectiveVariable::no_umbrella)
        .value("linear", CollectiveVariable::linear).value("harmonic", CollectiveVariable::harmonic).value("wall", CollectiveVariable::wall)
        .value("gaussian", CollectiveVariable::gaussian).export_values();
    py::class_<LamellarOrderParameterGPU, std::shared_ptr<LamellarOrderParameterGPU>>(m, "LamellarOrderParameterGPU", py::base<CollectiveVariable>())
        .def(py::init<std::shared_ptr<SystemDefinition>, const std::vector<Scalar> &, const std::vector<int3>, const std::string &>());
    py::class_<IntegratorMetaDynamics, std::shared_ptr<IntegratorMetaDynamics>> integ(m, "IntegratorMetaDynamics", py::base<IntegratorTwoStep>());
    integ.def(py::init<std::shared_ptr<SystemDefinition>, Scalar, Scalar, Scalar, Scalar, unsigned int, bool, const std::string &, bool, IntegratorMetaDynamics::Enum>())
        .def("registerCollectiveVariable", &IntegratorMetaDynamics::registerCollectiveVariable)
        .def("removeAllVariables", &IntegratorMetaDynamics::removeAllVariables).def("isInitialized", &IntegratorMetaDynamics::isInitialized)
        .def("setGrid", &IntegratorMetaDynamics::setGrid).def("setAddHills", &IntegratorMetaDynamics::setAddHills)
        .def("setMode", &IntegratorMetaDynamics::setMode).def("setStride", &IntegratorMetaDynamics::setStride)
        .def("setMultipleWalkers", &IntegratorMetaDynamics::setMultipleWalkers);
    py::enum_<IntegratorMetaDynamics::Enum>(integ, "mode").value("standard", IntegratorMetaDynamics::mode_standard)
        .value("well_tempered", IntegratorMetaDynamics::mode_well_tempered).export_values();
    }

} // namespace mtdhoomd

PYBIND11_MODULE(_metadynamics, m) { mtdhoomd::export_all(m); }
#endif // MTD_WITH_HOOMD
