// module.cc — pybind11 module `_metadynamics` (mirror of the reference's module.cc:23-41 and the
// export_* functions: CollectiveVariable.cc:109-130, IntegratorMetaDynamics.cc:1315-1349,
// LamellarOrderParameterGPU.cc:134-141, WellTemperedEnsemble.cc:190-197, AspectRatio.cc:132-140,
// Density.cc:56-64), plus the stand-alone SystemDefinition the reference borrows from HOOMD-blue.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>
#include <pybind11/stl_bind.h>

#include "metadynamics_host.h"
#include "grid_file.h"

#include <fstream>
#include <sstream>

namespace py = pybind11;
using namespace mtdhost;

PYBIND11_MAKE_OPAQUE(std::vector<int3>);

namespace
{

// HOOMD Scalar4 postype from positions + integer types: the type id is bit-cast into w (fp32) or into the
// low word of w (fp64), as __scalar_as_int reads it
py::array pack_postype(py::array_t<double, py::array::c_style | py::array::forcecast> pos,
                       py::array_t<int, py::array::c_style | py::array::forcecast> types, int dtype)
    {
    if (pos.ndim() != 2 || pos.shape(1) != 3) throw std::runtime_error("positions must have shape (N, 3)");
    const ssize_t N = pos.shape(0);
    if (types.ndim() != 1 || types.shape(0) != N) throw std::runtime_error("types must have shape (N,)");
    auto p = pos.unchecked<2>();
    auto t = types.unchecked<1>();
    if (dtype == MTD_F32)
        {
        py::array_t<float> out({N, (ssize_t)4});
        auto o = out.mutable_unchecked<2>();
        for (ssize_t i = 0; i < N; ++i)
            {
            o(i, 0) = (float)p(i, 0);
            o(i, 1) = (float)p(i, 1);
            o(i, 2) = (float)p(i, 2);
            int ti = t(i);
            float w;
            std::memcpy(&w, &ti, 4);
            o(i, 3) = w;
            }
        return out;
        }
    py::array_t<double> out({N, (ssize_t)4});
    auto o = out.mutable_unchecked<2>();
    for (ssize_t i = 0; i < N; ++i)
        {
        o(i, 0) = p(i, 0);
        o(i, 1) = p(i, 1);
        o(i, 2) = p(i, 2);
        long long ti = (unsigned int)t(i);
        double w;
        std::memcpy(&w, &ti, 8);
        o(i, 3) = w;
        }
    return out;
    }

void upload_array(DeviceBuffer &buf, py::array a, size_t expected_bytes, const char *what)
    {
    py::buffer_info info = a.request();
    const size_t bytes = (size_t)info.size * (size_t)info.itemsize;
    if (bytes != expected_bytes) throw std::runtime_error(std::string(what) + ": array has the wrong size or dtype");
    if (!(a.flags() & py::array::c_style)) throw std::runtime_error(std::string(what) + ": array must be C-contiguous");
    buf.upload(info.ptr, bytes);
    }

py::array download_scalar_array(const DeviceBuffer &buf, int dtype, std::vector<ssize_t> shape)
    {
    if (dtype == MTD_F32)
        {
        py::array_t<float> out(shape);
        buf.download(out.mutable_data(), buf.bytes());
        return out;
        }
    py::array_t<double> out(shape);
    buf.download(out.mutable_data(), buf.bytes());
    return out;
    }

} // namespace

PYBIND11_MODULE(_metadynamics, m)
    {
    m.doc() = "MI355X-native metadynamics plugin: host classes over libmtd_hip.so";
    m.attr("MTD_F32") = (int)MTD_F32;
    m.attr("MTD_F64") = (int)MTD_F64;

    py::class_<int3>(m, "int3")
        .def(py::init<>())
        .def_property("x", [](const int3 &v) { return v.x; }, [](int3 &v, int a) { v.x = a; })
        .def_property("y", [](const int3 &v) { return v.y; }, [](int3 &v, int a) { v.y = a; })
        .def_property("z", [](const int3 &v) { return v.z; }, [](int3 &v, int a) { v.z = a; });
    m.def("make_int3", [](int x, int y, int z) { return make_int3(x, y, z); });
    py::bind_vector<std::vector<int3>>(m, "std_vector_int3");          // module.cc:25
    m.def("pack_postype", &pack_postype);
    // the text formats of the grid dump and the hills log as host-only functions (grid_file.h): reachable without a GPU
    m.def("parse_grid_file", [](const std::string &path, size_t n_cv, size_t len)
        {
        std::ifstream f(path.c_str());
        GridFileData d;
        parse_grid_file(f, n_cv, len, d);
        py::dict out;
        out["num_gaussians"] = d.num_gaussians;
        out["grid"] = py::array_t<double>(d.grid.size(), d.grid.data());
        out["sigma_grid"] = py::array_t<double>(d.sigma_grid.size(), d.sigma_grid.data());
        out["reweighted"] = py::array_t<double>(d.rew.size(), d.rew.data());
        out["weight"] = py::array_t<double>(d.weight.size(), d.weight.data());
        out["hist"] = py::array_t<unsigned int>(d.hist.size(), d.hist.data());
        out["hist_gauss"] = py::array_t<unsigned int>(d.hist_gauss.size(), d.hist_gauss.data());
        return out;
        });
    m.def("format_grid_file", [](const std::string &path, const std::vector<std::string> &names, const std::vector<double> &cv_min,
                                 const std::vector<double> &cv_max, const std::vector<unsigned int> &num_points, unsigned int num_gaussians,
                                 const std::vector<double> &grid, const std::vector<double> &sigma_grid, const std::vector<double> &rew,
                                 const std::vector<double> &weight, const std::vector<unsigned int> &hist,
                                 const std::vector<unsigned int> &hist_gauss)
        {
        GridFileData d;
        d.num_gaussians = num_gaussians;
        d.grid = grid; d.sigma_grid = sigma_grid; d.rew = rew; d.weight = weight; d.hist = hist; d.hist_gauss = hist_gauss;
        std::ofstream f(path.c_str());
        format_grid_file(f, names, cv_min, cv_max, num_points, "\t", d);
        });
    m.def("format_hills_line", [](unsigned int timestep, double W, const std::vector<double> &cv, const std::vector<double> &sigma_inv)
        {
        std::ostringstream o;
        format_hills_line(o, timestep, W, cv, sigma_inv, "\t");
        return o.str();
        });

    py::class_<BoxDim>(m, "BoxDim")
        .def(py::init<double, double, double, double, double, double>(), py::arg("Lx") = 1.0, py::arg("Ly") = 1.0, py::arg("Lz") = 1.0,
             py::arg("xy") = 0.0, py::arg("xz") = 0.0, py::arg("yz") = 0.0)
        .def("setLo", &BoxDim::setLo)
        .def("getL", &BoxDim::getL)
        .def("getLo", &BoxDim::getLo)
        .def("getVolume", &BoxDim::getVolume)
        .def("scale", &BoxDim::scale);

    py::class_<ExecutionConfiguration, std::shared_ptr<ExecutionConfiguration>>(m, "ExecutionConfiguration")
        .def(py::init<>())
        .def("isCUDAEnabled", &ExecutionConfiguration::isCUDAEnabled)
        .def("setMailbox", &ExecutionConfiguration::setMailbox)
        .def("setStream", &ExecutionConfiguration::setStream)
        .def("setCommunicator", &ExecutionConfiguration::setCommunicator)
        .def("largeExchangeName", &ExecutionConfiguration::largeExchangeName)
        .def("setAllgather", [](ExecutionConfiguration &e, py::object fn) {
            // fn(bytes) -> bytes of every rank, concatenated in rank order (the launcher's control plane: torch.distributed / MPI)
            if (fn.is_none())
                {
                e.setAllgather(nullptr);
                return;
                }
            e.setAllgather([fn](const void *mine, size_t bytes, void *all) {
                py::gil_scoped_acquire gil;                             // (System::run releases the GIL)
                py::bytes got = fn(py::bytes((const char *)mine, bytes));
                const std::string blob = got;
                if (blob.size() % bytes != 0 || blob.empty()) throw std::runtime_error("setAllgather: the callback returned a blob of the wrong size");
                std::memcpy(all, blob.data(), blob.size());
            });
        })
        .def("setWalkerCommunicator", &ExecutionConfiguration::setWalkerCommunicator)
        .def("getNRanks", &ExecutionConfiguration::getNRanks)
        .def("getRank", &ExecutionConfiguration::getRank)
        .def("sync", &ExecutionConfiguration::sync);

    py::class_<ParticleData, std::shared_ptr<ParticleData>>(m, "ParticleData")
        .def(py::init<unsigned int, int, const std::vector<std::string> &, const BoxDim &>())
        .def("getN", &ParticleData::getN)
        .def("getNGhosts", &ParticleData::getNGhosts)
        .def("setNGhosts", &ParticleData::setNGhosts)
        .def("getNGlobal", &ParticleData::getNGlobal)
        .def("setNGlobal", &ParticleData::setNGlobal)
        .def("getNTypes", &ParticleData::getNTypes)
        .def("getNameByType", &ParticleData::getNameByType)
        .def("getDtype", &ParticleData::getDtype)
        .def("getGlobalBox", &ParticleData::getGlobalBox)
        .def("setGlobalBox", &ParticleData::setGlobalBox)
        .def("setPositions", [](ParticleData &p, py::array a) { upload_array(p.getPositions(), a, p.scalar4Bytes() * ((size_t)p.getN() + p.getNGhosts()), "setPositions"); },
             "Scalar4[N + n_ghosts]: the local particles followed by the ghost particles")
        .def("getPositions", [](ParticleData &p) { return download_scalar_array(p.getPositions(), p.getDtype(), {(ssize_t)p.getN() + (ssize_t)p.getNGhosts(), 4}); })
        .def("borrowPositions", [](ParticleData &p, size_t ptr) { p.borrowPositions((void *)ptr); },
             "use caller-owned device memory (Scalar4[N], e.g. tensor.data_ptr()) for the positions")
        .def("setNetForce", [](ParticleData &p, py::array a) { upload_array(p.getNetForce(), a, p.scalar4Bytes() * p.getN(), "setNetForce"); })
        .def("getNetForce", [](ParticleData &p) { return download_scalar_array(p.getNetForce(), p.getDtype(), {(ssize_t)p.getN(), 4}); })
        .def("setNetTorque", [](ParticleData &p, py::array a) { upload_array(p.getNetTorqueArray(), a, p.scalar4Bytes() * p.getN(), "setNetTorque"); })
        .def("getNetTorque", [](ParticleData &p) { return download_scalar_array(p.getNetTorqueArray(), p.getDtype(), {(ssize_t)p.getN(), 4}); })
        .def("setNetVirial", [](ParticleData &p, py::array a) { upload_array(p.getNetVirial(), a, p.scalarBytes() * 6 * p.getNetVirialPitch(), "setNetVirial"); })
        .def("getNetVirial", [](ParticleData &p) { return download_scalar_array(p.getNetVirial(), p.getDtype(), {6, (ssize_t)p.getNetVirialPitch()}); })
        .def("getExternalEnergy", &ParticleData::getExternalEnergy)
        .def("setExternalEnergy", &ParticleData::setExternalEnergy)
        .def("getPressureFlag", &ParticleData::getPressureFlag)
        .def("setPressureFlag", &ParticleData::setPressureFlag)
        .def("getExternalVirial", &ParticleData::getExternalVirial)
        .def("setExternalVirial", &ParticleData::setExternalVirial);

    py::class_<SystemDefinition, std::shared_ptr<SystemDefinition>>(m, "SystemDefinition")
        .def(py::init<std::shared_ptr<ParticleData>, std::shared_ptr<ExecutionConfiguration>>())
        .def("getParticleData", &SystemDefinition::getParticleData)
        .def("getExecConf", &SystemDefinition::getExecConf);

    py::class_<ForceCompute, std::shared_ptr<ForceCompute>>(m, "ForceCompute")
        .def("compute", &ForceCompute::compute)
        .def("getExternalVirial", &ForceCompute::getExternalVirial)
        .def("getExternalEnergy", &ForceCompute::getExternalEnergy)
        .def("getVirialPitch", &ForceCompute::getVirialPitch)
        .def("getForces", [](ForceCompute &f) { return download_scalar_array(f.getForceArray(), f.dtype(), {(ssize_t)f.numParticles(), 4}); })
        .def("getTorques", [](ForceCompute &f) { return download_scalar_array(f.getTorqueArray(), f.dtype(), {(ssize_t)f.numParticles(), 4}); })
        .def("getVirial", [](ForceCompute &f) { return download_scalar_array(f.getVirialArray(), f.dtype(), {6, (ssize_t)f.getVirialPitch()}); })
        .def("getProvidedLogQuantities", &ForceCompute::getProvidedLogQuantities)
        .def("getLogValue", &ForceCompute::getLogValue);

    // CollectiveVariable.cc:109-130
    py::class_<CollectiveVariable, ForceCompute, std::shared_ptr<CollectiveVariable>> collective_variable(m, "CollectiveVariable");
    collective_variable.def("getCurrentValue", &CollectiveVariable::getCurrentValue)
        .def("setUmbrella", &CollectiveVariable::setUmbrella)
        .def("setKappa", &CollectiveVariable::setKappa)
        .def("setWidthFlat", &CollectiveVariable::setWidthFlat)
        .def("setMinimum", &CollectiveVariable::setMinimum)
        .def("setScale", &CollectiveVariable::setScale)
        .def("requiresNetForce", &CollectiveVariable::requiresNetForce)
        .def("setBiasFactor", &CollectiveVariable::setBiasFactor)
        .def("computeDerivatives", &CollectiveVariable::computeDerivatives)
        .def("canComputeDerivatives", &CollectiveVariable::canComputeDerivatives)
        .def("getName", &CollectiveVariable::getName)
        .def("getUmbrellaPotential", &CollectiveVariable::getUmbrellaPotential)
        .def("getForceArray", [](CollectiveVariable &cv) {
            return download_scalar_array(cv.getForceArray(), cv.dtype(), {(ssize_t)cv.numParticles(), 4});
        });
    py::enum_<CollectiveVariable::umbrella_Enum>(collective_variable, "umbrella")
        .value("no_umbrella", CollectiveVariable::no_umbrella)
        .value("linear", CollectiveVariable::linear)
        .value("harmonic", CollectiveVariable::harmonic)
        .value("wall", CollectiveVariable::wall)
        .value("gaussian", CollectiveVariable::gaussian)
        .export_values();

    // LamellarOrderParameterGPU.cc:134-141 (the reference's CPU class LamellarOrderParameter has no
    // counterpart here: this build has no CPU path, cv.lamellar always creates the GPU class)
    py::class_<LamellarOrderParameterGPU, CollectiveVariable, std::shared_ptr<LamellarOrderParameterGPU>>(m, "LamellarOrderParameterGPU")
        .def(py::init<std::shared_ptr<SystemDefinition>, const std::vector<double> &, const std::vector<int3> &, const std::string &>())
        .def("setTrigMode", &LamellarOrderParameterGPU::setTrigMode)
        .def("getTrigMode", &LamellarOrderParameterGPU::getTrigMode);

    // OrderParameterMesh.cc:1181-1193 / OrderParameterMeshGPU.cc:571-584
    py::class_<OrderParameterMeshGPU, CollectiveVariable, std::shared_ptr<OrderParameterMeshGPU>>(m, "OrderParameterMeshGPU")
        .def(py::init<std::shared_ptr<SystemDefinition>, unsigned int, unsigned int, unsigned int, std::vector<double>, std::vector<int3>>())
        .def("setTable", &OrderParameterMeshGPU::setTable)
        .def("setUseTable", &OrderParameterMeshGPU::setUseTable)
        .def("setBugCompatible", &OrderParameterMeshGPU::setBugCompatible)
        .def("setSlabDecomposition", &OrderParameterMeshGPU::setSlabDecomposition)
        .def("getSlabDecomposition", &OrderParameterMeshGPU::getSlabDecomposition);

    py::class_<NeighborList, std::shared_ptr<NeighborList>> nlist(m, "NeighborList");
    nlist.def(py::init<std::shared_ptr<SystemDefinition>>())
        .def("setStorageMode", &NeighborList::setStorageMode)
        .def("getStorageMode", &NeighborList::getStorageMode)
        .def("compute", &NeighborList::compute)
        .def("setLists", [](NeighborList &n, py::array_t<unsigned int, py::array::c_style | py::array::forcecast> head,
                            py::array_t<unsigned int, py::array::c_style | py::array::forcecast> nneigh,
                            py::array_t<unsigned int, py::array::c_style | py::array::forcecast> list) {
            if (head.size() != nneigh.size()) throw std::runtime_error("setLists: head_list and n_neigh differ in length");
            n.setLists(head.data(), nneigh.data(), (size_t)head.size(), list.data(), (size_t)list.size());
        });
    py::enum_<NeighborList::storageMode>(nlist, "storageMode").value("half", NeighborList::half).value("full", NeighborList::full).export_values();

    py::class_<PrescribedForceCompute, ForceCompute, std::shared_ptr<PrescribedForceCompute>>(m, "PrescribedForceCompute")
        .def(py::init<std::shared_ptr<SystemDefinition>>())
        .def("setExternalEnergy", &PrescribedForceCompute::setExternalEnergy)
        .def("setArrays", [](PrescribedForceCompute &f, py::array force, py::array torque, py::array virial) {
            auto chk = [](py::array a, size_t bytes, const char *what) {
                py::buffer_info info = a.request();
                if ((size_t)info.size * (size_t)info.itemsize != bytes || !(a.flags() & py::array::c_style))
                    throw std::runtime_error(std::string(what) + ": array has the wrong size, dtype or layout");
                return info.ptr;
            };
            const size_t sb = f.dtype() == MTD_F32 ? 4 : 8, N = f.numParticles();
            f.setArrays(chk(force, 4 * sb * N, "force"), chk(torque, 4 * sb * N, "torque"), chk(virial, 6 * sb * f.getVirialPitch(), "virial"));
        });

    // CollectiveWrapper.cc:182-187
    py::class_<CollectiveWrapper, CollectiveVariable, std::shared_ptr<CollectiveWrapper>>(m, "CollectiveWrapper")
        .def(py::init<std::shared_ptr<SystemDefinition>, std::shared_ptr<ForceCompute>, const std::string &>());

    // SteinhardtQl.cc:341-347
    py::class_<SteinhardtQl, CollectiveVariable, std::shared_ptr<SteinhardtQl>>(m, "SteinhardtQl")
        .def(py::init<std::shared_ptr<SystemDefinition>, double, double, unsigned int, std::shared_ptr<NeighborList>, unsigned int,
                      const std::vector<double> &, const std::string &>());

    py::class_<WellTemperedEnsemble, CollectiveVariable, std::shared_ptr<WellTemperedEnsemble>>(m, "WellTemperedEnsemble")
        .def(py::init<std::shared_ptr<SystemDefinition>, const std::string &>());

    py::class_<AspectRatio, CollectiveVariable, std::shared_ptr<AspectRatio>>(m, "AspectRatio")
        .def(py::init<std::shared_ptr<SystemDefinition>, unsigned int, unsigned int>());

    py::class_<Density, CollectiveVariable, std::shared_ptr<Density>>(m, "Density")
        .def(py::init<std::shared_ptr<SystemDefinition>, const std::string &>());

    // IntegratorMetaDynamics.cc:1315-1349
    py::class_<IntegratorMetaDynamics, std::shared_ptr<IntegratorMetaDynamics>> integrator_metad(m, "IntegratorMetaDynamics");
    integrator_metad
        .def(py::init<std::shared_ptr<SystemDefinition>, double, double, double, double, unsigned int, bool, const std::string &, bool,
                      IntegratorMetaDynamics::Enum>())
        .def("registerCollectiveVariable", &IntegratorMetaDynamics::registerCollectiveVariable)
        .def("removeAllVariables", &IntegratorMetaDynamics::removeAllVariables)
        .def("isInitialized", &IntegratorMetaDynamics::isInitialized)
        .def("setGrid", &IntegratorMetaDynamics::setGrid)
        .def("dumpGrid", &IntegratorMetaDynamics::dumpGrid)
        .def("restartFromGridFile", &IntegratorMetaDynamics::restartFromGridFile)
        .def("setAddHills", &IntegratorMetaDynamics::setAddHills)
        .def("setMode", &IntegratorMetaDynamics::setMode)
        .def("setStride", &IntegratorMetaDynamics::setStride)
        .def("setAdaptive", &IntegratorMetaDynamics::setAdaptive)
        .def("setSigmaG", &IntegratorMetaDynamics::setSigmaG)
        .def("getSigmaInv", &IntegratorMetaDynamics::getSigmaInv)
        .def("getNumGaussians", &IntegratorMetaDynamics::getNumGaussians)
        .def("resetHistogram", &IntegratorMetaDynamics::resetHistogram)
        .def("setMultipleWalkers", &IntegratorMetaDynamics::setMultipleWalkers)
        // HOOMD Integrator interface used by System / analyze.log
        .def("prepRun", &IntegratorMetaDynamics::prepRun)
        .def("update", &IntegratorMetaDynamics::update)
        .def("addForceCompute", &IntegratorMetaDynamics::addForceCompute)
        .def("removeForceComputes", &IntegratorMetaDynamics::removeForceComputes)
        .def("getProvidedLogQuantities", &IntegratorMetaDynamics::getProvidedLogQuantities)
        .def("getLogValue", &IntegratorMetaDynamics::getLogValue)
        // this build
        .def("setFusedPath", &IntegratorMetaDynamics::setFusedPath)
        .def("usedFusedPath", &IntegratorMetaDynamics::usedFusedPath)
        .def("graphPeriod", &IntegratorMetaDynamics::graphPeriod)
        .def("getCurrentValues", &IntegratorMetaDynamics::getCurrentValues)
        .def("getBiasFactors", &IntegratorMetaDynamics::getBiasFactors)
        .def("getEngineHandle", [](IntegratorMetaDynamics &i) { return (size_t)i.getEngine(); });
    py::enum_<IntegratorMetaDynamics::Enum>(integrator_metad, "mode")
        .value("standard", IntegratorMetaDynamics::mode_standard)
        .value("well_tempered", IntegratorMetaDynamics::mode_well_tempered)
        .export_values();

    py::class_<System, std::shared_ptr<System>>(m, "System")
        .def(py::init<std::shared_ptr<SystemDefinition>, unsigned int>())
        .def("setIntegrator", &System::setIntegrator)
        .def("run", &System::run, py::call_guard<py::gil_scoped_release>())
        .def("getCurrentTimeStep", &System::getCurrentTimeStep)
        .def("setGraphMode", &System::setGraphMode)
        .def("lastRunGraphSteps", &System::lastRunGraphSteps)
        .def("lastRunGraphPeriod", &System::lastRunGraphPeriod);

    // the reference also exports the host-path class names (module.cc:29-31: export_LamellarOrderParameter,
    // export_OrderParameterMesh), which cv.py instantiates when the execution configuration has no GPU (cv.py:262-268,
    // 405-410).  This build has no CPU path: the names resolve to the device classes.
    m.attr("LamellarOrderParameter") = m.attr("LamellarOrderParameterGPU");
    m.attr("OrderParameterMesh") = m.attr("OrderParameterMeshGPU");
    }
