// MtdHoomd.h — SURVEY.md §8f N5: the plugin's classes on top of a real HOOMD-blue v2.x (ROCm) tree, calling libmtd_hip.so.
//
// Compiled ONLY with -DMTD_WITH_HOOMD inside a HOOMD plugin build (hoomd/ForceCompute.h, hoomd/md/IntegratorTwoStep.h on the
// include path); NEITHER BOX OF THIS PROJECT HAS A HOOMD TREE, so this file has never been compiled — it is kept thin and
// follows the reference's declarations line by line so that a maintainer can check it against the headers:
//   CollectiveVariable          CollectiveVariable.h:32-196
//   LamellarOrderParameterGPU   LamellarOrderParameterGPU.h, constructor LamellarOrderParameterGPU.cc:134-141
//   IntegratorMetaDynamics      IntegratorMetaDynamics.h:66-383, constructor IntegratorMetaDynamics.cc:1315-1349
// The stand-alone mirror of the same classes (host/metadynamics_host.h over host/mini_hoomd.h) is what the tests exercise.
#pragma once
#ifdef MTD_WITH_HOOMD

#include <hoomd/ForceCompute.h>
#include <hoomd/md/IntegratorTwoStep.h>

#include <memory>
#include <string>
#include <vector>

#include "mtd_abi.h"

namespace mtdhoomd
{

//! BoxDim -> the C ABI's POD
inline mtd_box to_mtd_box(const BoxDim &b)
    {
    mtd_box o;
    const Scalar3 L = b.getL(), lo = b.getLo();
    o.L[0] = L.x; o.L[1] = L.y; o.L[2] = L.z;
    o.lo[0] = lo.x; o.lo[1] = lo.y; o.lo[2] = lo.z;
    o.xy = b.getTiltFactorXY(); o.xz = b.getTiltFactorXZ(); o.yz = b.getTiltFactorYZ();
    const uchar3 p = b.getPeriodic();
    o.periodic[0] = p.x; o.periodic[1] = p.y; o.periodic[2] = p.z;
    return o;
    }

inline int mtd_dtype() { return sizeof(Scalar) == 4 ? MTD_F32 : MTD_F64; }

//! CollectiveVariable.h:32-196 — same interface; the bias factor may stay in device memory
class CollectiveVariable : public ForceCompute
    {
    public:
        enum umbrella_Enum { no_umbrella = 0, linear, harmonic, wall, gaussian };

        CollectiveVariable(std::shared_ptr<SystemDefinition> sysdef, const std::string &name)
            : ForceCompute(sysdef), m_bias(0.0), m_bias_device(nullptr), m_cv_name(name), m_umbrella(no_umbrella), m_cv0(0.0),
              m_kappa(1.0), m_width_flat(0.0), m_scale(1.0) {}
        virtual ~CollectiveVariable() {}

        virtual Scalar getCurrentValue(unsigned int timestep) { return Scalar(0.0); }
        //! device-resident form: make `engine` take collective variable `slot` from this object (no host read-back)
        virtual void enqueueCurrentValue(unsigned int timestep, mtd_metad *engine, unsigned int slot)
            {
            mtd_metad_set_cv_value(engine, slot, (double)getCurrentValue(timestep));
            }
        virtual void setBiasFactor(Scalar bias) { m_bias = bias; m_bias_device = nullptr; }
        virtual void setBiasFactorDevice(const double *d_bias) { m_bias_device = d_bias; }
        void setUmbrella(umbrella_Enum u) { m_umbrella = u; if (u == no_umbrella) m_bias = Scalar(0.0); }
        void setKappa(Scalar kappa) { m_kappa = kappa; }
        void setWidthFlat(Scalar width) { m_width_flat = width; }
        void setScale(Scalar scale) { m_scale = scale; }
        void setMinimum(Scalar cv0) { m_cv0 = cv0; }
        std::string getName() { return m_cv_name; }
        virtual bool requiresNetForce() { return false; }
        void computeDerivatives(unsigned int timestep) { setBiasFactor(Scalar(1.0)); computeBiasForces(timestep); }

    protected:
        virtual void computeForces(unsigned int timestep) { computeBiasForces(timestep); }   // umbrella add-on: CollectiveVariable.cc:22-66
        virtual void computeBiasForces(unsigned int timestep) {}
        Scalar m_bias;
        const double *m_bias_device;
        std::string m_cv_name;
        umbrella_Enum m_umbrella;
        Scalar m_cv0, m_kappa, m_width_flat, m_scale;
    };

//! LamellarOrderParameterGPU.h — constructor signature of LamellarOrderParameterGPU.cc:134-141
class LamellarOrderParameterGPU : public CollectiveVariable
    {
    public:
        LamellarOrderParameterGPU(std::shared_ptr<SystemDefinition> sysdef, const std::vector<Scalar> &mode,
                                  const std::vector<int3> lattice_vectors, const std::string &suffix = "");
        Scalar getCurrentValue(unsigned int timestep) override;                 // LamellarOrderParameter.h:75-79
        void enqueueCurrentValue(unsigned int timestep, mtd_metad *engine, unsigned int slot) override;
        const mtd_lamellar_set &getSet() const { return m_set; }

    protected:
        void computeBiasForces(unsigned int timestep) override;                 // LamellarOrderParameterGPU.cc:99-132
        void enqueuePartials();
        mtd_lamellar_set m_set;
        GPUArray<double> m_partials;                                            // mtd_lamellar_scratch_doubles()
        unsigned int m_n_partials;
    };

//! IntegratorMetaDynamics.h:66-383 — constructor signature of IntegratorMetaDynamics.cc:1315-1349
class IntegratorMetaDynamics : public IntegratorTwoStep
    {
    public:
        enum Enum { mode_standard, mode_well_tempered };

        IntegratorMetaDynamics(std::shared_ptr<SystemDefinition> sysdef, Scalar deltaT, Scalar W, Scalar T_shift, Scalar T,
                               unsigned int stride, bool add_bias = true, const std::string &filename = "", bool overwrite = false,
                               const Enum mode = mode_standard);
        virtual ~IntegratorMetaDynamics();

        virtual void update(unsigned int timestep);                              // :219-312
        virtual void prepRun(unsigned int timestep);                             // :121-217
        void registerCollectiveVariable(std::shared_ptr<CollectiveVariable> cv, Scalar sigma, Scalar cv_min = Scalar(0.0),
                                        Scalar cv_max = Scalar(0.0), int num_points = 0);
        void removeAllVariables() { m_variables.clear(); }
        bool isInitialized() { return m_is_initialized; }
        void setGrid(bool use_grid) { m_use_grid = use_grid; }
        void setAddHills(bool add_bias);
        void setMode(Enum mode);
        void setStride(unsigned int stride);
        void setMultipleWalkers(bool multiple) { m_multiple_walkers = multiple; }
        //! the communicator between walkers (m_partition_comm of the reference): an RCCL communicator built by the launcher
        void setWalkerCommunicator(mtd_rccl *walkers) { m_walkers = walkers; }

    private:
        struct Item { std::shared_ptr<CollectiveVariable> m_cv; Scalar m_sigma, m_cv_min, m_cv_max; unsigned int m_num_points; };
        void updateBiasPotential(unsigned int timestep);                         // :314-588 on the device
        void setupGrid();                                                        // :590-661 -> mtd_metad_create
        bool allLamellar() const;

        Scalar m_W, m_T_shift, m_temp;
        unsigned int m_stride;
        bool m_add_bias, m_use_grid, m_is_initialized, m_multiple_walkers;
        Enum m_mode;
        std::vector<Item> m_variables;
        mtd_metad *m_engine;
        mtd_rccl *m_walkers;
        GPUArray<double> m_scratch;
    };

void export_all(pybind11::module &m);

} // namespace mtdhoomd
#endif // MTD_WITH_HOOMD
