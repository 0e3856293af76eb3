// metadynamics_host.h — host-side mirror of the reference plugin's C++ classes for the hot path.
//
// Same class names, method names, argument meaning and error behaviour as the reference
// (CollectiveVariable.h, LamellarOrderParameterGPU.h, WellTemperedEnsemble.h, AspectRatio.h, Density.h,
// IntegratorMetaDynamics.h); the bodies call libmtd_hip.so through include/mtd_abi.h instead of the
// CUDA drivers, and the per-step hand-off between CVs and the integrator stays in device memory:
//
//   reference                                   here
//   Scalar getCurrentValue(t) [D2H + host sum]  enqueueCurrentValue(t, engine, slot)  (async, device)
//                                               getCurrentValue(t) = the same + lazy read-back
//   setBiasFactor(Scalar)  [host scalar]        setBiasFactorDevice(const double*)    (device pointer)
//                                               setBiasFactor(Scalar) kept for umbrella / derivatives
#pragma once

#include <fstream>
#include <memory>
#include <string>
#include <vector>

#include "mini_hoomd.h"

namespace mtdhost
{

class IntegratorMetaDynamics;

//! CollectiveVariable.h:32-196
class CollectiveVariable : public ForceCompute
    {
    public:
        enum umbrella_Enum
            {
            no_umbrella = 0,
            linear,
            harmonic,
            wall,
            gaussian
            };

        CollectiveVariable(std::shared_ptr<SystemDefinition> sysdef, const std::string &name);
        virtual ~CollectiveVariable() {}

        //! CollectiveVariable.h:55 — synchronising read-back of the CV value
        virtual double getCurrentValue(unsigned int timestep) { return 0.0; }

        //! device-resident form of getCurrentValue: make `engine` take CV `slot` from this variable.
        //! Default: the host value of getCurrentValue() travels in the next launch's arguments.
        virtual void enqueueCurrentValue(unsigned int timestep, mtd_metad *engine, unsigned int slot);
        //! A variable that is the ONLY one of the grid may run the engine's update itself, fused with the tail of its own value
        //! (cv.steinhardt: mtd_ql_finalize_update_bias): true = updateBiasPotential(timestep) is enqueued, the integrator skips
        //! mtd_metad_update_bias; false (default) = nothing fused, the integrator goes on with enqueueCurrentValue
        virtual bool enqueueValueAndBias(unsigned int timestep, mtd_metad *engine) { return false; }

        virtual void setBiasFactor(double bias)                        // CollectiveVariable.h:63-66
            {
            m_bias = bias;
            m_bias_device = nullptr;
            }
        //! the bias factor stays in device memory (written by the grid engine, read by the force kernels)
        virtual void setBiasFactorDevice(const double *d_bias) { m_bias_device = d_bias; }

        void setUmbrella(umbrella_Enum umbrella)                       // CollectiveVariable.h:71-76
            {
            m_umbrella = umbrella;
            if (umbrella == no_umbrella) m_bias = 0.0;
            }
        void setKappa(double kappa) { m_kappa = kappa; }
        void setWidthFlat(double width) { m_width_flat = width; }
        void setScale(double scale) { m_scale = scale; }
        void setMinimum(double cv0) { m_cv0 = cv0; }
        std::string getName() { return m_cv_name; }

        void computeDerivatives(unsigned int timestep)                // CollectiveVariable.h:120-125
            {
            setBiasFactor(1.0);
            computeBiasForces(timestep);
            }
        virtual bool canComputeDerivatives() { return true; }
        double getUmbrellaPotential(unsigned int timestep);           // CollectiveVariable.cc:68-106
        virtual bool requiresNetForce() { return false; }
        bool hasUmbrella() const { return m_umbrella != no_umbrella; }
        //! a domain-decomposed run (m_pdata->getDomainDecomposition() in the reference): this rank holds a shard of the
        //! particles and the per-step sums are reduced over the ranks of the execution configuration's mailbox
        bool distributed() const { return m_exec_conf->getMailbox() != nullptr; }

        std::vector<std::string> getProvidedLogQuantities() override
            {
            return {"umbrella_energy_" + m_cv_name};
            }
        double getLogValue(const std::string &quantity, unsigned int timestep) override
            {
            if (quantity == "umbrella_energy_" + m_cv_name) return getUmbrellaPotential(timestep);
            throw std::runtime_error("Error querying log quantity");     // CollectiveVariable.h:168-170
            }

    protected:
        void computeForces(unsigned int timestep) override;            // CollectiveVariable.cc:22-66
        virtual void computeBiasForces(unsigned int timestep) {}

        double m_bias;
        const double *m_bias_device;
        std::string m_cv_name;

    private:
        umbrella_Enum m_umbrella;
        double m_cv0, m_kappa, m_width_flat, m_scale;
    };

//! LamellarOrderParameterGPU.h / LamellarOrderParameter.h:30-101
class LamellarOrderParameterGPU : public CollectiveVariable
    {
    public:
        LamellarOrderParameterGPU(std::shared_ptr<SystemDefinition> sysdef, const std::vector<double> &mode,
                                  const std::vector<int3> &lattice_vectors, const std::string &suffix = "");
        void computeBiasForces(unsigned int timestep) override;        // LamellarOrderParameterGPU.cc:99-132
        double getCurrentValue(unsigned int timestep) override;        // LamellarOrderParameter.h:75-79 (always recomputes, Q4)
        void enqueueCurrentValue(unsigned int timestep, mtd_metad *engine, unsigned int slot) override;
        std::vector<std::string> getProvidedLogQuantities() override
            {
            auto l = CollectiveVariable::getProvidedLogQuantities();
            l.push_back(m_log_name);
            return l;
            }
        double getLogValue(const std::string &quantity, unsigned int timestep) override
            {
            if (quantity == m_log_name) return getCurrentValue(timestep);
            return CollectiveVariable::getLogValue(quantity, timestep);
            }
        const std::vector<double> &getMode() const { return m_mode; }
        const std::vector<int3> &getLatticeVectors() const { return m_lattice_vectors; }
        //! this build: trigonometry of THIS variable's kernels (MTD_TRIG_DEFAULT / _HARDWARE / _ACCURATE, mtd_abi.h); variables
        //! that share a fused launch take the accurate functions as soon as one of them asks for them
        void setTrigMode(int mode)
            {
            if (mode != MTD_TRIG_DEFAULT && mode != MTD_TRIG_HARDWARE && mode != MTD_TRIG_ACCURATE)
                throw std::runtime_error("cv.lamellar: trig mode must be 0 (default), 1 (hardware) or 2 (accurate)");
            m_set.trig_mode = mode;
            }
        int getTrigMode() const { return m_set.trig_mode; }

    protected:
        void enqueuePartials();
        std::string m_log_name;
        std::vector<double> m_mode;
        std::vector<int3> m_lattice_vectors;
        mtd_lamellar_set m_set;
        DeviceBuffer m_partials, m_cv_dev;
        unsigned int m_n_partials;
        double m_cv;
        unsigned int m_cv_last_updated;
    };

//! WellTemperedEnsemble.h:20-113
class WellTemperedEnsemble : public CollectiveVariable
    {
    public:
        WellTemperedEnsemble(std::shared_ptr<SystemDefinition> sysdef, const std::string &name);
        bool requiresNetForce() override { return true; }
        double getCurrentValue(unsigned int timestep) override;        // WellTemperedEnsemble.h:50-54
        void enqueueCurrentValue(unsigned int timestep, mtd_metad *engine, unsigned int slot) override;
        void computeBiasForces(unsigned int timestep) override;        // WellTemperedEnsemble.cc:135-188
        std::vector<std::string> getProvidedLogQuantities() override
            {
            auto l = CollectiveVariable::getProvidedLogQuantities();
            l.push_back(m_log_name);
            return l;
            }
        double getLogValue(const std::string &quantity, unsigned int timestep) override
            {
            if (quantity == m_log_name) return getCurrentValue(timestep);
            return CollectiveVariable::getLogValue(quantity, timestep);
            }

    protected:
        void enqueuePartials();
        double m_pe;
        std::string m_log_name;
        DeviceBuffer m_partials, m_sum;
        unsigned int m_n_partials;
    };

//! OrderParameterMeshGPU.h / OrderParameterMesh.h:20-180 (single rank: no ghost cells, no dfft)
class OrderParameterMeshGPU : public CollectiveVariable
    {
    public:
        OrderParameterMeshGPU(std::shared_ptr<SystemDefinition> sysdef, unsigned int nx, unsigned int ny, unsigned int nz,
                              std::vector<double> mode, std::vector<int3> zero_modes);
        virtual ~OrderParameterMeshGPU();
        double getCurrentValue(unsigned int timestep) override;        // OrderParameterMesh.cc:925-968 (cached per timestep)
        void enqueueCurrentValue(unsigned int timestep, mtd_metad *engine, unsigned int slot) override;
        void computeBiasForces(unsigned int timestep) override;        // OrderParameterMesh.cc:1052-1075
        //! convolution-kernel table: stored like the reference, which never applies it to the mesh (Q7)
        void setTable(const std::vector<double> &K, const std::vector<double> &d_K, double kmin, double kmax);   // :148-189
        void setUseTable(bool use_table);                             // OrderParameterMesh.h:60-63
        //! this build: false = interpolation function as intended instead of the reference's unsigned division (Q6)
        void setBugCompatible(bool on);
        //! this build, domain-decomposed runs: true = the mesh is DECOMPOSED into slabs over the ranks (mtd_mesh_slab_*: the
        //! reference's ghost-cell exchange + distributed FFT, OrderParameterMesh.cc:263-316, 659-746); false (default) = every
        //! rank keeps the whole mesh and the ranks sum their assignments (M + 1 doubles per step, :630).  Before the first step.
        void setSlabDecomposition(bool on);
        bool getSlabDecomposition() const { return m_slab; }
        //! event recorded when the CV partial sums of the next compute are complete (mtd_mesh_set_cv_event); nullptr clears
        void setCvEvent(hipEvent_t e) { mtd_mesh_set_cv_event(m_mesh, (void *)e); }
        //! the lamellar CVs of a mixed set (and the engine's deferred grid pass) ride in this mesh's next particle pass
        //! (mtd_mesh_set_lamellar_rider); false when the mesh cannot carry them or is already up to date for `timestep`
        bool armLamellarRider(unsigned int timestep, mtd_metad *engine, const mtd_lamellar_set *set, double *d_partials,
                              unsigned int *n_partials);
        //! true when riders armed earlier are still waiting (nothing consumed them): they are disarmed
        bool clearRider();
        //! The end of a mixed set's step in ONE launch (mtd_mesh_forces_update_bias): the grid engine's launch — scalar chain, first
        //! grid pass, the lamellar CVs' forces — inside this variable's force pass.  false: the shapes do not allow it (the caller
        //! runs the engine's launch, the force pass follows in computeBiasForces as always).  true: the forces of `timestep` are
        //! written (with the bias factor of this very step) and marked as computed.
        bool forcesWithBiasUpdate(unsigned int timestep, mtd_metad *engine, unsigned int mesh_slot, const mtd_lamellar_set *set,
                                  const unsigned int *slots, void *const *lamellar_forces);
        std::vector<std::string> getProvidedLogQuantities() override
            {
            auto l = CollectiveVariable::getProvidedLogQuantities();
            for (const char *n : {"cv_mesh", "qx_max", "qy_max", "qz_max", "sq_max"}) l.push_back(n);   // OrderParameterMesh.cc:118-122
            return l;
            }
        double getLogValue(const std::string &quantity, unsigned int timestep) override;   // :1077-1106

    private:
        void enqueueCV(unsigned int timestep);
        void attachSlab();
        bool m_slab = false, m_slab_attached = false;
        const double *m_slab_sum = nullptr;
        void needFourierMesh();
        bool m_keep_fourier;
        void computeQmax(unsigned int timestep);                      // :1108-1179
        void computeVirial();                                         // :970-1050
        unsigned int m_q_max_last_computed;
        double m_q_max[3], m_sq_max;
        mtd_mesh *m_mesh;
        std::vector<double> m_mode;
        std::vector<int3> m_zero_modes;          // stored and never read, like the reference (OrderParameterMesh.cc:59-63)
        std::vector<double> m_table, m_table_d;
        double m_k_min, m_k_max, m_delta_k;
        bool m_use_table, m_is_first_step;
        const double *m_partials;
        unsigned int m_n_partials, m_cv_last_updated;
        double m_cv;
        DeviceBuffer m_cv_dev;
    };

//! CollectiveWrapper.h / CollectiveWrapper.cc:13-188: the energy of any ForceCompute as collective variable
class CollectiveWrapper : public CollectiveVariable
    {
    public:
        CollectiveWrapper(std::shared_ptr<SystemDefinition> sysdef, std::shared_ptr<ForceCompute> fc, const std::string &name);
        double getCurrentValue(unsigned int timestep) override;        // CollectiveWrapper.h: computeCV then m_energy
        void enqueueCurrentValue(unsigned int timestep, mtd_metad *engine, unsigned int slot) override;
        void computeBiasForces(unsigned int timestep) override;        // CollectiveWrapper.cc:136-179
        std::vector<std::string> getProvidedLogQuantities() override
            {
            auto l = CollectiveVariable::getProvidedLogQuantities();
            l.push_back(m_cv_name);
            return l;
            }
        double getLogValue(const std::string &quantity, unsigned int timestep) override
            {
            if (quantity == m_cv_name) return getCurrentValue(timestep);
            return CollectiveVariable::getLogValue(quantity, timestep);
            }

    private:
        void enqueuePartials(unsigned int timestep);
        std::shared_ptr<ForceCompute> m_fc;
        double m_energy;
        DeviceBuffer m_partials, m_sum;
        unsigned int m_n_partials;
    };

//! SteinhardtQl.h:15-98 (the reference class is host-only; this one runs the same arithmetic on the device)
class SteinhardtQl : public CollectiveVariable
    {
    public:
        SteinhardtQl(std::shared_ptr<SystemDefinition> sysdef, double rcut, double ron, unsigned int lmax,
                     std::shared_ptr<NeighborList> nlist, unsigned int type, const std::vector<double> &Ql_ref,
                     const std::string &log_suffix = "");
        double getCurrentValue(unsigned int timestep) override;        // SteinhardtQl.h:44-48
        void enqueueCurrentValue(unsigned int timestep, mtd_metad *engine, unsigned int slot) override;
        bool enqueueValueAndBias(unsigned int timestep, mtd_metad *engine) override;
        //! this build: the steps of this variable can be replayed from a captured graph (no half-list scratch from a pool, no list rebuild pending)
        bool graphSafe() { return lists().mode != 1; }
        void computeBiasForces(unsigned int timestep) override;        // SteinhardtQl.cc:203-339
        std::vector<std::string> getProvidedLogQuantities() override;  // SteinhardtQl.h:34-42
        double getLogValue(const std::string &quantity, unsigned int timestep) override;   // :49-67

    private:
        void computeCV(unsigned int timestep);                         // SteinhardtQl.cc:62-201
        void accumulateSums(unsigned int timestep);                    // :62-171 + the sum over the ranks (:183-191): Q'_lm in m_scratch
        double m_rcut, m_ron;
        unsigned int m_lmax;
        std::shared_ptr<NeighborList> m_nlist;
        unsigned int m_type;
        std::vector<double> m_Ql_ref, m_Ql;
        unsigned int m_cv_last_updated;
        bool m_have_computed;
        double m_value;
        DeviceBuffer m_scratch;
        const double *m_d_value, *m_d_Ql, *m_d_Qlm;
        // half lists: the symmetric full list the half list stands for (mtd_ql_symmetrize_half_list), rebuilt when the
        // neighbour list changes; the passes then run in the symmetric-full-list mode (no atomics in the force pass)
        struct Lists
            {
            const unsigned int *head, *n_neigh, *nlist;
            int mode;                                                 // half_nlist of the mtd_ql_* calls
            };
        Lists lists();
        DeviceBuffer m_sym_head, m_sym_nneigh, m_sym_nlist, m_sym_work;
        unsigned int m_sym_version;
        bool m_sym_ok;
    };

//! AspectRatio.h / AspectRatio.cc:5-130 — box-shape CV, external virial only
class AspectRatio : public CollectiveVariable
    {
    public:
        AspectRatio(std::shared_ptr<SystemDefinition> sysdef, unsigned int dir1, unsigned int dir2);
        double getCurrentValue(unsigned int timestep) override;        // AspectRatio.cc:24-57
        void computeBiasForces(unsigned int timestep) override;        // AspectRatio.cc:59-130
        bool canComputeDerivatives() override { return false; }

    private:
        unsigned int m_dir1, m_dir2;
    };

//! Density.h / Density.cc:5-54 — N/V, external virial only (group = all particles here)
class Density : public CollectiveVariable
    {
    public:
        Density(std::shared_ptr<SystemDefinition> sysdef, const std::string &suffix);
        double getCurrentValue(unsigned int timestep) override;        // Density.cc:20-27
        void computeBiasForces(unsigned int timestep) override;        // Density.cc:29-54
        bool canComputeDerivatives() override { return false; }
    };

//! IntegratorMetaDynamics.h:66-383 (grid mode; the non-grid Gaussian resummation is unreachable from the
//! Python API, integrate.py:266-267 always calls setGrid(True))
class IntegratorMetaDynamics
    {
    public:
        enum Enum
            {
            mode_standard,
            mode_well_tempered
            };

        IntegratorMetaDynamics(std::shared_ptr<SystemDefinition> sysdef, double deltaT, double W, double T_shift, double T,
                               unsigned int stride, bool add_bias = true, const std::string &filename = "",
                               bool overwrite = false, const Enum mode = mode_standard);
        virtual ~IntegratorMetaDynamics();

        virtual void update(unsigned int timestep);                   // IntegratorMetaDynamics.cc:219-312
        virtual void prepRun(unsigned int timestep);                  // :121-217

        void registerCollectiveVariable(std::shared_ptr<CollectiveVariable> cv, double sigma, double cv_min = 0.0,
                                        double cv_max = 0.0, int num_points = 0);   // .h:117-134
        void removeAllVariables() { m_variables.clear(); }
        //! forces HOOMD's System would hand to computeNetForce (every ForceCompute, CVs included)
        void addForceCompute(std::shared_ptr<ForceCompute> f) { m_forces.push_back(f); }
        void removeForceComputes() { m_forces.clear(); }

        std::vector<std::string> getProvidedLogQuantities() { return m_log_names; }
        double getLogValue(const std::string &quantity, unsigned int timestep);     // .h:161-189

        void setGrid(bool use_grid);                                   // :778-815
        void setMode(Enum mode);
        void setStride(unsigned int stride);
        bool isInitialized() { return m_is_initialized; }
        void dumpGrid(const std::string &filename1, const std::string &filename2, unsigned int period);   // :817-829
        void restartFromGridFile(const std::string &filename) { m_restart_filename = filename; }
        void setAddHills(bool add_bias);
        void setAdaptive(bool adaptive) { m_adaptive = adaptive; }   // IntegratorMetaDynamics.h:248-251
        //! the inverse width matrix of the last deposit (row major n_cv x n_cv)
        std::vector<double> getSigmaInv() const { return m_sigma_inv; }
        //! hills deposited so far (m_num_gaussians, IntegratorMetaDynamics.cc:440)
        unsigned int getNumGaussians();
        void setSigmaG(double sigma_g) { m_sigma_g = sigma_g; }
        void setMultipleWalkers(bool multiple) { m_multiple_walkers = multiple; }
        void resetHistogram();                                         // :1195-1203

        //! this build's own knobs
        //! Steps of the stand-alone run loop as a HIP graph (System::run): the number of consecutive update() calls after which
        //! the sequence of launches repeats itself with the same arguments — lcm(2, stride): the mesh CV alternates between two
        //! sets of tile cursors, a deposit comes every `stride` steps — or 0 when the steps cannot be replayed from a capture:
        //! anything that reads a value back on the host or changes arguments from step to step (hills file, grid dumps, adaptive
        //! widths, umbrellas, walkers, a domain decomposition's exchange numbers, box CVs, the potential energy, wrapped computes),
        //! and pure lamellar sets, whose two-launch step measures SLOWER from a graph (profiles/r4/graph_ab.log).
        unsigned int graphPeriod() const;
        void setFusedPath(bool enable) { m_allow_fused = enable; }     // default on: two launches per step for lamellar CVs
        bool usedFusedPath() const { return m_used_fused; }
        mtd_metad *getEngine() { return m_engine; }
        //! CV values and dV/ds_c of the most recent bias update (synchronises)
        std::vector<double> getCurrentValues();
        std::vector<double> getBiasFactors();

    private:
        struct CollectiveVariableItem                                  // IntegratorMetaDynamics.h:21-28
            {
            std::shared_ptr<CollectiveVariable> m_cv;
            double m_sigma, m_cv_min, m_cv_max;
            unsigned int m_num_points;
            };

        void updateBiasPotential(unsigned int timestep);               // :314-588
        void computeNetForce(unsigned int timestep);
        void openOutputFile();                                         // :74-96
        void writeFileHeader();                                        // :98-119
        void setupGrid();                                              // :590-661 (device side: mtd_metad_create)
        void readGrid(const std::string &filename);                    // :928-1000
        void writeGrid(const std::string &filename, unsigned int timestep);   // :831-926
        bool fusedLamellarPossible() const;
        void fusedLamellarStep(unsigned int timestep);
        std::vector<unsigned int> mixedLamellarSlots() const;
        void buildMixedLamellarSet(const std::vector<unsigned int> &slots);
        void setMixedLamellarSources(const std::vector<unsigned int> &slots, unsigned int n_partials);
        void mixedLamellarCvPass(const std::vector<unsigned int> &slots, hipStream_t stream);
        void mixedLamellarForcePass(const std::vector<unsigned int> &slots, unsigned int timestep, hipStream_t stream,
                                    const std::shared_ptr<OrderParameterMeshGPU> &mesh = nullptr, unsigned int mesh_slot = 0);

        std::shared_ptr<SystemDefinition> m_sysdef;
        std::shared_ptr<ParticleData> m_pdata;
        std::shared_ptr<ExecutionConfiguration> m_exec_conf;
        double m_deltaT, m_W, m_T_shift;
        unsigned int m_stride;
        std::vector<CollectiveVariableItem> m_variables;
        std::vector<std::shared_ptr<ForceCompute>> m_forces;
        std::vector<std::string> m_log_names;
        bool m_is_initialized;
        std::string m_filename;
        bool m_overwrite, m_is_appending;
        std::ofstream m_file;
        std::string m_delimiter;
        bool m_use_grid, m_add_bias;
        std::string m_restart_filename, m_grid_fname1, m_grid_fname2;
        unsigned int m_grid_period, m_cur_file;
        double m_sigma_g;
        bool m_adaptive;
        void computeSigma();                                           // :1205-1294
        std::vector<double> m_sigma_inv;
        DeviceBuffer m_sigma_scratch, m_sigma_exchange;
        double m_temp;
        Enum m_mode;
        bool m_multiple_walkers, m_warned_single_walker;

        mtd_metad *m_engine;
        bool m_allow_fused, m_used_fused;
        mtd_lamellar_set m_fused_set;
        DeviceBuffer m_fused_partials;
        unsigned int m_fused_n_partials;
        std::vector<void *> m_fused_force_ptrs;
    };

//! the part of HOOMD's System the plugin relies on: the run loop calling Integrator::update once per step
class System
    {
    public:
        System(std::shared_ptr<SystemDefinition> sysdef, unsigned int initial_tstep) : m_sysdef(sysdef), m_cur_tstep(initial_tstep) {}
        void setIntegrator(std::shared_ptr<IntegratorMetaDynamics> integrator) { m_integrator = integrator; }
        std::shared_ptr<IntegratorMetaDynamics> getIntegrator() { return m_integrator; }
        ~System();
        void run(unsigned int nsteps);
        unsigned int getCurrentTimeStep() const { return m_cur_tstep; }
        //! this build: replay runs of at least four periods from a HIP graph where the integrator allows it
        //! (IntegratorMetaDynamics::graphPeriod).  OFF by default — measured slower than plain launches for every configuration of
        //! BASELINE.json (profiles/r4/graph_ab.log); MTD_GRAPH=1 or setGraphMode(1) switches it on
        void setGraphMode(int mode) { m_graph_mode = mode; }           // -1 default (environment), 0 off, 1 on
        //! steps of the last run() that were replayed from a graph (0: plain launches), and the period of the step sequence
        //! (a graph holds several periods, ~24 steps)
        unsigned int lastRunGraphSteps() const { return m_last_graph_steps; }
        unsigned int lastRunGraphPeriod() const { return m_last_graph_period; }

    private:
        unsigned int runGraph(unsigned int nsteps, unsigned int period);
        std::shared_ptr<SystemDefinition> m_sysdef;
        std::shared_ptr<IntegratorMetaDynamics> m_integrator;
        unsigned int m_cur_tstep;
        int m_graph_mode = -1;
        hipStream_t m_graph_stream = nullptr;
        unsigned int m_last_graph_steps = 0, m_last_graph_period = 0;
    };

} // namespace mtdhost
