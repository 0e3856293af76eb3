// mini_hoomd.h — the few HOOMD-blue v2 core types the plugin's host classes touch, as a stand-alone
// stand-in (HOOMD-blue is not part of the reference tree nor of this image; SURVEY.md App. B lists the
// semantics assumed).  In a real HOOMD-ROCm build these come from hoomd/*.h and this header is not used;
// the classes in metadynamics_host.h only rely on the members declared here.
#pragma once

#include <cstdint>
#include <hip/hip_runtime.h>

#include <array>
#include <cmath>
#include <cstring>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "mtd_abi.h"

namespace mtdhost
{

// int3 / make_int3: HIP's own vector type (hip/hip_runtime.h), the counterpart of the CUDA int3 HOOMD uses
using ::int3;
using ::make_int3;

inline void hip_check(hipError_t e, const char *what)
    {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
    }

inline void mtd_check(int rc, const char *what)
    {
    if (rc != 0) throw std::runtime_error(std::string(what) + ": " + mtd_status_string(rc));
    }

//! RAII device buffer (the role of HOOMD's GPUArray / GlobalArray for this plugin)
class DeviceBuffer
    {
    public:
        DeviceBuffer() : m_ptr(nullptr), m_bytes(0) {}
        explicit DeviceBuffer(size_t bytes) : m_ptr(nullptr), m_bytes(0) { resize(bytes); }
        ~DeviceBuffer() { if (m_ptr) (void)hipFree(m_ptr); }
        DeviceBuffer(const DeviceBuffer &) = delete;
        DeviceBuffer &operator=(const DeviceBuffer &) = delete;
        void resize(size_t bytes)
            {
            if (bytes == m_bytes) return;
            if (m_ptr) hip_check(hipFree(m_ptr), "hipFree");
            m_ptr = nullptr;
            m_bytes = bytes;
            if (bytes)
                {
                hip_check(hipMalloc(&m_ptr, bytes), "hipMalloc");
                hip_check(hipMemset(m_ptr, 0, bytes), "hipMemset");
                }
            }
        void upload(const void *host, size_t bytes) { hip_check(hipMemcpy(m_ptr, host, bytes, hipMemcpyHostToDevice), "hipMemcpy H2D"); }
        void download(void *host, size_t bytes) const { hip_check(hipMemcpy(host, m_ptr, bytes, hipMemcpyDeviceToHost), "hipMemcpy D2H"); }
        void *data() const { return m_ptr; }
        size_t bytes() const { return m_bytes; }

    private:
        void *m_ptr;
        size_t m_bytes;
    };

//! BoxDim (orthorhombic + triclinic), HOOMD conventions: a1=(Lx,0,0), a2=(xy Ly,Ly,0), a3=(xz Lz,yz Lz,Lz)
class BoxDim
    {
    public:
        BoxDim(double Lx = 1.0, double Ly = 1.0, double Lz = 1.0, double xy = 0.0, double xz = 0.0, double yz = 0.0)
            {
            m_L[0] = Lx; m_L[1] = Ly; m_L[2] = Lz;
            m_xy = xy; m_xz = xz; m_yz = yz;
            for (int i = 0; i < 3; ++i) m_lo[i] = -0.5 * m_L[i];
            }
        void setLo(double x, double y, double z) { m_lo[0] = x; m_lo[1] = y; m_lo[2] = z; }
        std::array<double, 3> getL() const { return {m_L[0], m_L[1], m_L[2]}; }
        std::array<double, 3> getLo() const { return {m_lo[0], m_lo[1], m_lo[2]}; }
        double getTiltFactorXY() const { return m_xy; }
        double getTiltFactorXZ() const { return m_xz; }
        double getTiltFactorYZ() const { return m_yz; }
        double getVolume() const { return m_L[0] * m_L[1] * m_L[2]; }
        //! scale all lengths (system.box = system.box.scale(s=...) in test/test_2d.py:32)
        BoxDim scale(double s) const
            {
            BoxDim b(m_L[0] * s, m_L[1] * s, m_L[2] * s, m_xy, m_xz, m_yz);
            b.setLo(m_lo[0] * s, m_lo[1] * s, m_lo[2] * s);
            return b;
            }
        mtd_box toMtd() const
            {
            mtd_box m;
            std::memset(&m, 0, sizeof(m));
            for (int i = 0; i < 3; ++i)
                {
                m.L[i] = m_L[i];
                m.lo[i] = m_lo[i];
                m.periodic[i] = 1;
                }
            m.xy = m_xy; m.xz = m_xz; m.yz = m_yz;
            return m;
            }

    private:
        double m_L[3], m_lo[3], m_xy, m_xz, m_yz;
    };

class ExecutionConfiguration
    {
    public:
        ExecutionConfiguration() : m_stream(nullptr)
            {
            int n = mtd_device_count();
            if (n <= 0) throw std::runtime_error("metadynamics: no HIP device (this build has no CPU path)");
            }
        ~ExecutionConfiguration()
            {
            for (hipEvent_t e : m_events) if (e) (void)hipEventDestroy(e);
            if (m_side) (void)hipStreamDestroy(m_side);
            }
        ExecutionConfiguration(const ExecutionConfiguration &) = delete;
        ExecutionConfiguration &operator=(const ExecutionConfiguration &) = delete;
        bool isCUDAEnabled() const { return true; }   // name kept from the reference (cv.py:260)
        hipStream_t getStream() const { return m_stream; }
        //! every launch of the host classes goes to this stream from now on (default: the null stream, like the reference's
        //! drivers).  A caller that captures steps into a hipGraph needs a stream of its own: the null stream cannot be captured.
        void setStream(uintptr_t stream) { m_stream = reinterpret_cast<hipStream_t>(stream); }
        //! A second stream for launches that depend on little of what the main stream is doing (mixed CV sets: the lamellar
        //! CV pass beside the mesh assignment, the grid-engine launch beside the mesh's inverse transforms), ordered against
        //! the main stream by the three events below.  OFF unless MTD_SIDE_STREAM=1: measured at config 3 (10^6 particles,
        //! 128^3 mesh + one lamellar CV) the step takes 187.9 us with it against 180.6 us on one stream — the kernels it lets
        //! run side by side are all memory-bound, and three cross-stream events cost more than the overlap returns.
        //! (unconditional = true: the stream itself, whatever the switch says — the integrator's narrower use of it, one launch
        //! beside the mesh's inverse transform, is decided there)
        hipStream_t getSideStream(bool unconditional = false)
            {
            static const bool on = [] { const char *e = std::getenv("MTD_SIDE_STREAM"); return e && e[0] == '1'; }();
            if (!on && !unconditional) return nullptr;
            if (!m_side)
                {
                hip_check(hipStreamCreateWithFlags(&m_side, hipStreamNonBlocking), "hipStreamCreateWithFlags");
                for (hipEvent_t &e : m_events) hip_check(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreateWithFlags");
                }
            return m_side;
            }
        hipEvent_t getEvent(unsigned int i) const { return m_events[i]; }
        void sync() const { hip_check(hipStreamSynchronize(m_stream), "hipStreamSynchronize"); }

        //! Domain decomposition: the role of HOOMD's MPI communicator (ExecutionConfiguration::getMPICommunicator) for this
        //! plugin's small per-step sums is played by the xGMI mailbox (mtd_comm, one rank per GPU of the node); the handle
        //! is created and connected by the launcher (metadynamics.xgmi.connect) and only borrowed here
        void setMailbox(uintptr_t comm) { m_comm = reinterpret_cast<mtd_comm *>(comm); }
        mtd_comm *getMailbox() const { return m_comm; }
        unsigned int getNRanks() const { return m_comm ? mtd_comm_world(m_comm) : 1; }
        unsigned int getRank() const { return m_comm ? mtd_comm_rank(m_comm) : 0; }
        //! Control plane of a domain-decomposed run — the role MPI_Allgather on getMPICommunicator() plays in HOOMD: gathers
        //! `bytes` bytes from every rank into `all` (world * bytes, rank order).  Only used at SET-UP (IPC handles of exported
        //! buffers: the mesh CV's large all-reduce and its slab decomposition); supplied by the launcher (metadynamics.xgmi.attach).
        using AllgatherFn = std::function<void(const void *mine, size_t bytes, void *all)>;
        void setAllgather(AllgatherFn f) { m_allgather = std::move(f); }
        void allgather(const void *mine, size_t bytes, void *all) const
            {
            if (getNRanks() == 1)
                {
                std::memcpy(all, mine, bytes);
                return;
                }
            if (!m_allgather)
                throw std::runtime_error("metadynamics: this collective variable needs the control plane of the domain-decomposed run "
                                         "(ExecutionConfiguration::setAllgather; metadynamics.xgmi.attach sets it)");
            m_allgather(mine, bytes, all);
            }
        //! a buffer of `bytes` on every rank that all ranks can read (mtd_comm_share + allgather of the handles + mtd_comm_open):
        //! returns this rank's buffer, peers[r] = rank r's as mapped here.  Collective.
        void *shareBuffer(size_t bytes, std::vector<void *> &peers)
            {
            if (!m_comm) throw std::runtime_error("metadynamics: shareBuffer needs the mailbox (ExecutionConfiguration::setMailbox)");
            const unsigned int world = getNRanks();
            void *local = nullptr;
            unsigned int slot = 0;
            unsigned char mine[MTD_COMM_HANDLE_BYTES];
            mtd_check(mtd_comm_share(m_comm, bytes, &local, &slot, mine), "mtd_comm_share");
            std::vector<unsigned char> all((size_t)world * MTD_COMM_HANDLE_BYTES);
            allgather(mine, MTD_COMM_HANDLE_BYTES, all.data());
            peers.assign(world, nullptr);
            mtd_check(mtd_comm_open(m_comm, slot, all.data(), peers.data()), "mtd_comm_open");
            return local;
            }
        //! sum over the ranks of a few device doubles (the xGMI mailbox; LamellarOrderParameterGPU.cc:69-77, SteinhardtQl.cc:183-191,
        //! WellTemperedEnsemble.cc:57-63, CollectiveWrapper.cc:64-70, IntegratorMetaDynamics.cc:1259-1268); no-op on one rank without a mailbox
        void allreduceSmall(double *d_values, unsigned int n, hipStream_t s) const
            {
            if (m_comm) mtd_check(mtd_comm_allreduce_small(m_comm, d_values, n, s), "mtd_comm_allreduce_small");
            }
        //! sum over the ranks of a LARGE device buffer (the replicated mesh, OrderParameterMesh.cc:263-316, 630): RCCL when a
        //! communicator was set (setCommunicator), else remote loads through the mailbox's exported buffers (mtd_comm_allreduce_pull,
        //! set up on first use with the capacity asked for — collective)
        void allreduceLarge(double *d_values, size_t count, hipStream_t s)
            {
            if (getNRanks() == 1) return;
            if (m_large)
                {
                mtd_check(mtd_comm_allreduce_large(m_large, d_values, count, MTD_ELEM_F64, s), "mtd_comm_allreduce_large");
                return;
                }
            if (count > m_pull_capacity)
                {
                if (m_pull_capacity) throw std::runtime_error("metadynamics: the large all-reduce was set up for a smaller buffer");
                std::vector<void *> in, out;
                const size_t bytes = mtd_comm_pull_bytes(count);
                shareBuffer(bytes, in);
                shareBuffer(bytes, out);
                mtd_check(mtd_comm_pull_attach(m_comm, count, in.data(), out.data()), "mtd_comm_pull_attach");
                m_pull_capacity = count;
                }
            mtd_check(mtd_comm_allreduce_pull(m_comm, d_values, count, s), "mtd_comm_allreduce_pull");
            }
        //! RCCL communicator between the ranks of the domain decomposition (mtd_rccl_create) for the large buffers; optional
        void setCommunicator(uintptr_t rccl) { m_large = reinterpret_cast<mtd_rccl *>(rccl); }
        mtd_rccl *getCommunicator() const { return m_large; }
        //! which path allreduceLarge takes (reported by bench.py)
        const char *largeExchangeName() const { return getNRanks() == 1 ? "none" : (m_large ? "rccl" : "xgmi-pull"); }
        //! the communicator between WALKERS (one simulation per GPU sharing one bias grid): HOOMD's m_partition_comm
        //! (IntegratorMetaDynamics.cc:393-409).  An RCCL communicator (mtd_rccl_create), borrowed.
        void setWalkerCommunicator(uintptr_t rccl) { m_walkers = reinterpret_cast<mtd_rccl *>(rccl); }
        mtd_rccl *getWalkerCommunicator() const { return m_walkers; }

    private:
        hipStream_t m_stream;
        hipStream_t m_side = nullptr;
        hipEvent_t m_events[3] = {nullptr, nullptr, nullptr};
        mtd_comm *m_comm = nullptr;
        mtd_rccl *m_walkers = nullptr;
        mtd_rccl *m_large = nullptr;
        AllgatherFn m_allgather;
        size_t m_pull_capacity = 0;
    };

//! Particle arrays in HOOMD layout; Scalar is chosen per system (dtype), the arrays live in HBM
class ParticleData
    {
    public:
        ParticleData(unsigned int N, int dtype, const std::vector<std::string> &type_names, const BoxDim &box)
            : m_N(N), m_N_global(N), m_dtype(dtype), m_type_names(type_names), m_box(box), m_external_energy(0.0), m_pressure_flag(false), m_virial_pitch(N)
            {
            if (dtype != MTD_F32 && dtype != MTD_F64) throw std::runtime_error("ParticleData: dtype must be MTD_F32 or MTD_F64");
            m_external_virial.fill(0.0);
            m_postype.resize(scalar4Bytes() * N);
            m_net_force.resize(scalar4Bytes() * N);
            m_net_torque.resize(scalar4Bytes() * N);
            m_net_virial.resize(scalarBytes() * 6 * m_virial_pitch);
            }
        unsigned int getN() const { return m_N; }
        //! ghost particles of a domain-decomposed run: stored BEHIND the local ones in the position array (HOOMD's layout);
        //! neighbour lists index them as N, N + 1, ...  Set before the positions are uploaded (the array is reallocated).
        unsigned int getNGhosts() const { return m_n_ghosts; }
        void setNGhosts(unsigned int n)
            {
            m_n_ghosts = n;
            m_postype.resize(scalar4Bytes() * ((size_t)m_N + n));
            }
        unsigned int getNGlobal() const { return m_N_global; }
        void setNGlobal(unsigned int n) { m_N_global = n; }   // a shard of a domain-decomposed system
        unsigned int getNTypes() const { return (unsigned int)m_type_names.size(); }
        std::string getNameByType(unsigned int t) const { return m_type_names.at(t); }
        int getDtype() const { return m_dtype; }
        size_t scalarBytes() const { return m_dtype == MTD_F32 ? 4 : 8; }
        size_t scalar4Bytes() const { return 4 * scalarBytes(); }
        const BoxDim &getGlobalBox() const { return m_box; }
        const BoxDim &getBox() const { return m_box; }
        void setGlobalBox(const BoxDim &b) { m_box = b; }
        DeviceBuffer &getPositions() { return m_postype; }
        DeviceBuffer &getNetForce() { return m_net_force; }
        DeviceBuffer &getNetTorqueArray() { return m_net_torque; }
        DeviceBuffer &getNetVirial() { return m_net_virial; }
        unsigned int getNetVirialPitch() const { return m_virial_pitch; }
        double getExternalEnergy() const { return m_external_energy; }
        void setExternalEnergy(double e) { m_external_energy = e; }
        //! PDataFlags pressure_tensor / isotropic_virial: set when an integrator or logger needs the virial
        bool getPressureFlag() const { return m_pressure_flag; }
        void setPressureFlag(bool f) { m_pressure_flag = f; }
        double getExternalVirial(unsigned int i) const { return m_external_virial.at(i); }
        void setExternalVirial(unsigned int i, double v) { m_external_virial.at(i) = v; }
        //! use caller-owned device memory for the positions (e.g. a torch tensor), no copy
        void borrowPositions(void *d_ptr) { m_borrowed_pos = d_ptr; }
        void *positionsPtr() { return m_borrowed_pos ? m_borrowed_pos : m_postype.data(); }

    private:
        unsigned int m_N, m_N_global;
        unsigned int m_n_ghosts = 0;
        int m_dtype;
        std::vector<std::string> m_type_names;
        BoxDim m_box;
        DeviceBuffer m_postype, m_net_force, m_net_torque, m_net_virial;
        double m_external_energy;
        bool m_pressure_flag;
        std::array<double, 6> m_external_virial;
        unsigned int m_virial_pitch;
        void *m_borrowed_pos = nullptr;
    };

class SystemDefinition
    {
    public:
        SystemDefinition(std::shared_ptr<ParticleData> pdata, std::shared_ptr<ExecutionConfiguration> exec)
            : m_pdata(pdata), m_exec(exec) {}
        std::shared_ptr<ParticleData> getParticleData() const { return m_pdata; }
        std::shared_ptr<ExecutionConfiguration> getExecConf() const { return m_exec; }
        unsigned int getNDimensions() const { return 3; }

    private:
        std::shared_ptr<ParticleData> m_pdata;
        std::shared_ptr<ExecutionConfiguration> m_exec;
    };

//! ForceCompute: owns force / virial arrays, compute(timestep) runs computeForces at most once per step
class ForceCompute
    {
    public:
        explicit ForceCompute(std::shared_ptr<SystemDefinition> sysdef)
            : m_sysdef(sysdef), m_pdata(sysdef->getParticleData()), m_exec_conf(sysdef->getExecConf()), m_last_computed(0),
              m_first_compute(true), m_external_energy(0.0)
            {
            m_force.resize(m_pdata->scalar4Bytes() * m_pdata->getN());
            m_external_virial.fill(0.0);
            }
        virtual ~ForceCompute() {}
        void compute(unsigned int timestep)
            {
            if (!m_first_compute && m_last_computed == timestep) return;
            m_first_compute = false;
            m_last_computed = timestep;
            computeForces(timestep);
            }
        //! mark the forces of `timestep` as already written (the fused step writes them itself)
        void markComputed(unsigned int timestep)
            {
            m_first_compute = false;
            m_last_computed = timestep;
            }
        DeviceBuffer &getForceArray() { return m_force; }
        //! torque Scalar4[N] and virial Scalar[6][pitch] of this compute; allocated (zeroed) on first use
        DeviceBuffer &getTorqueArray()
            {
            if (m_torque.bytes() == 0) m_torque.resize(m_pdata->scalar4Bytes() * m_pdata->getN());
            return m_torque;
            }
        DeviceBuffer &getVirialArray()
            {
            if (m_virial.bytes() == 0) m_virial.resize(m_pdata->scalarBytes() * 6 * getVirialPitch());
            return m_virial;
            }
        unsigned int getVirialPitch() const { return m_pdata->getN(); }
        double getExternalEnergy() const { return m_external_energy; }
        unsigned int numParticles() const { return m_pdata->getN(); }
        int dtype() const { return m_pdata->getDtype(); }
        double getExternalVirial(unsigned int i) const { return m_external_virial.at(i); }
        virtual std::vector<std::string> getProvidedLogQuantities() { return {}; }
        virtual double getLogValue(const std::string &quantity, unsigned int)
            {
            throw std::runtime_error("Error querying log quantity " + quantity);
            }

    protected:
        virtual void computeForces(unsigned int timestep) = 0;
        std::shared_ptr<SystemDefinition> m_sysdef;
        std::shared_ptr<ParticleData> m_pdata;
        std::shared_ptr<ExecutionConfiguration> m_exec_conf;
        DeviceBuffer m_force;
        std::array<double, 6> m_external_virial;
        unsigned int m_last_computed;
        bool m_first_compute;
        DeviceBuffer m_torque, m_virial;
        double m_external_energy;
    };

//! Stand-in for "any other HOOMD ForceCompute" (pair, bond, external ...): force (xyz + energy w), torque and virial
//! are prescribed from outside and restored on every compute(), as a real compute rewrites its arrays each step.
class PrescribedForceCompute : public ForceCompute
    {
    public:
        explicit PrescribedForceCompute(std::shared_ptr<SystemDefinition> sysdef) : ForceCompute(sysdef) {}
        void setArrays(const void *force, const void *torque, const void *virial)
            {
            const size_t b4 = m_pdata->scalar4Bytes() * m_pdata->getN(), bv = m_pdata->scalarBytes() * 6 * getVirialPitch();
            m_src_force.resize(b4);
            m_src_torque.resize(b4);
            m_src_virial.resize(bv);
            if (b4) m_src_force.upload(force, b4);
            if (b4) m_src_torque.upload(torque, b4);
            if (bv) m_src_virial.upload(virial, bv);
            }
        void setExternalEnergy(double e) { m_external_energy = e; }

    protected:
        void computeForces(unsigned int) override
            {
            if (m_src_force.bytes() == 0) throw std::runtime_error("PrescribedForceCompute: no arrays set");
            hipStream_t s = m_exec_conf->getStream();
            hip_check(hipMemcpyAsync(m_force.data(), m_src_force.data(), m_src_force.bytes(), hipMemcpyDeviceToDevice, s), "force copy");
            hip_check(hipMemcpyAsync(getTorqueArray().data(), m_src_torque.data(), m_src_torque.bytes(), hipMemcpyDeviceToDevice, s), "torque copy");
            hip_check(hipMemcpyAsync(getVirialArray().data(), m_src_virial.data(), m_src_virial.bytes(), hipMemcpyDeviceToDevice, s), "virial copy");
            }
        DeviceBuffer m_src_force, m_src_torque, m_src_virial;
    };

//! The part of HOOMD's md::NeighborList the plugin reads (SteinhardtQl.cc:80-85): head list, neighbour counts, flat list.
//! Building the list is HOOMD core; in the stand-alone system the arrays are supplied from outside (setLists).
class NeighborList
    {
    public:
        enum storageMode
            {
            half,
            full
            };
        explicit NeighborList(std::shared_ptr<SystemDefinition> sysdef) : m_N(sysdef->getParticleData()->getN()), m_mode(full), m_last(0), m_has(false) {}
        void setStorageMode(storageMode m) { m_mode = m; }
        storageMode getStorageMode() const { return m_mode; }
        void setLists(const unsigned int *head, const unsigned int *n_neigh, size_t n, const unsigned int *nlist, size_t n_list)
            {
            if (n != m_N) throw std::runtime_error("NeighborList::setLists: head_list / n_neigh must have N entries");
            m_head.resize(sizeof(unsigned int) * n);
            m_nneigh.resize(sizeof(unsigned int) * n);
            m_nlist.resize(sizeof(unsigned int) * (n_list ? n_list : 1));
            if (n) m_head.upload(head, sizeof(unsigned int) * n);
            if (n) m_nneigh.upload(n_neigh, sizeof(unsigned int) * n);
            if (n_list) m_nlist.upload(nlist, sizeof(unsigned int) * n_list);
            m_has = true;
            m_version++;
            // a full list of HOOMD is symmetric by construction ((i, j) listed <=> (j, i) listed) and, on one rank, indexes
            // local particles only; the stand-in checks what it is handed, since SteinhardtQl's CV pass then visits every pair once
            m_symmetric = true;
            for (size_t i = 0; i < n && m_symmetric; ++i)
                for (unsigned int k = 0; k < n_neigh[i] && m_symmetric; ++k)
                    {
                    const unsigned int j = nlist[head[i] + k];
                    bool back = false;
                    if (j < n)
                        for (unsigned int q = 0; q < n_neigh[j] && !back; ++q) back = nlist[head[j] + q] == i;
                    m_symmetric = back;
                    }
            }
        //! full list, (i, j) listed <=> (j, i) listed, no ghost particles
        bool isSymmetricFull() const { return m_mode == full && m_has && m_symmetric; }
        //! HOOMD rebuilds the list here when particles moved too far; the stand-in only checks that one was supplied
        void compute(unsigned int timestep)
            {
            if (!m_has) throw std::runtime_error("NeighborList: no neighbour list supplied (nlist.set_lists)");
            m_last = timestep;
            }
        //! counts the lists handed in (HOOMD: a rebuild of the list): consumers that derive something from a list redo it then
        unsigned int getVersion() const { return m_version; }
        DeviceBuffer &getHeadList() { return m_head; }
        DeviceBuffer &getNNeighArray() { return m_nneigh; }
        DeviceBuffer &getNListArray() { return m_nlist; }

    private:
        unsigned int m_N;
        storageMode m_mode;
        unsigned int m_last;
        bool m_has;
        bool m_symmetric = false;
        unsigned int m_version = 0;
        DeviceBuffer m_head, m_nneigh, m_nlist;
    };

} // namespace mtdhost
