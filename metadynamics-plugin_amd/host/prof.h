// prof.h — roctx ranges with the reference's Profiler names (HOOMD `Profiler` push/pop pairs, e.g.
// "Metadynamics" IntegratorMetaDynamics.cc:343-344, "Lamellar" LamellarOrderParameter.cc:44-45, "Mesh" / "forces"
// OrderParameterMesh.cc:930, OrderParameterMeshGPU.cc:368, "Well-Tempered Ensemble" WellTemperedEnsemble.cc:32-33,
// "Derivatives" IntegratorMetaDynamics.cc:1208).  Off unless MTD_ROCTX=1 (then `rocprofv3 --marker-trace` shows them);
// libroctx64 is bound at run time, no link-time dependency.
#pragma once

#include <dlfcn.h>

#include <cstdlib>

namespace mtdhost
{

class ProfRange
    {
    public:
        explicit ProfRange(const char *name) : m_on(api().push != nullptr)
            {
            if (m_on) api().push(name);
            }
        ~ProfRange()
            {
            if (m_on) api().pop();
            }
        ProfRange(const ProfRange &) = delete;
        ProfRange &operator=(const ProfRange &) = delete;

    private:
        struct Api
            {
            int (*push)(const char *) = nullptr;
            int (*pop)() = nullptr;
            };
        static Api &api()
            {
            static Api a = []
                {
                Api r;
                const char *e = std::getenv("MTD_ROCTX");
                if (!e || e[0] != '1') return r;
                void *h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
                if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
                if (!h) return r;
                r.push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
                r.pop = (int (*)())dlsym(h, "roctxRangePop");
                if (!r.push || !r.pop) r.push = nullptr;
                return r;
                }();
            return a;
            }
        bool m_on;
    };

} // namespace mtdhost
