// oracle/ref_shim.cc — TEST INFRASTRUCTURE ONLY.
//
// extern "C" veneer over the only two pieces of the reference that compile without HOOMD:
//   * IndexGrid            (/root/reference/metadynamics/IndexGrid.cc, IndexGrid.h)
//   * fsph::evaluate_SPH   (/root/reference/metadynamics/spherical_harmonics.hpp, SharedArray.hpp)
// The reference sources are compiled from where they lie (oracle/Makefile target `_ref`), never
// copied; the output goes to oracle/_ref/libmtd_refsrc.so (git-ignored, travels with gpurun).
// Used (a) to pin the oracle's ref_index_* / ref_sph_* restatements and (b) by
// tests/golden/make_golden.py to emit the committed fixtures.
#include "IndexGrid.h"
#include "spherical_harmonics.hpp"

#include <complex>
#include <vector>

extern "C" {

unsigned int refsrc_index_get(unsigned int dim, const unsigned int *lengths, const unsigned int *coords)
    {
    std::vector<unsigned int> l(lengths, lengths + dim), c(coords, coords + dim);
    IndexGrid g(l);
    return g.getIndex(c);
    }

void refsrc_index_coords(unsigned int dim, const unsigned int *lengths, unsigned int idx, unsigned int *coords)
    {
    std::vector<unsigned int> l(lengths, lengths + dim), c(dim);
    IndexGrid g(l);
    g.getCoordinates(idx, c);
    for (unsigned int i = 0; i < dim; i++) coords[i] = c[i];
    }

unsigned int refsrc_index_num_elements(unsigned int dim, const unsigned int *lengths)
    {
    std::vector<unsigned int> l(lengths, lengths + dim);
    IndexGrid g(l);
    return g.getNumElements();
    }

// out: (re,im) pairs; per point (lmax+1)^2 values when full_m, else (lmax+1)(lmax+2)/2.
// Argument naming follows the reference's call site SteinhardtQl.cc:143: (phi=polar, theta=azimuth).
void refsrc_evaluate_sph(double *out, unsigned int lmax, const double *phi, const double *theta,
                         unsigned int N, int full_m)
    {
    unsigned int per = full_m ? (lmax + 1) * (lmax + 1) : (lmax + 1) * (lmax + 2) / 2;
    std::vector<std::complex<double> > tmp((size_t)per * N);
    fsph::evaluate_SPH<double>(tmp.data(), lmax, phi, theta, N, full_m != 0);
    for (size_t i = 0; i < tmp.size(); i++)
        {
        out[2 * i] = tmp[i].real();
        out[2 * i + 1] = tmp[i].imag();
        }
    }

}
