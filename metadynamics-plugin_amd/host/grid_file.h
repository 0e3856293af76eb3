// grid_file.h — the text formats of the bias-grid dump (IntegratorMetaDynamics.cc:831-925 writeGrid, :928-1000 readGrid) and
// of one line of the hills log (:523-550) as PURE HOST functions over plain arrays: no device, no engine handle.  The
// integrator stages the arrays (mtd_metad_get_array / set_array) and calls these; the CPU tests and the sanitizer run
// (tools/asan.sh) exercise them without a GPU.
#pragma once

#include <iomanip>
#include <istream>
#include <ostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace mtdhost
{

struct GridFileData
    {
    unsigned int num_gaussians = 0;
    std::vector<double> grid, sigma_grid, rew, weight;
    std::vector<unsigned int> hist, hist_gauss;

    void resize(size_t len)
        {
        grid.assign(len, 0.0);
        sigma_grid.assign(len, 0.0);
        rew.assign(len, 0.0);
        weight.assign(len, 0.0);
        hist.assign(len, 0u);
        hist_gauss.assign(len, 0u);
        }
    };

// writeGrid (:847-921): three header lines, the column names, one line per grid cell (first collective variable fastest,
// IndexGrid.cc:46-58), ten significant digits.  sigma_grid holds the SUM of det(sigma^-1) over the cell's hills; the file holds
// the mean (:909-914).
inline void format_grid_file(std::ostream &file, const std::vector<std::string> &names, const std::vector<double> &cv_min,
                             const std::vector<double> &cv_max, const std::vector<unsigned int> &num_points, const std::string &delimiter,
                             const GridFileData &d)
    {
    const size_t dim = names.size();
    if (cv_min.size() != dim || cv_max.size() != dim || num_points.size() != dim) throw std::runtime_error("Error dumping grid.");
    size_t len = 1;
    for (size_t i = 0; i < dim; ++i) len *= num_points[i];
    if (d.grid.size() != len || d.sigma_grid.size() != len || d.rew.size() != len || d.weight.size() != len || d.hist.size() != len ||
        d.hist_gauss.size() != len)
        throw std::runtime_error("Error dumping grid.");
    file << "#n_cv: " << dim << std::endl;
    file << "#dim: ";
    for (size_t i = 0; i < dim; i++) file << " " << num_points[i];
    file << std::endl;
    file << "#num_gaussians: " << d.num_gaussians << std::endl;
    for (size_t i = 0; i < dim; i++) file << names[i] << delimiter;
    file << "grid_value" << delimiter << "det_sigma" << delimiter << "num_gaussians" << delimiter << "hist" << delimiter
         << "hist_reweight" << delimiter << "weight" << std::endl;

    std::vector<unsigned int> coords(dim);
    for (size_t grid_idx = 0; grid_idx < len; grid_idx++)
        {
        size_t rest = grid_idx;                                        // IndexGrid::getCoordinates, first CV fastest
        for (size_t i = 0; i < dim; ++i)
            {
            coords[i] = (unsigned int)(rest % num_points[i]);
            rest /= num_points[i];
            }
        for (size_t i = 0; i < dim; ++i)
            {
            double delta = (cv_max[i] - cv_min[i]) / (num_points[i] - 1);
            double val = cv_min[i] + coords[i] * delta;
            file << std::setprecision(10) << val << delimiter;
            }
        file << std::setprecision(10) << d.grid[grid_idx];
        double val = d.hist_gauss[grid_idx] > 0 ? d.sigma_grid[grid_idx] / (double)d.hist_gauss[grid_idx] : 0.0;   // :909-914
        file << delimiter << std::setprecision(10) << val;
        file << delimiter << d.hist_gauss[grid_idx];
        file << delimiter << d.hist[grid_idx];
        file << delimiter << std::setprecision(10) << d.rew[grid_idx];
        file << delimiter << std::setprecision(10) << d.weight[grid_idx];
        file << std::endl;
        }
    }

// readGrid (:928-1000): skips "#n_cv" and "#dim", takes the hill count of the third line, skips the column names, then reads
// exactly `len` cell lines — the leading n_cv columns (node coordinates) are skipped, a premature end of the file is an error
// (:973-977).  Like the reference it does not check the header against the grid in memory; a field that does not parse leaves
// zero in its place (operator>> of a failed stream), never anything uninitialised.
inline void parse_grid_file(std::istream &file, size_t n_cv, size_t len, GridFileData &d)
    {
    std::string line, tmp;
    getline(file, line);
    getline(file, line);
    getline(file, line);
    d.resize(len);
        {
        std::istringstream iss(line);
        unsigned int n = 0;
        iss >> tmp >> n;
        d.num_gaussians = iss.fail() ? 0u : n;
        }
    getline(file, line);
    for (size_t grid_idx = 0; grid_idx < len; grid_idx++)
        {
        if (!file.good()) throw std::runtime_error("Error reading grid.");   // premature end (:973-977)
        getline(file, line);
        std::istringstream iss(line);
        for (size_t i = 0; i < n_cv; i++) iss >> tmp;
        double g = 0.0, sg = 0.0, r = 0.0, w = 0.0;
        unsigned int hg = 0, h = 0;
        iss >> g >> sg >> hg >> h >> r >> w;
        if (iss.fail() && !iss.eof())
            {
            // a malformed field: everything from it on stays zero for this cell (the reference's stream does the same)
            }
        d.grid[grid_idx] = g;
        d.hist_gauss[grid_idx] = hg;
        d.hist[grid_idx] = h;
        d.sigma_grid[grid_idx] = sg * hg;                              // :992
        d.rew[grid_idx] = r;
        d.weight[grid_idx] = w;
        }
    }

// one line of the hills log (:523-550): timestep, W exp(-V / dT), then per collective variable its value and row i of the
// width matrix WITHOUT delimiters between the row's entries (Q16)
inline void format_hills_line(std::ostream &file, unsigned int timestep, double W, const std::vector<double> &cv,
                              const std::vector<double> &sigma_inv /* n_cv^2 */, const std::string &delimiter)
    {
    const size_t n = cv.size();
    if (sigma_inv.size() != n * n) throw std::runtime_error("hills log: width matrix of the wrong size");
    file << std::setprecision(10) << timestep << delimiter;
    file << std::setprecision(10) << W << delimiter;
    for (size_t i = 0; i < n; ++i)
        {
        file << std::setprecision(10) << cv[i] << delimiter;
        for (size_t j = 0; j < n; ++j) file << std::setprecision(10) << sigma_inv[i * n + j];
        if (i != n - 1) file << delimiter;
        }
    file << std::endl;
    }

} // namespace mtdhost
