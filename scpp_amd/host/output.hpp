// On-disk output tree of the reference drivers (scpp/src/SC_oneshot.cpp:31-63, SC_sim.cpp:75-103):
//   <root>/output/<Model>/<SC|SC_sim>/<YYYY_MM_DD_HH_MM_SS>/<iter>/{X.txt,U.txt,t.txt}
// rows ", "-separated with the stream's default precision (Eigen::StreamPrecision), one node per line, so that the
// reference's evaluation/ scripts read it unchanged.
#pragma once
#include <ctime>
#include <filesystem>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace scpp
{

// timing.cpp / commonFunctions: getTimeString
inline std::string getTimeString()
{
    char buf[64];
    const std::time_t t = std::time(nullptr);
    std::strftime(buf, sizeof(buf), "%Y_%m_%d_%H_%M_%S", std::localtime(&t));
    return buf;
}

template <class Row>
inline void writeRows(const std::filesystem::path &file, const std::vector<Row> &rows)
{
    std::ofstream f(file);
    for (const auto &r : rows)
    {
        for (size_t j = 0; j < r.size(); j++)
            f << (j ? ", " : "") << r[j];
        f << "\n";
    }
}

inline void makeDir(const std::filesystem::path &p)
{
    if (!std::filesystem::exists(p) && !std::filesystem::create_directories(p))
        throw std::runtime_error("Could not create output directory!");
}

} // namespace scpp
