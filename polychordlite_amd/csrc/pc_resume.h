// pc_resume.h -- the reference's .resume file (src/polychord/read_write.F90:219-288 writer, :384-476
// reader; pypolychord/polychord.py:650-789 writes the same grammar for `cube_samples`): labelled text
// sections, integers as I12, reals as E24.15E3, one array per line, 3-D arrays as one block per cluster
// introduced by a separator line.
#pragma once
#include <string>
#include <vector>

struct PcResume {
    int nDims = 0, nDerived = 0, ndead = 0, ncluster = 0, ncluster_dead = 0;
    std::vector<int> grade_dims, num_repeats;
    std::vector<long long> nlike;
    std::vector<int> nlive, nphantom, imin;          // per cluster (imin: 1-based position of the lowest point)
    double logZ = 0, logZ2 = 0, thin_posterior = 0, logX_last_update = 0;
    std::vector<double> logLp, logXp, logZXp, logZp, logZp2, logZpXp, logXpXq /* [nc*nc], column major */;
    std::vector<double> logZp_dead, logZp2_dead;
    std::vector<double> covmat, cholesky;            // [nc][D*D]: one matrix per cluster, written column by column
    std::vector<std::vector<double>> live, phantom;  // per cluster, [n][nTotal] row major (point = line)
    std::vector<double> dead, logweights;            // [ndead][nTotal], [ndead]
    int nTotal() const { return 2 * nDims + nDerived + 2; }
};

// false (with a message in `err`) when the file is missing or malformed
bool pc_resume_read(const std::string &path, PcResume &r, std::string &err);
bool pc_resume_write(const std::string &path, const PcResume &r, double logzero, std::string &err);
