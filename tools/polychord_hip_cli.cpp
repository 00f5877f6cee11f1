// polychord_hip_cli -- runs an ini file through polychord_c_interface_ini with one of the built-in
// device likelihoods; counterpart of the reference's src/drivers/polychord_CC_ini.cpp:10-18.
//   usage: polychord_hip_cli <file.ini> <gaussian|rastrigin|twin_gaussian|corr_gaussian> [batch]
#include "polychord_hip.h"
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <sys/stat.h>

int main(int argc, char **argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: %s <file.ini> <gaussian|rastrigin|twin_gaussian> [batch]\n", argv[0]); return 2; }
    polychord_loglike_fn like = nullptr;
    if (!std::strcmp(argv[2], "gaussian")) like = polychord_hip_gaussian;
    else if (!std::strcmp(argv[2], "rastrigin")) like = polychord_hip_rastrigin;
    else if (!std::strcmp(argv[2], "twin_gaussian")) like = polychord_hip_twin_gaussian;
    else { std::fprintf(stderr, "unknown likelihood %s\n", argv[2]); return 2; }
    if (argc > 3) polychord_hip_set_option("batch", std::atof(argv[3]));
    mkdir("chains", 0755); mkdir("chains/clusters", 0755);
    int comm = 0;
    polychord_c_interface_ini(like, nullptr, argv[1], &comm);
    return 0;
}
