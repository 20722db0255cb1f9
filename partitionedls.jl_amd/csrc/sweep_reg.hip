// sweep_reg.hip — register-resident tableau sweep (placeholder until the kernel lands).
#include "common.h"
namespace partls {
bool sweep_reg_supported(int) { return false; }
int sweep_reg_tiles(int n) { return (n + 15) / 16; }
size_t sweep_reg_t0_doubles(int) { return 8; }
hipError_t launch_layout_reg(const double *, int, int, double *, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_sweep_reg(const SweepParams &, int, int, hipStream_t) { return hipErrorNotSupported; }
}
