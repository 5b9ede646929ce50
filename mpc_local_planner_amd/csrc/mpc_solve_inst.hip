// mpc_solve_inst.hip -- one (arithmetic type, model) pair of the solve kernel per object file (split build):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DMPC_SPLIT_BUILD -DMPC_SOLVE_INST -DMPC_INST_T=double -DMPC_INST_MODEL=1 -c mpc_solve_inst.hip
#include "mpc_solve_kernel.hpp"

#if !defined(MPC_INST_T) || !defined(MPC_INST_MODEL)
#error "mpc_solve_inst.hip needs -DMPC_INST_T=<double|float> -DMPC_INST_MODEL=<0..3>"
#endif

namespace mpc {
template hipError_t launch_solve<MPC_INST_T, MPC_INST_MODEL>(const SolveLaunch&, const Problem<MPC_INST_T>&);
template hipError_t solve_occupancy<MPC_INST_T, MPC_INST_MODEL>(const SolveLaunch&, int*);
}
