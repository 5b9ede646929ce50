// ORACLE (test infrastructure only): what a maintainer of the reference would put into include/mpc_local_planner/controller.h to build the plugin on this repository's
// library -- see include/mpc_reference_binding.hpp.  With this directory first on the include path the reference's plugin source compiles against the binding.
#pragma once
#include <mpc_reference_binding.hpp>
