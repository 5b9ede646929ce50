#pragma once
namespace mpc_local_planner { struct FootprintReconfigureConfig { bool is_footprint_dynamic = false; }; }
