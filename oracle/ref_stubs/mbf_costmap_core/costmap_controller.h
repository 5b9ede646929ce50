// ORACLE (test infrastructure only): mbf_costmap_core::CostmapController as a base to derive from
#pragma once
namespace mbf_costmap_core { class CostmapController { public: virtual ~CostmapController() = default; }; }
