// ORACLE (test infrastructure only): stand-in for control_box_rst's corbo-core/types.h (absent from this image) -- only what the reference's
// sources compiled by oracle/ref_wrap.cpp need from it.
#pragma once
#include <Eigen/Core>
#include <memory>
#include <vector>

namespace corbo {
constexpr const double CORBO_INF_DBL = 2e30;      // control_box_rst: corbo-core/types.h ("representation for infinity"); only ever compared against
}
// corbo-core/console.h: console messages
#ifndef PRINT_WARNING_COND_ONCE
#define PRINT_WARNING_COND_ONCE(cond, msg) do { } while (0)
#define PRINT_WARNING(msg) do { } while (0)
#define PRINT_DEBUG(msg) do { } while (0)
#define PRINT_DEBUG_NAMED(msg) do { } while (0)
#endif
