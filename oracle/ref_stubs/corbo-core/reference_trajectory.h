// ORACLE (test infrastructure only): corbo::ReferenceTrajectoryInterface appears in the signature of StageInequalitySE2::update only (unused there).
#pragma once
#include <corbo-core/types.h>
namespace corbo {
class ReferenceTrajectoryInterface { public: virtual ~ReferenceTrajectoryInterface() = default; };
}
