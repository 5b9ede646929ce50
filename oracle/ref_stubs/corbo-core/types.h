// ORACLE (test infrastructure only): stand-in for control_box_rst's corbo-core/types.h (absent from this image) -- only what the reference's
// robot-model and collocation headers need from it.  See oracle/ref_wrap.cpp.
#pragma once
#include <Eigen/Core>
#include <memory>
#include <vector>
