// ORACLE (test infrastructure only): the outcome codes of mbf_msgs/ExePath.action the reference returns
#pragma once
#include <cstdint>
namespace mbf_msgs { struct ExePathResult { enum : uint32_t { SUCCESS = 0, NO_VALID_CMD = 100, INVALID_PATH = 103, NOT_INITIALIZED = 112, INTERNAL_ERROR = 114 }; }; }
