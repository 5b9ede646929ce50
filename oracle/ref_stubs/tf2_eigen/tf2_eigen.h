#pragma once
#include <tf2/utils.h>
