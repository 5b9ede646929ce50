#pragma once
#include <boost/shared_ptr.hpp>
