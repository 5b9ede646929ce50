#pragma once
namespace visualization_msgs { struct Marker {}; }
